#!/bin/bash
# GPU box: one-frame launches where the end of the launch dominates, per library variant in build_variants/:
# the dragon-like scene (glass: single paths hundreds of segments long), the teapot-like scene, and 1/8 shards of the benchmark frame.
# Usage: bash tools/tail_order_check.sh <variant names...>
mkdir -p gpurun_out/r03
for v in "$@"; do
  export RT_LIB_PATH=$PWD/build_variants/$v.so
  for wl in dragon_like_1920x1080_b16 teapot_like_1920x1080_b8; do
    timeout -k 10 300 python3 bench.py --workload $wl --steps 40 --warmup 24 --no-cpu-baseline --no-roofline-pass 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('%-8s %-28s %8.1f Mrays/s %7.3f ms/step (kernel %7.3f)' % ('$v', '$wl', d['value'], d['ms_per_step'], d['config']['trace_kernel_ms_per_step']))
" || exit 1
  done
  timeout -k 10 300 python3 tools/shard_probe.py 2>/dev/null | grep -E "^world (1|8|32):" | sed "s/^/$v sponza shard /" || exit 1
done
