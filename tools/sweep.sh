#!/bin/bash
# Runs bench.py once per library variant in build_variants/ (GPU box). Usage: bash tools/sweep.sh [names...]
# Prints throughput, then lane use and scheduler rounds per state (new, interior, leaf, shade) from the counting replay.
for v in "$@"; do
  RT_LIB_PATH=$PWD/build_variants/$v.so timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline ${SWEEP_ARGS} 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = (d.get('roofline') or {}).get('scheduler') or d.get('roofline') or {}
        print('%-12s %8.1f Mrays/s  %7.3f ms/step  use %s  rounds %s  cyc/round %s  share %s' % ('$v', d['value'], d['ms_per_step'], r.get('lane_use_new_interior_leaf_shade'), r.get('rounds_new_interior_leaf_shade'), r.get('wave_cycles_per_round_new_interior_leaf_shade'), r.get('wave_time_share_new_interior_leaf_shade')))
" || exit 1
done
