#!/bin/bash
# Runs bench.py once per library variant in build_variants/ (GPU box). Usage: bash tools/sweep.sh [names...]
for v in "$@"; do
  RT_LIB_PATH=$PWD/build_variants/$v.so timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline-pass 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('%-12s %8.1f Mrays/s  %7.3f ms/step' % ('$v', d['value'], d['ms_per_step']))
" || exit 1
done
