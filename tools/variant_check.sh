#!/bin/bash
# GPU box: parity tests on the library in the tree, then (only if they pass) a timing sweep over build_variants/.
# Usage: bash tools/variant_check.sh <tag> <variant names...>   (the tree's own library is added to the sweep as "tree")
TAG=$1; shift
mkdir -p gpurun_out/r03
LOG=gpurun_out/r03/${TAG}_tests.log
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q -m gpu --timeout 100 --timeout-method thread > $LOG 2>&1
RC=$?
echo "rc=$RC" >> $LOG
tail -n 6 $LOG
[ $RC -eq 0 ] || exit $RC
cp raytracing_amd/librt_mi355x.so build_variants/tree.so
bash tools/sweep.sh "$@" tree "$@" tree > gpurun_out/r03/${TAG}_sweep.log 2>&1
cat gpurun_out/r03/${TAG}_sweep.log
