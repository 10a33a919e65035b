#!/bin/bash
# GPU box: parity tests with a library variant from build_variants/. Usage: bash tools/variant_parity.sh <variant>
mkdir -p gpurun_out/r03
LOG=gpurun_out/r03/parity_$1.log
RT_LIB_PATH=$PWD/build_variants/$1.so timeout -k 10 300 python3 -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q -m gpu --timeout 100 --timeout-method thread > $LOG 2>&1
RC=$?
echo "rc=$RC" >> $LOG
tail -n 4 $LOG
exit $RC
