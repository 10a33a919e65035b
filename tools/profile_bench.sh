#!/bin/bash
# Profiles bench.py on the GPU box: one kernel-trace pass, then PMC passes (each in its own run).
# Usage (through gpurun): bash tools/profile_bench.sh <tag> [extra bench args]
set -o pipefail
TAG=${1:-r01}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline-pass $@"
python3 bench.py $ARGS > $OUT/plain.json 2> $OUT/plain.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || exit 1
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "FETCH_SIZE TCC_HIT_sum" \
            "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc$i -- python3 bench.py $ARGS > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.log; exit 1; }
done
find $OUT -name "*.csv" | head -50
