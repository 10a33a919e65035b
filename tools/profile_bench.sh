#!/bin/bash
# Profiles bench.py on the GPU box: one kernel-trace pass, then PMC passes (each in its own run).
# Usage (through gpurun): bash tools/profile_bench.sh <tag> [extra bench args]
set -o pipefail
TAG=${1:-r01}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
# bench.py with its default step counts: the counters then describe the launches the bench line times
# (the carried-counter check of bench.py is an error here: RT_BENCH_STRICT is set for the final confirming run below)
ARGS="--no-cpu-baseline --no-roofline-pass $@"
python3 bench.py $ARGS > $OUT/plain.json 2> $OUT/plain.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || exit 1
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "FETCH_SIZE TCC_HIT_sum" \
            "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
            "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" \
            "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum SQ_INSTS_SMEM SQ_IFETCH SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES" ; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc$i -- python3 bench.py $ARGS > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.log; exit 1; }
done
find $OUT -name "*.csv" | head -50
# condense, then confirm that a plain run reproduces its roofline from the counters just written
python3 tools/summarize_prof.py $TAG "$@" > $OUT/summary.log 2>&1 || { tail -5 $OUT/summary.log; exit 1; }
RT_BENCH_STRICT=1 python3 bench.py "$@" > $OUT/bench_line.json 2> $OUT/bench_line.err || { echo "confirming bench run failed"; tail -5 $OUT/bench_line.err; exit 1; }
cp $OUT/bench_line.json profiles/bench_$TAG.json
# gpurun only brings gpurun_out/ back: leave copies of what was written under profiles/ there (copy them to profiles/ at home)
mkdir -p $OUT/profiles && cp profiles/${TAG}_summary.txt profiles/counters.json profiles/hbm_traffic.json profiles/bench_$TAG.json $OUT/profiles/
tail -3 $OUT/summary.log; cut -c1-600 $OUT/bench_line.json
