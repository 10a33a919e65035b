#!/bin/bash
# Vector-memory counters of bench.py for library variants in build_variants/ (GPU box). Usage: bash tools/pmc_ta.sh name...
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in "$@"; do
  out=gpurun_out/pmcta_$v
  RT_LIB_PATH=$PWD/build_variants/$v.so timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY \
      --output-format csv -d $out -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline-pass > $out.log 2>&1 || { echo "pmc run failed for $v"; tail -3 $out.log; exit 1; }
  python3 - "$v" $out <<'PY'
import csv, glob, sys, collections
v, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "trace_kernel" in r["Kernel_Name"]]
n = len(set(r["Dispatch_Id"] for r in rows))
a = collections.defaultdict(float)
for r in rows: a[r["Counter_Name"]] += float(r["Counter_Value"]) / n
cu_cycles = a["GRBM_GUI_ACTIVE"] / 8 * 256
print("%-8s cycles %.3e  TA busy %.3f  TA stalled by TC %.3f  L1 accesses %.3e (%.2f /cycle/CU)  L1->L2 reads %.3e  VMEM_RD %.3e  LDS insts %.3e  LDS active %.3e  LDS bank conflict %.3e  wait %.3f" % (
    v, a["GRBM_GUI_ACTIVE"], a["TA_TA_BUSY_sum"] / cu_cycles, a["TA_ADDR_STALLED_BY_TC_CYCLES_sum"] / cu_cycles, a["TCP_TOTAL_CACHE_ACCESSES_sum"],
    a["TCP_TOTAL_CACHE_ACCESSES_sum"] / cu_cycles, a["TCP_TCC_READ_REQ_sum"], a["SQ_INSTS_VMEM_RD"], a["SQ_INSTS_LDS"], a["SQ_ACTIVE_INST_LDS"], a["SQ_LDS_BANK_CONFLICT"],
    a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"]))
PY
done
