#!/bin/bash
# Instruction and memory-path counters of bench.py for the two schedulers of the same library (GPU box).
# Usage: bash tools/pmc_sched.sh [pool] [wave]
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in "$@"; do
  for pass in 1 2; do
    out=gpurun_out/pmcs_${v}_$pass
    if [ $pass = 1 ]; then CTRS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE";
    else CTRS="TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"; fi
    RT_SCHEDULER=$v timeout -s KILL 200 rocprofv3 --pmc $CTRS --output-format csv -d $out -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-pass > $out.log 2>&1 || { echo "pmc run failed for $v"; tail -3 $out.log; exit 1; }
  done
  python3 - "$v" <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
a = collections.defaultdict(float)
for p in (1, 2):
    f = glob.glob("gpurun_out/pmcs_%s_%d/*/*_counter_collection.csv" % (v, p))[0]
    rows = [r for r in csv.DictReader(open(f)) if "trace_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]]
    ids = sorted(set(int(r["Dispatch_Id"]) for r in rows))[5:]
    for r in rows:
        if int(r["Dispatch_Id"]) in ids: a[r["Counter_Name"]] += float(r["Counter_Value"]) / len(ids) / (2 if r["Counter_Name"] == "GRBM_GUI_ACTIVE" else 1)
cyc = a["GRBM_GUI_ACTIVE"] / 8
tot = sum(a[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))
print("%-5s cycles %.3e | VALU %.3e SALU %.3e BRANCH %.3e LDS %.3e VMEM_RD %.3e VMEM_WR %.3e FLAT %.3e total %.3e issue %.3f | wait %.3f lanes %.3f | L1 acc %.3e (%.2f/cyc/CU) L1->L2 %.3e TA busy %.3f LDS conflict %.3e" % (
    v, cyc, a["SQ_INSTS_VALU"], a["SQ_INSTS_SALU"], a["SQ_INSTS_BRANCH"], a["SQ_INSTS_LDS"], a["SQ_INSTS_VMEM_RD"], a["SQ_INSTS_VMEM_WR"], a["SQ_INSTS_FLAT"], tot,
    tot * 2.2 / 1024 / cyc, a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_THREAD_CYCLES_VALU"] / (64 * a["SQ_ACTIVE_INST_VALU"]),
    a["TCP_TOTAL_CACHE_ACCESSES_sum"], a["TCP_TOTAL_CACHE_ACCESSES_sum"] / cyc / 256, a["TCP_TCC_READ_REQ_sum"], a["TA_TA_BUSY_sum"] / cyc / 256, a["SQ_LDS_BANK_CONFLICT"]))
PY
done
