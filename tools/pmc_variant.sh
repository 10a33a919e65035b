#!/bin/bash
# Instruction-count counters of bench.py for library variants in build_variants/ (GPU box). Usage: bash tools/pmc_variant.sh name...
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in "$@"; do
  out=gpurun_out/pmc_$v
  RT_LIB_PATH=$PWD/build_variants/$v.so timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
      --output-format csv -d $out -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline-pass > $out.log 2>&1 || { echo "pmc run failed for $v"; tail -3 $out.log; exit 1; }
  python3 - "$v" $out <<'PY'
import csv, glob, sys, collections
v, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "trace_kernel" in r["Kernel_Name"]]
n = len(set(r["Dispatch_Id"] for r in rows))
agg = collections.defaultdict(float)
for r in rows: agg[r["Counter_Name"]] += float(r["Counter_Value"]) / n
tot = sum(agg[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"))
print("%-8s VALU %.3e SALU %.3e BRANCH %.3e LDS %.3e VMEM_RD %.3e  total %.3e | wait %.3f  wait_inst %.3f" % (
    v, agg["SQ_INSTS_VALU"], agg["SQ_INSTS_SALU"], agg["SQ_INSTS_BRANCH"], agg["SQ_INSTS_LDS"], agg["SQ_INSTS_VMEM_RD"], tot,
    agg["SQ_WAIT_ANY"] / agg["SQ_WAVE_CYCLES"], agg["SQ_WAIT_INST_ANY"] / agg["SQ_WAVE_CYCLES"]))
PY
done
