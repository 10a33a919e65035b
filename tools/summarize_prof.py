#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag> directory (tools/profile_bench.sh) into profiles/<tag>_summary.txt."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", "prof_" + tag)
lines = ["# rocprofv3 summary for tag %s (command: python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline-pass %s)" % (tag, " ".join(sys.argv[2:])), ""]
lines.append("## bench line of the un-profiled run")
lines.append(open(os.path.join(src, "plain.json")).read().strip())
lines.append("")
lines.append("## rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    lines.append(open(f).read().strip())
lines.append("")
lines.append("## PMC passes (separate runs, one per counter group): per-dispatch mean over the trace_kernel dispatches")
for f in sorted(glob.glob(os.path.join(src, "pmc*", "*", "*_counter_collection.csv"))):
    rows = [r for r in csv.DictReader(open(f)) if "trace_kernel" in r["Kernel_Name"]]
    if not rows:
        continue
    n = len(set(r["Dispatch_Id"] for r in rows))
    agg = collections.defaultdict(float)
    for r in rows:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
    r0 = rows[0]
    lines.append("%s: %d dispatches, VGPR_Count %s SGPR_Count %s LDS %s grid %s wg %s" % (
        os.path.basename(os.path.dirname(os.path.dirname(f))), n, r0.get("VGPR_Count"), r0.get("SGPR_Count"),
        r0.get("LDS_Block_Size"), r0.get("Grid_Size"), r0.get("Workgroup_Size")))
    for k, v in sorted(agg.items()):
        lines.append("    %-34s %.6g" % (k, v / n))
v = {}
for f in sorted(glob.glob(os.path.join(src, "pmc*", "*", "*_counter_collection.csv"))):
    rows = [r for r in csv.DictReader(open(f)) if "trace_kernel" in r["Kernel_Name"]]
    n = max(1, len(set(r["Dispatch_Id"] for r in rows)))
    here = collections.defaultdict(float)
    for r in rows:
        here[r["Counter_Name"]] += float(r["Counter_Value"]) / n
    for k, val in here.items():
        v.setdefault(k, val)         # a counter collected in several passes is taken from the first one
lines.append("")
lines.append("## derived")
if "SQ_THREAD_CYCLES_VALU" in v:
    lines.append("VALU lane utilisation  = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU*64) = %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)))
    lines.append("wave time waiting      = SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f" % (v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
if "GRBM_GUI_ACTIVE" in v and "SQ_WAVE_CYCLES" in v:
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    lines.append("mean waves per SIMD    = 4*SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs) = %.2f" % (4 * v["SQ_WAVE_CYCLES"] / (cyc * 1024)))
    lines.append("VALU issue share       = 2*SQ_INSTS_VALU / (cycles * 1024 SIMDs) = %.3f" % (2 * v["SQ_INSTS_VALU"] / (cyc * 1024)))
if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
    lines.append("L2 hit rate            = TCC_HIT / (TCC_HIT+TCC_MISS) = %.3f" % (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
if "TCP_TCC_READ_REQ_sum" in v:
    lines.append("vector L1 hit rate     = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES = %.3f" % (1 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"]))
if "FETCH_SIZE" in v:
    lines.append("FETCH_SIZE             = %.4g KB per dispatch (fabric-side reads, Infinity-Cache hits included; x2 for wide coalesced reads on gfx950)" % v["FETCH_SIZE"])
if "WRITE_SIZE" in v:
    lines.append("WRITE_SIZE             = %.4g KB per dispatch" % v["WRITE_SIZE"])
os.makedirs("profiles", exist_ok=True)
if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
    # rocprofv3 reports both in KB. On gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so reads of
    # 16 bytes per lane are doubled; WRITE_SIZE is exact. Both count requests on the memory side of the L2, Infinity
    # Cache hits included, so the sum is an upper bound on what reached HBM.
    bench = json.loads(open(os.path.join(src, "plain.json")).read().strip().splitlines()[-1])
    fetch = v["FETCH_SIZE"] * 1024.0 * 2.0
    write = v["WRITE_SIZE"] * 1024.0
    traffic = {"workload": bench["config"]["workload"], "hbm_bytes_per_launch": int(fetch + write),
               "fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(write), "profile": tag,
               "note": "FETCH_SIZE*1024*2 + WRITE_SIZE*1024 per trace_kernel dispatch; memory-side L2 requests, Infinity-Cache hits included"}
    json.dump(traffic, open(os.path.join("profiles", "hbm_traffic.json"), "w"), indent=1)
    lines.append("memory-side traffic    = %.3f GB per dispatch (2*FETCH_SIZE + WRITE_SIZE, KB -> bytes)" % ((fetch + write) / 1e9))
out = os.path.join("profiles", tag + "_summary.txt")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-12:]))
