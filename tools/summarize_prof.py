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
lines = ["# rocprofv3 summary for tag %s (command: python3 bench.py --no-cpu-baseline --no-roofline-pass %s)" % (tag, " ".join(sys.argv[2:])), ""]
plain = json.loads(open(os.path.join(src, "plain.json")).read().strip().splitlines()[-1])
WARMUP = int(plain.get("warmup", 0))


def timed_rows(path):
    """Counter rows of the trace_kernel dispatches the bench line times: the first `warmup` dispatches are dropped."""
    rows = [r for r in csv.DictReader(open(path)) if "trace_kernel" in r["Kernel_Name"]]
    ids = sorted(set(int(r["Dispatch_Id"]) for r in rows))
    keep = set(ids[WARMUP:]) if len(ids) > WARMUP else set(ids)
    return [r for r in rows if int(r["Dispatch_Id"]) in keep]


lines.append("## bench line of the un-profiled run")
lines.append(open(os.path.join(src, "plain.json")).read().strip())
lines.append("")
lines.append("## rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    lines.append(open(f).read().strip())
lines.append("")
lines.append("## PMC passes (separate runs, one per counter group): per-dispatch mean over the timed trace_kernel dispatches (warm-up dispatches dropped)")
for f in sorted(glob.glob(os.path.join(src, "pmc*", "*", "*_counter_collection.csv"))):
    rows = timed_rows(f)
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
    rows = timed_rows(f)
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
# The counters bench.py carries into its roofline object (per trace_kernel launch, means over the profiled dispatches).
need = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE", "TCP_TOTAL_CACHE_ACCESSES_sum")
if all(k in v for k in need):
    bench = json.loads(open(os.path.join(src, "plain.json")).read().strip().splitlines()[-1])
    insts = sum(v[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")) + v.get("SQ_INSTS_VMEM_WR", 0.0) + v.get("SQ_INSTS_SMEM", 0.0)
    cycles = v["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
    c = {"workload": bench["config"]["workload"], "profile": tag,
         "source": "rocprofv3 --pmc passes of tools/profile_bench.sh, condensed by tools/summarize_prof.py into profiles/%s_summary.txt" % tag,
         "instructions_per_launch": int(insts), "valu_instructions_per_launch": int(v["SQ_INSTS_VALU"]),
         "shader_cycles_per_launch": int(cycles),
         "vector_l1_accesses_per_launch": int(v["TCP_TOTAL_CACHE_ACCESSES_sum"]),
         "l1_to_l2_read_requests_per_launch": int(v.get("TCP_TCC_READ_REQ_sum", 0)),
         "l2_requests_per_launch": int(v.get("TCC_REQ_sum", 0)),
         "l2_hit_rate": round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v else None,
         "vector_l1_hit_rate": round(1 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"], 4) if "TCP_TCC_READ_REQ_sum" in v else None,
         "valu_lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64), 4) if "SQ_THREAD_CYCLES_VALU" in v else None,
         "mean_waves_per_simd": round(4 * v["SQ_WAVE_CYCLES"] / (cycles * 1024), 3) if "SQ_WAVE_CYCLES" in v else None,
         "wave_time_waiting": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_ANY" in v else None,
         "ta_busy": round(v["TA_TA_BUSY_sum"] / (cycles * 256), 4) if "TA_TA_BUSY_sum" in v else None,
         "ta_busy_cycles_per_launch": int(v["TA_TA_BUSY_sum"]) if "TA_TA_BUSY_sum" in v else None,
         "rays_per_launch": bench["config"].get("rays_per_step"),
         "kernel_ms_when_profiled": bench["config"].get("trace_kernel_ms_per_step")}
    json.dump(c, open(os.path.join("profiles", "counters.json"), "w"), indent=1)
    lines.append("instructions / launch  = %.4g (vector %.4g) in %.4g shader cycles: issue load %.3f of one instruction per 2 cycles and SIMD" % (
        insts, v["SQ_INSTS_VALU"], cycles, insts * 2.0 / (1024 * cycles)))
    if "TA_TA_BUSY_sum" in v:
        lines.append("texture-address busy   = TA_TA_BUSY_sum / (cycles * 256 CUs) = %.3f" % (v["TA_TA_BUSY_sum"] / (cycles * 256)))
    if "SQ_THREAD_CYCLES_VALU" in v:
        lines.append("VALU lane rate share   = VALU issue share x lane utilisation = %.3f" % (
            2 * v["SQ_INSTS_VALU"] / (cycles * 1024) * v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)))
out = os.path.join("profiles", tag + "_summary.txt")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-12:]))
