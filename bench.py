#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

Metric (BASELINE.json): Mrays/s (whole node), Sponza 3840x2160, 8 bounces; ms/frame at 1 spp.
A "step" is one progressive frame (1 sample per pixel) of the whole image. A ray is one BVH
traversal (primary or bounce segment), counted exactly on the device.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: image tiles (8x8 pixels) are dealt round-robin over the ranks, every rank holds the whole
scene, and each step ends with an RCCL gather of the ranks' accumulation shards to rank 0
(framebuffer only). Total work is fixed as N grows -> "scaling": "strong".

The scene is a deterministic synthetic stand-in (the reference's Sponza.off is a Git-LFS stub),
written to and loaded from the reference's own .off/.txt formats.
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import raytracing_amd as ra  # noqa: E402
from raytracing_amd import scenes  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, 8 TB/s
# Ceilings of the units this kernel actually loads (the scene is cache-resident, so HBM is not one of them):
SIMDS, CUS = 1024, 256
ISSUE_CYCLES_PER_INSTRUCTION = 2.0    # a SIMD issues a wave64 vector instruction over 2 cycles (32 lanes per cycle): MI355X_MICROARCH.md, hardware model
L1_ACCESSES_PER_CYCLE_PER_CU = 1.48   # 16-byte lane accesses of a gather the vector L1 serves, measured: tools/microbench/l1_gather.hip
L2_PEAK_GBS = 34500.0                 # aggregate L2 bandwidth of the 8 XCDs
L2_BYTES_PER_REQUEST = 128            # a vector-L1 line
MULTI_SPP_FRAMES, MULTI_SPP_LAUNCHES = 16, 4   # N > 1 only: the extra measurement with several frames per launch
CARRIED_TIME_TOLERANCE = 0.03         # carried counters describe this run only if its kernel time is within 3 % of the profiled one

WORKLOADS = {
    # name: (scene generator, kwargs, width, height, bounce limit)
    "sponza_like_3840x2160_b8": ("sponza_like", {"target_tris": 260_000}, 3840, 2160, 8),
    "teapot_like_1920x1080_b8": ("teapot_like", {"target_tris": 140_000}, 1920, 1080, 8),
    "dragon_like_1920x1080_b16": ("dragon_like", {"target_tris": 870_000}, 1920, 1080, 16),
    "lightbox_640x360_b4": ("lightbox", {}, 640, 360, 4),
}


def algorithmic_bytes(c):
    """Algorithmic bytes moved by the work the counters describe: 32 B per distinct box tested,
    48 B per triangle tested, 4 B material index per hit, 12 B per noise fetch (4 RGB8 texels),
    16 B per sky fetch (4 RGBA8 texels), 32 B framebuffer traffic per pixel sample."""
    return (32 * c["boxes_tested"] + 48 * c["tris_tested"] + 4 * c["hits"] + 12 * c["noise_fetches"] +
            16 * c["sky_fetches"] + 32 * c["paths"])


def load_carried(workload):
    """Hardware counters of the dominant kernel, per launch, from the committed rocprofv3 profile of this same command
    (profiles/counters.json, written by tools/profile_bench.sh + tools/summarize_prof.py). bench.py cannot collect
    them itself (they need the profiler around the process), so they are carried, and every use is marked as such."""
    cf = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(cf):
        return None
    with open(cf) as f:
        c = json.load(f)
    if c.get("workload") != workload or not c.get("kernel_ms_when_profiled"):
        return None
    return c


def binding_roofs(c, avg_ms):
    """The ceilings that can bound the kernel, each as achieved / peak with the peak's source; numerators from the
    carried counters `c`, denominator this run's own kernel time; the shader clock is the profile's cycles over the
    profile's time. The carried numbers only describe this run if the kernel ran as long as when it was profiled: the
    check is part of the result, and RT_BENCH_STRICT=1 (set by tools/profile_bench.sh) turns a failed check into an error."""
    clock_hz = c["shader_cycles_per_launch"] / (c["kernel_ms_when_profiled"] * 1e-3)
    sec = avg_ms * 1e-3
    tag = c.get("profile")
    drift = avg_ms / c["kernel_ms_when_profiled"] - 1.0
    ok = abs(drift) <= CARRIED_TIME_TOLERANCE
    check = "%s: kernel %.4f ms in this run, %.4f ms when profiled (%+.1f %%, tolerance %.0f %%)" % (
        "ok" if ok else "FAILED", avg_ms, c["kernel_ms_when_profiled"], 100 * drift, 100 * CARRIED_TIME_TOLERANCE)
    if not ok:
        print("bench.py: carried counters of profile %s do not describe this run: %s" % (tag, check), file=sys.stderr)
        if os.environ.get("RT_BENCH_STRICT") == "1":
            raise SystemExit("carried counters are stale; re-run tools/profile_bench.sh")
    roofs = []
    issue_peak = SIMDS * clock_hz / ISSUE_CYCLES_PER_INSTRUCTION / 1e9
    issue = c["instructions_per_launch"] / sec / 1e9
    roofs.append({"name": "instruction issue", "achieved": round(issue, 1), "peak": round(issue_peak, 1), "unit": "G wave-instructions/s",
                  "frac": round(issue / issue_peak, 4), "carried_from": tag,
                  "formula": "instructions_per_launch / kernel time, over 1024 SIMDs x clock / 2 cycles per wave64 instruction",
                  "peak_source": "1024 SIMDs x %.2f GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md: 32 lanes per cycle); "
                                 "the clock is the profile's shader cycles over its kernel time" % (clock_hz / 1e9)})
    if c.get("ta_busy_cycles_per_launch"):
        ta = c["ta_busy_cycles_per_launch"] / sec / 1e9
        ta_peak = CUS * clock_hz / 1e9
        roofs.append({"name": "texture-address units busy", "achieved": round(ta, 1), "peak": round(ta_peak, 1), "unit": "G busy cycles/s",
                      "frac": round(ta / ta_peak, 4), "carried_from": tag,
                      "formula": "TA_TA_BUSY_sum per launch / kernel time, over 256 units x clock",
                      "peak_source": "one texture-address unit per CU, busy every cycle"})
    l1_peak = CUS * clock_hz * L1_ACCESSES_PER_CYCLE_PER_CU / 1e9
    l1 = c["vector_l1_accesses_per_launch"] / sec / 1e9
    roofs.append({"name": "vector L1 gather rate", "achieved": round(l1, 1), "peak": round(l1_peak, 1), "unit": "G lane accesses/s",
                  "frac": round(l1 / l1_peak, 4), "carried_from": tag,
                  "formula": "TCP_TOTAL_CACHE_ACCESSES_sum per launch / kernel time, over 256 CUs x clock x 1.48",
                  "peak_source": "256 CUs x %.2f GHz x 1.48 accesses per cycle and CU reached by a 16-byte gather (tools/microbench/l1_gather.hip)" % (clock_hz / 1e9)})
    if c.get("l2_requests_per_launch"):
        l2 = c["l2_requests_per_launch"] * L2_BYTES_PER_REQUEST / sec / 1e9
        roofs.append({"name": "L2", "achieved": round(l2, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(l2 / L2_PEAK_GBS, 4), "carried_from": tag,
                      "formula": "TCC_REQ_sum x 128 B per launch / kernel time",
                      "peak_source": "aggregate L2 bandwidth, 34.5 TB/s; bytes = TCC_REQ x 128 (a vector-L1 line; uncalibrated, an upper bound)"})
    binding = max(roofs, key=lambda r: r["frac"])
    valu_share = ISSUE_CYCLES_PER_INSTRUCTION * c["valu_instructions_per_launch"] / (SIMDS * clock_hz * sec) if c.get("valu_instructions_per_launch") else None
    lane = c.get("valu_lane_utilisation")
    return {"binding": binding, "roofs": roofs, "carried_ok": ok,
            "valu_lane_frac": round(valu_share * lane, 4) if valu_share and lane else None,
            "valu_issue_share": round(valu_share, 4) if valu_share else None,
            "carried": dict({k: c.get(k) for k in ("instructions_per_launch", "valu_instructions_per_launch", "shader_cycles_per_launch",
                                                   "vector_l1_accesses_per_launch", "ta_busy_cycles_per_launch", "l2_requests_per_launch",
                                                   "l2_hit_rate", "vector_l1_hit_rate", "valu_lane_utilisation", "mean_waves_per_simd",
                                                   "wave_time_waiting", "ta_busy", "kernel_ms_when_profiled")},
                            carried_from=tag, source=c.get("source"), kernel_ms_this_run=round(avg_ms, 4), check=check)}


def llvmpipe_recorded(workload):
    """The reference shader on Mesa llvmpipe for this workload's scene, as recorded in the build container (the shader
    text cannot travel to the GPU box, so it cannot be timed in this run)."""
    lf = os.path.join(ROOT, "profiles", "llvmpipe_recorded.json")
    if not os.path.exists(lf):
        return None
    with open(lf) as f:
        r = json.load(f)
    if r.get("workload") != workload:
        return None
    return {"value": r["llvmpipe_mrays_per_s"], "unit": "Mrays/s", "cores": r["host_cores"], "LP_NUM_THREADS": r["LP_NUM_THREADS"],
            "renderer": r["renderer"], "where": r["where"],
            "sample": "same scene, camera and bounce limit at %dx%d, %d frames after the JIT frame (%.0f ms/frame); the CPU restatement on the same "
                      "cores and sample: %.2f Mrays/s, images bit-identical: %s" % (r["resolution"][0], r["resolution"][1], r["frames_timed"],
                                                                                   r["llvmpipe_ms_per_frame"], r["cpu_restatement_mrays_per_s"], r["images_bit_identical"])}


def load_workload(name, threads):
    gen, kw, W, H, bl = WORKLOADS[name]
    cache = os.path.join(tempfile.gettempdir(), "rt_bench_" + name)
    paths_file = os.path.join(cache, "paths.json")
    if not os.path.exists(paths_file):
        paths = scenes.write_objects(scenes.SCENES[gen](**kw), cache + ".tmp%d" % os.getpid())
        os.makedirs(cache, exist_ok=True)
        final = []
        for p in paths:
            q = os.path.join(cache, os.path.basename(p))
            os.replace(p, q)
            final.append(q)
        with open(paths_file + ".tmp%d" % os.getpid(), "w") as f:
            json.dump(final, f)
        os.replace(f.name, paths_file)       # atomically: another process may be looking for it right now
    with open(paths_file) as f:
        paths = json.load(f)
    # parsed triangles and the built tree are kept next to the files, keyed by the files' bytes: later runs skip both steps
    return ra.load_scene(paths, build_threads=threads, cache_dir=os.path.join(cache, "scene_cache")), W, H, bl


def setup(r, sc, noise, sky, bl):
    r.set_scene_data(sc)
    fov = np.float32(120 * 0.01745329251994329576923690768489)
    r.set_camera((0, 0, 0), (1, 0, 0), float(fov), float(np.float32(2.0 * np.tan(float(fov) / 2.0))))
    r.set_noise(noise)
    r.set_skybox(sky)
    r.set_bounce_limit(bl)


def cpu_baseline(sc, noise, sky, bl, W, H):
    """The CPU restatement (oracle) timed on this host's cores on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from rt_oracle import Oracle
    cores = min(16, os.cpu_count() or 1)
    w, h, frames = W // 2, H // 2, 12      # about 10-15 s of work for 16 threads on the 4K workload
    o = Oracle(w, h, threads=cores)
    o.set_scene(sc.tri12, sc.matidx, sc.node8)
    o.set_materials(sc.materials[:16])
    fov = np.float32(120 * 0.01745329251994329576923690768489)
    o.set_camera((0, 0, 0), (1, 0, 0), float(fov), float(np.float32(2.0 * np.tan(float(fov) / 2.0))))
    o.set_noise(noise)
    o.set_skybox(sky)
    o.set_bounce_limit(bl)
    seeds = scenes.frame_seeds(frames)
    o.counters(reset=True)
    t0 = time.perf_counter()
    o.render(0, seeds)
    dt = time.perf_counter() - t0
    rays = o.counters()["rays"]
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "same scene, camera and bounce limit at %dx%d, %d frames (%d rays, %.1f s)" % (w, h, frames, rays, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="sponza_like_3840x2160_b8", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-pass", action="store_true", help="skip the instrumented replay that measures algorithmic bytes")
    ap.add_argument("--verify", action="store_true", help="N > 1: check the assembled image bit for bit against an unsharded render of the same frames")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # RT_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: ranks share devices and
    # the gather goes through host memory. The real thing is nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    # RT_BENCH_FORCE_DIST=1 runs the exchange path (gather, assembly, reductions) even with a single rank, so that a
    # one-GPU box can rehearse it with the real backend: python -m torch.distributed.run --nproc-per-node 1 bench.py
    sharded = world > 1 or (os.environ.get("RT_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if sharded:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    K, Wm = args.steps, args.warmup
    # The first use of a workload writes its scene files to a cache directory: rank 0 goes first, the others then
    # find the cache complete.
    if sharded and rank != 0:
        dist.barrier()
    sc, W, H, bl = load_workload(args.workload, threads=min(16, os.cpu_count() or 1))
    if sharded and rank == 0:
        dist.barrier()
    noise, sky = scenes.noise_texture(), scenes.sky_cubemap()
    seeds = scenes.frame_seeds(Wm + K)

    r = ra.Renderer(W, H, device=device)
    setup(r, sc, noise, sky, bl)
    root = None
    stage = flats = gathered = None
    pending = [None, None]
    if sharded:
        r.set_shard(rank, world)
        n = r.shard_tiles() * 64 * 4
        # torch tensors are only plumbing: device memory for the gather's send and receive sides.
        # Two of each, so that the gather of frame k runs while frame k+1 is being traced.
        stage = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(2)]
        if rank == 0:
            flats = [torch.empty(world * n, dtype=torch.float32, device="cuda") for _ in range(2)]
            gathered = [list(f.chunk(world)) for f in flats]     # the gather writes the shards back to back
            root = ra.Renderer(W, H, device=device)              # unsharded context that assembles the image

    # With nccl the whole exchange is ordered on the device: the renderers run on torch streams, so that torch events can
    # order them against the collective's stream, and the host never waits inside the loop (it only queues).
    #   tracer stream  rs : render k, [wait: gather k-2 done] export shard into stage[k&1], event `exported`
    #   torch stream      : wait exported, [rank 0: wait assembly k-2], gather k (its own stream, after this one), event `gathered`
    #   assembly stream as: (rank 0) wait gathered, scatter flats[k&1] into the image, event `assembled`
    on_device = sharded and backend == "nccl"
    rs = as_ = None
    gathered_ev = [None, None]
    assembled_ev = [None, None]
    if on_device:
        rs = torch.cuda.Stream()
        r.set_stream(rs.cuda_stream)
        if root is not None:
            as_ = torch.cuda.Stream()
            root.set_stream(as_.cuda_stream)

    def finish(k):
        """(host-staged rehearsal only) Completes the exchange of frame k: rank 0 scatters the shards into the image."""
        b = k & 1
        if pending[b] is None:
            return
        if rank == 0:
            root.assemble(flats[b].data_ptr(), world)
        pending[b] = None

    def step(k):
        r.render(k, seeds[k:k + 1])
        if not sharded:
            return
        b = k & 1
        if on_device:
            if gathered_ev[b] is not None:
                rs.wait_event(gathered_ev[b])        # stage[b] was the send buffer of gather k-2
            r.export_shard(stage[b].data_ptr())      # behind the frame on the tracer's stream; frame k+1 follows at once
            exported = torch.cuda.Event()
            exported.record(rs)
            cur = torch.cuda.current_stream()
            cur.wait_event(exported)
            if assembled_ev[b] is not None:
                cur.wait_event(assembled_ev[b])      # flats[b] was read by the assembly of frame k-2
            work = dist.gather(stage[b], gathered[b] if rank == 0 else None, dst=0, async_op=True)
            work.wait()                              # the current stream (not the host) waits for the collective
            gathered_ev[b] = torch.cuda.Event()
            gathered_ev[b].record(cur)
            if rank == 0:
                as_.wait_event(gathered_ev[b])
                root.assemble(flats[b].data_ptr(), world)
                assembled_ev[b] = torch.cuda.Event()
                assembled_ev[b].record(as_)
            return
        r.export_shard(stage[b].data_ptr())
        finish(k - 1)
        r.sync()
        if root is not None:
            root.sync()
        host = stage[b].cpu()
        parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
        dist.gather(host, parts, dst=0)
        if rank == 0:
            flats[b].copy_(torch.cat(parts))
        pending[b] = True

    def fence():
        finish(0)
        finish(1)
        r.sync()
        if root is not None:
            root.sync()
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()

    for k in range(Wm):
        step(k)
    fence()
    saved = r.read_accum() if (not sharded and not args.no_roofline_pass) else None
    saved_shard = None
    if sharded and not args.no_roofline_pass:
        # this rank's part of the image as the timed frames will find it, for the counting replay after them
        saved_shard = torch.empty(r.shard_tiles() * 64 * 4, dtype=torch.float32, device="cuda")
        r.export_shard(saved_shard.data_ptr())
        r.sync()
    r.counters(reset=True)
    r.kernel_time(reset=True)
    fence()
    t0 = time.perf_counter()
    for k in range(Wm, Wm + K):
        step(k)
    fence()
    dt = time.perf_counter() - t0

    if args.verify and sharded and rank == 0:
        whole = ra.Renderer(W, H, device=device)
        setup(whole, sc, noise, sky, bl)
        for k in range(Wm + K):
            whole.render(k, seeds[k:k + 1])
        a, b = root.read_accum(), whole.read_accum()
        same = bool((a.view(np.uint32) == b.view(np.uint32)).all())
        print("verify: image assembled from %d shards %s the unsharded render" % (world, "equals" if same else "DIFFERS FROM"), file=sys.stderr)
        whole.close()
        if not same:
            raise SystemExit("sharded and unsharded images differ")

    c = r.counters(reset=True)
    kms, launches = r.kernel_time(reset=True)
    stats = torch.tensor([dt, float(c["rays"]), float(c["paths"]), kms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if sharded:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, rays, paths, kms_max = float(tmax[0]), float(tsum[1]), float(tsum[2]), float(tmax[3])
    else:
        rays, paths, kms_max = float(c["rays"]), float(c["paths"]), kms

    # N > 1: the same exchange with 16 frames per launch and one gather per launch. A launch costs its rays plus an end phase
    # that depends on the length of single paths, not on their number (about 0.8 ms on this scene), and 1/N shards are small;
    # several frames per launch amortise that phase, which is how the offline Recorder path renders. Reported beside the
    # one-frame-per-step headline, never instead of it.
    multi_spp = None
    if sharded and MULTI_SPP_FRAMES > 0:
        spp, launches_m = MULTI_SPP_FRAMES, MULTI_SPP_LAUNCHES
        seeds_m = scenes.frame_seeds(Wm + K + spp * (launches_m + 1))

        def step_m(j):
            first = Wm + K + j * spp
            r.render(first, seeds_m[first:first + spp])
            if on_device:
                b = j & 1
                if gathered_ev[b] is not None:
                    rs.wait_event(gathered_ev[b])
                r.export_shard(stage[b].data_ptr())
                exported = torch.cuda.Event()
                exported.record(rs)
                cur = torch.cuda.current_stream()
                cur.wait_event(exported)
                if assembled_ev[b] is not None:
                    cur.wait_event(assembled_ev[b])
                work = dist.gather(stage[b], gathered[b] if rank == 0 else None, dst=0, async_op=True)
                work.wait()
                gathered_ev[b] = torch.cuda.Event()
                gathered_ev[b].record(cur)
                if rank == 0:
                    as_.wait_event(gathered_ev[b])
                    root.assemble(flats[b].data_ptr(), world)
                    assembled_ev[b] = torch.cuda.Event()
                    assembled_ev[b].record(as_)
            else:
                b = j & 1
                r.export_shard(stage[b].data_ptr())
                finish(j - 1)
                r.sync()
                if root is not None:
                    root.sync()
                host = stage[b].cpu()
                parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, parts, dst=0)
                if rank == 0:
                    flats[b].copy_(torch.cat(parts))
                pending[b] = True

        step_m(0)                      # warm-up launch
        fence()
        r.counters(reset=True)
        t0m = time.perf_counter()
        for j in range(1, launches_m + 1):
            step_m(j)
        fence()
        dtm = time.perf_counter() - t0m
        cm = r.counters(reset=True)
        sm = torch.tensor([dtm, float(cm["rays"])], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        smax, ssum = sm.clone(), sm.clone()
        dist.all_reduce(smax, op=dist.ReduceOp.MAX)
        dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
        multi_spp = {"frames_per_launch": spp, "launches": launches_m, "gathers": launches_m,
                     "value": round(float(ssum[1]) / float(smax[0]) / 1e6, 2), "unit": "Mrays/s",
                     "ms_per_frame": round(float(smax[0]) / (launches_m * spp) * 1e3, 4),
                     "note": "same tiles, ranks and gather; %d frames per launch, one gather per launch (the image rank 0 holds is %d frames behind "
                             "at most); frames of a pixel stay in one lane, results are bit-identical to one frame per launch" % (spp, spp)}

    roofline = None
    if sharded and saved_shard is not None:
        # Counting replay of exactly the timed frames, every rank on its own tiles from the state saved above; the
        # algorithmic bytes of the node are the sum over ranks, priced against the ranks' longest kernel time per step
        # and against N times the HBM peak.
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        r.sync()
        if hip.hipMemcpy(r.shard_device_ptr(), saved_shard.data_ptr(), saved_shard.numel() * 4, 3) != 0:
            raise SystemExit("restoring the shard for the counting replay failed")
        r.set_counting(True)
        r.counters(reset=True)
        for k in range(Wm, Wm + K):
            r.render(k, seeds[k:k + 1])
        r.sync()
        ci = r.counters(reset=True)
        r.set_counting(False)
        assert ci["rays"] == c["rays"], "instrumented replay traced a different number of rays"
        nb = torch.tensor([float(algorithmic_bytes(ci)), float(ci["boxes_tested"]), float(ci["tris_tested"])], dtype=torch.float64,
                          device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(nb, op=dist.ReduceOp.SUM)
        per_step = float(nb[0]) / K
        step_ms = kms_max / K
        achieved = per_step / (step_ms * 1e-3) / 1e9
        hbm_algorithmic = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                           "frac": round(achieved / (HBM_PEAK_GBS * world), 5), "is_a_bound": False,
                           "note": "NOT A BOUND: algorithmic bytes of all ranks per step over the slowest rank's kernel time per step, against %d x the HBM "
                                   "peak; the bytes are re-reads served by the caches" % world,
                           "algorithmic_bytes_per_step": int(per_step),
                           "bytes_per_ray": round(float(nb[0]) / rays, 1), "boxes_per_ray": round(float(nb[1]) / rays, 2), "tris_per_ray": round(float(nb[2]) / rays, 2)}
        carried = load_carried(args.workload)
        if carried and carried.get("rays_per_launch"):
            # The kernel is the one-GPU kernel on a 1/N share of the tiles, so its instructions per ray carry over from the one-GPU
            # profile; priced against N GPUs' issue rate over the slowest rank's kernel time per step.
            clock_hz = carried["shader_cycles_per_launch"] / (carried["kernel_ms_when_profiled"] * 1e-3)
            per_ray = carried["instructions_per_launch"] / carried["rays_per_launch"]
            issue = per_ray * (rays / K) / (step_ms * 1e-3) / 1e9
            peak = world * SIMDS * clock_hz / ISSUE_CYCLES_PER_INSTRUCTION / 1e9
            roofline = {"bound": "instruction issue", "achieved": round(issue, 1), "peak": round(peak, 1), "unit": "G wave-instructions/s",
                        "frac": round(issue / peak, 4), "traffic": None, "kernel": "trace_kernel", "kernel_ms_per_step_slowest_rank": round(step_ms, 4),
                        "carried_from": carried.get("profile"),
                        "formula": "instructions per ray of the one-GPU profile x rays per step of all ranks / slowest rank's kernel time per step, "
                                   "over N x 1024 SIMDs x clock / 2 cycles",
                        "note": "estimate: instructions per ray (%.1f) are carried from the one-GPU profile; small shards spend more of a launch "
                                "draining, which shows as a lower fraction" % per_ray,
                        "hbm_algorithmic": hbm_algorithmic}
        else:
            roofline = {"bound": "hbm", "achieved": hbm_algorithmic["achieved"], "peak": hbm_algorithmic["peak"], "unit": "GB/s", "frac": hbm_algorithmic["frac"],
                        "traffic": None, "kernel": "trace_kernel", "kernel_ms_per_step_slowest_rank": round(step_ms, 4),
                        "note": "no committed one-GPU profile with instructions per ray for this workload (profiles/counters.json): only the algorithmic "
                                "HBM figure, which is not a bound", "hbm_algorithmic": hbm_algorithmic}
    if not sharded and not args.no_roofline_pass:
        # instrumented replay of exactly the timed frames from the same accumulation state
        r.write_accum(saved)
        r.set_counting(True)
        r.counters(reset=True)
        for k in range(Wm, Wm + K):
            r.render(k, seeds[k:k + 1])
        ci = r.counters(reset=True)
        r.set_counting(False)
        assert ci["rays"] == c["rays"], "instrumented replay traced a different number of rays"
        per_launch = algorithmic_bytes(ci) / launches
        avg_ms = kms / launches
        achieved = per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            with open(tf) as f:
                t = json.load(f)
            if t.get("workload") == args.workload:
                traffic = t.get("hbm_bytes_per_launch")
        hbm_algorithmic = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                           "is_a_bound": False,
                           "note": "NOT A BOUND: algorithmic bytes (what the reference shader fetches: 32 B per box test, 48 B per triangle test, ...) over "
                                   "the kernel time. The scene is cache-resident and these bytes are re-reads served by LDS/L1/L2/Infinity Cache, so the "
                                   "figure exceeds the HBM peak; 'traffic' beside it is what was measured on the memory side of the L2",
                           "algorithmic_bytes_per_launch": int(per_launch),
                           "bytes_per_ray": round(algorithmic_bytes(ci) / ci["rays"], 1),
                           "boxes_per_ray": round(ci["boxes_tested"] / ci["rays"], 2),
                           "tris_per_ray": round(ci["tris_tested"] / ci["rays"], 2)}
        scheduler = {"lane_use_new_interior_leaf_shade": [round(l / (64.0 * max(1, n)), 3) for l, n in zip(ci["sched_lanes"], ci["sched_rounds"])],
                     "rounds_new_interior_leaf_shade": [n // launches for n in ci["sched_rounds"]],
                     "wave_cycles_per_round_new_interior_leaf_shade": [round(64.0 * c / max(1, n)) for c, n in zip(ci["sched_cycles"], ci["sched_rounds"])],
                     "wave_time_share_new_interior_leaf_shade": [round(c / max(1, sum(ci["sched_cycles"])), 3) for c in ci["sched_cycles"]],
                     "source": "counting variant of the same kernel replaying the timed frames from the same accumulation state (measured in this run)"}
        carried = load_carried(args.workload)
        if carried:
            br = binding_roofs(carried, avg_ms)
            b = br["binding"]
            roofline = {"bound": b["name"], "achieved": b["achieved"], "peak": b["peak"], "unit": b["unit"], "frac": b["frac"], "traffic": traffic,
                        "kernel": "trace_kernel", "kernel_avg_ms": round(avg_ms, 4), "formula": b["formula"], "peak_source": b["peak_source"],
                        "carried_from": b["carried_from"], "carried_ok": br["carried_ok"],
                        "note": "the roof with the highest measured load of those under 'roofs' (none of them is HBM: see 'hbm_algorithmic'); "
                                "numerators are rocprofv3 counters of the committed profile named in 'carried_from', divided by this run's kernel time",
                        "valu_lane_frac": br["valu_lane_frac"], "valu_issue_share": br["valu_issue_share"],
                        "roofs": br["roofs"], "hbm_algorithmic": hbm_algorithmic, "scheduler": scheduler, "carried": br["carried"]}
        else:
            roofline = {"bound": "hbm", "achieved": hbm_algorithmic["achieved"], "peak": hbm_algorithmic["peak"], "unit": "GB/s", "frac": hbm_algorithmic["frac"],
                        "traffic": traffic, "kernel": "trace_kernel", "kernel_avg_ms": round(avg_ms, 4),
                        "note": "no committed profile for this workload (profiles/counters.json): only the algorithmic HBM figure, which is not a bound",
                        "hbm_algorithmic": hbm_algorithmic, "scheduler": scheduler}

    if rank == 0:
        out = {
            "metric": "Mrays/s (whole node) Sponza 3840x2160 8 bounces; ms/frame at 1 spp",
            "value": round(rays / dt / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene_triangles": sc.n_triangles, "bvh_nodes": sc.n_nodes,
                       "resolution": [W, H], "bounce_limit": bl, "spp_per_step": 1,
                       "parallelism": "tiles%d" % world if sharded else "single",
                       "rays_per_step": round(rays / K), "mpaths_per_s": round(paths / dt / 1e6, 2),
                       "trace_kernel_ms_per_step": round(kms_max / K, 4)},
        }
        if roofline:
            out["roofline"] = roofline
        if multi_spp:
            out["multi_spp"] = multi_spp
        if not args.no_cpu_baseline and not sharded:
            out["cpu_baseline"] = cpu_baseline(sc, noise, sky, bl, W, H)
        rec = llvmpipe_recorded(args.workload)
        if rec and not sharded:
            out["llvmpipe_recorded"] = rec
        print(json.dumps(out), flush=True)
    if on_device:
        r.set_stream(None)          # back to the contexts' own streams before torch's go away
        if root is not None:
            root.set_stream(None)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
