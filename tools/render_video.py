#!/usr/bin/env python3
"""Offline video: the Recorder's playback path with whole frames sharded over the GPUs of a node.

  python tools/render_video.py --poses 16 --spp 64 --out shots                       (one GPU)
  python -m torch.distributed.run --nproc-per-node 8 tools/render_video.py --poses 600 --spp 256 --out shots

Pose k is rendered by rank k % N: `spp` progressive frames in ONE launch (the running mean of a pixel stays in
registers), then the bloom chain + final composite, then the 8-bit image is gathered to rank 0 (RCCL), which
writes shots/raytrace-out-<k>.png like Recorder::WriteTexture (Include/Recorder.h:93-103). There is no
communication while tracing. Pose k always uses frames k*spp .. (k+1)*spp-1 of the reference's seed stream and
starts from a cleared display, so the images do not depend on the number of GPUs.

Deviations from the interactive Recorder, on purpose: poses switch after `spp` frames instead of 5 s of wall
clock; bloom runs once per image instead of once per frame.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import raytracing_amd as ra  # noqa: E402
from raytracing_amd import controls, scenes, sharding  # noqa: E402


def camera_path(n):
    """A smooth fly-through of the sponza-like atrium: position on a spline, looking ahead."""
    t = np.linspace(0.0, 1.0, n)
    pos = np.stack([-2.0 + 20.0 * t, 2.2 * np.sin(2 * np.pi * t), 0.4 + 1.6 * np.sin(np.pi * t) ** 2], axis=1)
    ahead = np.stack([np.full(n, 1.0), 0.6 * np.cos(2 * np.pi * t), -0.15 + 0.1 * np.cos(np.pi * t)], axis=1)
    return pos.astype(np.float32), ahead.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=16, help="number of camera poses (ignored with --pose-file)")
    ap.add_argument("--pose-file", help="one pose per line: px py pz fx fy fz (what HeadlessApp/Recorder saves)")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--scene", default="sponza_like")
    ap.add_argument("--tris", type=int, default=260_000)
    ap.add_argument("--out", default="shots")
    ap.add_argument("--no-write", action="store_true", help="render and gather but do not encode PNGs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    device = local if backend == "nccl" else local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device)
    if world > 1:
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", device)} if backend == "nccl" else {}))

    import tempfile
    d = os.path.join(tempfile.gettempdir(), "rt_video_%s_%d" % (args.scene, args.tris))
    if not os.path.exists(os.path.join(d, "done")):
        if rank == 0:
            scenes.write_objects(scenes.SCENES[args.scene](target_tris=args.tris), d)
            open(os.path.join(d, "done"), "w").close()
    if world > 1:
        dist.barrier()
    paths = sorted(os.path.join(d, f) for f in os.listdir(d) if f != "done")
    sc = ra.load_scene(paths, build_threads=min(16, os.cpu_count() or 1))

    if args.pose_file:
        rows = np.loadtxt(args.pose_file, dtype=np.float32).reshape(-1, 6)
        pos, fwd = rows[:, :3], rows[:, 3:]
    else:
        pos, fwd = camera_path(args.poses)
    n = len(pos)
    W, H, spp = args.width, args.height, args.spp
    seeds = controls.rand_seed_triples(n * spp).reshape(n, spp, 3)

    r = ra.Renderer(W, H, device=device)
    r.set_scene_data(sc)
    r.set_noise(scenes.noise_texture())
    r.set_skybox(scenes.sky_cubemap())
    r.set_bounce_limit(args.bounces)
    cam = ra.Camera()
    if rank == 0 and not args.no_write:
        os.makedirs(args.out, exist_ok=True)

    mine = sharding.poses_of_rank(n, rank, world)
    rounds = (n + world - 1) // world
    frame = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda" if backend == "nccl" else "cpu")
    bufs = [torch.empty_like(frame) for _ in range(world)] if rank == 0 else None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for j in range(rounds):
        k = j * world + rank
        if k < n:
            cam.set_pose(pos[k], fwd[k])                 # SetPosition + SetFacing (normalises), Recorder.h:50-51
            c = cam.get()
            r.set_camera(c["position"], c["facing"], c["fov"], c["viewport_width"], c["viewport_distance"])
            r.clear()                                    # ResetFrameIndex: the first frame's weight is 1
            r.display_reset()
            r.render(0, seeds[k])
            _, rgb8 = r.postprocess(True)
            frame.copy_(torch.from_numpy(rgb8))
        if world > 1:
            dist.gather(frame, bufs, dst=0)              # finished 8-bit frames only
        if rank == 0 and not args.no_write:
            for q in range(world):
                kk = j * world + q
                if kk < n:
                    img = (bufs[q] if world > 1 else frame).cpu().numpy()
                    controls.write_png(os.path.join(args.out, "raytrace-out-%d.png" % kk), img)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    rays = torch.tensor([float(r.counters()["rays"])], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(rays)
    if rank == 0:
        print(json.dumps({"poses": n, "spp": spp, "resolution": [W, H], "bounce_limit": args.bounces, "n_gpus": world,
                          "seconds": round(dt, 3), "frames_per_s": round(n / dt, 3), "mrays_per_s": round(float(rays[0]) / dt / 1e6, 1),
                          "written_to": None if args.no_write else args.out}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
