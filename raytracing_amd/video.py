"""Offline video: the Recorder's playback path (Include/Recorder.h:36-55,93-103) with whole frames dealt over GPUs.

    python -m raytracing_amd.video --poses 16 --spp 64 --out shots                               (one GPU)
    python -m torch.distributed.run --nproc-per-node 8 -m raytracing_amd.video --poses 600 --spp 256 --out shots

Pose k is rendered by rank k % N: `spp` progressive frames in ONE launch (a pixel's running mean stays in
registers), one run of the bloom chain, 8-bit read-back, and the rank writes <out>/raytrace-out-<k>.png itself on a
writer thread while it traces its next pose. Nothing is exchanged between ranks: the ranks of a node share the file
system, so no rank waits for another one's encoder; torch.distributed only adds up the statistics at the end. Pose k
always uses frames k*spp .. (k+1)*spp-1 of the reference's seed stream and starts from a cleared display, so an image
does not depend on the number of ranks, and poses whose file already exists are skipped: running an interrupted job
again resumes it. The work itself is librt_mi355x.so's rth_video_render (C++); this module is its driver.

Deviations from the interactive Recorder, on purpose: a pose is shown for `spp` frames instead of 5 s of wall clock;
the bloom chain runs once per image instead of once per displayed frame; image k shows pose k (the reference writes
the image before it moves the camera, so its image k shows pose k - 1).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

from . import _lib
from ._lib import rth_video_stats

BLOOM, FLIP, RESUME, WRITE = 1, 2, 4, 8


def camera_path(n):
    """A smooth fly-through of the sponza-like atrium: position on a spline, looking ahead."""
    t = np.linspace(0.0, 1.0, n)
    pos = np.stack([-2.0 + 20.0 * t, 2.2 * np.sin(2 * np.pi * t), 0.4 + 1.6 * np.sin(np.pi * t) ** 2], axis=1)
    ahead = np.stack([np.full(n, 1.0), 0.6 * np.cos(2 * np.pi * t), -0.15 + 0.1 * np.cos(np.pi * t)], axis=1)
    return np.concatenate([pos, ahead], axis=1).astype(np.float32)


def read_poses(path):
    """(n, 6) float32 from a pose file: px py pz fx fy fz per line (what HeadlessApp.save_poses writes)."""
    L = _lib.lib()
    n = ctypes.c_uint32()
    if L.rth_read_poses(path.encode(), None, 0, ctypes.byref(n)) != 0:
        raise IOError("cannot read " + path)
    out = np.zeros((n.value, 6), dtype=np.float32)
    if n.value:
        L.rth_read_poses(path.encode(), out.ctypes.data, n.value, ctypes.byref(n))
    return out


def pose_seeds(first_pose, n_poses, spp):
    """(n_poses, spp, 3) u_RandSeed triples: pose k uses frames k*spp .. (k+1)*spp-1 of the reference's rand() stream."""
    out = np.zeros((n_poses, spp, 3), dtype=np.float32)
    if out.size:
        _lib.lib().rth_video_seeds(first_pose, n_poses, spp, out.ctypes.data)
    return out


def image_path(out_dir, pose):
    return os.path.join(out_dir, "raytrace-out-%d.png" % pose)


def render_poses(renderer, poses6, spp, out_dir, rank=0, world=1, bloom=True, flip=False, resume=True, write=True, compression=1,
                 writer_threads=2):
    """Renders this rank's share of `poses6` on `renderer` (scene, textures and bounce limit already set)."""
    poses6 = np.ascontiguousarray(poses6, dtype=np.float32).reshape(-1, 6)
    st = rth_video_stats()
    flags = (BLOOM if bloom else 0) | (FLIP if flip else 0) | (RESUME if resume else 0) | (WRITE if write else 0)
    rc = renderer.L.rth_video_render(renderer.h, renderer.W, renderer.H, poses6.ctypes.data, len(poses6), int(spp), int(rank), int(world),
                                     out_dir.encode(), flags, int(compression), int(writer_threads), ctypes.byref(st))
    renderer._ck(rc, "rth_video_render")
    return {f: getattr(st, f) for f, _ in rth_video_stats._fields_}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--poses", type=int, default=16, help="number of camera poses on the built-in path (ignored with --pose-file)")
    ap.add_argument("--pose-file", help="one pose per line: px py pz fx fy fz (what HeadlessApp/Recorder saves)")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--ray-count", type=int, default=1, help="rays per pixel and frame: the shader's compile-time RAY_COUNT (Fragment.glsl:5)")
    ap.add_argument("--scene", default="sponza_like")
    ap.add_argument("--tris", type=int, default=260_000)
    ap.add_argument("--out", default="shots")
    ap.add_argument("--no-write", action="store_true", help="render and read back but do not encode PNGs")
    ap.add_argument("--no-resume", action="store_true", help="render poses again whose image already exists")
    ap.add_argument("--flip", action="store_true", help="top row first (the reference's PNGs are upside down)")
    ap.add_argument("--compression", type=int, default=1)
    ap.add_argument("--writers", type=int, default=2)
    args = ap.parse_args(argv)

    import tempfile
    import raytracing_amd as ra
    from raytracing_amd import scenes

    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    dist = None
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    import torch
    device = local if backend == "nccl" else local % max(1, torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device)
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", device)} if backend == "nccl" else {}))

    d = os.path.join(tempfile.gettempdir(), "rt_video_%s_%d" % (args.scene, args.tris))
    if rank == 0 and not os.path.exists(os.path.join(d, "done")):
        scenes.write_objects(scenes.SCENES[args.scene](target_tris=args.tris), d)
        open(os.path.join(d, "done"), "w").close()
    if dist:
        dist.barrier()
    paths = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".txt", ".off")))
    sc = ra.load_scene(paths, build_threads=min(16, os.cpu_count() or 1), cache_dir=os.path.join(d, "scene_cache"))
    poses = read_poses(args.pose_file) if args.pose_file else camera_path(args.poses)

    r = ra.Renderer(args.width, args.height, device=device)
    r.set_scene_data(sc)
    r.set_noise(scenes.noise_texture())
    r.set_skybox(scenes.sky_cubemap())
    r.set_bounce_limit(args.bounces)
    r.set_ray_count(args.ray_count)
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    st = render_poses(r, poses, args.spp, args.out, rank, world, flip=args.flip, resume=not args.no_resume, write=not args.no_write,
                      compression=args.compression, writer_threads=args.writers)
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    keys = ("rendered", "skipped", "written", "rays", "trace_seconds", "encode_seconds", "stall_seconds", "wall_seconds")
    vals = torch.tensor([float(st[k]) for k in keys], dtype=torch.float64)
    vmax = vals.clone()
    if dist:
        dev = "cuda" if backend == "nccl" else "cpu"
        vals, vmax = vals.to(dev), vmax.to(dev)
        dist.all_reduce(vals)
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        vals, vmax = vals.cpu(), vmax.cpu()
    if rank == 0:
        tot = dict(zip(keys, vals.tolist()))
        print(json.dumps({"poses": len(poses), "spp": args.spp, "resolution": [args.width, args.height], "bounce_limit": args.bounces,
                          "n_gpus": world, "rendered": int(tot["rendered"]), "skipped": int(tot["skipped"]), "written": int(tot["written"]),
                          "seconds": round(dt, 3), "frames_per_s": round(tot["rendered"] / dt, 3) if dt > 0 else None,
                          "mrays_per_s": round(tot["rays"] / dt / 1e6, 1),
                          "slowest_rank": {"wall_s": round(float(vmax[7]), 3), "trace_s": round(float(vmax[4]), 3),
                                           "stalled_on_writers_s": round(float(vmax[6]), 3)},
                          "trace_share_of_wall": round(float(vmax[4]) / float(vmax[7]), 4) if float(vmax[7]) > 0 else None,
                          "encode_s_all_ranks": round(tot["encode_seconds"], 3),
                          "written_to": None if args.no_write else args.out}), flush=True)
    r.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
