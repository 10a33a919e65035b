#!/usr/bin/env python3
"""Times one rank's share of the benchmark frame on one GPU: shard 0 of WORLD (what each of WORLD GPUs would trace).
Usage: python3 tools/shard_probe.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracing_amd as ra
from raytracing_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_like_3840x2160_b8"
sc, W, H, bl = bench.load_workload(name, 16)
seeds = scenes.frame_seeds(12)
for world in (1, 2, 4, 8, 32, 128, 1024, 16384):
    r = ra.Renderer(W, H)
    bench.setup(r, sc, scenes.noise_texture(), scenes.sky_cubemap(), bl)
    r.set_shard(0, world)
    for k in range(2):
        r.render(k, seeds[k:k + 1])
    r.sync(); r.counters(reset=True); r.kernel_time(reset=True)
    t0 = time.perf_counter()
    for k in range(2, 12):
        r.render(k, seeds[k:k + 1])
    r.sync()
    wall = (time.perf_counter() - t0) * 1e3 / 10
    ms, n = r.kernel_time(reset=True); rays = r.counters()["rays"]
    print("world %d: %7.3f ms per frame on one rank  (ideal %7.3f)  %7.1f Mrays/s per rank   wall %7.3f ms per frame (+%.0f us over the kernel)" % (
        world, ms / n, ms / n if world == 1 else base / world, rays / ms / 1e3, wall, (wall - ms / n) * 1e3))
    if world == 1:
        base = ms / n
    r.close()
