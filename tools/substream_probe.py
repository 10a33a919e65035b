#!/usr/bin/env python3
"""Does splitting a rank's tiles over several contexts (own HIP streams, launches queued ahead) hide the drain at the end
of each launch? Times `frames` frames of 1/WORLD of the benchmark image rendered as SUB sub-shards on one GPU.
Usage: python3 tools/substream_probe.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracing_amd as ra
from raytracing_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_like_3840x2160_b8"
sc, W, H, bl = bench.load_workload(name, 16)
noise, sky = scenes.noise_texture(), scenes.sky_cubemap()
FR = 40
seeds = scenes.frame_seeds(FR + 8)
for world in (1, 8):
    for sub in (1, 2, 3, 4, 8):
        rs = []
        for j in range(sub):
            r = ra.Renderer(W, H)
            bench.setup(r, sc, noise, sky, bl)
            r.set_shard(j, world * sub)          # sub-shard j of this rank's share (rank 0 of `world`)
            rs.append(r)
        for k in range(8):
            for r in rs:
                r.render(k, seeds[k:k + 1])
        for r in rs:
            r.sync(); r.counters(reset=True)
        t0 = time.perf_counter()
        for k in range(8, 8 + FR):
            for r in rs:
                r.render(k, seeds[k:k + 1])
        for r in rs:
            r.sync()
        dt = (time.perf_counter() - t0) / FR * 1e3
        rays = sum(r.counters()["rays"] for r in rs)
        print("1/%d of the image as %d sub-shard(s): %7.3f ms per frame, %7.1f Mrays/s" % (world, sub, dt, rays / FR / dt / 1e3), flush=True)
        for r in rs:
            r.close()
