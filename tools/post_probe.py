#!/usr/bin/env python3
"""Times the bloom chain (rt_postprocess without read-back) next to the tracer on the benchmark frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracing_amd as ra
from raytracing_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_like_3840x2160_b8"
sc, W, H, bl = bench.load_workload(name, 16)
r = ra.Renderer(W, H)
bench.setup(r, sc, scenes.noise_texture(), scenes.sky_cubemap(), bl)
seeds = scenes.frame_seeds(14)
L, h = r.L, r.h
for k in range(2):
    r.render(k, seeds[k:k + 1]); L.rt_postprocess(h, 1, None, None)
r.sync()
t0 = time.perf_counter()
for k in range(2, 12):
    r.render(k, seeds[k:k + 1])
r.sync()
t1 = time.perf_counter()
for k in range(10):
    L.rt_postprocess(h, 1, None, None)
t2 = time.perf_counter()
print("%s: trace %.3f ms/frame, bloom chain %.3f ms/frame (%dx%d)" % (name, (t1 - t0) * 100, (t2 - t1) * 100, W, H))
