#!/usr/bin/env python3
"""How long one launch takes when the machine is all but empty: images of 1 pixel (one lane), one 8x8 tile (one wave),
8 and 64 tiles, one frame per launch. Separates the length of a single path from what a wave adds by serving 64 lanes
that are out of step with each other.   Usage: python3 tools/tail_probe.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracing_amd as ra
from raytracing_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_like_3840x2160_b8"
sc, _, _, bl = bench.load_workload(name, 16)
seeds = scenes.frame_seeds(44)
for (W, H) in ((1, 1), (2, 1), (4, 2), (8, 8), (64, 8), (64, 64), (256, 128)):
    r = ra.Renderer(W, H)
    bench.setup(r, sc, scenes.noise_texture(), scenes.sky_cubemap(), bl)
    for k in range(4):
        r.render(k, seeds[k:k + 1])
    r.sync(); r.counters(reset=True); r.kernel_time(reset=True)
    for k in range(4, 44):
        r.render(k, seeds[k:k + 1])
    r.sync()
    ms, n = r.kernel_time(reset=True); rays = r.counters()["rays"]
    print("%4dx%-4d %6d pixels: %7.1f us per launch, %5.2f rays per pixel" % (W, H, W * H, ms / n * 1e3, rays / n / (W * H)))
    r.close()
