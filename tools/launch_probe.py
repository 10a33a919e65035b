#!/usr/bin/env python3
"""Kernel time of consecutive one-frame launches, one line per launch (which launches pay for the tile re-ordering)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracing_amd as ra
from raytracing_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "teapot_like_1920x1080_b8"
sc, W, H, bl = bench.load_workload(name, 16)
r = ra.Renderer(W, H)
bench.setup(r, sc, scenes.noise_texture(), scenes.sky_cubemap(), bl)
seeds = scenes.frame_seeds(26)
out = []
for k in range(26):
    r.kernel_time(reset=True)
    r.render(k, seeds[k:k + 1]); r.sync()
    ms, _ = r.kernel_time(reset=True)
    out.append("%d:%.3f" % (k + 1, ms))
print(name, "launch:ms ", " ".join(out))
