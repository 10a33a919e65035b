#!/usr/bin/env python3
"""Times launches of 1, 2, 4, 8, 16 frames on the benchmark scene (how much of a 1-spp launch is tail)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracing_amd as ra
from raytracing_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_like_3840x2160_b8"
sc, W, H, bl = bench.load_workload(name, 16)
r = ra.Renderer(W, H)
bench.setup(r, sc, scenes.noise_texture(), scenes.sky_cubemap(), bl)
seeds = scenes.frame_seeds(64)
r.render(0, seeds[:2]); r.sync()
for n in (1, 2, 4, 8, 16):
    r.clear(); r.counters(reset=True); r.kernel_time(reset=True)
    r.render(0, seeds[:n]); r.sync()
    ms, _ = r.kernel_time(reset=True); rays = r.counters()["rays"]
    print("%-28s frames/launch %2d: %8.3f ms  %7.3f ms/frame  %7.1f Mrays/s" % (name, n, ms, ms / n, rays / ms / 1e3))
