#!/usr/bin/env python3
"""Where the time of a node visit goes (library built with -DRT_PROBE_STEP, counting mode): mean wave-cycles from the record
request to its data and for the whole visit, on the benchmark workload. Run it with RT_LIB_PATH pointing at the probe build and
once more at an ordinary build, whose sky / noise / hits tallies are subtracted (the probe adds to them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracing_amd as ra
from raytracing_amd import scenes
sc, W, H, bl = bench.load_workload("sponza_like_3840x2160_b8", 16)
r = ra.Renderer(W, H)
bench.setup(r, sc, scenes.noise_texture(), scenes.sky_cubemap(), bl)
seeds = scenes.frame_seeds(8)
for k in range(4):
    r.render(k, seeds[k:k + 1])
r.sync(); r.counters(reset=True)
r.set_counting(True)
for k in range(4, 6):
    r.render(k, seeds[k:k + 1])
r.sync()
c = r.counters(reset=True)
print("PROBE sky %d noise %d hits %d rounds_interior %d cycles_interior_x64 %d" % (c["sky_fetches"], c["noise_fetches"], c["hits"], c["sched_rounds"][1], c["sched_cycles"][1]))
