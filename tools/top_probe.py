#!/usr/bin/env python3
"""Prints the device counters of a few counted frames of the benchmark workload (library chosen by RT_LIB_PATH).
Run it once with the ordinary library and once with a probe build (tools/build_variant.sh probe -DRT_PROBE_TOP=1,
RT_LIB_PATH=build_variants/probe.so): the differences of sky_fetches, noise_fetches, hits, paths, tris_tested and
node_pops are the node visits that went to the first 32, 64, 128, 256, 1024 and 4096 records."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import raytracing_amd as ra
from raytracing_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_like_3840x2160_b8"
sc, W, H, bl = bench.load_workload(name, threads=16)
r = ra.Renderer(W, H)
bench.setup(r, sc, scenes.noise_texture(), scenes.sky_cubemap(), bl)
seeds = scenes.frame_seeds(4)
r.set_counting(True)
r.counters(reset=True)
for k in range(4):
    r.render(k, seeds[k:k + 1])
c = r.counters()
print(json.dumps({k: v for k, v in c.items() if not isinstance(v, list)}))
