#!/usr/bin/env python3
"""How the time of a node visit depends on how much of the scene the caches hold: the benchmark scene generated at
several triangle counts, 3840x2160, bounce limit 8, a few one-frame launches each. Prints ms per frame, rays per frame,
boxes per ray, nanoseconds per (ray x box pair), and wave cycles per interior step from the counting replay."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import raytracing_amd as ra
from raytracing_amd import scenes
W, H, BL = 3840, 2160, 8
for tris in [int(a) for a in sys.argv[1:]] or [2_000, 20_000, 80_000, 260_000]:
    d = tempfile.mkdtemp(prefix="rt_probe_")
    sc = ra.load_scene(scenes.write_objects(scenes.sponza_like(target_tris=tris), d), build_threads=16)
    r = ra.Renderer(W, H)
    bench.setup(r, sc, scenes.noise_texture(), scenes.sky_cubemap(), BL)
    seeds = scenes.frame_seeds(12)
    for k in range(4):
        r.render(k, seeds[k:k + 1])
    r.sync(); r.kernel_time(reset=True); r.counters(reset=True)
    for k in range(4, 8):
        r.render(k, seeds[k:k + 1])
    r.sync()
    ms, n = r.kernel_time(reset=True)
    rays = r.counters(reset=True)["rays"] / 4
    r.set_counting(True)
    for k in range(8, 10):
        r.render(k, seeds[k:k + 1])
    r.sync()
    c = r.counters(reset=True)
    boxes = c["boxes_tested"] / max(1, c["rays"])
    cyc = [64.0 * c["sched_cycles"][i] / max(1, c["sched_rounds"][i]) for i in range(4)]
    use = [c["sched_lanes"][i] / max(1, 64 * c["sched_rounds"][i]) for i in range(4)]
    print("tris %7d nodes %7d: %.3f ms/frame, %.1f Mrays/frame, %.2f Grays/s, %.1f boxes/ray, %.3f ns per ray-visit | counting build: cycles per interior/leaf/feed round %.0f %.0f %.0f, lane use %.2f %.2f %.2f"
          % (sc.n_triangles, len(sc.node8) // 8, ms / n, rays / 1e6, rays / (ms / n) / 1e6, boxes, 1e6 * (ms / n) / (rays * boxes / 2), cyc[1], cyc[2], cyc[3], use[1], use[2], use[3]), flush=True)
    r.close()
