#!/usr/bin/env python3
"""The one measurement the reference's tree still holds, done the same way: frames per second while the camera turns
once around the vertical axis in 1280 steps (fpsBenchmark.csv: `yaw, fps` per line; its generator is gone, only the
members it used are left in Include/Scene.h:55-61). Every step moves the camera, so every frame starts a new
accumulation: one tracer launch at frame index 0 plus the bloom chain, nothing read back. The reference's window is
1280x720 and its start-up bounce limit 4 (src/main.cpp:28-29, src/Scene.cpp:19).
  python3 tools/yaw_benchmark.py [workload] [out.csv]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import raytracing_amd as ra
from raytracing_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_like_3840x2160_b8"
out = sys.argv[2] if len(sys.argv) > 2 else None
sc, _, _, _ = bench.load_workload(name, 16)
W, H, BL, STEPS = 1280, 720, 4, 1280
r = ra.Renderer(W, H)
bench.setup(r, sc, scenes.noise_texture(), scenes.sky_cubemap(), BL)
seeds = scenes.frame_seeds(STEPS)
fov = float(np.float32(120 * 0.01745329251994329576923690768489))
vpw = float(np.float32(2.0 * np.tan(fov / 2.0)))
L, h = r.L, r.h
rows = []
for warm in (True, False):
    for k in range(64 if warm else STEPS):
        yaw = k * (2 * math.pi / STEPS)
        t0 = time.perf_counter()
        r.set_camera((0.0, 0.0, 0.0), (math.cos(yaw), math.sin(yaw), 0.0), fov, vpw)
        r.clear()
        r.render(0, seeds[k:k + 1])
        L.rt_postprocess(h, 1, None, None)          # returns when the frame is on the (virtual) screen
        dt = time.perf_counter() - t0
        if not warm:
            rows.append((yaw, 1.0 / dt))
fps = np.array([f for _, f in rows])
print("%s, %dx%d, bounce limit %d, %d yaw steps: fps min %.0f  median %.0f  max %.0f  (mean frame %.3f ms)" % (
    name, W, H, BL, STEPS, fps.min(), np.median(fps), fps.max(), 1e3 * float(np.mean(1.0 / fps))))
if out:
    with open(out, "w") as f:
        for yaw, v in rows:
            f.write("%g, %d\n" % (yaw, round(v)))
