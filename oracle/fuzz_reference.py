#!/usr/bin/env python3
"""Randomised live comparison of the CPU restatement with the reference's tracer shader on llvmpipe: random scene
generator and size, image size, camera pose, field of view, bounce limit (the two debug views included), frame count and position in the seed stream;
each case is one run of oracle/compare_reference.py. Build container only. Test infrastructure, not part of the product.

  python3 oracle/fuzz_reference.py --cases 200 --seed 1 [--log /tmp/fuzz_reference.log] [--ray-counts]

--ray-counts: every case also draws the shader's compile-time RAY_COUNT from 2..8 (Fragment.glsl:5; set in memory when the
shader file is loaded) and halves the image's sides.
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=50)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--log", default="")
ap.add_argument("--ray-counts", action="store_true")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
rng_rays = np.random.default_rng(args.seed + 7919)      # (a stream of its own: the cases stay what they are without the option)
log = open(args.log, "a") if args.log else sys.stdout
bad = 0
t0 = time.time()
for i in range(args.cases):
    scene = ["lightbox", "sponza_like", "dragon_like", "teapot_like"][int(rng.integers(0, 4))]
    tris = 0 if scene == "lightbox" else int(rng.integers(500, 60000))
    w, h = int(rng.integers(64, 700)), int(rng.integers(48, 420))
    pos = rng.uniform(-2.5, 2.5, 3)
    facing = rng.standard_normal(3)
    facing[2] *= 0.5
    cmd = [sys.executable, os.path.join(HERE, "compare_reference.py"), "--scene", scene, "--tris", str(tris), "--width", str(w), "--height", str(h),
           "--bounces", str([0, 1, 2, 4, 8, 16, 32][int(rng.integers(0, 7))]), "--frames", str(int(rng.integers(2, 9))),
           "--fov", str([120, 90][int(rng.integers(0, 2))]), "--skip-seeds", str(int(rng.integers(0, 500))),
           "--camera=" + ",".join("%.4f" % v for v in list(pos) + list(facing))]
    if args.ray_counts:
        cmd[cmd.index("--width") + 1], cmd[cmd.index("--height") + 1] = str(max(32, w // 2)), str(max(24, h // 2))
        cmd += ["--ray-count", str(int(rng_rays.integers(2, 9)))]
    run = subprocess.run(cmd, capture_output=True, text=True)
    ok = run.returncode == 0 and "triangle ids differing: 0; hit distances differing: 0" in run.stdout
    bad += not ok
    print("%s case %d (%.0f s): %s" % ("ok " if ok else "BAD", i, time.time() - t0, " ".join(cmd[2:])), file=log, flush=True)
    if not ok:
        print(run.stdout[-3000:] + run.stderr[-2000:], file=log, flush=True)
print("%d cases, %d differing" % (args.cases, bad), file=log, flush=True)
sys.exit(1 if bad else 0)
