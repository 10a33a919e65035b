#!/usr/bin/env python3
"""Randomised check of the bloom chain + final composite: GPU (through the C ABI) against the CPU oracle.
  python3 tests/fuzz_post.py [cases] [seed]
Random image sizes (from a few pixels to about 700x400), several consecutive calls per context so that the content
the chain leaves in its work image is carried over, bloom on and off mixed. Tolerance 2e-5 absolute on the float image
(two different powf implementations), one code on the 8-bit image. Test infrastructure: the oracle is only the checker."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import raytracing_amd as ra
from rt_oracle import PostOracle

TOL = 2e-5
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def hdr(w, h):
    img = (rng.random((h, w, 4)) * rng.uniform(0.2, 1.2)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(int(rng.integers(0, 10))):
        cx, cy, rad = rng.integers(0, w), rng.integers(0, h), rng.integers(1, max(2, w // 6))
        img[((xx - cx) ** 2 + (yy - cy) ** 2) < rad * rad, :3] = rng.uniform(0.5, 30.0, 3).astype(np.float32)
    img[..., 3] = 0
    return img


bad = 0
t0 = time.time()
for case in range(cases):
    small = rng.random() < 0.3
    w, h = (int(rng.integers(1, 45)), int(rng.integers(1, 45))) if small else (int(rng.integers(45, 700)), int(rng.integers(45, 400)))
    g, po = ra.Renderer(w, h), PostOracle(w, h)
    worst, worst8, calls = 0.0, 0, int(rng.integers(1, 5))
    for f in range(calls):
        a = hdr(w, h)
        g.write_accum(a)
        if rng.random() < 0.2:                          # the raw path leaves the chain's work image alone
            raw, raw8 = g.postprocess(False)
            worst = max(worst, float(np.abs(raw[..., :3] - a[..., :3]).max()))
            continue
        rgba, rgb8 = g.postprocess(True)
        o_rgba, o_rgb8 = po.run(a)
        worst = max(worst, float(np.abs(rgba[..., :3] - o_rgba[..., :3]).max()))
        worst8 = max(worst8, int(np.abs(rgb8.astype(int) - o_rgb8.astype(int)).max()))
    g.close()
    ok = worst <= TOL and worst8 <= 1
    print("case %2d: %3dx%-3d calls %d  max |diff| %.2e  8-bit %d  %s" % (case, w, h, calls, worst, worst8, "ok" if ok else "DIFFERS"), flush=True)
    bad += not ok
print("%d cases, %d with differences, %.0f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
