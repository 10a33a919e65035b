#!/usr/bin/env python3
"""Times the two CPU paths on the plumbing configuration (light box, 640x360, 4 bounces, 16 frames), in THIS container:
the reference shader on Mesa llvmpipe (oracle/_ref/libglref.so; needs /root/reference) and the CPU restatement.
Also checks that the two images are bit-identical. Test infrastructure, not part of the product.

  python3 oracle/time_cpu_paths.py            (LP_NUM_THREADS is honoured by Mesa; default = all cores)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(HERE, "glsl_ref"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from glref import GLRef  # noqa: E402
from reference_renderer import ReferenceRenderer  # noqa: E402
from rt_oracle import Oracle  # noqa: E402
from raytracing_amd import scenes  # noqa: E402
from util import DEFAULT_CAMERA, differing_pixels, make_scene  # noqa: E402

W, H, BL, FRAMES = 640, 360, 4, 16
sc = make_scene("lightbox")
noise, sky, seeds = scenes.noise_texture(), scenes.sky_cubemap(), scenes.frame_seeds(FRAMES + 1)
threads = os.cpu_count() or 1


def setup(r):
    r.set_scene(sc.tri12, sc.matidx, sc.node8)
    r.set_materials(sc.materials)
    r.set_camera(**DEFAULT_CAMERA)
    r.set_noise(noise)
    r.set_skybox(sky)
    r.set_bounce_limit(BL)


ref = ReferenceRenderer(W, H, gl=GLRef())
setup(ref)
ref.render(0, seeds[:1])                 # first frame: LLVM JIT of the shader
t0 = time.perf_counter()
ref.render(1, seeds[1:])
t_ref = (time.perf_counter() - t0) / FRAMES
a = ref.read_accum()

o = Oracle(W, H, threads=threads)
setup(o)
o.render(0, seeds[:1])
o.counters(reset=True)
t0 = time.perf_counter()
o.render(1, seeds[1:])
t_orc = (time.perf_counter() - t0) / FRAMES
rays = o.counters()["rays"]
b = o.read_accum()
print("light box %dx%d, bounce limit %d, %d frames after one warm-up frame, %d host threads (LP_NUM_THREADS=%s)" % (
    W, H, BL, FRAMES, threads, os.environ.get("LP_NUM_THREADS", "unset")))
print("llvmpipe (reference shader): %.2f ms/frame  %.2f Mrays/s" % (t_ref * 1e3, rays / FRAMES / t_ref / 1e6))
print("CPU restatement (oracle):    %.2f ms/frame  %.2f Mrays/s" % (t_orc * 1e3, rays / FRAMES / t_orc / 1e6))
print("pixels differing between the two after %d frames: %d" % (FRAMES + 1, differing_pixels(a, b)))
