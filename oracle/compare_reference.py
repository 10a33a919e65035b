#!/usr/bin/env python3
"""Live comparison, in the build container, of the CPU restatement with the reference's unmodified tracer shader on Mesa
llvmpipe at sizes too large to keep as fixtures: primary-hit ids and distances, then the accumulated radiance after
every frame, bit for bit. Needs /root/reference. Test infrastructure, not part of the product.

  LP_NUM_THREADS=8 python3 oracle/compare_reference.py --scene sponza_like --tris 260000 --width 960 --height 540 --bounces 8 --frames 5
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(HERE, "glsl_ref"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from glref import GLRef  # noqa: E402
from reference_renderer import ReferenceRenderer  # noqa: E402
from rt_oracle import Oracle  # noqa: E402
from raytracing_amd import scenes  # noqa: E402
from util import DEFAULT_CAMERA, bits, make_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sponza_like")
ap.add_argument("--tris", type=int, default=260000)
ap.add_argument("--width", type=int, default=960)
ap.add_argument("--height", type=int, default=540)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--camera", default="", help="px,py,pz,fx,fy,fz: position and facing (normalised here); default = the reference's start pose")
ap.add_argument("--fov", type=float, default=120.0, help="field of view in degrees (the reference uses 120, and 90 with the C key)")
ap.add_argument("--ray-count", type=int, default=1, help="the shader's compile-time RAY_COUNT (Fragment.glsl:5): rays per pixel and frame")
ap.add_argument("--skip-seeds", type=int, default=0, help="start this many draws into the seed stream")
args = ap.parse_args()
W, H = args.width, args.height
sc = make_scene(args.scene, **({"target_tris": args.tris} if args.tris else {}))
noise, sky, seeds = scenes.noise_texture(), scenes.sky_cubemap(), scenes.frame_seeds(args.frames + args.skip_seeds)[args.skip_seeds:]
camera = dict(DEFAULT_CAMERA)
if args.camera:
    c = np.array([float(x) for x in args.camera.split(",")], np.float32)
    facing = c[3:6] / np.float32(np.sqrt((c[3:6].astype(np.float64) ** 2).sum()))
    camera.update(position=tuple(float(x) for x in c[:3]), facing=tuple(float(x) for x in facing))
fov = np.float32(args.fov * 0.01745329251994329576923690768489)
camera.update(fov=float(fov), viewport_width=float(np.float32(2.0 * np.tan(float(fov) / 2.0))))


def setup(r):
    r.set_scene(sc.tri12, sc.matidx, sc.node8)
    r.set_materials(sc.materials)
    r.set_camera(**camera)
    r.set_noise(noise)
    r.set_skybox(sky)
    r.set_bounce_limit(args.bounces)


gl = GLRef()
print("%s, %d triangles, %dx%d, bounce limit %d; %s" % (args.scene, sc.n_triangles, W, H, args.bounces, gl.renderer()))
ids = ReferenceRenderer(W, H, ids=True, gl=gl)
setup(ids)
ids.set_bounce_limit(1)
ids.render(0, seeds[:1])
ia = ids.read_accum()
o = Oracle(W, H, threads=os.cpu_count() or 1)
setup(o)
ib = o.primary_ids(seeds[0])
hit = ia[..., 2] > 0
print("primary hits: %d pixels; triangle ids differing: %d; hit distances differing: %d" % (
    int(hit.sum()), int((ia[..., 0].astype(np.int32)[hit] != ib[..., 0][hit]).sum()), int((bits(ia[..., 1])[hit] != ib[..., 1].view(np.uint32)[hit]).sum())))
ref = ReferenceRenderer(W, H, gl=gl, ray_count=args.ray_count)
setup(ref)
o.set_ray_count(args.ray_count)
ref.clear()
o.clear()
worst = 0
for k in range(args.frames):
    ref.render(k, seeds[k:k + 1])
    o.render(k, seeds[k:k + 1])
    a, b = ref.read_accum(), o.read_accum()
    d = (bits(a[..., :3]) != bits(b[..., :3])).any(-1)
    worst = max(worst, int(d.sum()))
    ys, xs = np.nonzero(d)
    print("after frame %d: %d of %d pixels differ%s" % (k, int(d.sum()), W * H, "".join(
        "\n   (%d,%d) reference %s restatement %s" % (x, y, a[y, x, :3].tolist(), b[y, x, :3].tolist()) for y, x in list(zip(ys, xs))[:4])))
sys.exit(1 if worst else 0)
