"""Shared helpers for the tests: scene construction and renderer set-up on any back end
(GPU Renderer, CPU Oracle — same method names)."""
import math
import os
import tempfile

import numpy as np

import raytracing_amd as ra
from raytracing_amd import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FOV = np.float32(120 * 0.01745329251994329576923690768489)          # Camera.cpp:14
VPW = float(np.float32(2.0 * 1.0 * math.tan(float(FOV) / 2.0)))      # Camera.cpp:19
DEFAULT_CAMERA = dict(position=(0.0, 0.0, 0.0), facing=(1.0, 0.0, 0.0), fov=float(FOV), viewport_width=VPW)

_cache = {}


def make_scene(name, **kw):
    """Generates a synthetic scene, writes it in the reference's file formats, loads it back
    through the host model (loaders + BVH + flatten)."""
    key = (name, tuple(sorted(kw.items())))
    if key not in _cache:
        d = tempfile.mkdtemp(prefix="rt_scene_")
        paths = scenes.write_objects(scenes.SCENES[name](**kw), d)
        _cache[key] = ra.load_scene(paths, build_threads=4)
    return _cache[key]


def setup(r, scene, noise, sky, camera=None, bounce_limit=4):
    r.set_scene(scene.tri12, scene.matidx, scene.node8)
    r.set_materials(scene.materials[:16])
    r.set_camera(**(camera or DEFAULT_CAMERA))
    r.set_noise(noise)
    r.set_skybox(sky)
    r.set_bounce_limit(bounce_limit)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def differing_pixels(a, b):
    """Number of pixels whose RGB differs in any bit."""
    return int((bits(a[..., :3]) != bits(b[..., :3])).any(-1).sum())
