"""ctypes binding of oracle/librt_oracle.so — the scalar CPU restatement of the tracer.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py, as the checker. The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")


def build():
    subprocess.check_call(["make", "-C", _HERE, "librt_oracle.so"], stdout=subprocess.DEVNULL)


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    L = ctypes.CDLL(_LIB_PATH)
    L.orc_create.restype = ctypes.c_void_p
    L.orc_create.argtypes = [ctypes.c_int, ctypes.c_int]
    for name in ("orc_destroy", "orc_accum_clear"):
        getattr(L, name).argtypes = [ctypes.c_void_p]
    L.orc_set_scene.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    L.orc_set_materials.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    L.orc_set_camera.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float]
    L.orc_set_bounce_limit.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    L.orc_set_ray_count.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.orc_set_noise.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.orc_set_skybox.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.orc_accum_write.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.orc_accum_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.orc_render.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.orc_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.orc_sample_noise.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    L.orc_sample_sky.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.orc_sample_sky_rotated.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.orc_hit_triangles.argtypes = [ctypes.c_uint32] + [ctypes.c_void_p] * 5
    return L


class Oracle:
    """Same call surface as raytracing_amd.Renderer, computed on the CPU."""

    def __init__(self, width, height, threads=None):
        self.L = _load()
        self.W, self.H = width, height
        self.h = self.L.orc_create(width, height)
        self.threads = threads or min(16, os.cpu_count() or 1)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def set_scene(self, tri12, matidx, node8):
        tri12 = np.ascontiguousarray(tri12, dtype=np.float32)
        matidx = np.ascontiguousarray(matidx, dtype=np.int32)
        node8 = np.ascontiguousarray(node8, dtype=np.float32)
        self.L.orc_set_scene(self.h, tri12.ctypes.data, matidx.ctypes.data, matidx.size, node8.ctypes.data, node8.size // 8)

    def set_materials(self, mats11):
        m = np.ascontiguousarray(mats11, dtype=np.float32).reshape(-1, 11)
        self.L.orc_set_materials(self.h, m.ctypes.data, len(m))

    def set_camera(self, position, facing, fov, viewport_width, viewport_distance=1.0):
        p = np.asarray(position, dtype=np.float32)
        f = np.asarray(facing, dtype=np.float32)
        self.L.orc_set_camera(self.h, p.ctypes.data, f.ctypes.data, fov, viewport_width, viewport_distance)

    def set_bounce_limit(self, n):
        self.L.orc_set_bounce_limit(self.h, int(n))

    def set_ray_count(self, n):
        """RAY_COUNT of the shader (Fragment.glsl:5): rays per pixel and frame."""
        self.L.orc_set_ray_count(self.h, int(n))

    def set_noise(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        self.L.orc_set_noise(self.h, rgb.ctypes.data, rgb.shape[1], rgb.shape[0])

    def set_skybox(self, faces):
        faces = np.ascontiguousarray(faces, dtype=np.uint8)
        self.L.orc_set_skybox(self.h, faces.ctypes.data, faces.shape[1])

    def clear(self):
        self.L.orc_accum_clear(self.h)

    def write_accum(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        assert rgba.size == self.W * self.H * 4
        self.L.orc_accum_write(self.h, rgba.ctypes.data)

    def read_accum(self):
        out = np.empty((self.H, self.W, 4), dtype=np.float32)
        self.L.orc_accum_read(self.h, out.ctypes.data)
        return out

    def render(self, first_frame, seeds3):
        s = np.ascontiguousarray(seeds3, dtype=np.float32).reshape(-1, 3)
        self.L.orc_render(self.h, int(first_frame), len(s), s.ctypes.data, self.threads, None)

    def primary_ids(self, seed3):
        """(H, W, 3) int32: triangle index or -1, bits of t, iterationsCount. Renders one frame
        (frame index 0) into the accumulation as a side effect."""
        s = np.ascontiguousarray(seed3, dtype=np.float32).reshape(1, 3)
        out = np.empty((self.H, self.W, 3), dtype=np.int32)
        self.L.orc_render(self.h, 0, 1, s.ctypes.data, self.threads, out.ctypes.data)
        return out

    def counters(self, reset=False):
        c = np.zeros(7, dtype=np.uint64)
        self.L.orc_counters(self.h, c.ctypes.data, int(reset))
        return dict(zip(("rays", "node_pops", "boxes_tested", "tris_tested", "noise_fetches", "sky_fetches", "hits"), map(int, c)))

    def sample_noise(self, s, t):
        out = np.zeros(3, dtype=np.float32)
        self.L.orc_sample_noise(self.h, s, t, out.ctypes.data)
        return out

    def sample_sky(self, d):
        d = np.ascontiguousarray(d, dtype=np.float32)
        out = np.zeros(3, dtype=np.float32)
        self.L.orc_sample_sky(self.h, d.ctypes.data, out.ctypes.data)
        return out


    def sample_sky_rotated(self, d):
        """The sky lookup as the tracer makes it: rotation about +x, then the cube map."""
        d = np.ascontiguousarray(d, dtype=np.float32)
        out = np.zeros(3, dtype=np.float32)
        self.L.orc_sample_sky_rotated(self.h, d.ctypes.data, out.ctypes.data)
        return out


def hit_triangles(tri12, origins, directions):
    """hitTriangle of the tracer on n independent (triangle, ray) pairs: tri12 (n,12) as in the triangle buffer,
    origins and directions (n,3). Returns (hit (n,) bool, t (n,) float32, 0 where missed)."""
    L = _load()
    tri12 = np.ascontiguousarray(tri12, dtype=np.float32).reshape(-1, 12)
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    assert len(o) == len(tri12) == len(d)
    hit, t = np.zeros(len(o), dtype=np.uint8), np.zeros(len(o), dtype=np.float32)
    L.orc_hit_triangles(len(o), tri12.ctypes.data, o.ctypes.data, d.ctypes.data, hit.ctypes.data, t.ctypes.data)
    return hit.astype(bool), t


class PostOracle:
    """CPU restatement of the bloom chain + final composite (post_oracle.cpp). Keeps the `tmp`
    image between calls, as the reference does between frames."""

    def __init__(self, width, height):
        self.L = _load()
        self.L.post_create.restype = ctypes.c_void_p
        self.L.post_create.argtypes = [ctypes.c_int, ctypes.c_int]
        self.L.post_destroy.argtypes = [ctypes.c_void_p]
        self.L.post_run.argtypes = [ctypes.c_void_p] * 4
        self.L.post_level.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self.W, self.H = width, height
        self.h = self.L.post_create(width, height)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.post_destroy(self.h)
            self.h = None

    def run(self, accum_rgba):
        a = np.ascontiguousarray(accum_rgba, dtype=np.float32)
        assert a.size == self.W * self.H * 4
        rgba = np.empty((self.H, self.W, 4), dtype=np.float32)
        rgb8 = np.empty((self.H, self.W, 3), dtype=np.uint8)
        self.L.post_run(self.h, a.ctypes.data, rgba.ctypes.data, rgb8.ctypes.data)
        return rgba, rgb8

    def level(self, i):
        out = np.empty((self.H, self.W, 3), dtype=np.float32)
        self.L.post_level(self.h, i, out.ctypes.data)
        return out
