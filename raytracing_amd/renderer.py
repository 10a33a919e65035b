"""Python face of the C ABI: SceneData (host scene model) and Renderer (the GPU path tracer).

Thin by design — every method is one call into librt_mi355x.so. Images are numpy arrays of shape
(H, W, 4) float32 with row 0 at the BOTTOM of the picture (the reference's GL read-back order).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import rt_camera, rt_counters, rt_material, rth_bvh_stats


class RtError(RuntimeError):
    pass


def _as(ptr, n, ctype, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctype)), shape=(n,)).astype(dtype, copy=True)


def materials_to_rows(mats):
    """list[rt_material] -> (n, 11) float32 rows (the order of the shader's Material struct)."""
    rows = np.zeros((len(mats), 11), dtype=np.float32)
    for i, m in enumerate(mats):
        rows[i] = list(m.colour) + list(m.specularColour) + [m.roughness, m.metallic, m.transparency, m.refractionIndex, float(m.isLight)]
    return rows


def rows_to_materials(rows):
    rows = np.asarray(rows, dtype=np.float32).reshape(-1, 11)
    arr = (rt_material * len(rows))()
    for i, r in enumerate(rows):
        arr[i].colour[:] = [float(x) for x in r[0:3]]
        arr[i].specularColour[:] = [float(x) for x in r[3:6]]
        arr[i].roughness, arr[i].metallic, arr[i].transparency, arr[i].refractionIndex = (float(x) for x in r[6:10])
        arr[i].isLight = 1 if r[10] != 0 else 0
    return arr


class SceneData:
    """Flat scene arrays produced by the host model (object files -> BVH -> flatten)."""

    def __init__(self, handle):
        L = _lib.lib()
        n, m, k = L.rth_scene_triangle_count(handle), L.rth_scene_node_count(handle), L.rth_scene_material_count(handle)
        self.tri12 = _as(L.rth_scene_tri12(handle), n * 12, ctypes.c_float, np.float32)
        self.matidx = _as(L.rth_scene_matidx(handle), n, ctypes.c_int32, np.int32)
        self.node8 = _as(L.rth_scene_node8(handle), m * 8, ctypes.c_float, np.float32)
        self.materials = _as(L.rth_scene_materials(handle), k * 11, ctypes.c_float, np.float32).reshape(-1, 11)
        if k:   # isLight is an int32 in the ABI struct
            raw = _as(L.rth_scene_materials(handle), k * 11, ctypes.c_int32, np.int32).reshape(-1, 11)
            self.materials[:, 10] = raw[:, 10]
        st = rth_bvh_stats()
        L.rth_scene_stats(handle, ctypes.byref(st))
        self.stats = {f: getattr(st, f) for f, _ in rth_bvh_stats._fields_}
        self.warnings = [L.rth_scene_warning(handle, i).decode() for i in range(L.rth_scene_warning_count(handle))]
        L.rth_scene_free(handle)

    @property
    def n_triangles(self):
        return int(self.matidx.size)

    @property
    def n_nodes(self):
        return int(self.node8.size // 8)


def load_scene(paths, build_threads=1, cache_dir=None):
    """Scene::LoadObjects without the GPU part: parse files, merge, build the BVH, flatten. With cache_dir the result
    is kept there, keyed by the files' bytes, and read back on later calls (SceneData.cache_hit tells)."""
    L = _lib.lib()
    arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
    h = ctypes.c_void_p()
    hit = ctypes.c_int(0)
    rc = L.rth_scene_load_cached(arr, len(paths), build_threads, cache_dir.encode() if cache_dir else None, ctypes.byref(h), ctypes.byref(hit))
    if rc != 0:
        raise RtError("rth_scene_load failed: %d" % rc)
    sc = SceneData(h)
    sc.cache_hit = bool(hit.value)
    return sc


def build_scene(tri9, mat_slot, materials_rows=None, build_threads=1, max_leaf=0):
    L = _lib.lib()
    tri9 = np.ascontiguousarray(tri9, dtype=np.float32).reshape(-1, 9)
    mat_slot = np.ascontiguousarray(mat_slot, dtype=np.int32)
    assert len(tri9) == len(mat_slot)
    mats = rows_to_materials(materials_rows) if materials_rows is not None else None
    nm = len(mats) if mats is not None else 0
    h = ctypes.c_void_p()
    rc = L.rth_scene_build(tri9.ctypes.data, mat_slot.ctypes.data, len(tri9), ctypes.cast(mats, ctypes.c_void_p) if mats is not None else None,
                           nm, build_threads, max_leaf, ctypes.byref(h))
    if rc != 0:
        raise RtError("rth_scene_build failed: %d" % rc)
    return SceneData(h)


class Renderer:
    """One GPU path-tracing context (rt_context)."""

    def __init__(self, width, height, device=0, _borrowed=None):
        self.L = _lib.lib()
        self.W, self.H = width, height
        self._owned = _borrowed is None
        if _borrowed is not None:          # a context that belongs to a Group
            self.h = ctypes.c_void_p(_borrowed)
            return
        h = ctypes.c_void_p()
        rc = self.L.rt_create(device, width, height, ctypes.byref(h))
        if rc != 0:
            raise RtError("rt_create: " + self.L.rt_last_global_error().decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            if self._owned:
                self.L.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _ck(self, rc, what):
        if rc != 0:
            raise RtError("%s failed (%d): %s" % (what, rc, self.L.rt_last_error(self.h).decode()))

    def set_scene(self, tri12, matidx, node8):
        tri12 = np.ascontiguousarray(tri12, dtype=np.float32)
        matidx = np.ascontiguousarray(matidx, dtype=np.int32)
        node8 = np.ascontiguousarray(node8, dtype=np.float32)
        if tri12.size != matidx.size * 12 or node8.size % 8:
            raise RtError("scene arrays have inconsistent sizes")
        self._ck(self.L.rt_scene_upload(self.h, tri12.ctypes.data, matidx.ctypes.data, matidx.size, node8.ctypes.data, node8.size // 8), "rt_scene_upload")

    def set_materials(self, rows11):
        arr = rows_to_materials(rows11)
        self._ck(self.L.rt_set_materials(self.h, ctypes.cast(arr, ctypes.c_void_p), len(arr)), "rt_set_materials")

    def set_scene_data(self, scene):
        self.set_scene(scene.tri12, scene.matidx, scene.node8)
        self.set_materials(scene.materials[:16])

    def set_camera(self, position, facing, fov, viewport_width, viewport_distance=1.0):
        c = rt_camera()
        c.position[:] = [float(x) for x in position]
        c.facing[:] = [float(x) for x in facing]
        c.fov, c.viewportWidth, c.viewportDistance = float(fov), float(viewport_width), float(viewport_distance)
        self._ck(self.L.rt_set_camera(self.h, ctypes.byref(c)), "rt_set_camera")

    def set_bounce_limit(self, n):
        self._ck(self.L.rt_set_bounce_limit(self.h, int(n)), "rt_set_bounce_limit")

    def set_ray_count(self, n):
        """Rays per pixel and frame: the shader's compile-time RAY_COUNT (Fragment.glsl:5), 1 in the reference as checked in."""
        self._ck(self.L.rt_set_ray_count(self.h, int(n)), "rt_set_ray_count")

    def set_noise(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        self._ck(self.L.rt_set_noise(self.h, rgb.ctypes.data, rgb.shape[1], rgb.shape[0]), "rt_set_noise")

    def set_skybox(self, faces):
        faces = np.ascontiguousarray(faces, dtype=np.uint8)
        assert faces.ndim == 4 and faces.shape[0] == 6 and faces.shape[3] == 4 and faces.shape[1] == faces.shape[2]
        ptrs = (ctypes.c_void_p * 6)(*[faces[i].ctypes.data for i in range(6)])
        self._ck(self.L.rt_set_skybox(self.h, ptrs, faces.shape[1]), "rt_set_skybox")

    def clear(self):
        self._ck(self.L.rt_accum_clear(self.h), "rt_accum_clear")

    def write_accum(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        assert rgba.size == self.W * self.H * 4
        self._ck(self.L.rt_accum_write(self.h, rgba.ctypes.data), "rt_accum_write")

    def read_accum(self):
        out = np.empty((self.H, self.W, 4), dtype=np.float32)
        self._ck(self.L.rt_accum_read(self.h, out.ctypes.data), "rt_accum_read")
        return out

    def render(self, first_frame, seeds3):
        s = np.ascontiguousarray(seeds3, dtype=np.float32).reshape(-1, 3)
        self._ck(self.L.rt_render(self.h, int(first_frame), len(s), s.ctypes.data), "rt_render")

    def sync(self):
        self._ck(self.L.rt_sync(self.h), "rt_sync")

    def primary_ids(self, seed3):
        s = np.ascontiguousarray(seed3, dtype=np.float32).reshape(3)
        out = np.empty((self.H, self.W, 3), dtype=np.int32)
        self._ck(self.L.rt_primary_ids(self.h, s.ctypes.data, out.ctypes.data), "rt_primary_ids")
        return out

    def probe_triangles(self, tri12, origins, directions):
        """The tracer's ray/triangle test on n independent (triangle, ray) pairs: (hit (n,) bool, t (n,) float32)."""
        tri12 = np.ascontiguousarray(tri12, dtype=np.float32).reshape(-1, 12)
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        if not (len(tri12) == len(o) == len(d)):
            raise ValueError("probe_triangles: %d triangles, %d origins, %d directions" % (len(tri12), len(o), len(d)))
        hit, t = np.zeros(len(o), dtype=np.uint8), np.zeros(len(o), dtype=np.float32)
        self._ck(self.L.rt_probe_triangles(self.h, len(o), tri12.ctypes.data, o.ctypes.data, d.ctypes.data, hit.ctypes.data, t.ctypes.data),
                 "rt_probe_triangles")
        return hit.astype(bool), t

    def postprocess(self, bloom_on=True, want_rgb8=True):
        rgba = np.empty((self.H, self.W, 4), dtype=np.float32)
        rgb8 = np.empty((self.H, self.W, 3), dtype=np.uint8) if want_rgb8 else None
        self._ck(self.L.rt_postprocess(self.h, int(bloom_on), rgba.ctypes.data, rgb8.ctypes.data if want_rgb8 else None), "rt_postprocess")
        return rgba, rgb8

    def display_reset(self):
        self._ck(self.L.rt_display_reset(self.h), "rt_display_reset")

    def counters(self, reset=False):
        c = rt_counters()
        self._ck(self.L.rt_get_counters(self.h, ctypes.byref(c), int(reset)), "rt_get_counters")
        return {f: (int(getattr(c, f)) if not hasattr(getattr(c, f), "__len__") else [int(v) for v in getattr(c, f)])
                for f, _ in rt_counters._fields_}

    def set_counting(self, full):
        self._ck(self.L.rt_set_counting(self.h, int(full)), "rt_set_counting")

    def kernel_time(self, reset=False):
        ms, n = ctypes.c_double(), ctypes.c_uint64()
        self._ck(self.L.rt_kernel_time(self.h, ctypes.byref(ms), ctypes.byref(n), int(reset)), "rt_kernel_time")
        return ms.value, n.value

    # ---- multi-GPU tile sharding ----
    def set_shard(self, rank, world):
        self._ck(self.L.rt_set_shard(self.h, rank, world), "rt_set_shard")

    def shard_tiles(self):
        n = ctypes.c_uint32()
        self._ck(self.L.rt_shard_tiles(self.h, ctypes.byref(n)), "rt_shard_tiles")
        return n.value

    def shard_device_ptr(self):
        p = ctypes.c_void_p()
        self._ck(self.L.rt_shard_device_ptr(self.h, ctypes.byref(p)), "rt_shard_device_ptr")
        return p.value

    def export_shard(self, dst_device_ptr):
        self._ck(self.L.rt_shard_export(self.h, ctypes.c_void_p(dst_device_ptr)), "rt_shard_export")

    def assemble(self, gathered_device_ptr, world):
        self._ck(self.L.rt_accum_assemble(self.h, ctypes.c_void_p(gathered_device_ptr), world), "rt_accum_assemble")

    def set_stream(self, hip_stream):
        self._ck(self.L.rt_set_stream(self.h, ctypes.c_void_p(hip_stream)), "rt_set_stream")


class Group:
    """Several GPUs of this process rendering one image (rt_group): `ranks` are the sharded per-device renderers (upload
    scene, textures, camera to each), `root` is the unsharded renderer that holds the image after gather()."""

    def __init__(self, devices, width, height):
        self.L = _lib.lib()
        arr = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        rc = self.L.rt_group_create(arr, len(devices), width, height, ctypes.byref(h))
        if rc != 0:
            raise RtError("rt_group_create: " + self.L.rt_group_last_error(None).decode())
        self.h = h
        self.ranks = [Renderer(width, height, _borrowed=self.L.rt_group_rank(h, r)) for r in range(self.L.rt_group_size(h))]
        self.root = Renderer(width, height, _borrowed=self.L.rt_group_root(h))

    def _ck(self, rc, what):
        if rc != 0:
            raise RtError("%s failed (%d): %s" % (what, rc, self.L.rt_group_last_error(self.h).decode()))

    def render(self, first_frame, seeds3):
        s = np.ascontiguousarray(seeds3, dtype=np.float32).reshape(-1, 3)
        self._ck(self.L.rt_group_render(self.h, int(first_frame), len(s), s.ctypes.data), "rt_group_render")

    def gather(self):
        self._ck(self.L.rt_group_gather(self.h), "rt_group_gather")

    def sync(self):
        self._ck(self.L.rt_group_sync(self.h), "rt_group_sync")

    def close(self):
        if getattr(self, "h", None):
            for r in self.ranks + [self.root]:
                r.close()
            self.L.rt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()
