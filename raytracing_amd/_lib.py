"""Loads librt_mi355x.so (HIP kernels + C ABI + host scene model) and declares its signatures.

There is no fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_LIB_PATH") or os.path.join(_HERE, "librt_mi355x.so")   # RT_LIB_PATH: kernel-tuning experiments only


class rt_material(ctypes.Structure):
    _fields_ = [("colour", ctypes.c_float * 3), ("specularColour", ctypes.c_float * 3), ("roughness", ctypes.c_float),
                ("metallic", ctypes.c_float), ("transparency", ctypes.c_float), ("refractionIndex", ctypes.c_float),
                ("isLight", ctypes.c_int32)]


class rt_camera(ctypes.Structure):
    _fields_ = [("position", ctypes.c_float * 3), ("facing", ctypes.c_float * 3), ("fov", ctypes.c_float),
                ("viewportWidth", ctypes.c_float), ("viewportDistance", ctypes.c_float)]


class rt_counters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("rays", "node_pops", "boxes_tested", "tris_tested", "noise_fetches",
                                               "sky_fetches", "paths", "hits")] + [("sched_rounds", ctypes.c_uint64 * 4),
                                                                                    ("sched_lanes", ctypes.c_uint64 * 4),
                                                                                    ("sched_cycles", ctypes.c_uint64 * 4)]


class rth_bvh_stats(ctypes.Structure):
    _fields_ = [("max_depth", ctypes.c_int32), ("leaves", ctypes.c_uint32), ("splits", ctypes.c_uint64),
                ("degenerate_splits", ctypes.c_uint64), ("leaf_depth_sum", ctypes.c_uint64), ("build_ms", ctypes.c_double)]


class rth_video_stats(ctypes.Structure):
    _fields_ = [("rendered", ctypes.c_uint32), ("skipped", ctypes.c_uint32), ("written", ctypes.c_uint32), ("write_failures", ctypes.c_uint32),
                ("wall_seconds", ctypes.c_double), ("trace_seconds", ctypes.c_double), ("encode_seconds", ctypes.c_double),
                ("stall_seconds", ctypes.c_double), ("rays", ctypes.c_uint64)]


class rth_app_state(ctypes.Structure):
    _fields_ = [("camera", ctypes.c_float * 9), ("last_seed", ctypes.c_float * 3), ("next_frame_index", ctypes.c_uint32),
                ("bounce_limit", ctypes.c_uint32), ("recorder_state", ctypes.c_int32), ("poses_left", ctypes.c_uint32),
                ("images_written", ctypes.c_uint32)]


# name -> (restype, argtypes); this table is also what the symbol-export test checks against the headers
P, I, U32, U64P, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), ctypes.c_float
SIGNATURES = {
    "rt_last_global_error": (ctypes.c_char_p, []),
    "rt_last_error": (ctypes.c_char_p, [P]),
    "rt_create": (I, [I, U32, U32, ctypes.POINTER(P)]),
    "rt_destroy": (None, [P]),
    "rt_scene_upload": (I, [P, P, P, U32, P, U32]),
    "rt_set_materials": (I, [P, P, U32]),
    "rt_set_camera": (I, [P, ctypes.POINTER(rt_camera)]),
    "rt_set_bounce_limit": (I, [P, U32]),
    "rt_set_ray_count": (I, [P, U32]),
    "rt_set_noise": (I, [P, P, U32, U32]),
    "rt_set_skybox": (I, [P, ctypes.POINTER(P), U32]),
    "rt_accum_clear": (I, [P]),
    "rt_accum_write": (I, [P, P]),
    "rt_accum_read": (I, [P, P]),
    "rt_render": (I, [P, U32, U32, P]),
    "rt_sync": (I, [P]),
    "rt_primary_ids": (I, [P, P, P]),
    "rt_probe_triangles": (I, [P, U32, P, P, P, P, P]),
    "rt_probe_blit_planes": (I, [U32] * 8 + [P]),
    "rt_postprocess": (I, [P, I, P, P]),
    "rt_postprocess_static": (I, [P, U32, U32, P, P, P]),
    "rt_display_read": (I, [P, P, P]),
    "rt_display_reset": (I, [P]),
    "rt_get_counters": (I, [P, ctypes.POINTER(rt_counters), I]),
    "rt_set_counting": (I, [P, I]),
    "rt_set_shard": (I, [P, U32, U32]),
    "rt_shard_tiles": (I, [P, ctypes.POINTER(U32)]),
    "rt_shard_device_ptr": (I, [P, ctypes.POINTER(P)]),
    "rt_shard_export": (I, [P, P]),
    "rt_accum_assemble": (I, [P, P, U32]),
    "rt_group_create": (I, [ctypes.POINTER(ctypes.c_int), U32, U32, U32, ctypes.POINTER(P)]),
    "rt_group_destroy": (None, [P]),
    "rt_group_size": (U32, [P]),
    "rt_group_rank": (P, [P, U32]),
    "rt_group_root": (P, [P]),
    "rt_group_render": (I, [P, U32, U32, P]),
    "rt_group_gather": (I, [P]),
    "rt_group_sync": (I, [P]),
    "rt_group_last_error": (ctypes.c_char_p, [P]),
    "rt_set_stream": (I, [P, P]),
    "rt_kernel_time": (I, [P, ctypes.POINTER(ctypes.c_double), U64P, I]),
    "rth_scene_load": (I, [ctypes.POINTER(ctypes.c_char_p), U32, I, ctypes.POINTER(P)]),
    "rth_scene_load_cached": (I, [ctypes.POINTER(ctypes.c_char_p), U32, I, ctypes.c_char_p, ctypes.POINTER(P), ctypes.POINTER(I)]),
    "rth_scene_build": (I, [P, P, U32, P, U32, I, I, ctypes.POINTER(P)]),
    "rth_scene_free": (None, [P]),
    "rth_scene_triangle_count": (U32, [P]),
    "rth_scene_node_count": (U32, [P]),
    "rth_scene_material_count": (U32, [P]),
    "rth_scene_tri12": (P, [P]),
    "rth_scene_matidx": (P, [P]),
    "rth_scene_node8": (P, [P]),
    "rth_scene_materials": (P, [P]),
    "rth_scene_stats": (None, [P, ctypes.POINTER(rth_bvh_stats)]),
    "rth_scene_warning_count": (U32, [P]),
    "rth_scene_warning": (ctypes.c_char_p, [P, U32]),
    "rth_scene_send": (I, [P, P]),
    "rth_camera_create": (P, []),
    "rth_camera_destroy": (None, [P]),
    "rth_camera_key": (None, [P, I, I]),
    "rth_camera_tick": (I, [P]),
    "rth_camera_get": (None, [P, P]),
    "rth_camera_set_pose": (None, [P, P, P]),
    "rth_bounce_limit_from_key": (I, [I]),
    "rth_rand_seed_triples": (None, [U32, P]),
    "rth_ini_object_paths": (I, [ctypes.c_char_p, ctypes.c_char_p, U32, ctypes.c_char_p, U32]),
    "rth_ini_values": (I, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, U32, ctypes.c_char_p, U32]),
    "rth_ini_number": (I, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, U32, I, ctypes.POINTER(ctypes.c_double)]),
    "rth_write_png_rgb8": (I, [ctypes.c_char_p, U32, U32, P, I]),
    "rth_read_png": (I, [ctypes.c_char_p, I, ctypes.POINTER(U32), ctypes.POINTER(U32), P, ctypes.c_uint64]),
    "rth_app_create": (I, [P, U32, U32, ctypes.POINTER(P)]),
    "rth_app_destroy": (None, [P]),
    "rth_app_load_objects": (I, [P, ctypes.POINTER(ctypes.c_char_p), U32, I]),
    "rth_app_set_scene_cache": (None, [P, ctypes.c_char_p]),
    "rth_app_key": (None, [P, I, I]),
    "rth_app_tick": (I, [P]),
    "rth_app_read_display": (I, [P, P, P]),
    "rth_app_get_state": (None, [P, ctypes.POINTER(rth_app_state)]),
    "rth_app_configure": (None, [P, I, I, U32, U32, ctypes.c_char_p, I]),
    "rth_app_push_pose": (None, [P, P, P]),
    "rth_app_save_poses": (I, [P, ctypes.c_char_p]),
    "rth_app_load_poses": (I, [P, ctypes.c_char_p]),
    "rth_video_render": (I, [P, U32, U32, P, U32, U32, U32, U32, ctypes.c_char_p, U32, I, U32, ctypes.POINTER(rth_video_stats)]),
    "rth_video_seeds": (None, [U32, U32, U32, P]),
    "rth_read_poses": (I, [ctypes.c_char_p, P, U32, ctypes.POINTER(U32)]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if "emulated" in os.path.basename(LIB_PATH):
            raise RuntimeError("RT_LIB_PATH names the tests' host emulation of the library: the package only loads builds of the HIP library")
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "librt_mi355x.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C raytracing_amd/csrc`; there is no CPU fallback")
        # PyTorch ships its own copy of the HIP/HSA runtime under the same SONAMEs as the system one this
        # library links to. Whichever is loaded first serves the whole process; loading torch's first
        # keeps torch (device memory, streams, torch.distributed) and this library on one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
