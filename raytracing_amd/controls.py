"""Python face of the host-side control surface: Camera, key codes, seed stream, ini reader, and the
headless frame loop (HeadlessApp) with its Recorder. Each method is one call into librt_mi355x.so."""
import ctypes

import numpy as np

from . import _lib
from ._lib import rth_app_state

# GLFW key codes / actions the reference reacts to
KEY = dict(SPACE=32, K1=49, K2=50, K3=51, K4=52, K5=53, K6=54, A=65, B=66, C=67, D=68, P=80, R=82, S=83, W=87, GRAVE=96, RIGHT=262,
           LEFT=263, DOWN=264, UP=265, LEFT_SHIFT=340, LEFT_CONTROL=341, RIGHT_SHIFT=344)
RELEASE, PRESS, REPEAT = 0, 1, 2


class Camera:
    def __init__(self):
        self.L = _lib.lib()
        self.h = self.L.rth_camera_create()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.rth_camera_destroy(self.h)
            self.h = None

    def key(self, key, action):
        self.L.rth_camera_key(self.h, key, action)

    def tick(self):
        return bool(self.L.rth_camera_tick(self.h))

    def get(self):
        o = np.zeros(9, dtype=np.float32)
        self.L.rth_camera_get(self.h, o.ctypes.data)
        return dict(position=o[0:3].copy(), facing=o[3:6].copy(), fov=float(o[6]), viewport_width=float(o[7]), viewport_distance=float(o[8]))

    def set_pose(self, position, facing):
        p, f = np.asarray(position, np.float32), np.asarray(facing, np.float32)
        self.L.rth_camera_set_pose(self.h, p.ctypes.data, f.ctypes.data)


def bounce_limit_from_key(key):
    v = _lib.lib().rth_bounce_limit_from_key(key)
    return None if v < 0 else v


def rand_seed_triples(n):
    out = np.zeros((n, 3), dtype=np.float32)
    _lib.lib().rth_rand_seed_triples(n, out.ctypes.data)
    return out


def ini_object_paths(path):
    buf, sky = ctypes.create_string_buffer(1 << 16), ctypes.create_string_buffer(4096)
    n = _lib.lib().rth_ini_object_paths(path.encode(), buf, len(buf), sky, len(sky))
    if n < 0:
        raise ValueError("cannot parse " + path)
    text = buf.value.decode()
    return (text.split("\n") if n else []), sky.value.decode()


def ini_values(path, section, name):
    """Values under [section] name as the reference's aConfigVec<std::string> returns them; ValueError (carrying the
    reference's message) when the file does not parse."""
    buf, err = ctypes.create_string_buffer(1 << 16), ctypes.create_string_buffer(4096)
    n = _lib.lib().rth_ini_values(path.encode(), section.encode(), name.encode(), buf, len(buf), err, len(err))
    if n < 0:
        raise ValueError(err.value.decode())
    return buf.value.decode().split("\n")[:n]


def ini_number(path, section, name, pos=0, kind=int):
    out = ctypes.c_double()
    rc = _lib.lib().rth_ini_number(path.encode(), section.encode(), name.encode(), pos, {int: 0, float: 1, bool: 2}[kind], ctypes.byref(out))
    if rc != 0:
        raise ValueError("cannot read %s / %s from %s" % (section, name, path))
    return kind(out.value)


def write_png(path, rgb, flip=False):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    rc = _lib.lib().rth_write_png_rgb8(path.encode(), rgb.shape[1], rgb.shape[0], rgb.ctypes.data, int(flip))
    if rc != 0:
        raise IOError("cannot write " + path)


def read_png(path, channels=3):
    """Decodes a PNG like the reference's stbi_load(path, ..., channels): (H, W, channels) uint8, file row order."""
    L = _lib.lib()
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    if L.rth_read_png(path.encode(), channels, ctypes.byref(w), ctypes.byref(h), None, 0) != 0:
        raise IOError("cannot decode " + path)
    out = np.empty((h.value, w.value, channels), dtype=np.uint8)
    if L.rth_read_png(path.encode(), channels, None, None, out.ctypes.data, out.nbytes) != 0:
        raise IOError("cannot decode " + path)
    return out


def load_skybox(directory):
    """The six faces of a sky-box directory in the reference's order (src/main.cpp:52-59)."""
    return np.stack([read_png("%s/%s.png" % (directory, n), 4) for n in ("right", "left", "top", "bottom", "front", "back")])


class HeadlessApp:
    """The reference's frame loop without a window, driving a Renderer."""

    def __init__(self, renderer):
        self.L = _lib.lib()
        self.r = renderer
        h = ctypes.c_void_p()
        if self.L.rth_app_create(renderer.h, renderer.W, renderer.H, ctypes.byref(h)) != 0:
            raise RuntimeError("rth_app_create failed")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.rth_app_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def load_objects(self, paths, build_threads=1):
        arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
        self.r._ck(self.L.rth_app_load_objects(self.h, arr, len(paths), build_threads), "rth_app_load_objects")

    def key(self, key, action):
        self.L.rth_app_key(self.h, key, action)

    def tap(self, key):
        self.key(key, PRESS)
        self.key(key, RELEASE)

    def tick(self, n=1):
        for _ in range(n):
            self.r._ck(self.L.rth_app_tick(self.h), "rth_app_tick")

    def read_display(self):
        rgba = np.empty((self.r.H, self.r.W, 4), dtype=np.float32)
        rgb8 = np.empty((self.r.H, self.r.W, 3), dtype=np.uint8)
        self.r._ck(self.L.rth_app_read_display(self.h, rgba.ctypes.data, rgb8.ctypes.data), "rth_app_read_display")
        return rgba, rgb8

    def state(self):
        s = rth_app_state()
        self.L.rth_app_get_state(self.h, ctypes.byref(s))
        c = np.array(list(s.camera), dtype=np.float32)
        return dict(position=c[0:3], facing=c[3:6], fov=float(c[6]), viewport_width=float(c[7]), viewport_distance=float(c[8]),
                    last_seed=np.array(list(s.last_seed), dtype=np.float32), next_frame_index=s.next_frame_index,
                    bounce_limit=s.bounce_limit, recorder_state=s.recorder_state, poses_left=s.poses_left, images_written=s.images_written)

    def configure(self, bloom=True, post_every_frame=True, frames_per_image=256, record_every_frames=1, output_dir="shots", flip=False):
        self.L.rth_app_configure(self.h, int(bloom), int(post_every_frame), frames_per_image, record_every_frames, output_dir.encode(), int(flip))

    def push_pose(self, position, facing):
        p, f = np.asarray(position, np.float32), np.asarray(facing, np.float32)
        self.L.rth_app_push_pose(self.h, p.ctypes.data, f.ctypes.data)

    def save_poses(self, path):
        self.r._ck(self.L.rth_app_save_poses(self.h, path.encode()), "rth_app_save_poses")

    def load_poses(self, path):
        self.r._ck(self.L.rth_app_load_poses(self.h, path.encode()), "rth_app_load_poses")
