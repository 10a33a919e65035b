"""ctypes binding of oracle/_ref/libglref.so (headless Mesa llvmpipe GL 4.3 context).

TEST INFRASTRUCTURE ONLY: used to generate golden fixtures from the reference's own GLSL
shaders inside the build container. Nothing in the product path imports this module.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "..", "_ref", "libglref.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_ubyte_p = ctypes.POINTER(ctypes.c_ubyte)


class GLRef:
    def __init__(self):
        L = ctypes.CDLL(_LIB)
        L.glref_error.restype = ctypes.c_char_p
        L.glref_renderer.restype = ctypes.c_char_p
        L.glref_version.restype = ctypes.c_char_p
        L.glref_program.restype = ctypes.c_uint
        L.glref_program.argtypes = [ctypes.c_char_p] * 4
        L.glref_uniform1f.argtypes = [ctypes.c_uint, ctypes.c_char_p, ctypes.c_float]
        L.glref_uniform2f.argtypes = [ctypes.c_uint, ctypes.c_char_p] + [ctypes.c_float] * 2
        L.glref_uniform3f.argtypes = [ctypes.c_uint, ctypes.c_char_p] + [ctypes.c_float] * 3
        L.glref_uniform1i.argtypes = [ctypes.c_uint, ctypes.c_char_p, ctypes.c_int]
        L.glref_uniform1ui.argtypes = [ctypes.c_uint, ctypes.c_char_p, ctypes.c_uint]
        L.glref_uniform_location.argtypes = [ctypes.c_uint, ctypes.c_char_p]
        L.glref_tex2d_rgba32f.restype = ctypes.c_uint
        L.glref_tex2d_rgba32f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.glref_tex2d_upload_rgba32f.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.glref_tex2d_rgb8.restype = ctypes.c_uint
        L.glref_tex2d_rgb8.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.glref_cubemap_rgba8.restype = ctypes.c_uint
        L.glref_cubemap_rgba8.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.glref_texbuffer.restype = ctypes.c_uint
        L.glref_texbuffer.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int]
        L.glref_ssbo.restype = ctypes.c_uint
        L.glref_ssbo.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
        L.glref_fbo.restype = ctypes.c_uint
        L.glref_fbo.argtypes = [ctypes.c_uint]
        L.glref_draw.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
        L.glref_blit.argtypes = [ctypes.c_uint, ctypes.c_uint] + [ctypes.c_int] * 9
        L.glref_read_rgba32f.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
        L.glref_read_rgb8.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
        self.L = L
        if L.glref_init() != 0:
            raise RuntimeError("glref_init: " + L.glref_error().decode())

    def err(self):
        return self.L.glref_error().decode()

    def renderer(self):
        return self.L.glref_renderer().decode() + " / " + self.L.glref_version().decode()

    def program(self, vs_path, fs_path, find=None, repl=None):
        p = self.L.glref_program(vs_path.encode(), fs_path.encode(),
                                 find.encode() if find else None, repl.encode() if repl else None)
        if not p:
            raise RuntimeError("glref_program: " + self.err())
        return p

    def u1f(self, prog, name, a):
        self.L.glref_uniform1f(prog, name.encode(), a)

    def u2f(self, prog, name, a, b):
        self.L.glref_uniform2f(prog, name.encode(), a, b)

    def u3f(self, prog, name, a, b, c):
        self.L.glref_uniform3f(prog, name.encode(), a, b, c)

    def u1i(self, prog, name, a):
        self.L.glref_uniform1i(prog, name.encode(), a)

    def u1ui(self, prog, name, a):
        self.L.glref_uniform1ui(prog, name.encode(), a)

    def tex_f32(self, unit, w, h, data=None, linear=False):
        ptr = None
        if data is not None:
            data = np.ascontiguousarray(data, dtype=np.float32)
            assert data.size == w * h * 4
            ptr = data.ctypes.data
        return self.L.glref_tex2d_rgba32f(unit, w, h, ptr, int(linear))

    def tex_upload_f32(self, unit, tex, w, h, data):
        data = np.ascontiguousarray(data, dtype=np.float32)
        assert data.size == w * h * 4
        self.L.glref_tex2d_upload_rgba32f(unit, tex, w, h, data.ctypes.data)

    def tex_rgb8(self, unit, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        h, w, c = rgb.shape
        assert c == 3
        return self.L.glref_tex2d_rgb8(unit, w, h, rgb.ctypes.data)

    def cubemap(self, unit, faces):
        faces = np.ascontiguousarray(faces, dtype=np.uint8)
        assert faces.ndim == 4 and faces.shape[0] == 6 and faces.shape[1] == faces.shape[2] and faces.shape[3] == 4
        return self.L.glref_cubemap_rgba8(unit, faces.shape[1], faces.ctypes.data)

    def texbuffer(self, unit, arr, kind):
        arr = np.ascontiguousarray(arr)
        return self.L.glref_texbuffer(unit, arr.ctypes.data, arr.nbytes, kind)

    def ssbo(self, binding, arr):
        arr = np.ascontiguousarray(arr)
        return self.L.glref_ssbo(binding, arr.ctypes.data, arr.nbytes)

    def fbo(self, tex):
        f = self.L.glref_fbo(tex)
        if not f:
            raise RuntimeError("glref_fbo: " + self.err())
        return f

    def draw(self, prog, fbo, w, h):
        if self.L.glref_draw(prog, fbo, w, h) != 0:
            raise RuntimeError("glref_draw: " + self.err())

    def blit(self, src, dst, s, d, linear):
        if self.L.glref_blit(src, dst, *s, *d, int(linear)) != 0:
            raise RuntimeError("glref_blit: " + self.err())

    def finish(self):
        self.L.glref_finish()

    def read_f32(self, unit, tex, w, h):
        out = np.empty((h, w, 4), dtype=np.float32)
        if self.L.glref_read_rgba32f(unit, tex, out.ctypes.data) != 0:
            raise RuntimeError("glref_read: " + self.err())
        return out

    def read_rgb8(self, unit, tex, w, h):
        out = np.empty((h, w, 3), dtype=np.uint8)
        if self.L.glref_read_rgb8(unit, tex, out.ctypes.data) != 0:
            raise RuntimeError("glref_read: " + self.err())
        return out
