"""Runs the reference renderer's own GLSL passes on Mesa llvmpipe, headless.

TEST INFRASTRUCTURE ONLY (fixture generation in the build container; needs /root/reference).

This module replaces the reference's src/main.cpp for that purpose: it issues the same GL calls
in the same order — accumulation FBO, scene buffers, material/camera/bounce uniforms, one
full-screen draw per frame with u_FrameIndex / u_RandSeed, and the bloom chain — against the
UNMODIFIED shader files under <reference>/src/shaders. The only optional changes are one of two textual
substitutions made in memory while the file is loaded: `ids=True` makes the tracer's normal-view return
(Fragment.glsl:389) emit (triangle index, t, hitAnything) instead of |normal|, because the
reference has no output that carries triangle ids; `ray_count=n` sets the shader's own compile-time
constant RAY_COUNT (Fragment.glsl:5, checked in as 1) to n.
"""
import os

import numpy as np

from glref import GLRef

REFERENCE = os.environ.get("RT_REFERENCE_DIR", "/root/reference")
SHADERS = os.path.join(REFERENCE, "src", "shaders")

_IDS_FIND = "return vec3(abs(hitRecord.normal));"
_IDS_REPL = "return vec3(float(hitRecord.index), hitRecord.t, hitAnything ? 1.0 : 0.0);"
_RAYS_FIND = "#define RAY_COUNT 1\n"

# texture units (any distinct values work; the reference hands them out with a counter)
U_ACCUM, U_SKY, U_NOISE, U_MATIDX, U_BOXES, U_TMP, U_BLOOM0 = 0, 1, 2, 3, 4, 5, 6

DOWNSAMPLE = (5, 10, 20, 40, 80, 160)   # src/main.cpp:183


class ReferenceRenderer:
    def __init__(self, width, height, ids=False, gl=None, ray_count=1):
        self.gl = gl or GLRef()
        self.W, self.H = width, height
        g = self.gl
        vs = os.path.join(SHADERS, "Vertex.glsl")
        fs = os.path.join(SHADERS, "Fragment.glsl")
        assert not (ids and ray_count != 1), "the loader makes one substitution"
        if ray_count != 1:
            self.prog = g.program(vs, fs, _RAYS_FIND, "#define RAY_COUNT %d\n" % int(ray_count))
        else:
            self.prog = g.program(vs, fs, _IDS_FIND if ids else None, _IDS_REPL if ids else None)
        g.u2f(self.prog, "screenResolution", float(width), float(height))          # main.cpp:38-39
        self.accum = g.tex_f32(U_ACCUM, width, height, None, linear=False)            # main.cpp:148-158
        self.fbo = g.fbo(self.accum)
        g.u1i(self.prog, "u_Accumulation", U_ACCUM)
        g.u1ui(self.prog, "u_BounceLimit", 4)                                          # Scene.cpp:19
        self.frame_index = 0
        self._post = None

    # ---- scene / uniforms -------------------------------------------------------------
    def set_scene(self, tri12, matidx, node8):
        g = self.gl
        tri12 = np.ascontiguousarray(tri12, dtype=np.float32)
        matidx = np.ascontiguousarray(matidx, dtype=np.int32)
        node8 = np.ascontiguousarray(node8, dtype=np.float32)
        g.L.glref_use(self.prog)
        if tri12.size:
            g.ssbo(0, tri12)                                                           # Scene.cpp:124
        if matidx.size:
            g.texbuffer(U_MATIDX, matidx, 0)                                           # Scene.cpp:125
            g.u1i(self.prog, "u_MaterialsIndex", U_MATIDX)
            g.u1ui(self.prog, "u_MaterialsIndexCount", matidx.size)
        if node8.size:
            g.texbuffer(U_BOXES, node8, 1)                                             # Scene.cpp:126
            g.u1i(self.prog, "u_BoundingBoxes", U_BOXES)
        g.u1ui(self.prog, "u_BoundingBoxesCount", node8.size // 8)

    def set_materials(self, mats11):
        """mats11: (n, 11) float rows colour[3] specular[3] roughness metallic transparency ior isLight."""
        g = self.gl
        mats11 = np.asarray(mats11, dtype=np.float32).reshape(-1, 11)
        for i, m in enumerate(mats11):                                                 # Scene.cpp:133-153
            p = f"u_Materials[{i}]."
            g.u3f(self.prog, p + "colour", *map(float, m[0:3]))
            g.u3f(self.prog, p + "specularColour", *map(float, m[3:6]))
            g.u1f(self.prog, p + "roughness", float(m[6]))
            g.u1f(self.prog, p + "metallic", float(m[7]))
            g.u1f(self.prog, p + "transparency", float(m[8]))
            g.u1f(self.prog, p + "refractionIndex", float(m[9]))
            g.u1i(self.prog, p + "isLight", 1 if m[10] != 0 else 0)
        g.u1i(self.prog, "u_MaterialsCount", len(mats11))

    def set_camera(self, position, facing, fov, viewport_width, viewport_distance=1.0):
        g = self.gl
        g.u3f(self.prog, "u_Camera.position", *map(float, position))                   # Camera.cpp:129-135
        g.u3f(self.prog, "u_Camera.facing", *map(float, facing))
        g.u1f(self.prog, "u_Camera.fov", float(fov))
        g.u1f(self.prog, "u_Camera.viewportWidth", float(viewport_width))
        g.u1f(self.prog, "u_Camera.viewportDistance", float(viewport_distance))        # Scene.cpp:21

    def set_bounce_limit(self, n):
        self.gl.u1ui(self.prog, "u_BounceLimit", int(n))

    def set_noise(self, rgb):
        g = self.gl
        h, w, _ = rgb.shape
        g.tex_rgb8(U_NOISE, rgb)                                                       # main.cpp:89-114
        g.u1i(self.prog, "u_RgbNoise", U_NOISE)
        g.u2f(self.prog, "u_RgbNoiseResolution", float(w), float(h))

    def set_skybox(self, faces):
        g = self.gl
        g.cubemap(U_SKY, faces)                                                        # main.cpp:45-87
        g.u1i(self.prog, "u_Skybox", U_SKY)

    # ---- frames -----------------------------------------------------------------------
    def clear(self):
        self.gl.tex_upload_f32(U_ACCUM, self.accum, self.W, self.H, np.zeros((self.H, self.W, 4), np.float32))

    def write_accum(self, rgba):
        self.gl.tex_upload_f32(U_ACCUM, self.accum, self.W, self.H, rgba)

    def render(self, first_frame, seeds3):
        """One draw per seed triple (Scene::Tick uniforms + main.cpp:232-233)."""
        g = self.gl
        seeds3 = np.asarray(seeds3, dtype=np.float32).reshape(-1, 3)
        for k, s in enumerate(seeds3):
            g.u1ui(self.prog, "u_FrameIndex", int(first_frame) + k)                    # Scene.cpp:30
            g.u3f(self.prog, "u_RandSeed", float(s[0]), float(s[1]), float(s[2]))      # Scene.cpp:31
            g.draw(self.prog, self.fbo, self.W, self.H)
            # The reference presents every frame (blits + glfwSwapBuffers, main.cpp:235-278), which
            # flushes the pipeline; without a flush llvmpipe may bin several draws into one scene and
            # the accumulation read-back of draw k+1 can then miss what draw k wrote.
            g.finish()

    def read_accum(self):
        return self.gl.read_f32(U_ACCUM, self.accum, self.W, self.H)

    # ---- bloom chain + final composite (main.cpp:161-171,183-217,242-275) ---------------
    def _setup_post(self):
        g = self.gl
        W, H = self.W, self.H
        p = {}
        p["final"] = g.program(os.path.join(SHADERS, "finalVertex.glsl"), os.path.join(SHADERS, "finalFragment.glsl"))
        p["threshold"] = g.program(os.path.join(SHADERS, "extractBrightVertex.glsl"), os.path.join(SHADERS, "extractBrightFragment.glsl"))
        p["bloom"] = g.program(os.path.join(SHADERS, "bloomVertex.glsl"), os.path.join(SHADERS, "bloomFragment.glsl"))
        g.u1i(p["final"], "u_Original", U_ACCUM)
        g.u2f(p["final"], "u_Resolution", float(W), float(H))
        p["level_tex"], p["level_fbo"] = [], []
        for i in range(len(DOWNSAMPLE)):
            t = g.tex_f32(U_BLOOM0 + i, W, H, None, linear=True)
            p["level_tex"].append(t)
            p["level_fbo"].append(g.fbo(t))
            g.u1i(p["final"], f"u_Bloom{i}", U_BLOOM0 + i)
        p["tmp_tex"] = g.tex_f32(U_TMP, W, H, None, linear=True)
        p["tmp_fbo"] = g.fbo(p["tmp_tex"])
        self._post = p

    def postprocess(self, bloom_on=True):
        """Returns (rgba32f of the tmp texture, rgb8 read-back as Recorder::WriteTexture makes it)."""
        g = self.gl
        W, H = self.W, self.H
        if self._post is None:
            self._setup_post()
        p = self._post
        if not bloom_on:
            img = self.read_accum()
            return img, None
        for i, ds in enumerate(DOWNSAMPLE):
            w, h = W // ds, H // ds
            g.blit(self.fbo, p["tmp_fbo"], (0, 0, W, H), (0, 0, w, h), linear=False)
            g.u1i(p["threshold"], "u_Image", U_TMP)
            g.draw(p["threshold"], p["level_fbo"][i], W, H)
            g.u1i(p["bloom"], "u_Image", U_BLOOM0 + i)
            g.u2f(p["bloom"], "u_ImageResolution", float(w), float(h))
            g.draw(p["bloom"], p["tmp_fbo"], W, H)
            g.blit(p["tmp_fbo"], p["level_fbo"][i], (0, 0, w, h), (0, 0, W, H), linear=True)
        g.draw(p["final"], p["tmp_fbo"], W, H)
        g.finish()
        return g.read_f32(U_TMP, p["tmp_tex"], W, H), g.read_rgb8(U_TMP, p["tmp_tex"], W, H)

    def level_image(self, i):
        return self.gl.read_f32(U_BLOOM0 + i, self._post["level_tex"][i], self.W, self.H)
