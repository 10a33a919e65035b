"""Probe llvmpipe's LINEAR filtering of the RGB8 REPEAT noise texture and the RGBA8 cube map.

TEST INFRASTRUCTURE ONLY. Compares texture() results bit-for-bit with candidate formulas.
Run twice:  python oracle/glsl_ref/probe_tex.py
            GALLIVM_PERF=no_aos_sampling python oracle/glsl_ref/probe_tex.py
"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from glref import GLRef  # noqa: E402

f32 = np.float32
VS = "#version 430 core\nin vec4 position;\nvoid main(){ gl_Position = position; }\n"


def prog(gl, src):
    d = tempfile.mkdtemp()
    open(os.path.join(d, "v.glsl"), "w").write(VS)
    open(os.path.join(d, "f.glsl"), "w").write(src)
    return gl.program(os.path.join(d, "v.glsl"), os.path.join(d, "f.glsl"))


def bilinear_float(tex, s, t, conv):
    H, W, _ = tex.shape
    out = []
    for (coord, n) in ((s, W), (t, H)):
        c = f32(coord) - np.floor(f32(coord))          # fract
        c = f32(c * f32(n))
        c = f32(c - f32(0.5))
        i0 = int(np.floor(c))
        w = f32(c - f32(i0))
        if i0 < 0:
            i0 = n - 1
        i1 = 0 if i0 == n - 1 else i0 + 1
        out.append((i0, i1, w))
    (x0, x1, wx), (y0, y1, wy) = out
    t00, t10, t01, t11 = (conv(tex[y0, x0]), conv(tex[y0, x1]), conv(tex[y1, x0]), conv(tex[y1, x1]))
    top = t00 + wx * (t10 - t00)
    bot = t01 + wx * (t11 - t01)
    return (top + wy * (bot - top)).astype(f32), (x0, x1, y0, y1, wx, wy)


def main():
    gl = GLRef()
    print(gl.renderer(), "GALLIVM_PERF=" + os.environ.get("GALLIVM_PERF", ""))
    rng = np.random.default_rng(3)
    W, H = 600, 400
    tex = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    p = prog(gl, "#version 430 core\nout vec4 color;\nuniform sampler2D T;\nuniform vec2 P;\n"
                 "void main(){ color = vec4(texture(T, P).rgb, 1.0); }\n")
    gl.tex_rgb8(2, tex)
    gl.u1i(p, "T", 2)
    t = gl.tex_f32(0, 4, 4)
    fb = gl.fbo(t)
    convs = {
        "u8*(1/255)": lambda v: v.astype(f32) * f32(1.0 / 255.0),
        "u8/255": lambda v: v.astype(f32) / f32(255.0),
    }
    tally = {}
    worst = 0.0
    for trial in range(300):
        s, tt = rng.uniform(-3, 3, size=2).astype(f32)
        gl.u2f(p, "P", float(s), float(tt))
        gl.draw(p, fb, 4, 4)
        got = gl.read_f32(0, t, 4, 4)[0, 0, :3]
        for name, cv in convs.items():
            exp, dbg = bilinear_float(tex, s, tt, cv)
            ok = (exp.view(np.uint32) == got.view(np.uint32)).all()
            tally[name] = tally.get(name, 0) + int(ok)
            if name == "u8*(1/255)":
                worst = max(worst, float(np.abs(exp - got).max()))
        # fixed-point candidate
        x0, x1, y0, y1, wx, wy = dbg
        for shift_name, fx in (("w8 trunc", lambda w: int(w * 256.0)), ("w8 round", lambda w: int(w * 256.0 + 0.5))):
            wxi, wyi = fx(float(wx)), fx(float(wy))
            a, b_, c, d = [tex[yy, xx].astype(np.int32) for (yy, xx) in ((y0, x0), (y0, x1), (y1, x0), (y1, x1))]
            top = a + (((b_ - a) * wxi) >> 8)
            bot = c + (((d - c) * wxi) >> 8)
            r = top + (((bot - top) * wyi) >> 8)
            for cn, cv in convs.items():
                e = cv(r.astype(np.uint8))
                ok = (e.view(np.uint32) == got.view(np.uint32)).all()
                key = f"fixed {shift_name} {cn}"
                tally[key] = tally.get(key, 0) + int(ok)
        if trial < 3:
            print("sample", s, tt, "got", got, "float-exp", exp, "got*255", got * 255)
    print("2D RGB8 REPEAT LINEAR:", tally, "worst |float-got| =", worst)

    # cube map
    S = 16
    faces = rng.integers(0, 256, size=(6, S, S, 4), dtype=np.uint8)
    pc = prog(gl, "#version 430 core\nout vec4 color;\nuniform samplerCube T;\nuniform vec3 D;\n"
                  "void main(){ color = vec4(texture(T, D).rgb, 1.0); }\n")
    gl.cubemap(3, faces)
    gl.u1i(pc, "T", 3)
    np.save("/tmp/cube_faces.npy", faces)
    res = []
    dirs = rng.standard_normal((400, 3)).astype(f32)
    for d in dirs:
        gl.u3f(pc, "D", *[float(x) for x in d])
        gl.draw(pc, fb, 4, 4)
        res.append(gl.read_f32(0, t, 4, 4)[0, 0, :3].copy())
    np.save("/tmp/cube_dirs.npy", dirs)
    np.save("/tmp/cube_res.npy", np.array(res))
    print("cube results saved")


if __name__ == "__main__":
    main()
