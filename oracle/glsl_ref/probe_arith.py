"""Probe how Mesa llvmpipe evaluates the GLSL built-ins the tracer's hit decisions depend on.

TEST INFRASTRUCTURE ONLY. Each probe renders a tiny fragment shader (written here, not taken
from the reference) whose inputs are uniforms, and compares the result bit-for-bit with a few
candidate float32 evaluation orders. The findings are recorded in oracle/README.md and are
what oracle/rt_oracle.cpp implements.

Run:  python oracle/glsl_ref/probe_arith.py
"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from glref import GLRef  # noqa: E402

f32 = np.float32
VS = "#version 430 core\nin vec4 position;\nvoid main(){ gl_Position = position; }\n"


def run(gl, body, A, B, C=(0, 0, 0, 0), decl=""):
    src = ("#version 430 core\nout vec4 color;\nuniform vec3 A; uniform vec3 B; uniform vec3 C;\n"
           "uniform uint U;\n" + decl + "\nvoid main(){\n" + body + "\n}\n")
    d = tempfile.mkdtemp()
    with open(os.path.join(d, "v.glsl"), "w") as f:
        f.write(VS)
    with open(os.path.join(d, "f.glsl"), "w") as f:
        f.write(src)
    p = gl.program(os.path.join(d, "v.glsl"), os.path.join(d, "f.glsl"))
    gl.u3f(p, "A", *[float(x) for x in A[:3]])
    gl.u3f(p, "B", *[float(x) for x in B[:3]])
    gl.u3f(p, "C", *[float(x) for x in C[:3]])
    t = gl.tex_f32(0, 4, 4)
    fb = gl.fbo(t)
    gl.draw(p, fb, 4, 4)
    return gl.read_f32(0, t, 4, 4)[0, 0].copy()


def bits(x):
    return np.asarray(x, dtype=f32).view(np.uint32)


def report(name, got, cands):
    hits = [k for k, v in cands.items() if bits(v) == bits(got)]
    print(f"{name:28s} got={float(got)!r:24} match={hits}")
    return hits


def main():
    gl = GLRef()
    print(gl.renderer(), "GALLIVM_PERF=" + os.environ.get("GALLIVM_PERF", ""))
    rng = np.random.default_rng(7)
    tally = {}
    for trial in range(200):
        a = rng.standard_normal(3).astype(f32) * f32(3.7)
        b = rng.standard_normal(3).astype(f32) * f32(1.3)
        c = rng.standard_normal(3).astype(f32)
        res = {}
        # dot
        r = run(gl, "color = vec4(dot(A,B), 0,0,0);", a, b)
        res["dot3"] = [k for k, v in {
            "(xx+yy)+zz": (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2],
            "(zz+yy)+xx": (a[2] * b[2] + a[1] * b[1]) + a[0] * b[0],
            "xx+(yy+zz)": a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]),
            "fma(z,fma(y,xx))": f32(np.float64(a[2]) * np.float64(b[2]) + np.float64(f32(np.float64(a[1]) * np.float64(b[1]) + np.float64(a[0] * b[0])))),
        }.items() if bits(v) == bits(r[0])]
        # cross
        r = run(gl, "color = vec4(cross(A,B), 0);", a, b)
        res["cross"] = [k for k, v in {
            "a.yzx*b.zxy-a.zxy*b.yzx": np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=f32),
        }.items() if (bits(v) == bits(r[:3])).all()]
        # normalize
        r = run(gl, "color = vec4(normalize(A), 0);", a, b)
        d_xyz = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]
        d_zyx = (a[2] * a[2] + a[1] * a[1]) + a[0] * a[0]
        res["normalize"] = [k for k, v in {
            "a*(1/sqrt(d_xyz))": a * (f32(1) / np.sqrt(d_xyz)),
            "a*(1/sqrt(d_zyx))": a * (f32(1) / np.sqrt(d_zyx)),
            "a/sqrt(d_xyz)": a / np.sqrt(d_xyz),
            "a/sqrt(d_zyx)": a / np.sqrt(d_zyx),
        }.items() if (bits(v) == bits(r[:3])).all()]
        # division
        r = run(gl, "color = vec4(A/B, 1.0/B.x);", a, b)
        res["div"] = [k for k, v in {
            "a/b": a / b,
            "a*(1/b)": a * (f32(1) / b),
        }.items() if (bits(v) == bits(r[:3])).all()]
        # mix
        t = f32(abs(c[0]) % 1.0)
        r = run(gl, "color = vec4(mix(A,B,C.x), 0);", a, b, (t, 0, 0))
        res["mix"] = [k for k, v in {
            "a*(1-t)+b*t": a * (f32(1) - t) + b * t,
            "a+t*(b-a)": a + t * (b - a),
            "b*t+a*(1-t)": b * t + a * (f32(1) - t),
        }.items() if (bits(v) == bits(r[:3])).all()]
        # reflect (B normalised)
        n = b / np.sqrt((b * b).sum(dtype=f32))
        r = run(gl, "color = vec4(reflect(A,B), 0);", a, n)
        dxyz = (n[0] * a[0] + n[1] * a[1]) + n[2] * a[2]
        dzyx = (n[2] * a[2] + n[1] * a[1]) + n[0] * a[0]
        res["reflect"] = [k for k, v in {
            "I-(2*d_xyz)*N": a - (f32(2) * dxyz) * n,
            "I-(2*d_zyx)*N": a - (f32(2) * dzyx) * n,
            "I-2*(d_xyz*N)": a - f32(2) * (dxyz * n),
            "I-2*(d_zyx*N)": a - f32(2) * (dzyx * n),
        }.items() if (bits(v) == bits(r[:3])).all()]
        # mod
        x = f32(abs(a[0]) * 1e6)
        r = run(gl, "color = vec4(mod(A.x, B.x), 0,0,0);", (x, 0, 0), (600.0, 0, 0))
        y = f32(600)
        res["mod"] = [k for k, v in {
            "x-y*floor(x/y)": x - y * np.floor(x / y),
            "x-y*floor(x*(1/y))": x - y * np.floor(x * (f32(1) / y)),
        }.items() if bits(v) == bits(r[0])]
        for k, v in res.items():
            tally.setdefault(k, {})
            for h in (v or ["<none>"]):
                tally[k][h] = tally[k].get(h, 0) + 1
    for k, v in tally.items():
        print(k, v)

    # single-shot probes
    nan_probe = run(gl, "float z = A.x*B.x; color = vec4(min(z, 5.0), max(z, 5.0), min(5.0, z), max(5.0, z));",
                    (0, 0, 0), (np.inf, 0, 0))
    print("min/max with NaN (0*inf):", nan_probe)
    den = run(gl, "color = vec4(A.x*B.x, A.y+A.y, A.y, abs(A.y) < 1e-8 ? 1.0 : 0.0);", (1e-30, 1e-40, 0), (1e-10, 0, 0))
    print("denormals (1e-30*1e-10, 1e-40+1e-40, 1e-40):", den, bits(den))
    sc = run(gl, "float a = -3.1415/2; color = vec4(cos(a), sin(a), cos(A.x), sin(A.x));", (-3.1415 / 2, 0, 0), (0, 0, 0))
    print("cos/sin const vs uniform:", [repr(float(v)) for v in sc], "libm:", repr(float(f32(np.cos(f32(-3.1415 / 2))))),
          repr(float(f32(np.sin(f32(-3.1415 / 2))))))
    print("   float(-3.1415/2) =", repr(float(f32(-3.1415 / 2))), "cosf/sinf(double arg):", np.cos(-3.1415 / 2), np.sin(-3.1415 / 2))
    u = run(gl, "uint a = floatBitsToUint(A.x); uint b = floatBitsToUint(A.y); color = vec4(float(a), float(b), float(a+b), A.z*float(floatBitsToUint(B.x)));",
            np.array([0xFFFFFF7F, 0x81000041, 0], dtype=np.uint32).view(f32), np.array([4, 0, 0], dtype=np.uint32).view(f32))
    print("uint->float:", [repr(float(v)) for v in u], "numpy:", float(f32(np.uint32(0xFFFFFF7F))), float(f32(np.uint32(0x81000041))))
    rf = run(gl, "color = vec4(refract(A,B,C.x), 0);", (0.6, 0.0, -0.8), (0, 0, 1), (0.66, 0, 0))
    print("refract:", [repr(float(v)) for v in rf])
    pw = run(gl, "color = vec4(pow(A.x, B.x), pow(A.y, B.x), sqrt(A.x), length(A.xy));", (0.7, 2.5, 0), (1 / 1.61803, 0, 0))
    print("pow:", [repr(float(v)) for v in pw], "numpy:", float(f32(0.7) ** f32(1 / 1.61803)), float(f32(2.5) ** f32(1 / 1.61803)))


if __name__ == "__main__":
    main()
