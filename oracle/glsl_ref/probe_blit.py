"""Measures which texels llvmpipe's framebuffer blits read (TEST INFRASTRUCTURE ONLY, build container).

The reference's bloom chain (src/main.cpp:244-248) shrinks the accumulation image into the corner of
its tmp texture with glBlitNamedFramebuffer(GL_NEAREST) and enlarges the blurred corner again with
GL_LINEAR (src/main.cpp:261-265). For the even factors the NEAREST sample point lies exactly on a
texel boundary, so the texel taken is decided by the rasteriser's interpolation arithmetic. This
probe blits an index ramp (texel value = its own x, y) and reports, for every destination texel,
the source texel it received, next to candidate formulas; for the LINEAR blit it blits ramps and
unit impulses to recover the two source texels and the weight of every destination texel.

    python oracle/glsl_ref/probe_blit.py [--sizes 1280x720,160x80 ...]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from glref import GLRef  # noqa: E402

DOWNSAMPLE = (5, 10, 20, 40, 80, 160)


def ramp(W, H):
    img = np.zeros((H, W, 4), np.float32)
    img[..., 0] = np.arange(W, dtype=np.float32)[None, :]
    img[..., 1] = np.arange(H, dtype=np.float32)[:, None]
    return img


def nearest_selection(g, W, H, ds):
    """Returns (sx[w], sy[h]): the source column/row each destination column/row of the down-blit received."""
    src = g.tex_f32(0, W, H, ramp(W, H), linear=False)
    sfbo = g.fbo(src)
    dst = g.tex_f32(1, W, H, np.full((H, W, 4), -1.0, np.float32), linear=True)
    dfbo = g.fbo(dst)
    w, h = W // ds, H // ds
    g.blit(sfbo, dfbo, (0, 0, W, H), (0, 0, w, h), linear=False)
    g.finish()
    out = g.read_f32(1, dst, W, H)
    corner = out[:h, :w]
    sx, sy = corner[0, :, 0].astype(np.int64), corner[:, 0, 1].astype(np.int64)
    # the selection must be separable: every row uses the same columns and vice versa
    assert (corner[..., 0] == sx[None, :]).all() and (corner[..., 1] == sy[:, None]).all(), "selection is not separable"
    assert (out[h:, :, 0] == -1).all() and (out[:, w:, 0] == -1).all(), "blit wrote outside the corner"
    return sx, sy


def candidates(n_dst, n_src):
    x = np.arange(n_dst)
    f32 = np.float32
    scale32 = f32(n_src) / f32(n_dst)
    return {
        "floor((x+0.5)*S/D) f32": np.floor((x.astype(f32) + f32(0.5)) * scale32).astype(np.int64),
        "exact floor((2x+1)*S/(2D))": ((2 * x + 1) * n_src) // (2 * n_dst),
        "exact ceil-1": -((-(2 * x + 1) * n_src) // (2 * n_dst)) - 1,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1280x720,1920x1080,3840x2160,160x80,250x130,123x77")
    a = ap.parse_args()
    g = GLRef()
    print(g.renderer())
    for s in a.sizes.split(","):
        W, H = map(int, s.split("x"))
        for ds in DOWNSAMPLE:
            if W // ds == 0 or H // ds == 0:
                continue
            sx, sy = nearest_selection(g, W, H, ds)
            for name, sel, nd, ns in (("x", sx, W // ds, W), ("y", sy, H // ds, H)):
                line = f"{W}x{H} ds={ds} {name}: dst {nd} <- src {ns}; first {sel[:6].tolist()} last {sel[-3:].tolist()}"
                for cname, c in candidates(nd, ns).items():
                    bad = int((c != sel).sum())
                    line += f" | {cname}: {bad} differ"
                print(line)


if __name__ == "__main__":
    main()
