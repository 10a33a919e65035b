"""Bloom chain + final composite.

CPU: the post oracle against llvmpipe output of the reference's three post shaders
(tests/golden/post_reference.npz, three consecutive frames so that the carried-over tmp content
is exercised). Tolerance 2e-5 absolute on float images (llvmpipe approximates pow(); everything
else agrees to rounding), 8-bit read-back within one code.
GPU: the HIP chain against the oracle, tolerance 2e-5 absolute (powf of two different libms)."""
import os

import numpy as np
import pytest

from rt_oracle import PostOracle
from util import GOLDEN

FX = np.load(os.path.join(GOLDEN, "post_reference.npz"))
W, H = int(FX["width"]), int(FX["height"])
TOL = 2e-5


def test_post_oracle_matches_reference_shaders():
    po = PostOracle(W, H)
    for f in range(3):
        rgba, rgb8 = po.run(FX["accum%d" % f])
        assert np.abs(rgba[..., :3] - FX["final%d" % f]).max() <= TOL
        assert np.abs(rgb8.astype(int) - FX["rgb8_%d" % f].astype(int)).max() <= 1
        assert (rgb8 != FX["rgb8_%d" % f]).any(-1).mean() < 0.002
    for i in range(4):
        assert np.abs(po.level(i) - FX["level%d_2" % i]).max() <= TOL, i
    assert FX["final2"].max() > 2.0      # the fixture does contain bloom


def smooth_hdr(w, h, seed):
    r = np.random.default_rng(seed)
    img = (r.random((h, w, 4)) * 0.8).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(8):
        cx, cy, rad = r.integers(0, w), r.integers(0, h), r.integers(2, max(3, w // 8))
        img[((xx - cx) ** 2 + (yy - cy) ** 2) < rad * rad, :3] = r.uniform(1.0, 9.0, 3).astype(np.float32)
    img[..., 3] = 0
    return img


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(160, 80), (200, 120), (96, 54), (333, 77)])
def test_gpu_post_matches_oracle(size):
    import raytracing_amd as ra
    w, h = size
    g, po = ra.Renderer(w, h), PostOracle(w, h)
    for f in range(3):
        a = smooth_hdr(w, h, 100 + f)
        g.write_accum(a)
        rgba, rgb8 = g.postprocess(True)
        o_rgba, o_rgb8 = po.run(a)
        assert np.abs(rgba[..., :3] - o_rgba[..., :3]).max() <= TOL, f
        assert np.abs(rgb8.astype(int) - o_rgb8.astype(int)).max() <= 1
    raw, raw8 = g.postprocess(False)
    assert (raw[..., :3] == a[..., :3]).all()
    assert (raw8 == np.rint(np.clip(a[..., :3], 0, 1) * 255).astype(np.uint8)).all()
    g.close()


@pytest.mark.gpu
def test_gpu_post_matches_reference_fixture():
    import raytracing_amd as ra
    g = ra.Renderer(W, H)
    for f in range(3):
        g.write_accum(FX["accum%d" % f])
        rgba, rgb8 = g.postprocess(True)
        assert np.abs(rgba[..., :3] - FX["final%d" % f]).max() <= TOL
        assert np.abs(rgb8.astype(int) - FX["rgb8_%d" % f].astype(int)).max() <= 1
    g.close()
