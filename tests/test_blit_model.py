"""The two framebuffer blits of the bloom chain (src/main.cpp:244-248 NEAREST down, :261-265 LINEAR up) as Mesa llvmpipe
executes them. tests/golden/blit_reference.npz holds, per destination pixel, which source texel llvmpipe's NEAREST blit
read (index ramps blitted at 1280x720, 1920x1080, 3840x2160, 250x130 and 64x64, the bloom factors 5..40: at the even
factors the sample point lies exactly on a texel boundary) and the first texel and weights of its LINEAR blit's
footprint. The CPU restatement must reproduce every pixel, and the product's plane coefficients (host arithmetic,
exposed as the parity channel rt_probe_blit_planes) must equal the restatement's bit for bit; the per-pixel evaluation
of those planes on the GPU is covered by the chain fixtures in tests/test_post.py."""
import ctypes
import os
import re

import numpy as np
import pytest

from raytracing_amd import _lib
from rt_oracle import _load as load_oracle
from util import GOLDEN

FX = np.load(os.path.join(GOLDEN, "blit_reference.npz"))
NEAREST = [str(k) for k in FX["nearest_cases"]]
LINEAR = [str(k) for k in FX["linear_cases"]]


def geometry(key):
    W, H, ds = map(int, re.match(r"\w+?_(\d+)x(\d+)_ds(\d+)", key).groups())
    return W, H, W // ds, H // ds


def oracle_map(sw, sh, texW, texH, dw, dh, surfW, surfH, linear):
    L = load_oracle()
    L.post_blit_map.argtypes = [ctypes.c_int] * 9 + [ctypes.c_void_p] * 4
    ix, iy = np.zeros((dh, dw), np.int32), np.zeros((dh, dw), np.int32)
    wx, wy = np.zeros((dh, dw), np.float32), np.zeros((dh, dw), np.float32)
    L.post_blit_map(sw, sh, texW, texH, dw, dh, surfW, surfH, int(linear), ix.ctypes.data, iy.ctypes.data, wx.ctypes.data, wy.ctypes.data)
    return ix, iy, wx, wy


def test_fixture_contains_the_boundary_cases():
    assert len(NEAREST) >= 12 and len(LINEAR) >= 4
    x = FX["nearest_1280x720_ds20_x"]
    assert (x == 20 * np.arange(64)[None, :] + 9).all()          # never the 20x + 10 of exact arithmetic
    x = FX["nearest_1920x1080_ds10_x"]
    assert 0.1 < (x == 10 * np.arange(192)[None, :] + 4).mean() < 0.9      # mixed: decided per column by rounding


@pytest.mark.parametrize("key", NEAREST)
def test_nearest_down_blit_reads_llvmpipes_texels(key):
    W, H, w, h = geometry(key)
    ix, iy, _, _ = oracle_map(W, H, W, H, w, h, W, H, False)
    assert (ix == FX[key + "_x"]).all() and (iy == FX[key + "_y"]).all()


@pytest.mark.parametrize("key", LINEAR)
def test_linear_up_blit_footprint_and_weights(key):
    W, H, w, h = geometry(key)
    ix, iy, wx, wy = oracle_map(w, h, W, H, W, H, W, H, True)
    assert (ix == FX[key + "_ix"]).all() and (iy == FX[key + "_iy"]).all()
    assert (wx.view(np.uint32) == FX[key + "_wx"].view(np.uint32)).all() and (wy.view(np.uint32) == FX[key + "_wy"].view(np.uint32)).all()


@pytest.mark.parametrize("key", NEAREST + LINEAR)
def test_product_planes_equal_the_restatements(key):
    W, H, w, h = geometry(key)
    args = (W, H, W, H, w, h, W, H) if key.startswith("nearest") else (w, h, W, H, W, H, W, H)
    L = load_oracle()
    L.post_blit_planes.argtypes = [ctypes.c_int] * 8 + [ctypes.c_void_p]
    want, got = np.zeros(12, np.float32), np.zeros(12, np.float32)
    L.post_blit_planes(*args, want.ctypes.data)
    assert _lib.lib().rt_probe_blit_planes(*args, got.ctypes.data) == 0
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


def test_probe_rejects_bad_geometry():
    out = np.zeros(12, np.float32)
    assert _lib.lib().rt_probe_blit_planes(0, 1, 1, 1, 1, 1, 1, 1, out.ctypes.data) != 0
    assert _lib.lib().rt_probe_blit_planes(1, 1, 1, 1, 1, 1, 1, 1, None) != 0
