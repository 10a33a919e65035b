"""GPU path against the llvmpipe-rendered fixtures directly (no oracle in between). Bit-exact."""
import glob
import os

import numpy as np
import pytest

import raytracing_amd as ra
from test_oracle_golden import FIXTURES, load_into, radiance_cases
from util import bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f)[7:-4] for f in FIXTURES])
def test_gpu_reproduces_reference_fixture(path):
    fx = np.load(path)
    W, H = int(fx["width"]), int(fx["height"])
    g = ra.Renderer(W, H)
    load_into(g, fx)
    g.set_bounce_limit(1)
    ids = g.primary_ids(fx["seeds"][0])
    assert (ids[..., 0] == fx["ids_index"]).all()
    assert (ids[..., 1].view(np.uint32) == bits(fx["ids_t"])).all()
    for bl, key in ((0, "heat"), (1, "normals")):
        g.clear()
        g.set_bounce_limit(bl)
        g.render(0, fx["seeds"][:1])
        same = (bits(g.read_accum()[..., :3]) == bits(fx[key])).all(-1)
        if key == "normals":
            same |= fx["ids_hit"] == 0
        assert same.all(), key
    for key, bl, frames in radiance_cases(fx):
        g.clear()
        g.set_bounce_limit(bl)
        g.render(0, fx["seeds"][:frames])
        assert (bits(g.read_accum()[..., :3]) == bits(fx[key])).all(), key
    g.close()
