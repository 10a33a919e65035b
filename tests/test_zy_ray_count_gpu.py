"""More than one ray per pixel and frame through the C ABI (rt_set_ray_count = the shader's compile-time RAY_COUNT,
Fragment.glsl:5,459-485): the kernels built for it against the llvmpipe fixtures rendered with that constant changed,
and against the oracle. Bit-exact. The bodies need no torch device except where they say so, so tests/test_emulated.py
runs them on the host emulation of the library as well."""
import os

import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import scenes, sharding
from rt_oracle import Oracle
from test_oracle_golden import RAY_FIXTURES, load_into, ray_count_cases
from util import bits, differing_pixels, make_scene, setup

pytestmark = pytest.mark.gpu

NOISE = scenes.noise_texture(64, 48)
SKY = scenes.sky_cubemap(32)
SEEDS = scenes.frame_seeds(8)


def _emulated():
    from raytracing_amd import _lib
    return _lib.lib()._name.endswith("librt_emulated.so")


@pytest.mark.parametrize("path", RAY_FIXTURES, ids=[os.path.basename(f)[9:-4] for f in RAY_FIXTURES])
def test_reproduces_reference_fixture(path):
    fx = np.load(path)
    W, H = int(fx["width"]), int(fx["height"])
    g = ra.Renderer(W, H)
    load_into(g, fx)
    n = 0
    for key, rays, bl, frames in ray_count_cases(fx):
        g.set_ray_count(rays)
        g.set_bounce_limit(bl)
        g.clear()
        g.render(0, fx["seeds"][:frames])                     # all frames in one launch
        assert (bits(g.read_accum()[..., :3]) == bits(fx[key])).all(), key
        g.clear()
        for k in range(frames):                               # one launch per frame, as the reference draws
            g.render(k, fx["seeds"][k:k + 1])
        assert (bits(g.read_accum()[..., :3]) == bits(fx[key])).all(), key + " (frame by frame)"
        n += 1
    assert n >= 1
    g.close()


def test_counters_shards_and_return_to_one_ray(W=56, H=27, ray_counts=(2, 7)):
    sc = make_scene("sponza_like", target_tris=6000)
    g, o = ra.Renderer(W, H), Oracle(W, H)
    setup(g, sc, NOISE, SKY, bounce_limit=8)
    setup(o, sc, NOISE, SKY, bounce_limit=8)
    for rays in ray_counts:
        for r in (g, o):
            r.set_ray_count(rays)
            r.clear()
            r.counters(reset=True)
        g.set_counting(True)
        g.render(0, SEEDS[:3]); o.render(0, SEEDS[:3])
        whole = g.read_accum()
        assert differing_pixels(whole, o.read_accum()) == 0
        cg, co = g.counters(reset=True), o.counters(reset=True)
        for k in ("rays", "node_pops", "boxes_tested", "tris_tested", "noise_fetches", "sky_fetches", "hits"):
            assert cg[k] == co[k], (rays, k)
        # every pixel traces between 1 and `rays` paths per frame (the loop's early-out skips some)
        assert W * H * 3 < cg["paths"] <= W * H * 3 * rays
        g.set_counting(False)
        g.clear()
        g.render(0, SEEDS[:1]); g.render(1, SEEDS[1:3])
        assert differing_pixels(whole, g.read_accum()) == 0
        c2 = g.counters(reset=True)
        assert c2["paths"] == cg["paths"] and c2["rays"] == cg["rays"]      # without the full counters the kernel still counts its paths

        # three shards, each traced on its own context, assembled = unsharded
        world = 3
        per = sharding.shard_floats(W, H, world)
        flat = np.zeros(world * per, dtype=np.float32)
        for rank in range(world):
            s = ra.Renderer(W, H)
            s.set_shard(rank, world)
            setup(s, sc, NOISE, SKY, bounce_limit=8)
            s.set_ray_count(rays)
            s.clear()
            s.render(0, SEEDS[:3])
            if _emulated():
                part = np.empty(per, dtype=np.float32)
                s.export_shard(part.ctypes.data); s.sync()
            else:
                import torch
                t = torch.empty(per, dtype=torch.float32, device="cuda")
                s.export_shard(t.data_ptr()); s.sync()
                part = t.cpu().numpy()
            flat[rank * per:(rank + 1) * per] = part
            s.close()
        assert differing_pixels(sharding.assemble_numpy(flat, W, H, world), whole) == 0

    # back to the reference's setting: the same image as a context that never left it
    for r in (g, o):
        r.set_ray_count(1)
        r.clear()
        r.render(0, SEEDS[:2])
    assert differing_pixels(g.read_accum(), o.read_accum()) == 0
    assert (g.primary_ids(SEEDS[0]) == o.primary_ids(SEEDS[0])).all()
    with pytest.raises(ra.RtError, match="outside 1"):
        g.set_ray_count(0)
    with pytest.raises(ra.RtError, match="outside 1"):
        g.set_ray_count(65)
    g.close()


def test_many_frames_of_many_rays_in_one_launch(frames=70):
    """The table of per-ray jitters behind the frame records grows with frames x rays: 70 frames of 64 rays (and already 6)
    need more than the 64 records a context starts with."""
    W, H = 8, 8
    sc = make_scene("lightbox")
    g, o = ra.Renderer(W, H), Oracle(W, H)
    setup(g, sc, NOISE, SKY, bounce_limit=4)
    setup(o, sc, NOISE, SKY, bounce_limit=4)
    seeds = scenes.frame_seeds(frames)
    for r in (g, o):
        r.set_ray_count(64)
        r.clear()
        r.render(0, seeds)
    assert differing_pixels(g.read_accum(), o.read_accum()) == 0
    g.close()


def test_random_configurations_with_several_rays_bit_exact(cases=16):
    """The randomised parity check (random scene, image size, camera, bounce limit, launch split, 1-5 shards), every case
    with a random number of rays per pixel from 2 to 9."""
    import fuzz_parity
    lines = []
    bad = fuzz_parity.run(cases, 4242, log=lines.append, host_buffers=_emulated(), ray_counts=True)
    assert bad == 0, "\n".join(l for l in lines if "DIFFERS" in l)
