"""GPU parity at the benchmark's full size: the 3840x2160, 8-bounce workload bench.py times.

At this size the checks are the ones that do not get more expensive than the render itself:
  * one frame against the CPU oracle, bit for bit (a few seconds on the GPU box's host cores);
  * properties that hold for any image size: the same seeds give the same image, frames rendered one launch
    at a time equal the same frames in one launch, an image assembled from 3 or 8 tile shards equals the
    unsharded image, and the device counters satisfy their identities (one path per pixel and frame, at least
    one ray per path, no more hits than rays).
Bar: bit-exact (tolerance 0), as for the small cases in test_gpu_parity.py."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import raytracing_amd as ra
from raytracing_amd import scenes, sharding
from rt_oracle import Oracle
from util import differing_pixels

pytestmark = pytest.mark.gpu

WORKLOAD = "sponza_like_3840x2160_b8"
FRAMES = 3


@pytest.fixture(scope="module")
def workload():
    import bench
    sc, W, H, bl = bench.load_workload(WORKLOAD, threads=min(16, os.cpu_count() or 1))
    noise, sky = scenes.noise_texture(), scenes.sky_cubemap()
    return bench, sc, W, H, bl, noise, sky, scenes.frame_seeds(FRAMES)


@pytest.fixture(scope="module")
def whole(workload):
    """The unsharded image of FRAMES frames in one launch, with its counters."""
    bench, sc, W, H, bl, noise, sky, seeds = workload
    r = ra.Renderer(W, H)
    bench.setup(r, sc, noise, sky, bl)
    r.set_counting(True)
    r.counters(reset=True)
    r.render(0, seeds)
    img = r.read_accum()
    counters = r.counters(reset=True)
    r.set_counting(False)
    yield r, img, counters
    r.close()


def test_size_is_the_benchmark_size(workload):
    _, sc, W, H, bl, *_ = workload
    assert (W, H, bl) == (3840, 2160, 8)
    assert sc.n_triangles > 200_000


def test_first_frame_equals_oracle(workload):
    """One whole 4K frame, bit for bit against the CPU restatement of the reference shader."""
    bench, sc, W, H, bl, noise, sky, seeds = workload
    g, o = ra.Renderer(W, H), Oracle(W, H, threads=min(16, os.cpu_count() or 1))
    bench.setup(g, sc, noise, sky, bl)
    o.set_scene(sc.tri12, sc.matidx, sc.node8)
    o.set_materials(sc.materials[:16])
    fov = np.float32(120 * 0.01745329251994329576923690768489)
    o.set_camera((0, 0, 0), (1, 0, 0), float(fov), float(np.float32(2.0 * np.tan(float(fov) / 2.0))))
    o.set_noise(noise)
    o.set_skybox(sky)
    o.set_bounce_limit(bl)
    g.render(0, seeds[:1])
    o.render(0, seeds[:1])
    a, b = g.read_accum(), o.read_accum()
    assert np.isfinite(b[..., :3]).all()
    assert differing_pixels(a, b) == 0
    g.close()


def test_counter_identities(workload, whole):
    _, _, W, H, *_ = workload
    _, _, c = whole
    assert c["paths"] == W * H * FRAMES
    assert c["rays"] >= c["paths"]
    assert c["hits"] <= c["rays"]
    assert c["boxes_tested"] >= c["rays"]          # every ray tests at least the root box
    assert c["sky_fetches"] > 0 and c["noise_fetches"] >= c["paths"]


def test_same_seeds_same_image(workload, whole):
    bench, sc, W, H, bl, noise, sky, seeds = workload
    _, want, _ = whole
    r = ra.Renderer(W, H)
    bench.setup(r, sc, noise, sky, bl)
    r.render(0, seeds)
    assert differing_pixels(r.read_accum(), want) == 0
    r.close()


def test_split_launches_equal_one_launch(workload, whole):
    _, _, _, _, _, _, _, seeds = workload
    r, want, _ = whole
    r.clear()
    for k in range(FRAMES):
        r.render(k, seeds[k:k + 1])
    assert differing_pixels(r.read_accum(), want) == 0


@pytest.mark.parametrize("world", [3, 8])
def test_sharded_equals_unsharded(workload, whole, world):
    import torch
    bench, sc, W, H, bl, noise, sky, seeds = workload
    _, want, counters = whole
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    per = sharding.shard_floats(W, H, world)
    flat = torch.zeros(world * per, dtype=torch.float32, device="cuda")
    r = ra.Renderer(W, H)
    bench.setup(r, sc, noise, sky, bl)
    total_rays = 0
    for rank in range(world):
        r.set_shard(rank, world)
        assert r.shard_tiles() * 64 * 4 == per
        r.clear()
        r.counters(reset=True)
        r.render(0, seeds)
        r.sync()
        total_rays += r.counters()["rays"]
        assert hip.hipMemcpy(flat.data_ptr() + rank * per * 4, r.shard_device_ptr(), per * 4, 3) == 0   # device to device
    r.close()
    torch.cuda.synchronize()
    root = ra.Renderer(W, H)
    root.assemble(flat.data_ptr(), world)
    assert differing_pixels(root.read_accum(), want) == 0
    assert total_rays == counters["rays"]
    root.close()


@pytest.mark.parametrize("name", ["teapot_like_1920x1080_b8", "dragon_like_1920x1080_b16"])
def test_other_full_size_workloads_equal_oracle(name):
    """BASELINE configurations 2 and 3 at their own size: one 1920x1080 frame and the primary-hit ids, bit for bit.
    The dragon-like tree (about 870 000 triangles, over 30 levels) makes rays spill stack entries to memory on real
    geometry; its glass paths run to hundreds of segments."""
    import bench
    threads = min(16, os.cpu_count() or 1)
    sc, W, H, bl = bench.load_workload(name, threads=threads)
    assert (W, H) == (1920, 1080)
    if name.startswith("dragon"):
        assert sc.n_triangles > 800_000 and sc.stats["max_depth"] > 12 + 8    # deeper than the on-chip stack entries
    noise, sky, seeds = scenes.noise_texture(), scenes.sky_cubemap(), scenes.frame_seeds(1)
    g, o = ra.Renderer(W, H), Oracle(W, H, threads=threads)
    bench.setup(g, sc, noise, sky, bl)
    o.set_scene(sc.tri12, sc.matidx, sc.node8)
    o.set_materials(sc.materials[:16])
    fov = np.float32(120 * 0.01745329251994329576923690768489)
    o.set_camera((0, 0, 0), (1, 0, 0), float(fov), float(np.float32(2.0 * np.tan(float(fov) / 2.0))))
    o.set_noise(noise)
    o.set_skybox(sky)
    o.set_bounce_limit(bl)
    g.render(0, seeds)
    o.render(0, seeds)
    a, b = g.read_accum(), o.read_accum()
    assert np.isfinite(b[..., :3]).all()
    assert differing_pixels(a, b) == 0
    ids_g, ids_o = g.primary_ids(seeds[0]), o.primary_ids(seeds[0])
    assert (ids_g[..., 0] == ids_o[..., 0]).all(), "triangle ids differ"
    assert (ids_g[..., 1] == ids_o[..., 1]).all(), "hit distances differ"
    assert (ids_g[..., 2] == ids_o[..., 2]).all(), "iterationsCount differs"
    assert (ids_g[..., 0] >= 0).mean() > 0.05
    g.close()
