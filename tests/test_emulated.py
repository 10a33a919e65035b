"""The product's own kernels and host code on the CPU: the HIP sources compiled as host C++ against stand-in headers
(tests/emu), lanes run as fibers that meet at every wave-wide operation, checked against the same oracle and to the same
bar (bit for bit) as the GPU tests. What this covers that the oracle-only CPU tests cannot: the scheduler, the on-chip and
spilled stack, the tile queues, record building, sharding and the exchange between devices (with an in-process imitation
of RCCL). What it cannot cover: the compiler's device code, the hardware's division and square-root sequences
(exact_div.h; the host build divides), timing. Sizes are small: a lane here costs about a microsecond per step."""
import numpy as np
import pytest

import raytracing_amd as ra
from emulation import Emulated
from raytracing_amd import scenes, sharding
from rt_oracle import Oracle, PostOracle
from test_gpu_parity import _chain_scene
from test_post import TOL_GPU_ORACLE, noisy_hdr
from util import DEFAULT_CAMERA, bits, differing_pixels, make_scene, setup

NOISE = scenes.noise_texture(64, 48)
SKY = scenes.sky_cubemap(32)
SEEDS = scenes.frame_seeds(12)

CASES = [
    ("lightbox", {}, 40, 24),
    ("sponza_like", {"target_tris": 6000}, 45, 27),      # neither side a multiple of 8
]


@pytest.fixture(scope="module", autouse=True)
def _emulated_module():
    with Emulated(devices=3):
        yield


@pytest.fixture(scope="module", params=CASES, ids=[c[0] for c in CASES])
def pair(request):
    name, kw, W, H = request.param
    sc = make_scene(name, **kw)
    g, o = ra.Renderer(W, H), Oracle(W, H)
    setup(g, sc, NOISE, SKY)
    setup(o, sc, NOISE, SKY)
    yield g, o, sc
    g.close()


def test_the_emulation_itself():
    """Wave- and block-wide operations return what the hardware returns (ballot over running lanes only, shuffle, first
    lane, mbcnt, a barrier that orders two waves); lanes that reach different wave-wide operations, and waves that can never
    complete one, end the process with a message instead of a wrong answer."""
    import os
    import subprocess
    from emulation import EMU_BUILD, EMU_DIR, library
    library()
    exe = os.path.join(EMU_DIR, EMU_BUILD, "selftest")
    ok = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and "0 wrong values" in ok.stdout, ok.stdout + ok.stderr
    for mode, message in (("divergent", "different wave-wide operations"), ("deadlock", "deadlock")):
        r = subprocess.run([exe, mode], capture_output=True, text=True, timeout=120)
        assert r.returncode == -6 and message in r.stderr, (mode, r.returncode, r.stderr)


def test_it_is_the_emulated_library_that_runs():
    from raytracing_amd import _lib
    assert _lib.lib()._name.endswith("librt_emulated.so")


def test_primary_ids_bit_exact(pair):
    g, o, _ = pair
    a, b = g.primary_ids(SEEDS[0]), o.primary_ids(SEEDS[0])
    assert (a == b).all()
    assert (a[..., 0] >= 0).any()


@pytest.mark.parametrize("bounce_limit,frames", [(0, 1), (1, 1), (2, 3), (8, 2), (16, 1)])
def test_radiance_bit_exact(pair, bounce_limit, frames):
    g, o, _ = pair
    for r in (g, o):
        r.set_bounce_limit(bounce_limit)
        r.clear()
        r.render(0, SEEDS[:frames])
    a, b = g.read_accum(), o.read_accum()
    assert np.isfinite(b[..., :3]).all()
    assert differing_pixels(a, b) == 0
    assert (a[..., 3] == 0).all()


def test_progressive_split_equals_batch_and_counters_match(pair):
    g, o, _ = pair
    for r in (g, o):
        r.set_bounce_limit(4)
        r.clear()
    g.set_counting(True)
    g.counters(reset=True); o.counters(reset=True)
    g.render(0, SEEDS[:4]); o.render(0, SEEDS[:4])
    whole = g.read_accum()
    cg, co = g.counters(reset=True), o.counters(reset=True)
    g.set_counting(False)
    for k in ("rays", "node_pops", "boxes_tested", "tris_tested", "noise_fetches", "sky_fetches", "hits"):
        assert cg[k] == co[k], k
    assert cg["paths"] == g.W * g.H * 4
    assert sum(cg["sched_lanes"]) > 0 and cg["sched_rounds"][1] > 0
    g.clear()
    for k in range(4):
        g.render(k, SEEDS[k:k + 1])
    assert differing_pixels(whole, g.read_accum()) == 0
    assert differing_pixels(whole, o.read_accum()) == 0
    assert g.counters(reset=True)["rays"] == co["rays"]


def test_empty_and_single_leaf_scenes():
    W, H = 24, 16
    g, o = ra.Renderer(W, H), Oracle(W, H)
    one = np.array([[4, -1, -1, 0, 4, 1, -1, 0, 4, 0, 1, 0]], dtype=np.float32)
    node = np.array([[4, 1, 1, 4, -1, -1, 1, 0]], dtype=np.float32)
    mats = np.array([[0.8, 0.3, 0.2, 1, 1, 1, 1, 0, 0, 0, 0]], dtype=np.float32)
    for tri12, matidx, node8 in ((np.zeros((0, 12), np.float32), np.zeros(0, np.int32), np.zeros((0, 8), np.float32)),
                                 (one, np.zeros(1, np.int32), node)):
        for r in (g, o):
            r.set_scene(tri12, matidx, node8)
            r.set_materials(mats)
            r.set_camera(**DEFAULT_CAMERA)
            r.set_noise(NOISE)
            r.set_skybox(SKY)
        for bl in (0, 1, 4):
            for r in (g, o):
                r.set_bounce_limit(bl)
                r.clear()
                r.render(0, SEEDS[:2])
            assert differing_pixels(g.read_accum(), o.read_accum()) == 0
        assert (g.primary_ids(SEEDS[0]) == o.primary_ids(SEEDS[0])).all()
    g.close()


def test_deepest_tree_spills_the_stack_bit_exact():
    """Interior nodes 63 levels deep, both children entered at every level: 51 of a lane's deferred nodes live in the
    spill area (Fragment.glsl:9,270); one level more is refused at upload."""
    tri12, matidx, node8 = _chain_scene(63)
    mats = np.array([[0.8, 0.6, 0.3, 1, 1, 1, 1, 0, 0, 0, 0]], dtype=np.float32)
    W, H = 24, 16
    camera = dict(DEFAULT_CAMERA, viewport_width=0.5)
    g, o = ra.Renderer(W, H), Oracle(W, H)
    for r in (g, o):
        r.set_scene(tri12, matidx, node8)
        r.set_materials(mats)
        r.set_camera(**camera)
        r.set_noise(NOISE)
        r.set_skybox(SKY)
        r.set_bounce_limit(4)
    a, b = g.primary_ids(SEEDS[0]), o.primary_ids(SEEDS[0])
    assert (a == b).all()
    assert (a[..., 0] == 63).all() and (a[..., 2] >= 2 * 63).all()
    for r in (g, o):
        r.render(0, SEEDS[:2])
    assert differing_pixels(g.read_accum(), o.read_accum()) == 0
    with pytest.raises(ra.RtError, match="63"):
        g.set_scene(*_chain_scene(64))
    g.close()


def test_glass_deck_bit_exact():
    """A deck of parallel quads, some of glass: long paths through many overlapping boxes, deferred nodes beyond the
    on-chip entries on the way."""
    n = 6000
    x = (1.0 + 0.003 * np.arange(n, dtype=np.float64)).astype(np.float32)
    lo, hi = np.float32(-6.0), np.float32(6.0)
    tri9 = np.zeros((2 * n, 9), dtype=np.float32)
    tri9[0::2] = np.stack([x, np.full(n, lo), np.full(n, lo), x, np.full(n, hi), np.full(n, lo), x, np.full(n, hi), np.full(n, hi)], 1)
    tri9[1::2] = np.stack([x, np.full(n, lo), np.full(n, lo), x, np.full(n, hi), np.full(n, hi), x, np.full(n, lo), np.full(n, hi)], 1)
    mats = np.zeros((2, 11), dtype=np.float32)
    mats[0] = [0.8, 0.7, 0.6, 1, 1, 1, 1.0, 0.0, 0.0, 0.0, 0]
    mats[1] = [0.9, 0.9, 1.0, 1, 1, 1, 0.0, 0.5, 0.9, 1.5, 0]
    slots = (np.arange(2 * n) // 2 % 7 != 0).astype(np.int32)
    sc = ra.build_scene(tri9, slots, mats, build_threads=2)
    assert sc.stats["max_depth"] >= 13
    W, H = 32, 16
    g, o = ra.Renderer(W, H), Oracle(W, H)
    for r in (g, o):
        setup(r, sc, NOISE, SKY, bounce_limit=3)
        r.render(0, SEEDS[:2])
    assert differing_pixels(g.read_accum(), o.read_accum()) == 0
    assert (g.primary_ids(SEEDS[0]) == o.primary_ids(SEEDS[0])).all()
    g.close()


@pytest.mark.parametrize("size", [(1, 1), (8, 70), (5, 33), (150, 9), (1, 65)], ids=lambda s: "%dx%d" % s)
def test_extreme_image_shapes_alone_and_in_three_shards(size):
    W, H = size
    sc = make_scene("sponza_like", target_tris=6000)
    o = Oracle(W, H)
    setup(o, sc, NOISE, SKY, bounce_limit=3)
    o.render(0, SEEDS[:2])
    want = o.read_accum()
    g = ra.Renderer(W, H)
    setup(g, sc, NOISE, SKY, bounce_limit=3)
    g.render(0, SEEDS[:2])
    assert differing_pixels(g.read_accum(), want) == 0
    g.close()
    world = 3
    per = sharding.shard_floats(W, H, world)
    flat = np.zeros(world * per, dtype=np.float32)
    for rank in range(world):
        r = ra.Renderer(W, H)
        setup(r, sc, NOISE, SKY, bounce_limit=3)
        r.set_shard(rank, world)
        r.render(0, SEEDS[:2])
        part = np.empty(per, dtype=np.float32)
        r.export_shard(part.ctypes.data); r.sync()          # "device" memory is host memory here
        flat[rank * per:(rank + 1) * per] = part
        r.close()
    assert differing_pixels(sharding.assemble_numpy(flat, W, H, world), want) == 0


def test_tile_order_feedback_changes_nothing(pair):
    """After the first launches the tiles are handed out most expensive first (costs from earlier launches, sorted on
    the device): the image must not depend on it."""
    g, o, _ = pair
    for r in (g, o):
        r.set_bounce_limit(4)
        r.clear()
    for k in range(10):
        g.render(k, SEEDS[k:k + 1])
    o.render(0, SEEDS[:10])
    assert differing_pixels(g.read_accum(), o.read_accum()) == 0


def test_group_of_three_devices_equals_unsharded():
    """rt_group with three ranks: sharded contexts, staging buffers, sends and receives (imitated RCCL: a send without its
    receive, or sizes that differ, is an error), double-buffered receive, assembly on the root."""
    W, H = 52, 36
    sc = make_scene("sponza_like", target_tris=6000)
    g = ra.Group([0, 1, 2], W, H)
    assert len(g.ranks) == 3
    for r in g.ranks:
        setup(r, sc, NOISE, SKY, bounce_limit=4)
    o = Oracle(W, H)
    setup(o, sc, NOISE, SKY, bounce_limit=4)
    g.render(0, SEEDS[:2]); g.gather(); g.sync()
    o.render(0, SEEDS[:2])
    assert differing_pixels(g.root.read_accum(), o.read_accum()) == 0
    for k in range(2, 7):                       # render + gather pairs queued back to back: both halves of every double buffer
        g.render(k, SEEDS[k:k + 1])
        g.gather()
    g.sync()
    o.render(2, SEEDS[2:7])
    assert differing_pixels(g.root.read_accum(), o.read_accum()) == 0
    g.render(7, SEEDS[7:12]); g.gather(); g.sync()      # a multi-frame launch between gathers
    o.render(7, SEEDS[7:12])
    assert differing_pixels(g.root.read_accum(), o.read_accum()) == 0
    assert sum(r.counters()["rays"] for r in g.ranks) == o.counters()["rays"]
    rgba, _ = g.root.postprocess(True)
    assert np.isfinite(rgba).all()
    with pytest.raises(ra.RtError):
        ra.Group([0, 0], W, H)
    with pytest.raises(ra.RtError):
        ra.Group([0, 3], W, H)                  # the imitated runtime has three devices
    g.close()


@pytest.mark.parametrize("size", [(96, 54), (33, 17)])
def test_bloom_chain_matches_oracle(size):
    """The post kernels (threshold, blits, blur, composite) on the host: same tolerance as on the GPU; pow() is libm's
    here, the hardware's exp2/log2 there."""
    w, h = size
    g, po = ra.Renderer(w, h), PostOracle(w, h)
    for f in range(3):
        a = noisy_hdr(w, h, 100 + f)
        g.write_accum(a)
        rgba, rgb8 = g.postprocess(True)
        o_rgba, o_rgb8 = po.run(a)
        assert np.abs(rgba[..., :3] - o_rgba[..., :3]).max() <= TOL_GPU_ORACLE, f
        assert np.abs(rgb8.astype(int) - o_rgb8.astype(int)).max() <= 1
    raw, raw8 = g.postprocess(False)
    assert (bits(raw[..., :3]) == bits(a[..., :3])).all()
    g.close()


def test_entry_point_that_allocates_nothing_equals_the_ordinary_one():
    """rt_postprocess_static on two images in a row, the second smaller (both share the library's fixed array), against
    rt_postprocess: the same kernels on the same values, so bit for bit."""
    import ctypes
    from raytracing_amd import _lib
    from test_zz_no_alloc import hdr_image, reference_chain
    L = _lib.lib()
    for img in (hdr_image(160, 90, 1), hdr_image(51, 29, 2), hdr_image(3, 2, 4)):
        h, w = img.shape[:2]
        rgba = np.full((h, w, 4), -1.0, dtype=np.float32)
        rgb8 = np.full((h, w, 3), 7, dtype=np.uint8)
        rc = L.rt_postprocess_static(img.ctypes.data, w, h, rgba.ctypes.data, rgb8.ctypes.data, None)
        assert rc == 0, L.rt_last_global_error().decode()
        want_rgba, want_rgb8 = reference_chain(img)
        assert (bits(rgba) == bits(want_rgba)).all()
        assert (rgb8 == want_rgb8).all()
    assert L.rt_postprocess_static(img.ctypes.data, 8000, 4500, None, rgb8.ctypes.data, None) == -4
    assert L.rt_postprocess_static(None, 3, 2, None, rgb8.ctypes.data, None) == -1


def test_tallies_are_folded_before_they_can_wrap():
    """The per-wave tally columns are 32 bits wide and folded into 64-bit totals every 4096 tile-frames per wave: a fold
    before a launch that would cross the line, one after a launch that crossed it, and a counting launch far beyond it
    refused. One tile, an empty scene: every path is one ray into the sky."""
    W = H = 8
    g = ra.Renderer(W, H)
    g.set_scene(np.zeros((0, 12), np.float32), np.zeros(0, np.int32), np.zeros((0, 8), np.float32))
    g.set_materials(np.array([[0.8, 0.3, 0.2, 1, 1, 1, 1, 0, 0, 0, 0]], dtype=np.float32))
    g.set_camera(**DEFAULT_CAMERA)
    g.set_noise(NOISE)
    g.set_skybox(SKY)
    g.set_bounce_limit(4)
    seeds = scenes.frame_seeds(9000)
    g.counters(reset=True)
    first = 0
    for n in (1500, 1500, 1500, 4200, 3):
        g.render(first, seeds[first:first + n])
        first += n
    c = g.counters(reset=True)
    assert c["rays"] == first * W * H and c["paths"] == first * W * H
    g.set_counting(True)
    g.render(first, seeds[first:first + 200])
    c = g.counters(reset=True)
    assert c["rays"] == c["paths"] == c["sky_fetches"] // 2 == 200 * W * H      # primary rays that miss look the sky up twice
    with pytest.raises(ra.RtError, match="split the launch"):
        g.render(0, np.zeros((8 * 4096 + 1, 3), np.float32))
    g.set_counting(False)
    g.close()


def test_headless_executable_on_three_imitated_devices_writes_the_same_image(tmp_path):
    """The C++ host program (ini -> PNG textures -> objects -> frames -> bloom -> PNG) linked against the emulated library:
    a plain context against --gpus 3 (rt_group: three sharded ranks, gather to the root context, bloom chain there), with
    the scene cache on."""
    import os
    import subprocess
    from PIL import Image
    from emulation import EMU_DIR, EMU_BUILD
    W, H = 48, 32
    objects = scenes.write_objects(scenes.lightbox(), str(tmp_path / "objs"))
    root = tmp_path / "install"
    (root / "Sky").mkdir(parents=True)
    for f, name in enumerate(("right", "left", "top", "bottom", "front", "back")):
        Image.fromarray(SKY[f], "RGBA").save(str(root / "Sky" / (name + ".png")))
    Image.fromarray(NOISE, "RGB").save(str(root / "noise.png"))
    scenes.write_ini(str(root / "RayTracer.ini"), os.path.dirname(objects[0]), [os.path.basename(p) for p in objects], skybox_path=str(root / "Sky"))
    exe = os.path.join(EMU_DIR, EMU_BUILD, "raytracer_headless_emulated")
    common = [exe, "--ini", str(root / "RayTracer.ini"), "--noise", str(root / "noise.png"), "--width", str(W), "--height", str(H),
              "--cache", str(tmp_path / "cache")]
    env = dict(os.environ, RT_EMU_DEVICES="3")
    images = {}
    # (with bloom the plain loop runs the chain per displayed frame and carries its tmp texture over, the group once on the
    # gathered image: same image for one frame only; without bloom for any number of frames)
    for tag, extra in (("bloom1", ["--frames", "1"]), ("plain5", ["--frames", "5", "--no-bloom"]), ("rays3", ["--frames", "2", "--no-bloom", "--ray-count", "3"])):
        one = subprocess.run(common + extra + ["--out", str(tmp_path / ("one" + tag))], capture_output=True, text=True, timeout=300, env=env)
        assert one.returncode == 0, one.stdout + one.stderr
        grp = subprocess.run(common + extra + ["--gpus", "3", "--out", str(tmp_path / ("grp" + tag))], capture_output=True, text=True, timeout=300, env=env)
        assert grp.returncode == 0, grp.stdout + grp.stderr
        assert "on 3 GPUs" in grp.stdout and len(os.listdir(str(tmp_path / "cache"))) == 1
        a, b = np.asarray(Image.open(str(tmp_path / ("one" + tag + ".png")))), np.asarray(Image.open(str(tmp_path / ("grp" + tag + ".png"))))
        assert a.shape == (H, W, 3) and a.max() > 40
        assert (a == b).all(), tag
        images[tag] = a
    assert (images["rays3"] != images["plain5"]).any()          # (the option reached the kernels)
    extra = ["--frames", "1"]
    four = subprocess.run(common + extra + ["--gpus", "4", "--out", str(tmp_path / "four")], capture_output=True, text=True, timeout=300, env=env)
    assert four.returncode != 0 and "device" in (four.stdout + four.stderr).lower()      # more ranks than devices: refused, not crashed


# ---- the bodies of GPU tests that need no torch device, run here against the emulated library ----
import test_app_gpu
import test_gpu_golden
import test_post
import test_video_gpu


@pytest.mark.parametrize("path", test_gpu_golden.FIXTURES, ids=[p.split("tracer_")[-1][:-4] for p in test_gpu_golden.FIXTURES])
def test_kernels_reproduce_the_llvmpipe_fixtures(path):
    """Ids, hit distances, heat map, normals and multi-frame radiance as the reference's shader rendered them on llvmpipe:
    the product's kernel source, on the host, bit for bit."""
    test_gpu_golden.test_gpu_reproduces_reference_fixture(path)


def test_reference_frame_loop_tracer_and_bloom_chain():
    """The reference's main loop on the light box (tracer draw + bloom chain per frame, 16 frames): tracer bit-exact, chain
    within the GPU test's tolerance of llvmpipe's output."""
    test_post.test_gpu_frame_loop_matches_reference()
    test_post.test_gpu_post_matches_reference_fixture()


def test_headless_app_ticks_keys_and_recorder(tmp_path):
    objects = scenes.write_objects(scenes.lightbox(), str(tmp_path / "objs"))
    test_app_gpu.test_ticks_equal_direct_rendering(objects)
    test_app_gpu.test_keys_restart_accumulation(objects)


def test_offline_video_path_matches_tracer_and_post_oracle(tmp_path):
    test_video_gpu.test_every_pose_matches_tracer_and_post_oracle(tmp_path)


import test_zy_ray_count_gpu


@pytest.mark.parametrize("path", test_zy_ray_count_gpu.RAY_FIXTURES, ids=[p.split("raycount_")[-1][:-4] for p in test_zy_ray_count_gpu.RAY_FIXTURES])
def test_kernels_reproduce_the_llvmpipe_fixtures_with_several_rays_per_pixel(path):
    """The shader's RAY_COUNT above 1 (rt_set_ray_count): what llvmpipe rendered with the reference's constant changed."""
    test_zy_ray_count_gpu.test_reproduces_reference_fixture(path)


def test_several_rays_per_pixel_counters_shards_and_long_launches():
    test_zy_ray_count_gpu.test_counters_shards_and_return_to_one_ray(W=27, H=13, ray_counts=(3,))
    test_zy_ray_count_gpu.test_many_frames_of_many_rays_in_one_launch(frames=6)


def test_the_drivers_smoke_entry_point(capsys):
    """__graft_entry__.smoke(), the call the driver makes on a GPU before the benchmark, with the emulated library."""
    import __graft_entry__
    __graft_entry__.smoke()
    assert "pixels differing from the oracle: 0; primary ids equal: True" in capsys.readouterr().out


@pytest.mark.parametrize("cus,threads", [(1, 1), (3, 2), (40, 4)])
def test_any_grid_size_gives_the_same_image(cus, threads):
    """The persistent grid is sized to the machine: one work-group that takes every tile, a few, and far more waves than
    tiles (most find their queue empty at once). Frames 0..5 in launches of 1, 2 and 3, so that the tile order is rebuilt
    in between."""
    W, H = 56, 24          # 21 tiles
    sc = make_scene("sponza_like", target_tris=6000)
    o = Oracle(W, H)
    setup(o, sc, NOISE, SKY, bounce_limit=8)
    o.render(0, SEEDS[:6])
    with Emulated(devices=1, cus=cus, threads=threads):
        g = ra.Renderer(W, H)
        setup(g, sc, NOISE, SKY, bounce_limit=8)
        g.render(0, SEEDS[:1]); g.render(1, SEEDS[1:3]); g.render(3, SEEDS[3:6])
        assert differing_pixels(g.read_accum(), o.read_accum()) == 0
        assert g.counters()["rays"] == o.counters()["rays"]
        g.close()


def test_random_configurations_bit_exact():
    """The randomised parity check of the GPU tests (random scene, image size, camera, bounce limit 0-32, launch split,
    1-5 shards) on the emulated library, with seeds of its own (14 cases here; 400 were run once, none differing)."""
    import fuzz_parity
    lines = []
    bad = fuzz_parity.run(14, 31, log=lines.append, host_buffers=True)
    assert bad == 0, "\\n".join(l for l in lines if "DIFFERS" in l)
    assert sum("shards 1" not in l for l in lines[:-1]) >= 4      # sharded cases were among them


def test_the_package_itself_never_loads_the_emulated_library():
    """RT_LIB_PATH (kernel-tuning builds of the HIP library) cannot be pointed at the emulation: only the tests put it under
    the Python layer, by hand."""
    import os
    import subprocess
    import sys
    from emulation import EMU_LIB
    r = subprocess.run([sys.executable, "-c", "import raytracing_amd._lib as l; l.lib()"], env=dict(os.environ, RT_LIB_PATH=EMU_LIB),
                       capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode != 0 and "host emulation" in r.stderr
