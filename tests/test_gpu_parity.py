"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar: bit-exact. Triangle ids, hit distances and traversal counts are integers/bit patterns; the
radiance is fp32 computed with the same operation order on both sides, so it is compared
bit-for-bit too (tolerance 0)."""
import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import scenes
from rt_oracle import Oracle
from util import DEFAULT_CAMERA, bits, differing_pixels, make_scene, setup

pytestmark = pytest.mark.gpu

NOISE = scenes.noise_texture(64, 48)
SKY = scenes.sky_cubemap(32)
SEEDS = scenes.frame_seeds(16)

CASES = [
    ("lightbox", {}, 160, 90),
    ("sponza_like", {"target_tris": 20000}, 160, 96),
    ("dragon_like", {"target_tris": 12000}, 96, 56),
    ("teapot_like", {"target_tris": 12000}, 100, 52),   # width/height not multiples of 8
]


@pytest.fixture(scope="module", params=CASES, ids=[c[0] for c in CASES])
def pair(request):
    name, kw, W, H = request.param
    sc = make_scene(name, **kw)
    g, o = ra.Renderer(W, H), Oracle(W, H)
    setup(g, sc, NOISE, SKY)
    setup(o, sc, NOISE, SKY)
    yield g, o, sc
    g.close()


def test_primary_ids_bit_exact(pair):
    g, o, _ = pair
    a, b = g.primary_ids(SEEDS[0]), o.primary_ids(SEEDS[0])
    assert (a[..., 0] == b[..., 0]).all(), "triangle ids differ"
    assert (a[..., 1] == b[..., 1]).all(), "hit distances differ"
    assert (a[..., 2] == b[..., 2]).all(), "iterationsCount differs"
    assert (a[..., 0] >= 0).any()


@pytest.mark.parametrize("bounce_limit,frames", [(0, 1), (1, 1), (2, 3), (4, 4), (8, 3), (16, 2)])
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


def test_progressive_split_equals_batch(pair):
    """Frames rendered one launch at a time equal the same frames rendered in one launch."""
    g, o, _ = pair
    g.set_bounce_limit(4)
    g.clear()
    g.render(0, SEEDS[:6])
    whole = g.read_accum()
    g.clear()
    for k in range(6):
        g.render(k, SEEDS[k:k + 1])
    assert differing_pixels(whole, g.read_accum()) == 0


def test_counters_match_oracle(pair):
    g, o, _ = pair
    for r in (g, o):
        r.set_bounce_limit(4)
        r.clear()
    g.set_counting(True)
    g.counters(reset=True)
    o.counters(reset=True)
    g.render(0, SEEDS[:2])
    o.render(0, SEEDS[:2])
    cg, co = g.counters(reset=True), o.counters(reset=True)
    g.set_counting(False)
    for k in ("rays", "node_pops", "boxes_tested", "tris_tested", "noise_fetches", "sky_fetches", "hits"):
        assert cg[k] == co[k], k
    assert cg["paths"] == g.W * g.H * 2
    # the uninstrumented kernel counts rays and paths only, and the same number of them
    g.clear()
    g.render(0, SEEDS[:2])
    c2 = g.counters(reset=True)
    assert c2["rays"] == co["rays"] and c2["paths"] == g.W * g.H * 2


def test_accum_write_read_roundtrip(pair):
    g, _, _ = pair
    img = np.random.default_rng(0).random((g.H, g.W, 4), dtype=np.float32)
    g.write_accum(img)
    assert (bits(g.read_accum()) == bits(img)).all()
    g.clear()
    assert (g.read_accum() == 0).all()


def test_empty_and_single_leaf_scenes():
    W, H = 40, 24
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


def test_errors_are_reported_not_fatal():
    g = ra.Renderer(16, 16)
    with pytest.raises(ra.RtError):
        g.render(0, SEEDS[:1])            # nothing uploaded yet
    bad_node = np.array([[1, 1, 1, 0, 0, 0, 0, 7]], dtype=np.float32)   # interior node with a child out of range
    with pytest.raises(ra.RtError):
        g.set_scene(np.zeros((0, 12), np.float32), np.zeros(0, np.int32), bad_node)
    with pytest.raises(ra.RtError):
        g.set_materials(np.zeros((17, 11), np.float32))
    g.close()


def test_reupload_on_one_context_equals_fresh_context():
    """State is replaced, not merged: after another scene, noise texture and sky have been uploaded and rendered,
    uploading the first ones again gives the image a fresh context gives."""
    W, H = 96, 56
    a_scene, b_scene = make_scene("lightbox"), make_scene("teapot_like", target_tris=12000)
    other_noise, other_sky = scenes.noise_texture(40, 24), scenes.sky_cubemap(16)
    fresh = ra.Renderer(W, H)
    setup(fresh, a_scene, NOISE, SKY)
    fresh.render(0, SEEDS[:3])
    want = fresh.read_accum()
    fresh.close()
    r = ra.Renderer(W, H)
    setup(r, b_scene, other_noise, other_sky, bounce_limit=8)
    r.render(0, SEEDS[:2])
    setup(r, a_scene, NOISE, SKY)            # bounce limit back to 4 as well
    r.clear()
    r.render(0, SEEDS[:3])
    assert differing_pixels(r.read_accum(), want) == 0
    r.close()


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return tuple(np.float32(v / np.linalg.norm(v)))


CAMERAS = [
    # position, facing (unit), field of view in degrees
    ((1.5, -0.8, 0.6), _unit((1.0, 0.7, -0.2)), 90.0),
    ((4.0, 0.5, 1.2), _unit((-1.0, 0.1, 0.05)), 60.0),       # looking back towards the origin
    ((0.0, 0.0, 2.5), _unit((0.3, 0.2, -1.0)), 140.0),       # steeply downwards, very wide
]


@pytest.mark.parametrize("cam", range(len(CAMERAS)))
def test_moved_and_rotated_cameras_bit_exact(cam):
    """Other poses and fields of view than the default one (Include/Camera.h:7-11), as Camera::UploadInfo sends them."""
    import math
    pos, facing, fov_deg = CAMERAS[cam]
    fov = np.float32(fov_deg * 0.01745329251994329576923690768489)
    camera = dict(position=pos, facing=facing, fov=float(fov), viewport_width=float(np.float32(2.0 * math.tan(float(fov) / 2.0))))
    W, H = 120, 72
    sc = make_scene("sponza_like", target_tris=20000)
    g, o = ra.Renderer(W, H), Oracle(W, H)
    for r in (g, o):
        setup(r, sc, NOISE, SKY, camera=camera, bounce_limit=8)
        r.render(0, SEEDS[:3])
    a, b = g.read_accum(), o.read_accum()
    assert np.isfinite(b[..., :3]).all()
    assert differing_pixels(a, b) == 0
    # the id pass last: the oracle's renders a frame into its accumulation image as a side effect, the product's does not
    assert (g.primary_ids(SEEDS[0]) == o.primary_ids(SEEDS[0])).all()
    assert differing_pixels(g.read_accum(), a) == 0
    g.close()


def test_bounce_limit_32_and_a_long_launch_bit_exact():
    """The largest limit the reference's keys select (src/BounceLimitManager.cpp:14-48) and 40 frames in one launch."""
    W, H = 64, 40
    sc = make_scene("dragon_like", target_tris=12000)
    seeds = scenes.frame_seeds(40)
    g, o = ra.Renderer(W, H), Oracle(W, H)
    for r in (g, o):
        setup(r, sc, NOISE, SKY, bounce_limit=32)
        r.render(0, seeds)
    assert differing_pixels(g.read_accum(), o.read_accum()) == 0
    g.close()


@pytest.mark.parametrize("size", [(1, 1), (8, 600), (5, 333), (1500, 9), (2047, 3), (1, 257), (4100, 8)],
                         ids=lambda s: "%dx%d" % s)
def test_extreme_image_shapes_bit_exact(size):
    """One tile column, one tile row, single pixels: the tile -> pixel arithmetic at its edges, alone and in 3 shards."""
    from raytracing_amd import sharding
    import torch
    W, H = size
    sc = make_scene("sponza_like", target_tris=6000)
    o = Oracle(W, H, threads=16)
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
        t = torch.empty(per, dtype=torch.float32, device="cuda")
        r.export_shard(t.data_ptr()); r.sync()
        flat[rank * per:(rank + 1) * per] = t.cpu().numpy()
        r.close()
    assert differing_pixels(sharding.assemble_numpy(flat, W, H, world), want) == 0


def test_deep_stack_spills_to_memory_bit_exact():
    """A deck of parallel quads facing the camera: every ray enters both children of every node on its way down, so
    the deferred-node stack reaches the depth of the tree, beyond the entries kept on chip, on every primary ray."""
    n = 20000
    x = (1.0 + 0.001 * np.arange(n, dtype=np.float64)).astype(np.float32)
    lo, hi = np.float32(-6.0), np.float32(6.0)
    tri9 = np.zeros((2 * n, 9), dtype=np.float32)
    tri9[0::2] = np.stack([x, np.full(n, lo), np.full(n, lo), x, np.full(n, hi), np.full(n, lo), x, np.full(n, hi), np.full(n, hi)], 1)
    tri9[1::2] = np.stack([x, np.full(n, lo), np.full(n, lo), x, np.full(n, hi), np.full(n, hi), x, np.full(n, lo), np.full(n, hi)], 1)
    mats = np.zeros((2, 11), dtype=np.float32)
    mats[0] = [0.8, 0.7, 0.6, 1, 1, 1, 1.0, 0.0, 0.0, 0.0, 0]       # lambertian
    mats[1] = [0.9, 0.9, 1.0, 1, 1, 1, 0.0, 0.5, 0.9, 1.5, 0]       # glass: paths go through the whole deck
    slots = (np.arange(2 * n) // 2 % 7 == 0).astype(np.int32)
    sc = ra.build_scene(tri9, slots, mats, build_threads=4)
    assert sc.stats["max_depth"] >= 14
    W, H = 72, 40
    g, o = ra.Renderer(W, H), Oracle(W, H, threads=16)
    for r in (g, o):
        setup(r, sc, NOISE, SKY, bounce_limit=3)
        r.render(0, SEEDS[:2])
    assert differing_pixels(g.read_accum(), o.read_accum()) == 0
    assert (g.primary_ids(SEEDS[0]) == o.primary_ids(SEEDS[0])).all()
    g.close()


def _chain_scene(levels):
    """A hand-built tree that is one chain of `levels` interior nodes: every interior node's left child is the next
    interior node (the last one's is a leaf), its right child a leaf. All boxes overlap along the view axis and the
    interior boxes start nearer than the leaf boxes, so a ray enters the left child first and defers the right one at
    every level: the deferred-node stack grows to `levels` entries. The leaves' triangles come nearer the further up
    their leaf hangs, so every deferred leaf is still worth visiting when it is popped."""
    n_leaf = levels + 1
    n_node = 2 * levels + 1
    node8 = np.zeros((n_node, 8), dtype=np.float32)
    tri12 = np.zeros((n_leaf, 12), dtype=np.float32)
    # pre-order: interior 1..levels at indices 0..levels-1, then the deepest left leaf, then the right leaves from the
    # deepest interior node upwards
    def tri_at(x):
        return [x, -40, -40, 0, x, 40, -40, 0, x, 0, 40, 0]
    xs = 60.0 - 0.5 * np.arange(n_leaf)             # leaf j (in visiting order) holds a triangle at x = xs[j]
    for j in range(n_leaf):
        tri12[j] = tri_at(xs[j])
        node = levels + j
        node8[node] = [xs[j] + 0.25, 50, 50, xs[j] - 0.25, -50, -50, 1, j]
    for k in range(levels):                          # interior node at depth k + 1
        right = levels + (levels - k)                # its right leaf: visited (levels - k)-th after the left-most leaf
        node8[k] = [100, 50, 50, 1, -50, -50, 0, right]
    return tri12, np.zeros(n_leaf, np.int32), node8


def test_deepest_tree_the_stack_can_hold_bit_exact():
    """Interior nodes 63 levels deep, every ray entering both children of each: what the shader's stack[64] can just
    hold (Fragment.glsl:9,270). One level more is refused at upload."""
    tri12, matidx, node8 = _chain_scene(63)
    mats = np.array([[0.8, 0.6, 0.3, 1, 1, 1, 1, 0, 0, 0, 0]], dtype=np.float32)
    W, H = 64, 40
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
    assert (a[..., 0] == 63).all(), "every primary ray must end on the last leaf visited"
    assert (a[..., 2] >= 2 * 63).all(), "every level's deferred leaf must have been popped and visited"
    for r in (g, o):
        r.render(0, SEEDS[:2])
    assert differing_pixels(g.read_accum(), o.read_accum()) == 0
    # a second, sharded context renders next to it: with a stack entry too many, one wave's deepest entries would
    # land in its neighbour's spill rows
    deeper = _chain_scene(64)
    with pytest.raises(ra.RtError, match="63"):
        g.set_scene(*deeper)
    g.close()


def test_triangle_edges_bit_exact():
    """The device's ray/triangle test on rays aimed at triangle edges: against the llvmpipe fixture (the reference's own
    hitTriangle), and against the CPU restatement on a fresh million of such pairs. Bar: same hit, same bits of t."""
    import os
    from rt_oracle import hit_triangles
    t = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "triangle_edges.npz"))
    g = ra.Renderer(64, 64)
    hit, dist = g.probe_triangles(t["tri12"], t["origins"], t["directions"])
    assert (hit == t["hit"]).all(), int((hit != t["hit"]).sum())
    assert (bits(dist) == bits(t["t"])).all()

    rng = np.random.default_rng(77)
    n = 1 << 20
    v0 = (np.round(rng.uniform(-8, 8, (n, 3)) * 2) / 2).astype(np.float32)
    e1, e2 = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    axis = rng.integers(0, 3, n)
    rows = np.arange(n)
    e1[rows, (axis + 1) % 3] = np.round(rng.uniform(1, 24, n)) / 2
    e2[rows, (axis + 2) % 3] = np.round(rng.uniform(1, 24, n)) / 2
    free = rng.random(n) < 0.5                                   # the other half: triangles in general position
    e1[free], e2[free] = rng.uniform(-12, 12, (int(free.sum()), 3)), rng.uniform(-12, 12, (int(free.sum()), 3))
    v1, v2 = v0 + e1, v0 + e2
    s = rng.random((n, 1)).astype(np.float32)
    target = v1 + s * (v2 - v1)                                  # on the edge opposite the first corner: u + v = 1
    origin = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    d = (target - origin).astype(np.float64)
    d = (d / np.sqrt((d * d).sum(-1, keepdims=True))).astype(np.float32)
    tri12 = np.zeros((n, 12), np.float32)
    tri12[:, 0:3], tri12[:, 4:7], tri12[:, 8:11] = v0, v1, v2
    hit_g, t_g = g.probe_triangles(tri12, origin, d)
    hit_o, t_o = hit_triangles(tri12, origin, d)
    assert 0.2 < hit_o.mean() < 0.8
    assert (hit_g == hit_o).all(), int((hit_g != hit_o).sum())
    assert (bits(t_g) == bits(t_o)).all()
    assert g.probe_triangles(np.zeros((0, 12)), np.zeros((0, 3)), np.zeros((0, 3)))[0].shape == (0,)
    g.close()
