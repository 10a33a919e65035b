"""The kernels as the compiler built them, on the CPU: hipcc's gfx950 assembly of trace_kernels.hip executed by an
interpreter of the instructions it uses (tests/emu/isa), inside the host emulation of the library, against the oracle, bit
for bit. The fiber emulation (tests/test_emulated.py) checks the kernels' source; this checks their machine code: the
registers and flags a wave really has, EXEC-masked lock step, scalar/vector split, LDS and scratch addressing, and
that every memory result is waited for before a register it lands in is named again. Registers, LDS and scratch start with
random bits, global addresses are checked against the runtime's allocations. What it cannot see: timing, and hazards that
need idle cycles rather than waits. Needs hipcc (no GPU): skipped where there is none."""
import os
import sys

import numpy as np
import pytest

import raytracing_amd as ra
from emulation import Emulated
from raytracing_amd import scenes
from rt_oracle import Oracle
from util import differing_pixels, make_scene, setup

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "isa"))

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc to produce the device assembly")

NOISE = scenes.noise_texture(64, 48)
SKY = scenes.sky_cubemap(32)
SEEDS = scenes.frame_seeds(6)
# every kernel of trace_kernels.hip: the tracer, the two that prepare a launch, and the helpers (image layout, tile order, tallies)
KERNELS = ("trace_kernel", "prepare_frames_kernel", "prepare_rays_kernel", "untile_kernel", "retile_kernel", "reduce_tally_kernel",
           "order_keys_kernel", "probe_triangles_kernel")


@pytest.fixture(scope="module")
def assembly():
    from harness import assemble
    return assemble()


@pytest.fixture(scope="module", autouse=True)
def _emulated_module():
    with Emulated(devices=1, cus=2):
        yield


def test_the_interpreter_notices_a_missing_wait(assembly):
    """The wait checker on a kernel whose every third vector-memory wait is taken away: it must object."""
    import gcn_interp as gi
    from harness import FromAssembly
    original = gi.Machine._check_wait

    def lossy(self, w, pc):
        ins = self.k.instrs[pc]
        if ins.op == "s_waitcnt" and "lgkmcnt" not in ins.text and ins.line % 3 == 0:
            return
        return original(self, w, pc)
    g = ra.Renderer(16, 8)
    setup(g, make_scene("lightbox"), NOISE, SKY, bounce_limit=4)
    gi.Machine._check_wait = lossy
    try:
        with FromAssembly(assembly, check_waits=True) as run:
            g.render(0, SEEDS[:1]); run.check()
            assert len(run.wait_violations()) > 100
    finally:
        gi.Machine._check_wait = original
    with FromAssembly(assembly, check_waits=True) as run:
        g.render(1, SEEDS[1:2]); run.check()
        assert run.wait_violations() == []
    g.close()


@pytest.mark.parametrize("scene,kw,W,H", [("lightbox", {}, 16, 8), ("sponza_like", {"target_tris": 6000}, 19, 9)], ids=["lightbox", "sponza_like"])
def test_machine_code_of_every_tracer_kernel_equals_the_oracle(assembly, scene, kw, W, H):
    """All eleven instantiations: one frame / several frames per launch, the heat-map view, full counters, primary ids, and
    the four kernels for several rays per pixel; the two kernels that prepare a launch run from their assembly too."""
    from harness import FromAssembly
    sc = make_scene(scene, **kw)
    g, o = ra.Renderer(W, H), Oracle(W, H)
    setup(g, sc, NOISE, SKY)
    setup(o, sc, NOISE, SKY)
    seen = set()
    with FromAssembly(assembly, kernels=KERNELS, garbage=13, check_waits=True) as run:
        #        rays, bounce limit, frames, full counters
        cases = [(1, 4, 1, False), (1, 2, 2, False), (1, 0, 1, False), (1, 0, 2, False), (1, 4, 1, True), (1, 2, 2, True),
                 (3, 4, 1, False), (2, 2, 2, False), (2, 4, 1, True), (2, 1, 2, True)]
        if scene != "lightbox":          # the larger tree (node records from memory, not only from the LDS copy): four of the kernels
            cases = [(1, 8, 1, False), (1, 2, 2, False), (3, 4, 1, False), (1, 4, 1, True)]
        for rays, bl, frames, count in cases:
            for r in (g, o):
                r.set_ray_count(rays); r.set_bounce_limit(bl); r.clear(); r.counters(reset=True)
            g.set_counting(count)
            n0 = len(run.launches)
            g.render(0, SEEDS[:frames]); run.check()
            o.render(0, SEEDS[:frames])
            assert differing_pixels(g.read_accum(), o.read_accum()) == 0, (rays, bl, frames, count)
            cg, co = g.counters(reset=True), o.counters(reset=True)
            for k in (co if count else ("rays",)):
                assert cg[k] == co[k], (rays, bl, frames, count, k)
            seen.update(l[0] for l in run.launches[n0:])
            assert sum(l[3] for l in run.launches[n0:]) == 0      # no LDS access beyond the work-group's allocation
        g.set_counting(False)
        assert (g.primary_ids(SEEDS[0]) == o.primary_ids(SEEDS[0])).all(); run.check()
        seen.add(run.launches[-1][0])
        assert run.wait_violations() == []
    assert len([s for s in seen if "trace_kernel" in s]) == (11 if scene == "lightbox" else 5), sorted(seen)
    for helper in ("prepare_rays_kernel", "prepare_frames_kernel", "untile_kernel", "reduce_tally_kernel", "order_keys_kernel"):
        assert any(helper in s for s in seen), helper
    g.close()


def test_a_stray_address_is_reported_not_followed(assembly):
    """A triangle reference that points past the triangle array (a damaged record, written behind the library's back): the
    interpreter names the load and the lanes instead of reading whatever lies there."""
    import ctypes
    import gcn_interp as gi
    from harness import FromAssembly
    W, H = 16, 8
    g = ra.Renderer(W, H)
    one = np.array([[4, -1, -1, 0, 4, 1, -1, 0, 4, 0, 1, 0]], dtype=np.float32)
    node = np.array([[4, 1, 1, 4, -1, -1, 1, 0]], dtype=np.float32)
    g.set_scene(one, np.zeros(1, np.int32), node)
    g.set_materials(np.array([[0.8, 0.3, 0.2, 1, 1, 1, 1, 0, 0, 0, 0]], dtype=np.float32))
    from util import DEFAULT_CAMERA
    g.set_camera(**DEFAULT_CAMERA); g.set_noise(NOISE); g.set_skybox(SKY); g.set_bounce_limit(4)
    with FromAssembly(assembly) as run:
        g.render(0, SEEDS[:1]); run.check()                  # the intact scene renders
        # the scene is a single leaf, referenced by the launch's scene parameters: make the next launch's copy claim the
        # triangle 2^20 records further on
        L = run.lib
        original = run._hook

        def damaged(name, dims, args, nbytes):
            if b"trace_kernel" in name:
                raw = bytearray(args[:nbytes])
                words = np.frombuffer(raw, np.uint32)
                hits = np.nonzero(words == np.uint32(0x80000000))[0]          # leaf bit | count 1 | first triangle 0
                assert len(hits) == 1
                words = words.copy(); words[hits[0]] = 0x80000000 | (1 << 20)
                buf = (ctypes.c_ubyte * nbytes).from_buffer_copy(words.tobytes())
                return original(name, dims, buf, nbytes)
            return original(name, dims, args, nbytes)
        from harness import HOOK
        cb = HOOK(damaged)
        L.emu_set_launch_hook(cb)
        g.render(1, SEEDS[1:2])
        with pytest.raises(gi.StrayAccess, match="outside every allocation"):
            run.check()
    g.close()


def test_the_benchmarked_kernel_keeps_its_register_and_lds_budget(assembly):
    """What the design rests on for the one-frame kernel (five waves per SIMD, five work-groups per CU): at most 96 vector
    registers, no scratch, exactly 25 LDS granules of 1 280 bytes. An edit that spills or grows the LDS copy shows here, on the
    CPU, before it shows as a slower frame. The other instantiations stay within 96 registers too (they may spill)."""
    import gcn_interp as gi
    kernels, _ = gi.parse_assembly(assembly)
    tracer = {k: v for k, v in kernels.items() if "12trace_kernelI" in k}
    assert len(tracer) == 11
    for name, k in tracer.items():
        assert k.desc["next_free_vgpr"] <= 96, name
        assert k.desc["group_segment_fixed_size"] <= 32000, name
    bench = [k for n, k in tracer.items() if "trace_kernelILb0ELb0ELb1ELb0ELb0EE" in n]
    assert len(bench) == 1
    assert bench[0].desc["private_segment_fixed_size"] == 0
    assert bench[0].desc["group_segment_fixed_size"] == 32000
    assert bench[0].meta[".vgpr_spill_count"] == 0


def test_one_ray_kernels_are_the_instruction_streams_that_ran_on_the_gpu(assembly):
    """profiles/r03_kernel_isa_hashes.json holds fingerprints of the seven one-ray kernels as they were built for the profile
    r03a and the GPU test run before it (the sources of that commit give the same fingerprints). The current sources must
    still compile to exactly those instructions: what was added since (the kernels for several rays per pixel) lives in
    instantiations of its own. Skipped under another compiler version, where the streams differ for other reasons."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_isa_hashes as kh
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_kernel_isa_hashes.json")) as f:
        recorded = json.load(f)
    if recorded["toolchain"] != kh.toolchain():
        pytest.skip("recorded with another compiler: " + recorded["toolchain"])
    now = kh.fingerprints(assembly)
    assert len(recorded["kernels"]) == 7
    for name, fp in recorded["kernels"].items():
        assert now[name] == fp, name


POST_KERNELS = ("down_kernel", "pre_kernel", "blur_kernel", "snap_kernel", "final_kernel", "to_rgb8_kernel")


def test_machine_code_of_the_bloom_chain():
    """post_kernels.hip from its assembly: three consecutive frames (the chain's tmp image carries over) against the post
    oracle within the GPU test's tolerance (the hardware's exp2/log2 are within an ulp of the interpreter's), the 8-bit image
    within one code; then the entry point that allocates nothing on two images, the second smaller, bit-equal to the
    ordinary one run the same way. Two-dimensional launches, the LDS-tiled blur, byte and short stores, SDWA."""
    from harness import CSRC, FromAssembly, assemble
    from raytracing_amd import _lib
    from rt_oracle import PostOracle
    from test_post import TOL_GPU_ORACLE, noisy_hdr
    from test_zz_no_alloc import hdr_image, reference_chain
    from util import bits
    asm = assemble(source=os.path.join(CSRC, "post_kernels.hip"), tag="post_kernels")
    W, H = 123, 77
    g, po = ra.Renderer(W, H), PostOracle(W, H)
    with FromAssembly(asm, kernels=POST_KERNELS, garbage=17, check_waits=True) as run:
        for frame in range(3):
            a = noisy_hdr(W, H, 60 + frame)
            g.write_accum(a)
            rgba, rgb8 = g.postprocess(True); run.check()
            o_rgba, o_rgb8 = po.run(a)
            assert np.abs(rgba[..., :3] - o_rgba[..., :3]).max() <= TOL_GPU_ORACLE, frame
            assert np.abs(rgb8.astype(int) - o_rgb8.astype(int)).max() <= 1
        assert {l[0] for l in run.launches} == set(POST_KERNELS)
        L = _lib.lib()
        for img in (hdr_image(96, 54, 1), hdr_image(31, 19, 2)):
            h, w = img.shape[:2]
            rgba = np.full((h, w, 4), -1.0, dtype=np.float32)
            rgb8 = np.full((h, w, 3), 7, dtype=np.uint8)
            run.share(img, rgba, rgb8)
            assert L.rt_postprocess_static(img.ctypes.data, w, h, rgba.ctypes.data, rgb8.ctypes.data, None) == 0; run.check()
            want_rgba, want_rgb8 = reference_chain(img); run.check()
            assert (bits(rgba) == bits(want_rgba)).all()
            assert (rgb8 == want_rgb8).all()
        assert run.wait_violations() == []
        assert sum(l[3] for l in run.launches) == 0
    g.close()


def test_machine_code_of_the_triangle_test_on_shared_edges(assembly):
    """probe_triangles_kernel (the leaf body's ray/triangle test on caller-supplied pairs) from its assembly, on rays aimed at
    triangle edges, against what the reference's own hitTriangle returned on llvmpipe (tests/golden/triangle_edges.npz):
    which side of an edge a hit falls on is decided by the last bit, so the division sequence and the order of the
    operations in the compiled code are what is being checked."""
    from harness import FromAssembly
    from util import GOLDEN, bits
    t = np.load(os.path.join(GOLDEN, "triangle_edges.npz"))
    n = len(t["hit"])
    g = ra.Renderer(8, 8)
    with FromAssembly(assembly, kernels=("probe_triangles_kernel",), garbage=3, check_waits=True) as run:
        hit, dist = g.probe_triangles(t["tri12"][:n], t["origins"][:n], t["directions"][:n]); run.check()
        assert len(run.launches) == 1 and run.wait_violations() == []
    assert (hit == t["hit"][:n]).all()
    assert (bits(dist)[hit] == bits(t["t"][:n])[hit]).all()
    assert 0.2 < hit.mean() < 0.95
    g.close()


def test_random_configurations_from_machine_code():
    """Two cases of tests/emu/isa/fuzz_kernels.py (random scene, size, bounce limit, launch split, rays per pixel, counters,
    shards; every kernel of trace_kernels.hip from its assembly)."""
    import fuzz_kernels
    lines = []
    assert fuzz_kernels.run(2, 2026, log=lines.append) == 0, "\n".join(lines)
