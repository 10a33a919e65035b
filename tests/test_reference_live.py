"""The CPU restatement against the LIVE reference: the reference's unmodified tracer shader run on headless Mesa
llvmpipe (oracle/glsl_ref), on scenes and cameras that are not among the committed fixtures. Runs only where the
reference's sources and the harness built from them (oracle/_ref/libglref.so) are present, i.e. in the build container;
skipped elsewhere (the reference does not travel). Bar: primary-hit ids and distances and every frame's radiance bit
for bit (oracle/compare_reference.py exits non-zero on any difference)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("RT_REFERENCE_DIR", "/root/reference")
HAVE = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libglref.so")) and os.path.exists(os.path.join(REFERENCE, "src", "shaders", "Fragment.glsl"))

pytestmark = pytest.mark.skipif(not HAVE, reason="needs the reference's sources and oracle/_ref (build container only)")

CASES = [
    # scene, triangles, width, height, bounce limit, frames, camera, fov
    ("lightbox", 0, 200, 120, 4, 6, "0.5,0.3,0.2,1,0.4,-0.3", 120),
    ("sponza_like", 9000, 176, 100, 8, 4, "1.0,0.5,0.4,-1,0.2,0.1", 90),
    ("dragon_like", 7000, 144, 90, 16, 3, "", 120),
    ("teapot_like", 5000, 150, 84, 2, 3, "1.5,-2,0.7,0.8,0.55,-0.2", 120),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_restatement_equals_live_shader(case):
    scene, tris, w, h, bounces, frames, camera, fov = case
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "compare_reference.py"), "--scene", scene, "--tris", str(tris), "--width", str(w),
           "--height", str(h), "--bounces", str(bounces), "--frames", str(frames), "--fov", str(fov), "--skip-seeds", "7"]
    if camera:
        cmd += ["--camera=" + camera]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "triangle ids differing: 0; hit distances differing: 0" in run.stdout
    assert run.stdout.count("pixels differ") == frames


RAY_CASES = [
    # scene, triangles, width, height, bounce limit, frames, RAY_COUNT
    ("lightbox", 0, 120, 72, 4, 3, 2),
    ("sponza_like", 9000, 110, 64, 8, 2, 5),
    ("dragon_like", 7000, 96, 60, 16, 2, 3),
]


@pytest.mark.parametrize("case", RAY_CASES, ids=["%s-rays%d" % (c[0], c[6]) for c in RAY_CASES])
def test_restatement_equals_live_shader_with_several_rays_per_pixel(case):
    """The shader's loop over RAY_COUNT rays per pixel (Fragment.glsl:459-485), with the reference's own compile-time
    constant set to 2, 3 and 5 while its file is loaded: per-ray jitter, per-ray seed term, the early-out and the mean."""
    scene, tris, w, h, bounces, frames, rays = case
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "compare_reference.py"), "--scene", scene, "--tris", str(tris), "--width", str(w),
           "--height", str(h), "--bounces", str(bounces), "--frames", str(frames), "--skip-seeds", "3", "--ray-count", str(rays)]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert run.stdout.count("pixels differ") == frames


POST_LIVE = r'''
import sys
import numpy as np
ROOT = sys.argv[1]
for p in (ROOT, ROOT + "/oracle", ROOT + "/oracle/glsl_ref", ROOT + "/tests"):
    sys.path.insert(0, p)
from glref import GLRef
from reference_renderer import ReferenceRenderer
from rt_oracle import PostOracle
import make_golden as mg
gl = GLRef()
for (W, H) in ((320, 160), (200, 120), (250, 130), (123, 77), (64, 64), (640, 360)):
    ref, po = ReferenceRenderer(W, H, gl=gl), PostOracle(W, H)
    for frame in range(4):                       # consecutive frames: the tmp texture carries over in both
        a = mg.noisy_hdr(W, H, 90 + frame)
        ref.write_accum(a)
        rgba, rgb8 = ref.postprocess(True)
        o_rgba, o_rgb8 = po.run(a)
        d = float(np.abs(rgba[..., :3] - o_rgba[..., :3]).max())
        codes = int(np.abs(rgb8.astype(int) - o_rgb8.astype(int)).max())
        print("%dx%d frame %d: max abs diff %.3g, 8-bit codes %d" % (W, H, frame, d, codes))
        assert d <= 5e-6 and codes <= 1
print("post chain ok")
'''


def test_bloom_chain_restatement_equals_live_shaders(tmp_path):
    """The bloom chain of the CPU restatement against the reference's own post shaders run on llvmpipe with main.cpp's
    sequence of passes, at sizes other than the fixture's (one of them square: pixel centres on the blits' diagonal), four
    consecutive frames each, on images whose every texel differs from its neighbours. Tolerance 5e-6 absolute (llvmpipe
    approximates pow), 8-bit output within one code."""
    script = tmp_path / "post_live.py"
    script.write_text(POST_LIVE)
    run = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "post chain ok" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]
