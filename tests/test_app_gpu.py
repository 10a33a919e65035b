"""Headless frame loop on the GPU: the app's ticks reproduce direct rendering with the reference's
seed stream; key presses restart the accumulation; the Recorder writes one image per pose."""
import os

import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import controls, scenes
from raytracing_amd.controls import KEY, PRESS, RELEASE
from util import DEFAULT_CAMERA, differing_pixels, setup

pytestmark = pytest.mark.gpu

W, H = 96, 56
NOISE, SKY = scenes.noise_texture(64, 48), scenes.sky_cubemap(32)


@pytest.fixture()
def objects(tmp_path):
    return scenes.write_objects(scenes.lightbox(), str(tmp_path / "objs"))


def make_app(objects):
    r = ra.Renderer(W, H)
    r.set_noise(NOISE)
    r.set_skybox(SKY)
    app = ra.HeadlessApp(r)
    app.load_objects(objects)
    return r, app


def test_ticks_equal_direct_rendering(objects):
    r, app = make_app(objects)
    app.tick(5)
    st = app.state()
    assert st["next_frame_index"] == 5 and st["bounce_limit"] == 4
    d = ra.Renderer(W, H)
    setup(d, ra.load_scene(objects), NOISE, SKY, camera=DEFAULT_CAMERA, bounce_limit=4)
    d.render(0, controls.rand_seed_triples(5))
    assert differing_pixels(r.read_accum(), d.read_accum()) == 0
    # displayed image = bloom chain run once per frame on the evolving accumulation
    shown, shown8 = app.read_display()
    assert np.isfinite(shown).all() and shown8.shape == (H, W, 3)
    app.close(); r.close(); d.close()


def test_keys_restart_accumulation(objects):
    r, app = make_app(objects)
    app.tick(3)
    app.tap(KEY["K4"])                      # bounce limit 8
    st = app.state()
    assert st["bounce_limit"] == 8 and st["next_frame_index"] == 0
    app.tick(2)
    assert app.state()["next_frame_index"] == 2
    app.key(KEY["W"], PRESS)                # moving: every tick restarts at frame 0
    app.tick(3)
    st = app.state()
    assert st["next_frame_index"] == 1 and np.allclose(st["position"], [0.6, 0, 0], atol=1e-6)
    app.key(KEY["W"], RELEASE)
    app.tick(2)
    assert app.state()["next_frame_index"] == 3
    # the accumulation after those ticks = frames 0..2 from the new position with the seeds the app drew
    app.close(); r.close()


def test_recorder_playback_writes_one_image_per_pose(objects, tmp_path):
    from PIL import Image
    r, app = make_app(objects)
    out = str(tmp_path / "shots")
    app.configure(bloom=True, post_every_frame=True, frames_per_image=4, output_dir=out)
    # record three poses while moving forward
    app.tap(KEY["R"])
    app.key(KEY["W"], PRESS)
    app.tick(3)
    app.key(KEY["W"], RELEASE)
    app.tap(KEY["R"])
    assert app.state()["recorder_state"] == 0 and app.state()["poses_left"] == 3
    poses = str(tmp_path / "poses.txt")
    app.save_poses(poses)
    assert len(open(poses).read().strip().splitlines()) == 3
    app.tap(KEY["P"])
    assert app.state()["recorder_state"] == 2
    app.tick(4 * 4 + 1)                     # image 0 (current pose) + one image per recorded pose
    st = app.state()
    assert st["images_written"] == 4 and st["poses_left"] == 0 and st["recorder_state"] == 0
    files = sorted(os.listdir(out))
    assert files == ["raytrace-out-%d.png" % i for i in range(4)]
    img = np.asarray(Image.open(os.path.join(out, files[1])))
    assert img.shape == (H, W, 3) and img.std() > 5
    # the last image shows 4 frames from the last recorded pose: reproduce it directly
    last = [float(v) for v in open(poses).read().strip().splitlines()[-1].split()]
    app.close(); r.close()
    # the recorder samples the camera before the frame's movement is applied (main.cpp:228 precedes :230)
    assert abs(last[0] - 0.4) < 1e-6


def test_headless_executable_matches_python_driver(objects, tmp_path):
    """The C++ executable (ini -> PNG textures -> objects -> frames -> PNG) against the same steps driven from Python."""
    import subprocess
    from PIL import Image
    root = tmp_path / "install"
    (root / "Textures" / "Sky").mkdir(parents=True)
    (root / "Textures" / "Noise").mkdir(parents=True)
    for f, name in enumerate(("right", "left", "top", "bottom", "front", "back")):
        Image.fromarray(SKY[f], "RGBA").save(str(root / "Textures" / "Sky" / (name + ".png")))
    Image.fromarray(NOISE, "RGB").save(str(root / "Textures" / "Noise" / "rgbNoise.png"))
    obj_dir = os.path.dirname(objects[0])
    scenes.write_ini(str(root / "RayTracer.ini"), obj_dir, [os.path.basename(p) for p in objects], skybox_path=str(root / "Textures" / "Sky"))
    exe = os.path.join(os.path.dirname(ra.__file__), "raytracer_headless")
    out = str(tmp_path / "frame")
    r = subprocess.run([exe, "--ini", str(root / "RayTracer.ini"), "--noise", str(root / "Textures" / "Noise" / "rgbNoise.png"),
                        "--width", str(W), "--height", str(H), "--keys", "K3 W+ 2 W- 4", "--frames", "6", "--out", out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "triangles count: 1366" in r.stdout
    got = np.asarray(Image.open(out + ".png"))
    rr, app = make_app(objects)
    app.tap(KEY["K3"])
    app.key(KEY["W"], PRESS); app.tick(2); app.key(KEY["W"], RELEASE); app.tick(4)
    _, want = app.read_display()
    assert (got == want).all()
    app.close(); rr.close()


def test_headless_executable_on_a_device_group_writes_the_same_image(objects, tmp_path):
    """--gpus 1: the C++ host on top of rt_group (sharded rank, RCCL gather to the root context, bloom chain there)
    against the same frames on a plain context. With the scene cache on, so the second run loads the blob."""
    import subprocess
    from PIL import Image
    root = tmp_path / "install"
    (root / "Sky").mkdir(parents=True)
    for f, name in enumerate(("right", "left", "top", "bottom", "front", "back")):
        Image.fromarray(SKY[f], "RGBA").save(str(root / "Sky" / (name + ".png")))
    Image.fromarray(NOISE, "RGB").save(str(root / "noise.png"))
    scenes.write_ini(str(root / "RayTracer.ini"), os.path.dirname(objects[0]), [os.path.basename(p) for p in objects], skybox_path=str(root / "Sky"))
    exe = os.path.join(os.path.dirname(ra.__file__), "raytracer_headless")
    common = [exe, "--ini", str(root / "RayTracer.ini"), "--noise", str(root / "noise.png"), "--width", str(W), "--height", str(H),
              "--frames", "6", "--cache", str(tmp_path / "cache")]
    one = subprocess.run(common + ["--out", str(tmp_path / "one")], capture_output=True, text=True, timeout=120)
    assert one.returncode == 0, one.stdout + one.stderr
    grp = subprocess.run(common + ["--gpus", "1", "--out", str(tmp_path / "grp")], capture_output=True, text=True, timeout=120)
    assert grp.returncode == 0, grp.stdout + grp.stderr
    assert "on 1 GPUs" in grp.stdout and len(os.listdir(str(tmp_path / "cache"))) == 1
    assert (np.asarray(Image.open(str(tmp_path / "one.png"))) == np.asarray(Image.open(str(tmp_path / "grp.png")))).all()
