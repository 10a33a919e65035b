"""Host-side control surface (CPU): camera keys, bounce-limit keys, seed stream, ini reader, PNG writer.
The reference's Camera.cpp needs GLFW/GLEW headers and cannot be compiled here, so these
expectations are derived from reading src/Camera.cpp:78-127, src/BounceLimitManager.cpp:14-48,
src/ConfigParser/ConfigParser.cpp:36-86 ("parity unpinned" by reference output)."""
import math
import os

import numpy as np
import pytest

from raytracing_amd import controls, scenes
from raytracing_amd.controls import KEY, PRESS, RELEASE, Camera
from util import FOV, VPW

f32 = np.float32


def test_camera_defaults():
    c = Camera().get()
    assert (c["position"] == 0).all() and (c["facing"] == [1, 0, 0]).all()
    assert c["fov"] == float(FOV) and c["viewport_width"] == VPW and c["viewport_distance"] == 1.0


def test_camera_translation_keys():
    cam = Camera()
    assert cam.tick() is False                     # nothing pressed: not moved
    cam.key(KEY["W"], PRESS)
    assert cam.tick() is True
    assert np.allclose(cam.get()["position"], [0.2, 0, 0], atol=0)
    cam.key(KEY["LEFT_SHIFT"], PRESS)              # doubled speed
    cam.tick()
    assert cam.get()["position"][0] == f32(f32(0.2) + f32(1) * f32(0.4))
    cam.key(KEY["LEFT_SHIFT"], RELEASE)
    cam.key(KEY["W"], RELEASE)
    cam.key(KEY["D"], PRESS)                       # right = facing x (0,0,1) = (0,-1,0)
    cam.tick()
    assert cam.get()["position"][1] == f32(-0.2)
    cam.key(KEY["D"], RELEASE)
    cam.key(KEY["SPACE"], PRESS)                   # up at half speed
    cam.tick()
    assert cam.get()["position"][2] == f32(0.1)
    cam.key(KEY["SPACE"], RELEASE)
    cam.key(KEY["RIGHT_SHIFT"], PRESS)             # halved speed
    cam.key(KEY["LEFT_CONTROL"], PRESS)
    cam.tick()
    assert cam.get()["position"][2] == f32(f32(0.1) - f32(0.05))
    cam.key(KEY["LEFT_CONTROL"], RELEASE)
    assert cam.tick() is False


def test_camera_rotation_and_zoom():
    cam = Camera()
    cam.key(KEY["LEFT"], PRESS)
    assert cam.tick()
    f = cam.get()["facing"]
    # yaw to the left about the up axis by 0.02 rad: cos/sin evaluated in double, narrowed to float
    assert f[0] == f32(math.cos(f32(0.02))) and f[1] == f32(math.sin(f32(0.02))) and f[2] == 0
    cam.key(KEY["LEFT"], RELEASE)
    cam.key(KEY["UP"], PRESS)
    cam.tick()
    assert cam.get()["facing"][2] > 0
    cam.key(KEY["UP"], RELEASE)
    cam.key(KEY["C"], PRESS)                       # zoom: fov 90 degrees while held
    assert cam.tick() is True
    c = cam.get()
    assert c["fov"] == float(f32(90 * 0.01745329251994329576923690768489))
    assert c["viewport_width"] == float(f32(2.0 * math.tan(c["fov"] / 2.0)))
    assert cam.tick() is False                     # still held: nothing changes
    cam.key(KEY["C"], RELEASE)
    assert cam.tick() is True and cam.get()["fov"] == float(FOV)


def test_set_facing_normalises():
    cam = Camera()
    cam.set_pose((1, 2, 3), (0, 3, 4))
    c = cam.get()
    assert np.allclose(c["facing"], [0, 0.6, 0.8], atol=1e-7) and (c["position"] == [1, 2, 3]).all()


def test_bounce_limit_keys():
    want = {KEY["GRAVE"]: 0, KEY["K1"]: 1, KEY["K2"]: 2, KEY["K3"]: 4, KEY["K4"]: 8, KEY["K5"]: 16, KEY["K6"]: 32}
    for k, v in want.items():
        assert controls.bounce_limit_from_key(k) == v
    assert controls.bounce_limit_from_key(KEY["W"]) is None


def test_seed_stream_matches_glibc():
    a = controls.rand_seed_triples(50)
    assert (a == scenes.frame_seeds(50)).all()                       # independent Python restatement
    assert np.allclose(a[0], [0.783099234, 0.394382924, 0.840187728], atol=1e-9)
    import ctypes
    libc = ctypes.CDLL("libc.so.6")                                  # and the real thing
    libc.srand(1)
    vals = [libc.rand() for _ in range(6)]
    assert a[0][2] == f32(vals[0]) / f32(2147483647) and a[0][0] == f32(vals[2]) / f32(2147483647)
    assert a[1][1] == f32(vals[4]) / f32(2147483647)


def test_ini_reader(tmp_path):
    ini = tmp_path / "RayTracer.ini"
    ini.write_text("; comment\n[ShaderPaths]\nTracerVert = ./src/shaders/Vertex.glsl\n\n[SkyBox]\nPath = ./Textures/Day light Box\n"
                   "[Objects]\nDirectoryPath = ./Objects\nFilenames =  DragonGlass.off, Teapot.txt ,LightBox.txt\nFilenames = ignored.txt\n"
                   "[NotUsed]\nFilenames =\n")
    paths, sky = controls.ini_object_paths(str(ini))
    assert paths == ["./Objects/DragonGlass.off", "./Objects/Teapot.txt", "./Objects/LightBox.txt"]   # first definition wins
    assert sky == "./Textures/DaylightBox"                                                         # all whitespace is stripped
    bad = tmp_path / "bad.ini"
    bad.write_text("[A]\nthis line has no equals sign\n")
    with pytest.raises(ValueError):
        controls.ini_object_paths(str(bad))
    scenes.write_ini(str(tmp_path / "gen.ini"), "./Objects", ["a.txt", "b.off"])
    assert controls.ini_object_paths(str(tmp_path / "gen.ini"))[0] == ["./Objects/a.txt", "./Objects/b.off"]


def test_png_writer_roundtrip(tmp_path):
    from PIL import Image
    img = np.random.default_rng(0).integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "a.png")
    controls.write_png(p, img)
    assert (np.asarray(Image.open(p)) == img).all()
    controls.write_png(p, img, flip=True)
    assert (np.asarray(Image.open(p)) == img[::-1]).all()
    big = np.random.default_rng(1).integers(0, 256, size=(200, 300, 3), dtype=np.uint8)   # several deflate blocks
    controls.write_png(p, big)
    assert (np.asarray(Image.open(p)) == big).all()


def test_png_reader_matches_pil(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    for mode, shape in (("RGB", (19, 31, 3)), ("RGBA", (16, 16, 4)), ("L", (9, 40))):
        a = rng.integers(0, 256, size=shape, dtype=np.uint8)
        p = str(tmp_path / (mode + ".png"))
        Image.fromarray(a, mode).save(p)
        for ch in (3, 4):
            want = np.asarray(Image.open(p).convert("RGB" if ch == 3 else "RGBA"))
            assert (controls.read_png(p, ch) == want).all(), (mode, ch)
    pal = Image.fromarray(rng.integers(0, 256, size=(12, 12, 3), dtype=np.uint8), "RGB").quantize(16)
    p = str(tmp_path / "pal.png")
    pal.save(p)
    assert (controls.read_png(p, 3) == np.asarray(Image.open(p).convert("RGB"))).all()
    with pytest.raises(IOError):
        controls.read_png(str(tmp_path / "missing.png"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/Textures"), reason="the reference's textures only exist in the build container")
def test_png_reader_on_reference_textures():
    from PIL import Image
    for rel, ch in (("Noise/rgbNoise.png", 3), ("Noise/rgbSmall.png", 3), ("DaylightBox/top.png", 4), ("BlackStudio/left.png", 4)):
        p = "/root/reference/Textures/" + rel
        want = np.asarray(Image.open(p).convert("RGB" if ch == 3 else "RGBA"))
        assert (controls.read_png(p, ch) == want).all(), rel
    assert controls.load_skybox("/root/reference/Textures/DaylightBox").shape == (6, 511, 511, 4)
