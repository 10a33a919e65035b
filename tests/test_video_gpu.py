"""The offline video path (BASELINE configuration 5: camera poses x N samples per pixel + bloom, poses dealt over
ranks) end to end on the GPU, through rth_video_render.

Oracle comparison at small size: every pose's accumulation bit-exact against the CPU oracle, every written image
against oracle tracer + post oracle within one 8-bit code (the post chain's tolerance: powf of two libms, see
tests/test_post.py). At 1920x1080: images are the same files whether a pose is rendered by the only rank or by
rank k of 8, and an interrupted job resumes."""
import os

import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import controls, scenes, video
from rt_oracle import Oracle, PostOracle
from util import differing_pixels, make_scene, setup

pytestmark = pytest.mark.gpu

NOISE, SKY = scenes.noise_texture(64, 48), scenes.sky_cubemap(32)
POSES = np.array([[0.0, 0.0, 0.0, 1.0, 0.0, 0.0],
                  [0.6, -0.3, 0.2, 2.0, 1.0, -0.2],      # facing not normalised: the Recorder's SetFacing normalises
                  [1.5, 0.4, 0.5, -1.0, 0.3, 0.1],
                  [0.3, 0.2, 0.9, 0.5, -0.5, -1.0]], dtype=np.float32)


def oracle_camera(pose):
    cam = controls.Camera()
    cam.set_pose(pose[:3], pose[3:])
    return cam.get()


def test_every_pose_matches_tracer_and_post_oracle(tmp_path):
    W, H, spp, bl = 112, 64, 6, 4
    sc = make_scene("sponza_like", target_tris=20000)
    out = str(tmp_path / "shots")
    g, o = ra.Renderer(W, H), Oracle(W, H, threads=16)
    setup(g, sc, NOISE, SKY, bounce_limit=bl)
    setup(o, sc, NOISE, SKY, bounce_limit=bl)
    n = len(POSES)
    seeds = video.pose_seeds(0, n, spp)
    assert (seeds.reshape(-1, 3) == controls.rand_seed_triples(n * spp)).all()
    bright = 0
    for k in range(n):
        # as rank k of n this call renders pose k alone, with pose k's slice of the seed stream
        st = video.render_poses(g, POSES, spp, out, rank=k, world=n, compression=k % 2)
        assert (st["rendered"], st["skipped"], st["written"], st["write_failures"]) == (1, 0, 1, 0)
        assert st["rays"] >= W * H * spp and st["trace_seconds"] > 0
        o.set_camera(**oracle_camera(POSES[k]))
        o.clear()
        o.render(0, seeds[k])
        want_accum = o.read_accum()
        assert differing_pixels(g.read_accum(), want_accum) == 0, "pose %d: accumulation differs from the oracle" % k
        _, want8 = PostOracle(W, H).run(want_accum)          # a fresh chain per image: nothing carried over
        got8 = controls.read_png(video.image_path(out, k), 3)
        assert got8.shape == (H, W, 3)
        diff = np.abs(got8.astype(int) - want8.astype(int))
        assert diff.max() <= 1 and (diff != 0).any(-1).mean() < 0.005, "pose %d" % k
        bright += int((want_accum[..., :3] > 1.5).any())
    assert sorted(os.listdir(out)) == sorted("raytrace-out-%d.png" % k for k in range(n))
    g.close()


def test_flip_writes_the_mirrored_image(tmp_path):
    W, H, spp = 64, 40, 3
    sc = make_scene("lightbox")
    g = ra.Renderer(W, H)
    setup(g, sc, NOISE, SKY, bounce_limit=4)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    video.render_poses(g, POSES[:1], spp, a)
    video.render_poses(g, POSES[:1], spp, b, flip=True)
    assert (controls.read_png(video.image_path(a, 0)) == controls.read_png(video.image_path(b, 0))[::-1]).all()
    g.close()


def test_full_hd_images_do_not_depend_on_the_rank_count_and_jobs_resume(tmp_path):
    W, H, spp, n = 1920, 1080, 4, 8
    sc = make_scene("sponza_like", target_tris=20000)
    poses = video.camera_path(n)
    g = ra.Renderer(W, H)
    setup(g, sc, scenes.noise_texture(), scenes.sky_cubemap(64), bounce_limit=8)
    one, eight = str(tmp_path / "one"), str(tmp_path / "eight")
    st = video.render_poses(g, poses, spp, one, rank=0, world=1)
    assert (st["rendered"], st["written"], st["skipped"]) == (n, n, 0)
    for rank in range(8):                       # one-device rehearsal of the eight ranks of a node
        st = video.render_poses(g, poses, spp, eight, rank=rank, world=8)
        assert (st["rendered"], st["written"]) == (1, 1)
    files = ["raytrace-out-%d.png" % k for k in range(n)]
    assert sorted(os.listdir(eight)) == sorted(files) and sorted(os.listdir(one)) == sorted(files)
    images = {}
    for f in files:
        images[f] = open(os.path.join(one, f), "rb").read()
        assert images[f] == open(os.path.join(eight, f), "rb").read(), f
    assert len({images[f] for f in files}) == n, "different poses must give different images"
    # resume: everything is there -> nothing is rendered; after losing two files exactly those are rendered again
    st = video.render_poses(g, poses, spp, one)
    assert (st["rendered"], st["skipped"], st["written"]) == (0, n, 0)
    os.remove(os.path.join(one, files[2]))
    os.remove(os.path.join(one, files[5]))
    open(os.path.join(one, files[6] + ".part999"), "wb").write(b"half")     # a crashed writer's left-over does not count as an image
    st = video.render_poses(g, poses, spp, one)
    assert (st["rendered"], st["skipped"], st["written"]) == (2, n - 2, 2)
    for f in (files[2], files[5]):
        assert open(os.path.join(one, f), "rb").read() == images[f]
    # without resume every pose is rendered again, into the same files
    st = video.render_poses(g, poses[:3], spp, one, resume=False)
    assert (st["rendered"], st["skipped"]) == (3, 0)
    assert open(os.path.join(one, files[1]), "rb").read() == images[files[1]]
    g.close()
