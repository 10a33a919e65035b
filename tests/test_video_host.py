"""Host side of the offline video path that needs no GPU: the seed stream is cut per pose, pose files round-trip,
the PNG writer's output decodes to what went in."""
import numpy as np

from raytracing_amd import controls, video


def test_pose_seeds_are_slices_of_the_reference_stream():
    spp = 5
    whole = controls.rand_seed_triples(7 * spp).reshape(7, spp, 3)
    assert (video.pose_seeds(0, 7, spp) == whole).all()
    assert (video.pose_seeds(3, 2, spp) == whole[3:5]).all()
    assert video.pose_seeds(0, 0, spp).shape == (0, spp, 3)


def test_pose_file_roundtrip(tmp_path):
    poses = video.camera_path(9)
    p = tmp_path / "poses.txt"
    p.write_text("\n".join(" ".join("%.9g" % v for v in row) for row in poses) + "\n")
    assert (video.read_poses(str(p)) == poses).all()
    assert len(video.read_poses(str(p))) == 9


def test_png_writer_roundtrip(tmp_path):
    from raytracing_amd import _lib
    L = _lib.lib()
    img = np.random.default_rng(3).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    img[10:20] = 200                                   # something compressible too
    p = str(tmp_path / "img.png")
    assert L.rth_write_png_rgb8(p.encode(), 53, 37, img.ctypes.data, 0) == 0
    assert (controls.read_png(p) == img).all()
    from PIL import Image
    assert (np.asarray(Image.open(p)) == img).all()    # and an independent decoder agrees
