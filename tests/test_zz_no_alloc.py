"""rt_postprocess_static, the bloom chain that allocates nothing, against rt_postprocess on the same images.

Bar: bit-exact (the two entry points run the same kernels on the same values). Two calls in a row share the
library's fixed device array, the second with a smaller image, so anything the first call leaves behind in it
would show."""
import ctypes

import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import _lib
from util import bits

pytestmark = pytest.mark.gpu


def hdr_image(w, h, seed):
    """Mostly dim pixels with bright spots above the bloom threshold, like a rendered interior with lamps."""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w, 4), dtype=np.float32)
    img[..., :3] = rng.random((h, w, 3), dtype=np.float32) * 0.9
    for _ in range(max(3, w * h // 4000)):
        x, y, r = int(rng.integers(0, w)), int(rng.integers(0, h)), int(rng.integers(1, 6))
        img[max(0, y - r):y + r, max(0, x - r):x + r, :3] = rng.random(3, dtype=np.float32) * 6.0 + 0.5
    return img


def reference_chain(img):
    """The existing entry point on a fresh context: write the image, run the chain once."""
    h, w = img.shape[:2]
    r = ra.Renderer(w, h)
    r.write_accum(img)
    rgba, rgb8 = r.postprocess(True)
    r.close()
    return rgba, rgb8


def static_chain(img, want_rgba=True, want_rgb8=True):
    import torch
    L = _lib.lib()
    h, w = img.shape[:2]
    src = torch.from_numpy(img).cuda()
    rgba = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda") if want_rgba else None
    rgb8 = torch.full((h, w, 3), 7, dtype=torch.uint8, device="cuda") if want_rgb8 else None
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    rc = L.rt_postprocess_static(src.data_ptr(), w, h, rgba.data_ptr() if want_rgba else None, rgb8.data_ptr() if want_rgb8 else None,
                                 ctypes.c_void_p(stream))
    assert rc == 0, L.rt_last_global_error().decode()
    torch.cuda.synchronize()
    return (rgba.cpu().numpy() if want_rgba else None), (rgb8.cpu().numpy() if want_rgb8 else None)


def test_two_calls_in_a_row_second_one_smaller_equal_rt_postprocess():
    big, small = hdr_image(640, 360, 1), hdr_image(203, 117, 2)
    got_big = static_chain(big)
    got_small = static_chain(small)            # shares the fixed array with the call before
    for img, (rgba, rgb8) in ((big, got_big), (small, got_small)):
        want_rgba, want_rgb8 = reference_chain(img)
        assert (want_rgba[..., :3] > 1.0).any(), "the test image must trigger bloom"
        assert (bits(rgba) == bits(want_rgba)).all()
        assert (rgb8 == want_rgb8).all()


def test_full_hd_and_single_outputs():
    img = hdr_image(1920, 1080, 3)
    want_rgba, want_rgb8 = reference_chain(img)
    rgba, _ = static_chain(img, want_rgb8=False)
    _, rgb8 = static_chain(img, want_rgba=False)
    assert (bits(rgba) == bits(want_rgba)).all()
    assert (rgb8 == want_rgb8).all()
    tiny = hdr_image(3, 2, 4)                 # smaller than every down-sampling factor
    a, b = static_chain(tiny)
    wa, wb = reference_chain(tiny)
    assert (bits(a) == bits(wa)).all() and (b == wb).all()


def test_no_device_memory_is_taken_and_oversized_images_are_refused():
    import torch
    L = _lib.lib()
    img = hdr_image(320, 200, 5)
    static_chain(img)                          # warm-up: code object, symbol lookup
    src = torch.from_numpy(img).cuda()
    out = torch.empty((200, 320, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        assert L.rt_postprocess_static(src.data_ptr(), 320, 200, None, out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    # beyond what the fixed array holds: a clear error, nothing launched
    rc = L.rt_postprocess_static(src.data_ptr(), 8000, 4500, None, out.data_ptr(), None)
    assert rc == -4 and b"3840x2160" in L.rt_last_global_error()
    assert L.rt_postprocess_static(None, 320, 200, None, out.data_ptr(), None) == -1
    assert L.rt_postprocess_static(src.data_ptr(), 320, 200, None, None, None) == -1
