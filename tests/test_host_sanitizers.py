"""The host-side readers (object files, scene cache, PNG, INI) under AddressSanitizer and UndefinedBehaviorSanitizer:
tests/host_sanitize_main.cpp is built from the host sources with g++ -fsanitize=address,undefined and run on the good
files, on truncations of them and on a few thousand randomly damaged copies. Host code only (no GPU code is built with
sanitizers). Skipped where g++ has no sanitizer runtime."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "raytracing_amd", "host")
SOURCES = ["bvh_tree.cpp", "object_loader.cpp", "scene_data.cpp", "scene_cache.cpp", "config_parser.cpp", "image_io.cpp"]


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def test_host_readers_survive_damaged_files_under_sanitizers(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I" + HOST, "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "host_sanitize_main.cpp")] + \
          [os.path.join(HOST, s) for s in SOURCES] + ["-lpthread", "-lz"]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    # skipped only when the sanitizer runtimes themselves are missing (a linker complaint about libasan / libubsan), never
    # for a compile error in the sources
    if build.returncode != 0 and " error: " not in build.stderr and ("-lasan" in build.stderr or "-lubsan" in build.stderr or "libasan" in build.stderr or "libubsan" in build.stderr):
        pytest.skip("g++ cannot link the sanitizer runtimes here")
    assert build.returncode == 0, build.stderr[-3000:]
    png = str(tmp_path / "seed.png")
    write_png(png, np.random.default_rng(1).integers(0, 256, (13, 17, 3), dtype=np.uint8))
    ini = str(tmp_path / "seed.ini")
    with open(ini, "w") as f:
        f.write("; comment\n[SkyBox]\nPath = Textures/DaylightBox/\n[Objects]\nDirectoryPath = Objects/\nFilenames = a.off, b.txt, c.txt\n[Shaders]\nCount = 0x10\n")
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "objects"), str(scratch), png, ini], capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, (run.stdout[-2000:] + run.stderr[-6000:])
    assert "host_sanitize:" in run.stdout


def test_kernels_on_the_host_emulation_under_address_sanitizer():
    """The library's kernels and C ABI compiled as host C++ (tests/emu) with AddressSanitizer, on a selection of the
    emulated parity cases: device buffers are heap blocks there, so a kernel or host index past the end of one (tally
    columns, spill rows, tile order, staging and receive buffers of a device group, the fixed post-processing array) is
    reported instead of passing by luck. Skipped where clang's sanitizer runtime is missing."""
    import glob
    import sys
    runtimes = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not runtimes:
        pytest.skip("no AddressSanitizer runtime for clang here")
    env = dict(os.environ, RT_EMU_BUILD="_build_asan", LD_PRELOAD=runtimes[-1],
               ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0")
    cases = "(radiance and lightbox) or deepest or glass or three_devices or folded or allocates or (shapes and 5x33)"
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_emulated.py"), "-q", "-x", "-p", "no:cacheprovider", "-k", cases],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    out = run.stdout[-3000:] + run.stderr[-6000:]
    assert "ERROR: AddressSanitizer" not in run.stdout + run.stderr, out
    assert run.returncode == 0, out
    assert " passed" in run.stdout and "failed" not in run.stdout, out
