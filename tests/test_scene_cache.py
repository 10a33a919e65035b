"""The binary scene cache: what comes back from it is what the loaders and the builder produced, bit for bit; a changed,
reordered or damaged input never yields a stale scene."""
import hashlib
import os
import time

import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import scenes


def same_scene(a, b):
    return ((a.tri12.view(np.uint32) == b.tri12.view(np.uint32)).all() and (a.matidx == b.matidx).all() and
            (a.node8.view(np.uint32) == b.node8.view(np.uint32)).all() and (a.materials.view(np.uint32) == b.materials.view(np.uint32)).all() and
            a.warnings == b.warnings and {k: v for k, v in a.stats.items() if k != "build_ms"} == {k: v for k, v in b.stats.items() if k != "build_ms"})


def blob_files(d):
    return sorted(f for f in os.listdir(d) if f.endswith(".rtscene"))


def test_second_load_comes_from_the_cache_and_is_identical(tmp_path):
    paths = scenes.write_objects(scenes.SCENES["sponza_like"](target_tris=20000), str(tmp_path / "objs"))
    cache = str(tmp_path / "cache")
    plain = ra.load_scene(paths, build_threads=2)
    first = ra.load_scene(paths, build_threads=2, cache_dir=cache)
    assert not first.cache_hit and len(blob_files(cache)) == 1
    second = ra.load_scene(paths, build_threads=1, cache_dir=cache)
    assert second.cache_hit
    assert same_scene(plain, first) and same_scene(plain, second)
    assert second.stats["build_ms"] == first.stats["build_ms"]          # the stored figure, not a new measurement


def test_key_is_sha256_of_version_and_file_bytes(tmp_path):
    """The blob's name can be recomputed from the inputs alone."""
    import struct
    paths = scenes.write_objects(scenes.lightbox(), str(tmp_path / "objs"))
    cache = str(tmp_path / "cache")
    ra.load_scene(paths, cache_dir=cache)
    h = hashlib.sha256(struct.pack("<3I", 1, 3, len(paths)))
    for p in paths:
        data = open(p, "rb").read()
        h.update(struct.pack("<3Q", 1, len(data), int(".off" in p or ".OFF" in p)))
        h.update(data)
    assert blob_files(cache) == [h.hexdigest() + ".rtscene"]
    # the portable SHA-256 code (used on CPUs without SHA instructions) gives the same key: a fresh process with the
    # switch set finds the blob
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import raytracing_amd as ra; "
            "print(ra.load_scene(%r, cache_dir=%r).cache_hit)" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), paths, cache))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RT_SHA256_PORTABLE="1"), capture_output=True, text=True, check=True)
    assert out.stdout.strip() == "True"


def test_changed_reordered_or_missing_inputs_miss(tmp_path):
    paths = scenes.write_objects(scenes.lightbox(), str(tmp_path / "objs"))
    assert len(paths) >= 2
    cache = str(tmp_path / "cache")
    base = ra.load_scene(paths, cache_dir=cache)
    assert ra.load_scene(paths, cache_dir=cache).cache_hit
    # an edited file: one vertex moved
    text = open(paths[0]).read()
    lines = text.split("\n")
    for i, line in enumerate(lines):
        parts = line.split()
        if len(parts) == 9:
            parts[0] = repr(float(parts[0]) + 0.125)
            lines[i] = " ".join(parts)
            break
    os.utime(paths[0])                                                  # a touch alone changes nothing ...
    assert ra.load_scene(paths, cache_dir=cache).cache_hit
    open(paths[0], "w").write("\n".join(lines))                          # ... changed bytes do
    edited = ra.load_scene(paths, cache_dir=cache)
    assert not edited.cache_hit and not same_scene(base, edited)
    assert same_scene(edited, ra.load_scene(paths))
    # another order of the same files is another scene (material slots follow the order)
    swapped = ra.load_scene(paths[::-1], cache_dir=cache)
    assert not swapped.cache_hit and same_scene(swapped, ra.load_scene(paths[::-1]))
    # a file that cannot be read is part of the key too
    gone = ra.load_scene(paths + [str(tmp_path / "nope.txt")], cache_dir=cache)
    assert not gone.cache_hit and gone.warnings
    again = ra.load_scene(paths + [str(tmp_path / "nope.txt")], cache_dir=cache)
    assert again.cache_hit and again.warnings == gone.warnings


def test_damaged_or_foreign_blobs_are_ignored_and_rewritten(tmp_path):
    paths = scenes.write_objects(scenes.lightbox(), str(tmp_path / "objs"))
    cache = str(tmp_path / "cache")
    good = ra.load_scene(paths, cache_dir=cache)
    blob = os.path.join(cache, blob_files(cache)[0])
    data = open(blob, "rb").read()
    for bad in (data[:len(data) // 2],                                   # truncated
                data[:200] + bytes([data[200] ^ 1]) + data[201:],        # one bit flipped in the payload
                data[:4] + b"\x63\0\0\0" + data[8:],                      # another format version
                data[:8] + bytes(32) + data[40:],                         # another key inside
                b""):
        open(blob, "wb").write(bad)
        sc = ra.load_scene(paths, cache_dir=cache)
        assert not sc.cache_hit and same_scene(sc, good)
        again = open(blob, "rb").read()                                   # rewritten: as before but for the measured build
        assert len(again) == len(data) and again[:88] == data[:88] and again[96:-8] == data[96:-8]   # time (and with it the checksum)
    assert ra.load_scene(paths, cache_dir=cache).cache_hit


def test_cache_is_at_least_ten_times_faster_on_a_large_scene(tmp_path):
    """The dragon-like scene of BASELINE configuration 3 (about 870 000 triangles, a 30 MB .off file)."""
    paths = scenes.write_objects(scenes.SCENES["dragon_like"](target_tris=870_000), str(tmp_path / "objs"))
    cache = str(tmp_path / "cache")
    t0 = time.perf_counter()
    first = ra.load_scene(paths, build_threads=8, cache_dir=cache)
    slow = time.perf_counter() - t0
    t0 = time.perf_counter()
    second = ra.load_scene(paths, build_threads=8, cache_dir=cache)
    fast = time.perf_counter() - t0
    assert not first.cache_hit and second.cache_hit and first.n_triangles > 800_000
    assert same_scene(first, second)
    # a generous bound (typically 30 times faster): the point is that a hit skips parse and build, not a timing race
    assert fast * 3 <= slow, "cached load %.2f s, uncached %.2f s" % (fast, slow)
