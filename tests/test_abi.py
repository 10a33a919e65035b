"""The C-ABI library loads (no GPU needed for that) and exports every function the headers declare."""
import ctypes
import os
import re

import pytest

from raytracing_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in ("rt_mi355x.h", "rt_host.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_loads_and_exports_all_declared_symbols():
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), "missing export: " + n


def test_python_binding_covers_the_headers():
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_create_without_gpu_fails_cleanly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = _lib.lib()
    h = ctypes.c_void_p()
    rc = L.rt_create(0, 64, 64, ctypes.byref(h))
    assert rc != 0 and not h.value
    assert b"HIP device" in L.rt_last_global_error() or L.rt_last_global_error()


def test_oracle_is_not_reachable_from_the_product_package():
    pkg = os.path.join(ROOT, "raytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "rt_oracle" not in text and "librt_oracle" not in text and "orc_" not in text, os.path.join(dirpath, f)


def test_ctypes_structs_match_the_c_headers(tmp_path):
    """Size and every field offset of the Python mirrors, against what a C compiler makes of include/*.h."""
    import subprocess
    structs = {"rt_material": _lib.rt_material, "rt_camera": _lib.rt_camera, "rt_counters": _lib.rt_counters,
               "rth_bvh_stats": _lib.rth_bvh_stats, "rth_app_state": _lib.rth_app_state}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rt_host.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (name, name))
        for field, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (name, field, name, field))
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for line in filter(None, out):
        name, field, value = line.split()
        cls = structs[name]
        want = ctypes.sizeof(cls) if field == "size" else getattr(cls, field).offset
        assert int(value) == want, line
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in structs.values())
