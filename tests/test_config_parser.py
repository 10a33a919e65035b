"""The ini reader (raytracing_amd/host/config_parser.cpp, through the host C ABI) against the reference's own
ConfigParser: tests/golden/ini_reference.npz holds, for the ini files under tests/golden/ini/, what the reference's
src/ConfigParser/ConfigParser.cpp (compiled from its sources by oracle/Makefile) returned: the value list of every
queried key (white space inside values, ';' in the middle of a line, duplicate keys, empty and trailing-comma values,
keys outside any section), typed values (decimal, hexadecimal, float, bool spellings, text that is no number), and the
exception message for lines without '=' (with the reference's line count, which skips section lines)."""
import json
import os

import numpy as np
import pytest

from raytracing_amd import controls
from util import GOLDEN

CASES = json.loads(str(np.load(os.path.join(GOLDEN, "ini_reference.npz"))["cases_json"]))
KIND = {"int": int, "float": float, "bool": bool}


@pytest.fixture(autouse=True)
def in_golden_dir(monkeypatch):
    monkeypatch.chdir(GOLDEN)      # the reference quotes the path as given in its messages; the fixture used relative ones


def test_fixture_covers_the_quirks():
    assert len(CASES) >= 12
    assert sum(1 for c in CASES.values() if c["error"]) >= 3
    assert ["A", "trail", ["a", ""]] in CASES["empty_values"]["values"]
    assert ["A", "k", ["first", "1"]] in CASES["duplicates"]["values"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_ini_reader_equals_reference(name):
    case = CASES[name]
    if case["error"]:
        with pytest.raises(ValueError) as e:
            controls.ini_values(case["path"], "A", "k")
        assert str(e.value) == case["error"]
        return
    for section, key, values in case["values"]:
        assert controls.ini_values(case["path"], section, key) == values, (section, key)
    for section, key, pos, kind, value in case["typed"]:
        got = controls.ini_number(case["path"], section, key, pos, KIND[kind])
        assert float(got) == pytest.approx(value, rel=0, abs=0), (section, key, pos, kind)


def test_reference_parser_live_when_present():
    """Where the reference tree and its compiled parser exist (build container), the fixture is re-derived and compared."""
    lib = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libref_config.so")
    if not (os.path.isdir("/root/reference/src") and os.path.exists(lib)):
        pytest.skip("reference tree not present")
    import ctypes
    L = ctypes.CDLL(lib)
    L.ref_ini_open.restype = ctypes.c_void_p
    L.ref_ini_open.argtypes = [ctypes.c_char_p]
    L.ref_ini_close.argtypes = [ctypes.c_void_p]
    L.ref_ini_error.restype = ctypes.c_char_p
    L.ref_ini_error.argtypes = [ctypes.c_void_p]
    L.ref_ini_values.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p]
    L.ref_ini_text.restype = ctypes.c_char_p
    L.ref_ini_text.argtypes = [ctypes.c_void_p]
    for name, case in CASES.items():
        h = L.ref_ini_open(case["path"].encode())
        assert L.ref_ini_error(h).decode() == case["error"], name
        if not case["error"]:
            for section, key, values in case["values"]:
                n = L.ref_ini_values(h, section.encode(), key.encode())
                assert L.ref_ini_text(h).decode().split("\n")[:n] == values, (name, section, key)
        L.ref_ini_close(h)
