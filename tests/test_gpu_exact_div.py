"""The short quotient sequences of csrc/exact_div.h against the compiler's correctly rounded division, on every one
of the 2^32 fp32 inputs (raytracing_amd/exact_div_check, built with the library, runs the comparison on the GPU).
1/x, 0.5/x, sqrt(x) and 1/sqrt(x) must agree everywhere; a/c for launch constants c must agree wherever the residual cannot underflow:
differences are allowed only for |a| < 2^-100, for infinite a, and in the sign of a flushed negative denormal."""
import os
import re
import subprocess

import pytest

EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracing_amd", "exact_div_check")


@pytest.mark.gpu
def test_short_quotients_equal_division_on_all_inputs():
    assert os.path.exists(EXE), "raytracing_amd/exact_div_check is missing: run __graft_entry__.build()"
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=150, check=True).stdout
    sections, cur = {}, None
    for line in out.splitlines():
        m = re.match(r"(\S+)\s+differing inputs: (\d+)", line)
        if m:
            cur = m.group(1)
            sections[cur] = {"bad": int(m.group(2)), "runs": []}
            continue
        m = re.match(r"\s+([+-]) exponent (\d+)\.\.(\d+): (\d+)", line)
        if m and cur:
            sections[cur]["runs"].append((m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))))
    assert sections["1/x"]["bad"] == 0
    assert sections["0.5/x"]["bad"] == 0
    assert sections["sqrt(x)"]["bad"] == 0
    assert sections["1/sqrt(x)"]["bad"] == 0
    consts = [k for k in sections if k.startswith("x/")]
    assert len(consts) >= 20
    for k in consts:
        for sign, e0, e1, n in sections[k]["runs"]:
            tiny = e1 <= 26                       # |a| < 2^-100 (biased exponent 27), denormals included
            infinite = e0 == 255 and e1 == 255 and n == 1
            assert tiny or infinite, (k, sign, e0, e1, n)
