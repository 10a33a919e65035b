"""The kernel experiments that are kept as patches (profiles/*.patch, described in profiles/EXPERIMENTS.md) are work to be
taken up again: each must still apply to the sources as they are."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CURRENT = ("r03_two_rays_per_lane.patch", "r03_gather_memory_lanes.patch", "r03_texel_pairs.patch")


@pytest.mark.skipif(not os.path.isdir(os.path.join(ROOT, ".git")), reason="needs the git work tree")
@pytest.mark.parametrize("name", CURRENT)
def test_patch_applies(name):
    path = os.path.join(ROOT, "profiles", name)
    assert os.path.exists(path)
    r = subprocess.run(["git", "apply", "--check", path], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
