"""Randomised parity on the GPU: random scene, image size, camera, bounce limit, launch split and shard count,
each compared bit for bit with the CPU oracle (tests/fuzz_parity.py, fixed seeds here)."""
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("seed,cases,large", [(11, 40, False), (12, 6, True)])
def test_random_configurations_bit_exact(seed, cases, large):
    import fuzz_parity
    lines = []
    bad = fuzz_parity.run(cases, seed, large=large, log=lines.append)
    assert bad == 0, "\n".join(l for l in lines if "DIFFERS" in l)
