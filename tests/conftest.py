import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_addoption(parser):
    parser.addoption("--emulated", action="store_true", default=False,
                     help="run the session against the host emulation of the HIP library (tests/emu): lets the tests marked "
                          "gpu that need no torch device run on a machine without a GPU, slowly (-m gpu --emulated)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _emulated_session(request):
    if not request.config.getoption("--emulated"):
        yield
        return
    from emulation import Emulated
    with Emulated(devices=int(os.environ.get("RT_EMU_DEVICES", "1"))):
        yield


@pytest.fixture
def emulated():
    """The package bound to the emulated library with one imitated device, for the duration of a test."""
    from emulation import Emulated
    with Emulated():
        yield
