"""The library's own HIP and C++ sources compiled as host C++ (tests/emu/) and loaded under the Python layer in
place of librt_mi355x.so, for CPU tests: the kernels run lane by lane on fibers that meet at every wave-wide
operation, device memory is host memory, several devices and RCCL are imitated in-process. The tests that use it
check the product's kernel and host logic against the same oracle as the GPU tests; they say nothing about speed,
and nothing in the package can reach this library."""
import ctypes
import os
import subprocess

from raytracing_amd import _lib

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
EMU_BUILD = os.environ.get("RT_EMU_BUILD", "_build")           # "_build_asan": the AddressSanitizer build (tests/emu/Makefile)
EMU_LIB = os.path.join(EMU_DIR, EMU_BUILD, "librt_emulated.so")
_loaded = None


def library():
    """Builds (if stale) and loads the emulated library, with the package's signature table applied."""
    global _loaded
    if _loaded is None:
        r = subprocess.run(["make", "-C", EMU_DIR, "-j4"] + (["asan"] if EMU_BUILD == "_build_asan" else ["B=" + EMU_BUILD]), capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building the emulated library failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
        L = ctypes.CDLL(EMU_LIB)
        for name, (res, args) in _lib.SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _loaded = L
    return _loaded


class Emulated:
    """Context manager: inside it the package talks to the emulated library. `devices` and `cus` are what the
    imitated runtime reports; `threads` host threads run the work-groups of a launch."""

    def __init__(self, devices=1, cus=2, threads=4):
        self.env = {"RT_EMU_DEVICES": str(devices), "RT_EMU_CUS": str(cus), "RT_EMU_THREADS": str(threads)}

    def __enter__(self):
        self.saved_lib = _lib._lib
        self.saved_env = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)
        _lib._lib = library()
        return self

    def __exit__(self, *exc):
        _lib._lib = self.saved_lib
        for k, v in self.saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return False
