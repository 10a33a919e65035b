"""Runs selected kernels of a launch sequence from their gfx950 assembly (gcn_interp) instead of their C++ source, inside
the host emulation of the library: the emulated runtime offers every launch to a hook; the hook finds the kernel in the
assembly file by its name and template arguments, executes it with the launch's own argument bytes against the runtime's
memory, and reports it as done. Everything else (host code, the other kernels, the Python layer, the oracle comparison)
runs as in the other emulated tests."""
import ctypes
import os
import re
import subprocess

import gcn_interp as gi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
CSRC = os.path.join(ROOT, "raytracing_amd", "csrc")
# the flags the library's device code is built with (raytracing_amd/csrc/Makefile)
DEVFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fgpu-flush-denormals-to-zero", "-fno-slp-vectorize", "-std=c++17",
            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "raytracing_amd", "host"), "-I" + CSRC]
HOOK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_ubyte), ctypes.c_size_t)


def assemble(source=os.path.join(CSRC, "trace_kernels.hip"), out_dir=os.path.join(HERE, "_build"), extra=(), tag="trace_kernels"):
    """Compiles a HIP source to device assembly with the library's flags (plus `extra`); returns the .s path."""
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, tag + ".s")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + DEVFLAGS + list(extra) + ["-S", "--cuda-device-only", "-o", out, source])
    return out


def mangled_fragment(launch_name):
    """'(trace_kernel<true, false, true, true, false>)' -> '12trace_kernelILb1ELb0ELb1ELb1ELb0EE'; 'retile_kernel' -> '13retile_kernelE'."""
    n = launch_name.strip().strip("()").replace(" ", "")
    m = re.fullmatch(r"(\w+)<([\w,]+)>", n)
    if m:
        base, params = m.group(1), m.group(2).split(",")
        if not all(p in ("true", "false") for p in params):
            return None
        return "%d%sI%sE" % (len(base), base, "".join("Lb%dE" % (p == "true") for p in params))
    return "%d%sE" % (len(n), n) if re.fullmatch(r"\w+", n) else None


class FromAssembly:
    """with Emulated(...), FromAssembly(asm_path, kernels=("trace_kernel",)) as run: ... ; run.launches lists what it executed."""

    def __init__(self, asm_path, kernels=("trace_kernel",), budget=None, trace=None, garbage=None, check_waits=False):
        self.garbage = garbage
        self.check_waits = check_waits
        self.kernels, self.blobs = gi.parse_assembly(asm_path)
        self.select = tuple(kernels)
        self.budget = budget
        self.trace = trace
        self.launches = []          # (kernel name, grid, instructions executed, LDS accesses out of range)
        self.error = None
        self.machines = {}
        self._shared = []

    def _hook(self, name, dims, args, nbytes):
        try:
            name = name.decode()
            if not any(name.strip("() ").startswith(s) for s in self.select):
                return 0
            frag = mangled_fragment(name)
            hits = [k for k in self.kernels if frag and frag in k]
            if len(hits) != 1:
                raise gi.Unsupported("no kernel in the assembly for the launch of %s" % name)
            if hits[0] not in self.machines:
                self.machines[hits[0]] = gi.Machine(self.kernels[hits[0]], self.blobs, self.lib, trace=self.trace, garbage=self.garbage, check_waits=self.check_waits)
            m = self.machines[hits[0]]
            before, oob = m.executed, m.lds_out_of_range
            m.launch((dims[0], dims[1], dims[2]), (dims[3], dims[4], dims[5]), bytes(bytearray(args[:nbytes])), max_instructions=(before + self.budget) if self.budget else None)
            self.launches.append((name, dims[0], m.executed - before, m.lds_out_of_range - oob))
            return 1
        except Exception as e:          # an exception cannot cross the C frames between here and the test
            self.error = e
            return 1

    def __enter__(self):
        from raytracing_amd import _lib
        self.lib = _lib.lib()
        assert self.lib._name.endswith("librt_emulated.so"), "only inside the host emulation"
        self._cb = HOOK(self._hook)
        self.lib.emu_set_launch_hook.argtypes = [HOOK]
        self.lib.emu_set_launch_hook(self._cb)
        return self

    def __exit__(self, *exc):
        self.lib.emu_set_launch_hook(HOOK(0))
        for a in self._shared:
            self.lib.emu_forget_range(ctypes.c_void_p(a.ctypes.data))
        for m in self.machines.values():
            m.close()
        return False

    def share(self, *arrays):
        """Host arrays handed to the library as if they were device buffers (the emulation's device memory is host memory)
        are no allocations of the runtime: this enters them in its table, so that the interpreter's address check accepts
        them, until the block ends."""
        for a in arrays:
            self.lib.emu_register_range(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes))
            self._shared.append(a)

    def wait_violations(self):
        """Instructions that named a register whose memory result had not been waited for (check_waits=True)."""
        return [v for m in self.machines.values() for v in m.wait_violations]

    def check(self):
        """Raises what a launch raised (a stray address, an instruction that is not modelled)."""
        if self.error is not None:
            e, self.error = self.error, None
            raise e
