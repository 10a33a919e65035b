"""An interpreter for the gfx950 assembly the compiler makes of the library's kernels (hipcc -S), for tests only.

The host emulation next door (tests/emu) runs the kernels' C++ SOURCE lane by lane; this runs what the compiler made of
it, wave by wave, with the machine's execution model: 64 lanes in lock step under an EXEC mask, scalar registers, VCC,
SCC, LDS per work-group, per-lane scratch, and global memory that is the emulated runtime's host memory, every address
checked against the runtime's table of allocations. It exists to look below source level without a GPU: a kernel whose
source is right (the fiber emulation agrees with the oracle) and whose machine code is not shows here, and a stray
address is reported with its instruction instead of faulting a machine.

Scope: the instructions these kernels use (about 200 opcodes of SOP*, VOP* with SDWA, DS, GLOBAL, SCRATCH, SMEM), single
precision with the mode register's denormal control honoured, launches of up to three dimensions, no timing, no hardware
hazards (a missing s_nop cannot show; a missing s_waitcnt can, see check_waits). v_rcp_f32, v_rsq_f32, v_exp_f32 and
v_log_f32 return correctly rounded results where the hardware is within one ulp: the tracer's division and square-root
sequences correct their estimate to the exact result either way, the bloom chain is compared within a tolerance.
"""
import ctypes
import re
import struct

import numpy as np
import yaml

U32, F32, U64, I32 = np.uint32, np.float32, np.uint64, np.int32
LANES = np.arange(64, dtype=np.uint64)
FULL = (1 << 64) - 1
TINY = np.float32(1.17549435e-38)


class StrayAccess(Exception):
    pass


class Unsupported(Exception):
    pass


def mask_to_bools(m):
    return np.unpackbits(np.frombuffer(int(m).to_bytes(8, "little"), np.uint8), bitorder="little").astype(bool)


def bools_to_mask(b):
    return int.from_bytes(np.packbits(b, bitorder="little").tobytes(), "little")


# ---------------------------------------------------------------- parsing
class Instr:
    __slots__ = ("op", "args", "mods", "text", "line")

    def __init__(self, op, args, mods, text, line):
        self.op, self.args, self.mods, self.text, self.line = op, args, mods, text, line


class Kernel:
    def __init__(self, name):
        self.name = name
        self.instrs = []
        self.labels = {}
        self.desc = {}
        self.meta = None


def _split_operands(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "[(":
            depth += 1
        elif ch in "])":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _split_blanks(s):
    """Splits at blanks that are outside parentheses: 'hwreg(HW_REG_MODE, 4, 2)' stays whole."""
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch == "("
        depth -= ch == ")"
        if ch.isspace() and depth == 0:
            if cur:
                out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur:
        out.append(cur)
    return out or [""]


def parse_assembly(path):
    """Returns ({kernel name: Kernel}, {data symbol: bytes}) for one device assembly file."""
    kernels, data = {}, {}
    cur, cur_data, desc_for = None, None, None      # function being read, data symbol being read, kernel whose descriptor is being read
    meta_lines, in_meta = [], False
    for ln, raw in enumerate(open(path), 1):
        line = raw.rstrip("\n")
        if in_meta:
            if ".end_amdgpu_metadata" in line:
                in_meta = False
            else:
                meta_lines.append(line)
            continue
        if line.strip() == ".amdgpu_metadata":
            in_meta = True
            continue
        code = line.split(";")[0].rstrip()
        st = code.strip()
        if not st:
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", code)
        if m and not code.startswith("\t"):
            name = m.group(1)
            if name.startswith(".Lfunc_end"):
                cur = None
            elif name.startswith(".L"):
                if cur is not None:
                    cur.labels[name] = len(cur.instrs)      # a branch target inside the current function
            else:
                # a function or a data symbol: what follows decides (instructions, or .long words)
                cur = kernels.setdefault(name, Kernel(name))
                cur_data = name
                data[name] = bytearray()
            continue
        if st.startswith(".amdhsa_kernel"):
            desc_for = kernels.get(st.split()[1])
            continue
        if st.startswith(".end_amdhsa_kernel"):
            desc_for = None
            continue
        if desc_for is not None and st.startswith(".amdhsa_"):
            k, v = st.split()[:2]
            desc_for.desc[k[len(".amdhsa_"):]] = int(v, 0)
            continue
        if st.startswith(".long") and cur_data is not None:
            data[cur_data] += struct.pack("<I", int(st.split()[1], 0) & 0xffffffff)
            continue
        if st.startswith("."):
            continue
        if cur is None:
            continue
        parts = st.split(None, 1)
        op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        # operands are separated by commas; modifiers follow the last operand, separated by blanks ("v10 dst_sel:DWORD",
        # "off offset:24", "s[6:7] offset:16 sc0"); an instruction may also have modifiers only ("vmcnt(0) lgkmcnt(0)")
        toks = _split_operands(rest)
        mods = {}
        if toks:
            words = _split_blanks(toks[-1])
            flags = ("clamp", "sc0", "sc1", "nt", "glc", "slc")
            is_operand = not (words[0] in flags or (":" in words[0] and not words[0].startswith(("v[", "s["))))
            toks = toks[:-1] + ([words[0]] if is_operand else [])
            for t in words[1 if is_operand else 0:]:
                k, _, v = t.partition(":")
                mods[k] = v if v else True
        args = [a for a in toks if a != ""]
        cur.instrs.append(Instr(op, args, mods, st, ln))
    meta = yaml.safe_load("\n".join(meta_lines)) if meta_lines else {"amdhsa.kernels": []}
    out = {}
    for km in meta.get("amdhsa.kernels", []):
        k = kernels.get(km[".name"])
        if k is not None and k.instrs:
            k.meta = km
            out[km[".name"]] = k
    blobs = {n: bytes(b) for n, b in data.items() if b and n not in out}
    return out, blobs


# ---------------------------------------------------------------- machine state
class Wave:
    def __init__(self, nv):
        self.V = np.zeros((nv, 64), U32)
        self.S = [0] * 128
        self.vcc = 0
        self.exec = FULL
        self.em = np.ones(64, bool)
        self.full = True
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.vm, self.lgkm, self.in_flight = [], [], set()      # memory results not yet waited for (Machine.check_waits)
        self.done = False
        self.at_barrier = False
        self.flush = True          # FP32 denormals flushed (mode register bits 4-5 == 0)
        self.scratch = None

    def set_exec(self, m):
        m &= FULL
        self.exec = m
        self.full = m == FULL
        self.em = mask_to_bools(m)


def _flush(x):
    """FTZ/DAZ on an array of float32: denormals become zeros of their sign."""
    return np.where(np.abs(x) < TINY, np.copysign(F32(0), x), x)


class Machine:
    """Runs one kernel launch. `lib` is the emulated runtime (ctypes) whose allocation table guards global memory."""

    def __init__(self, kernel, blobs, lib, trace=None, garbage=None, check_waits=False):
        """garbage: a seed; registers, LDS and scratch then start with random bits instead of zeros, as on the machine, so
        that a result that depends on something never written shows as a difference."""
        self.k = kernel
        self.lib = lib
        self.trace = trace
        self.blobs = {}
        self._keep = []
        vp = ctypes.c_void_p
        lib.emu_register_range.argtypes, lib.emu_register_range.restype = [vp, ctypes.c_size_t], None
        lib.emu_forget_range.argtypes, lib.emu_forget_range.restype = [vp], None
        lib.emu_known.argtypes, lib.emu_known.restype = [vp, ctypes.c_size_t], ctypes.c_int
        lib.emu_gather.argtypes, lib.emu_gather.restype = [vp, ctypes.c_uint64, ctypes.c_int, vp], ctypes.c_uint64
        lib.emu_scatter.argtypes, lib.emu_scatter.restype = [vp, ctypes.c_uint64, ctypes.c_int, vp], ctypes.c_uint64
        lib.emu_atomic_add_u32.argtypes, lib.emu_atomic_add_u32.restype = [vp, ctypes.c_uint64, vp, vp], ctypes.c_uint64
        lib.emu_vfma.argtypes, lib.emu_vfma.restype = [vp, vp, vp, vp, ctypes.c_int], None
        for name, b in blobs.items():
            buf = ctypes.create_string_buffer(b, len(b))
            self._keep.append(buf)
            lib.emu_register_range(ctypes.addressof(buf), len(b))
            self.blobs[name] = ctypes.addressof(buf)
        self.nv = max(int(kernel.desc.get("next_free_vgpr", 256)), 1)
        self.lds_bytes = int(kernel.desc.get("group_segment_fixed_size", 0))
        self.scratch_bytes = int(kernel.desc.get("private_segment_fixed_size", 0))
        self.garbage = None if garbage is None else np.random.default_rng(garbage)
        # check_waits: memory results arrive late on the machine and s_waitcnt is what makes them safe to use. Every load
        # marks its destination registers as in flight (vector memory: vmcnt, LDS and scalar memory: lgkmcnt, results of one
        # counter arriving in the order issued); an instruction that names a register still in flight is recorded in
        # `wait_violations`: the interpreter itself would have read the right value, the machine a stale one.
        self.histogram = None         # set to [0] * len(kernel.instrs) to count how often each instruction runs (instruction_mix)
        self.check_waits = check_waits
        self.wait_violations = []
        self.access = [self._registers_named(ins) for ins in kernel.instrs] if check_waits else None
        self.executed = 0
        self.lds_debug = None         # a list: collects (instruction index, addresses) of LDS accesses out of range
        self.lds_out_of_range = 0     # lane accesses beyond the work-group's LDS (the hardware returns 0 / drops the write)
        self.compiled = [self._compile(i, ins) for i, ins in enumerate(kernel.instrs)]

    @staticmethod
    def _registers_named(ins):
        """(kind, registers written by a memory result, all registers named), registers as ('v'|'s', index)."""
        def regs(text):
            out = set()
            for m in re.finditer(r"\b([vs])(\d+)\b", text):
                out.add((m.group(1), int(m.group(2))))
            for m in re.finditer(r"\b([vs])\[(\d+):(\d+)\]", text):
                out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
            if re.search(r"\bvcc\b", text): out.add(("s", "vcc"))
            return out
        op = ins.op
        named = regs(" ".join(ins.args))
        kind, dst = None, set()
        if op.startswith(("global_load", "scratch_load", "flat_load")):
            kind, dst = "vm", regs(ins.args[0])
        elif op.startswith(("global_store", "scratch_store", "flat_store")):
            kind = "vm"
        elif op.startswith("global_atomic"):
            kind, dst = "vm", (regs(ins.args[0]) if len(ins.args) == 4 else set())
        elif op.startswith(("ds_read", "ds_bpermute")):
            kind, dst = "lgkm", regs(ins.args[0])
        elif op.startswith("ds_write"):
            kind = "lgkm"
        elif op.startswith("s_load"):
            kind, dst = "lgkm", regs(ins.args[0])
        return kind, frozenset(dst), frozenset(named)

    def _check_wait(self, w, pc):
        ins = self.k.instrs[pc]
        kind, dst, named = self.access[pc]
        if ins.op == "s_waitcnt":
            for name, queue in (("vmcnt", w.vm), ("lgkmcnt", w.lgkm)):
                m = re.search(name + r"\((\d+)\)", ins.text)
                if m:
                    del queue[:max(0, len(queue) - int(m.group(1)))]
            w.in_flight = set().union(*w.vm, *w.lgkm) if (w.vm or w.lgkm) else set()
            return
        if w.in_flight:
            hit = named & w.in_flight
            if hit and kind is not None:
                # results of one counter arrive in the order issued: a later load may name an earlier one's destination as
                # its own destination; what it may not do is read it (an address, store data) or belong to the other counter
                same = set().union(*(w.vm if kind == "vm" else w.lgkm)) if (w.vm if kind == "vm" else w.lgkm) else set()
                hit = {r for r in hit if not (r in dst and r in same and r not in self._sources(ins))}
            if hit:
                self.wait_violations.append((ins.line, ins.text, sorted(hit, key=str)))
        if kind == "vm":
            w.vm.append(dst); w.in_flight |= dst
        elif kind == "lgkm":
            w.lgkm.append(dst); w.in_flight |= dst

    @staticmethod
    def _sources(ins):
        out = set()
        for a in ins.args[1:]:
            for m in re.finditer(r"\b([vs])(\d+)\b", a):
                out.add((m.group(1), int(m.group(2))))
            for m in re.finditer(r"\b([vs])\[(\d+):(\d+)\]", a):
                out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        return out

    def instruction_mix(self):
        """Executed wave-instructions by kind (needs `histogram`): vector ALU, scalar ALU, branches, vector memory (global and
        scratch), LDS, scalar memory, waits and other."""
        mix = {}
        for ins, n in zip(self.k.instrs, self.histogram):
            op = ins.op
            kind = ("vmem" if op.startswith(("global_", "flat_")) else "scratch" if op.startswith("scratch_") else "lds" if op.startswith("ds_") else
                    "smem" if op.startswith("s_load") else "wait" if op in ("s_waitcnt", "s_nop") else
                    "branch" if op.startswith(("s_cbranch", "s_branch")) else "valu" if op.startswith("v_") else "salu")
            mix[kind] = mix.get(kind, 0) + n
        return mix

    def close(self):
        for buf in self._keep:
            self.lib.emu_forget_range(ctypes.addressof(buf))
        self._keep = []

    # ---- operand access ----
    def _reg(self, t):
        """('v'|'s', first, count) or a special name, or None."""
        m = re.fullmatch(r"([vs])(\d+)", t)
        if m:
            return (m.group(1), int(m.group(2)), 1)
        m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", t)
        if m:
            return (m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1)
        return None

    def _imm32(self, t):
        if re.fullmatch(r"-?0x[0-9a-fA-F]+", t) or re.fullmatch(r"-?\d+", t):
            return int(t, 0) & 0xffffffff
        if re.fullmatch(r"-?\d+\.\d*(e[-+]?\d+)?", t) or re.fullmatch(r"-?\d+e[-+]?\d+", t):
            return struct.unpack("<I", struct.pack("<f", float(t)))[0]
        return None

    def src32(self, t):
        """Returns f(wave) -> uint32 array[64] or python int (scalar), for a 32-bit source operand without modifiers."""
        r = self._reg(t)
        if r:
            kind, n, cnt = r
            if kind == "v":
                return lambda w, n=n: w.V[n]
            return lambda w, n=n: w.S[n]
        imm = self._imm32(t)
        if imm is not None:
            return lambda w, imm=imm: imm
        if t == "vcc_lo": return lambda w: w.vcc & 0xffffffff
        if t == "vcc_hi": return lambda w: w.vcc >> 32
        if t == "exec_lo": return lambda w: w.exec & 0xffffffff
        if t == "exec_hi": return lambda w: w.exec >> 32
        if t == "m0": return lambda w: w.m0
        if t == "scc": return lambda w: w.scc
        if t == "src_scc": return lambda w: w.scc
        m = re.fullmatch(r"(\w+)@rel32@(lo|hi)\+(\d+)", t)
        if m:
            # s_getpc_b64 + this pair form the address of a data symbol: getpc yields 0 here, the pair the symbol's address
            addr = self.blobs[m.group(1)]
            return (lambda w, v=addr & 0xffffffff: v) if m.group(2) == "lo" else (lambda w, v=addr >> 32: v)
        raise Unsupported("operand " + t)

    def srcf(self, t):
        """f(wave) -> float32 array[64] for a float source with optional -x, |x|, -|x|."""
        neg = absv = False
        if t.startswith("-"):
            probe = t[1:]
            if probe.startswith("|") or self._reg(probe):
                neg, t = True, probe
        if t.startswith("|") and t.endswith("|"):
            absv, t = True, t[1:-1]
        g = self.src32(t)

        def get(w):
            x = g(w)
            if isinstance(x, int):
                x = np.full(64, x, U32)
            f = x.view(F32)
            if absv: f = np.abs(f)
            if neg: f = -f
            return f
        return get

    def srcu(self, t):
        g = self.src32(t)

        def get(w):
            x = g(w)
            return np.full(64, x, U32) if isinstance(x, int) else x
        return get

    def src64s(self, t):
        """Scalar 64-bit source -> python int."""
        r = self._reg(t)
        if r and r[0] == "s":
            n = r[1]
            return lambda w, n=n: w.S[n] | (w.S[n + 1] << 32)
        if t == "vcc": return lambda w: w.vcc
        if t == "exec": return lambda w: w.exec
        imm = self._imm32(t)
        if imm is not None:
            v = imm if imm < 0x80000000 or not t.startswith("-") else imm | 0xffffffff00000000
            if t.startswith("-"):
                v = int(t, 0) & FULL
            return lambda w, v=v: v
        raise Unsupported("64-bit scalar operand " + t)

    def dst64s(self, t):
        r = self._reg(t)
        if r and r[0] == "s":
            n = r[1]

            def put(w, v, n=n):
                w.S[n] = v & 0xffffffff; w.S[n + 1] = (v >> 32) & 0xffffffff
            return put
        if t == "vcc":
            def put(w, v): w.vcc = v & FULL
            return put
        if t == "exec":
            return lambda w, v: w.set_exec(v)
        raise Unsupported("64-bit scalar destination " + t)

    def dst32s(self, t):
        r = self._reg(t)
        if r and r[0] == "s":
            n = r[1]

            def put(w, v, n=n): w.S[n] = v & 0xffffffff
            return put
        if t == "vcc_lo":
            def put(w, v): w.vcc = (w.vcc & ~0xffffffff) | (v & 0xffffffff)
            return put
        if t == "vcc_hi":
            def put(w, v): w.vcc = (w.vcc & 0xffffffff) | ((v & 0xffffffff) << 32)
            return put
        if t == "m0":
            def put(w, v): w.m0 = v & 0xffffffff
            return put
        if t == "exec_lo":
            return lambda w, v: w.set_exec((w.exec & ~0xffffffff) | (v & 0xffffffff))
        if t == "exec_hi":
            return lambda w, v: w.set_exec((w.exec & 0xffffffff) | ((v & 0xffffffff) << 32))
        raise Unsupported("scalar destination " + t)

    @staticmethod
    def vwrite(w, n, val):
        if not isinstance(val, np.ndarray):
            val = np.full(64, val, U32)
        elif val.dtype != U32:
            raise TypeError("vector results are written as uint32 bit patterns, not %s" % val.dtype)
        if w.full:
            w.V[n][:] = val
        else:
            np.copyto(w.V[n], val, where=w.em)

    def lanemask_dst(self, t):
        return self.dst64s(t)

    # ---- compile one instruction into a closure f(wave) -> next pc or None ----
    def _compile(self, idx, ins):
        op, a, mods = ins.op, ins.args, ins.mods
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        sdwa = op.endswith("_sdwa")
        fn = getattr(self, "op_" + base, None)
        four = fn is not None and fn.__code__.co_argcount == 5
        if fn is None:
            fn = self._generic(base)
            four = True
        if fn is None:
            def bad(w, ins=ins):
                raise Unsupported("%s (assembly line %d)" % (ins.text, ins.line))
            return bad
        try:
            run = fn(a, mods, sdwa, ins) if four else fn(a, mods)
            if sdwa and mods.get("dst_sel", "DWORD") != "DWORD":
                run = self._sdwa_destination(run, a, mods)
            if sdwa and not four and any(mods.get("src%d_sel" % i, "DWORD") != "DWORD" for i in range(2)):
                raise Unsupported("sdwa source selector on this instruction")
            return run
        except Unsupported as e:
            msg = "%s: %s (assembly line %d)" % (ins.text, e, ins.line)

            def bad(w, msg=msg):
                raise Unsupported(msg)
            return bad

    # ---- tables of simple operations ----
    def _generic(self, base):
        # float VOP2/VOP3 with two or three sources
        f2 = {
            "v_add_f32": lambda x, y: x + y, "v_sub_f32": lambda x, y: x - y, "v_subrev_f32": lambda x, y: y - x,
            "v_mul_f32": lambda x, y: x * y,
            "v_max_f32": lambda x, y: np.fmax(x, y), "v_min_f32": lambda x, y: np.fmin(x, y),
        }
        if base in f2:
            return lambda a, mods, sdwa, ins, f=f2[base]: self._valu_f(a, mods, f, 2)
        f3 = {"v_min3_f32": lambda x, y, z: np.fmin(np.fmin(x, y), z), "v_max3_f32": lambda x, y, z: np.fmax(np.fmax(x, y), z),
              # (the median of three; with a NaN among them the hardware returns min3, which the fmin chain gives as well)
              "v_med3_f32": lambda x, y, z: np.where(np.isnan(x) | np.isnan(y) | np.isnan(z), np.fmin(np.fmin(x, y), z),
                                                     np.fmax(np.fmin(x, y), np.fmin(np.fmax(x, y), z)))}
        if base in f3:
            return lambda a, mods, sdwa, ins, f=f3[base]: self._valu_f(a, mods, f, 3)
        f1 = {"v_floor_f32": np.floor, "v_rndne_f32": np.rint, "v_trunc_f32": np.trunc}
        if base in f1:
            return lambda a, mods, sdwa, ins, f=f1[base]: self._valu_f(a, mods, f, 1)
        u2 = {
            "v_add_u32": lambda x, y: x + y, "v_sub_u32": lambda x, y: x - y, "v_subrev_u32": lambda x, y: y - x,
            "v_and_b32": lambda x, y: x & y, "v_or_b32": lambda x, y: x | y, "v_xor_b32": lambda x, y: x ^ y,
            "v_lshlrev_b32": lambda x, y: y << (x & U32(31)), "v_lshrrev_b32": lambda x, y: y >> (x & U32(31)),
            "v_ashrrev_i32": lambda x, y: (y.view(I32) >> (x & U32(31)).view(I32)).view(U32),
            "v_mul_lo_u32": lambda x, y: (x.astype(U64) * y.astype(U64)).astype(U32),
            "v_mul_hi_u32": lambda x, y: ((x.astype(U64) * y.astype(U64)) >> U64(32)).astype(U32),
            "v_mul_u32_u24": lambda x, y: ((x & U32(0xffffff)).astype(U64) * (y & U32(0xffffff)).astype(U64)).astype(U32),
            "v_min_u32": np.minimum, "v_max_u32": np.maximum,
            "v_mul_lo_u16": lambda x, y: (x * y) & U32(0xffff), "v_sub_u16": lambda x, y: (x - y) & U32(0xffff),
            "v_add_u16": lambda x, y: (x + y) & U32(0xffff),
            "v_min_i32": lambda x, y: np.minimum(x.view(I32), y.view(I32)).view(U32),
            "v_max_i32": lambda x, y: np.maximum(x.view(I32), y.view(I32)).view(U32),
        }
        if base in u2:
            return lambda a, mods, sdwa, ins, f=u2[base]: self._valu_u(a, mods, f, 2, sdwa)
        u3 = {
            "v_add3_u32": lambda x, y, z: x + y + z,
            "v_min3_i32": lambda x, y, z: np.minimum(np.minimum(x.view(I32), y.view(I32)), z.view(I32)).view(U32),
            "v_max3_i32": lambda x, y, z: np.maximum(np.maximum(x.view(I32), y.view(I32)), z.view(I32)).view(U32),
            "v_bfi_b32": lambda x, y, z: (x & y) | (~x & z),
            "v_lshl_add_u32": lambda x, y, z: (x << (y & U32(31))) + z,
            "v_add_lshl_u32": lambda x, y, z: (x + y) << (z & U32(31)),
            "v_lshl_or_b32": lambda x, y, z: (x << (y & U32(31))) | z,
            "v_and_or_b32": lambda x, y, z: (x & y) | z,
            "v_or3_b32": lambda x, y, z: x | y | z,
            "v_xad_u32": lambda x, y, z: (x ^ y) + z,
            "v_mad_u32_u24": lambda x, y, z: ((x & U32(0xffffff)).astype(U64) * (y & U32(0xffffff)).astype(U64)).astype(U32) + z,
            "v_bfe_u32": lambda x, y, z: (x >> (y & U32(31))) & ((U64(1) << (z & U32(31)).astype(U64)) - U64(1)).astype(U32),
            "v_alignbit_b32": lambda x, y, z: (((x.astype(U64) << U64(32)) | y.astype(U64)) >> (z & U32(31)).astype(U64)).astype(U32),
        }
        if base in u3:
            return lambda a, mods, sdwa, ins, f=u3[base]: self._valu_u(a, mods, f, 3, sdwa)
        cmpf = {"lt": lambda x, y: x < y, "le": lambda x, y: x <= y, "gt": lambda x, y: x > y, "ge": lambda x, y: x >= y,
                "eq": lambda x, y: x == y, "lg": lambda x, y: (x < y) | (x > y),
                "neq": lambda x, y: ~(x == y), "nlt": lambda x, y: ~(x < y), "nle": lambda x, y: ~(x <= y),
                "ngt": lambda x, y: ~(x > y), "nge": lambda x, y: ~(x >= y), "nlg": lambda x, y: ~((x < y) | (x > y)),
                "o": lambda x, y: ~(np.isnan(x) | np.isnan(y)), "u": lambda x, y: np.isnan(x) | np.isnan(y)}
        m = re.fullmatch(r"v_cmp_(\w+)_f32", base)
        if m and m.group(1) in cmpf:
            return lambda a, mods, sdwa, ins, f=cmpf[m.group(1)]: self._vcmp(a, f, "f")
        cmpi = {"lt": lambda x, y: x < y, "le": lambda x, y: x <= y, "gt": lambda x, y: x > y, "ge": lambda x, y: x >= y,
                "eq": lambda x, y: x == y, "ne": lambda x, y: x != y}
        m = re.fullmatch(r"v_cmp_(\w+)_([iu])(32|64)", base)
        if m and m.group(1) in cmpi:
            return lambda a, mods, sdwa, ins, f=cmpi[m.group(1)], k=m.group(2) + m.group(3): self._vcmp(a, f, k)
        s2 = {
            "s_and_b32": lambda x, y: x & y, "s_or_b32": lambda x, y: x | y, "s_xor_b32": lambda x, y: x ^ y,
            "s_andn2_b32": lambda x, y: x & ~y & 0xffffffff, "s_orn2_b32": lambda x, y: (x | ~y) & 0xffffffff,
            "s_nor_b32": lambda x, y: ~(x | y) & 0xffffffff, "s_nand_b32": lambda x, y: ~(x & y) & 0xffffffff,
        }
        if base in s2:
            return lambda a, mods, sdwa, ins, f=s2[base]: self._salu2(a, f, 32)
        s2q = {
            "s_and_b64": lambda x, y: x & y, "s_or_b64": lambda x, y: x | y, "s_xor_b64": lambda x, y: x ^ y,
            "s_andn2_b64": lambda x, y: x & ~y & FULL, "s_orn2_b64": lambda x, y: (x | ~y) & FULL,
            "s_nor_b64": lambda x, y: ~(x | y) & FULL, "s_nand_b64": lambda x, y: ~(x & y) & FULL,
            "s_xnor_b64": lambda x, y: ~(x ^ y) & FULL,
        }
        if base in s2q:
            return lambda a, mods, sdwa, ins, f=s2q[base]: self._salu2(a, f, 64)
        scmp = {"eq": lambda x, y: x == y, "lg": lambda x, y: x != y, "gt": lambda x, y: x > y, "ge": lambda x, y: x >= y,
                "lt": lambda x, y: x < y, "le": lambda x, y: x <= y}
        m = re.fullmatch(r"s_cmp_(\w+)_([iu])(32|64)", base)
        if m and m.group(1) in scmp:
            return lambda a, mods, sdwa, ins, f=scmp[m.group(1)], sg=m.group(2) == "i", q=m.group(3) == "64": self._scmp(a, f, sg, q)
        return None

    # ---- VALU helpers ----
    def _valu_f(self, a, mods, f, n):
        d = self._reg(a[0])[1]
        srcs = [self.srcf(t) for t in a[1:1 + n]]
        clamp = "clamp" in mods
        omod = float(mods["mul"]) if "mul" in mods else (1.0 / float(mods["div"]) if "div" in mods else None)

        def run(w):
            xs = [g(w) for g in srcs]
            if w.flush:
                xs = [_flush(x) for x in xs]
            with np.errstate(all="ignore"):
                r = f(*xs).astype(F32, copy=False)
                if omod: r = r * F32(omod)
                if clamp: r = np.where(np.isnan(r), F32(0), np.clip(r, F32(0), F32(1)))
            if w.flush:
                r = _flush(r)
            self.vwrite(w, d, r.view(U32))
        return run

    def _sdwa_destination(self, run, a, mods):
        """The instruction's result goes to one byte or half of the destination; the rest is zeroed or kept."""
        d = self._reg(a[0])[1]
        sel, unused = mods["dst_sel"], mods.get("dst_unused", "UNUSED_PAD")
        m = re.fullmatch(r"(BYTE|WORD)_(\d)", sel)
        if not m or unused not in ("UNUSED_PAD", "UNUSED_PRESERVE"):
            raise Unsupported("sdwa destination %s %s" % (sel, unused))
        bits_ = 8 if m.group(1) == "BYTE" else 16
        shift = U32(bits_ * int(m.group(2)))
        field = U32(((1 << bits_) - 1))

        def wrapped(w):
            old = w.V[d].copy()
            run(w)
            placed = (w.V[d] & field) << shift
            if unused == "UNUSED_PRESERVE":
                placed |= old & ~(field << shift)
            w.V[d][:] = np.where(w.em, placed, old)
        return wrapped

    def _sdwa_sel(self, x, sel):
        if sel in (None, "DWORD"): return x
        m = re.fullmatch(r"BYTE_(\d)", sel)
        if m: return (x >> U32(8 * int(m.group(1)))) & U32(0xff)
        m = re.fullmatch(r"WORD_(\d)", sel)
        if m: return (x >> U32(16 * int(m.group(1)))) & U32(0xffff)
        raise Unsupported("sdwa selector " + sel)

    def _valu_u(self, a, mods, f, n, sdwa=False):
        d = self._reg(a[0])[1]
        srcs = [self.srcu(t) for t in a[1:1 + n]]
        sels = [mods.get("src%d_sel" % i) for i in range(n)] if sdwa else None

        def run(w):
            xs = [g(w) for g in srcs]
            if sels:
                xs = [self._sdwa_sel(x, s) for x, s in zip(xs, sels)]
            with np.errstate(over="ignore"):
                r = f(*xs)
            self.vwrite(w, d, r.astype(U32, copy=False))
        return run

    def _vcmp(self, a, f, kind):
        put = self.lanemask_dst(a[0])
        if kind == "f":
            ga, gb = self.srcf(a[1]), self.srcf(a[2])
        elif kind in ("u32", "i32"):
            ga, gb = self.srcu(a[1]), self.srcu(a[2])
        else:
            ga, gb = self.src64v(a[1]), self.src64v(a[2])

        def run(w):
            x, y = ga(w), gb(w)
            if kind == "f" and w.flush:
                x, y = _flush(x), _flush(y)
            if kind == "i32":
                x, y = x.view(I32), y.view(I32)
            elif kind == "i64":
                x, y = x.view(np.int64), y.view(np.int64)
            with np.errstate(invalid="ignore"):
                b = f(x, y)
            put(w, bools_to_mask(b) & w.exec)
        return run

    def src64v(self, t):
        """64-bit vector source: VGPR pair, SGPR pair or constant -> uint64 array."""
        r = self._reg(t)
        if r and r[0] == "v":
            n = r[1]
            return lambda w, n=n: w.V[n].astype(U64) | (w.V[n + 1].astype(U64) << U64(32))
        g = self.src64s(t)
        return lambda w: np.full(64, g(w), U64)

    def vwrite64(self, w, n, val):
        self.vwrite(w, n, (val & U64(0xffffffff)).astype(U32))
        self.vwrite(w, n + 1, (val >> U64(32)).astype(U32))

    # ---- SALU helpers ----
    def _salu2(self, a, f, bits):
        if bits == 32:
            put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])
        else:
            put, ga, gb = self.dst64s(a[0]), self.src64s(a[1]), self.src64s(a[2])

        def run(w):
            r = f(ga(w), gb(w))
            put(w, r)
            w.scc = 1 if r != 0 else 0
        return run

    def _scmp(self, a, f, signed, q):
        ga, gb = (self.src64s(a[0]), self.src64s(a[1])) if q else (self.src32(a[0]), self.src32(a[1]))

        def run(w):
            x, y = ga(w), gb(w)
            if signed:
                x = x - (1 << 32) if x & 0x80000000 else x
                y = y - (1 << 32) if y & 0x80000000 else y
            w.scc = 1 if f(x, y) else 0
        return run

    # ---- SOP ----
    def op_s_mov_b32(self, a, mods):
        put, g = self.dst32s(a[0]), self.src32(a[1])
        return lambda w: put(w, g(w))

    def op_s_movk_i32(self, a, mods):
        put = self.dst32s(a[0]); v = int(a[1], 0) & 0xffff
        v = v | 0xffff0000 if v & 0x8000 else v
        return lambda w: put(w, v)

    def op_s_mov_b64(self, a, mods):
        put, g = self.dst64s(a[0]), self.src64s(a[1])
        return lambda w: put(w, g(w))

    def op_s_brev_b32(self, a, mods):
        put, g = self.dst32s(a[0]), self.src32(a[1])
        return lambda w: put(w, int("{:032b}".format(g(w))[::-1], 2))

    def op_s_not_b64(self, a, mods):
        put, g = self.dst64s(a[0]), self.src64s(a[1])

        def run(w):
            r = ~g(w) & FULL; put(w, r); w.scc = 1 if r else 0
        return run

    def op_s_bcnt1_i32_b64(self, a, mods):
        put, g = self.dst32s(a[0]), self.src64s(a[1])

        def run(w):
            r = bin(g(w)).count("1"); put(w, r); w.scc = 1 if r else 0
        return run

    def op_s_bcnt1_i32_b32(self, a, mods):
        put, g = self.dst32s(a[0]), self.src32(a[1])

        def run(w):
            r = bin(g(w)).count("1"); put(w, r); w.scc = 1 if r else 0
        return run

    def op_s_ff1_i32_b64(self, a, mods):
        put, g = self.dst32s(a[0]), self.src64s(a[1])

        def run(w):
            x = g(w); put(w, (x & -x).bit_length() - 1 if x else 0xffffffff)
        return run

    def _saveexec(self, a, f):
        put, g = self.dst64s(a[0]), self.src64s(a[1])

        def run(w):
            old = w.exec
            s = g(w)
            put(w, old)
            w.set_exec(f(s, old))
            w.scc = 1 if w.exec else 0
        return run

    def op_s_and_saveexec_b64(self, a, mods): return self._saveexec(a, lambda s, e: s & e)
    def op_s_or_saveexec_b64(self, a, mods): return self._saveexec(a, lambda s, e: s | e)
    def op_s_xor_saveexec_b64(self, a, mods): return self._saveexec(a, lambda s, e: s ^ e)
    def op_s_andn2_saveexec_b64(self, a, mods): return self._saveexec(a, lambda s, e: s & ~e & FULL)
    def op_s_andn1_saveexec_b64(self, a, mods): return self._saveexec(a, lambda s, e: ~s & e & FULL)
    def op_s_orn2_saveexec_b64(self, a, mods): return self._saveexec(a, lambda s, e: (s | ~e) & FULL)

    def op_s_cselect_b32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])
        return lambda w: put(w, ga(w) if w.scc else gb(w))

    def op_s_cselect_b64(self, a, mods):
        put, ga, gb = self.dst64s(a[0]), self.src64s(a[1]), self.src64s(a[2])
        return lambda w: put(w, ga(w) if w.scc else gb(w))

    def op_s_add_u32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])

        def run(w):
            r = ga(w) + gb(w); put(w, r); w.scc = 1 if r > 0xffffffff else 0
        return run

    def op_s_addc_u32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])

        def run(w):
            r = ga(w) + gb(w) + w.scc; put(w, r); w.scc = 1 if r > 0xffffffff else 0
        return run

    def op_s_subb_u32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])

        def run(w):
            x, y = ga(w), gb(w) + w.scc; put(w, x - y); w.scc = 1 if y > x else 0
        return run

    def op_s_memtime(self, a, mods):
        put = self.dst64s(a[0])
        return lambda w: put(w, self.executed * 4)      # a clock that only moves forward; nothing here is timed

    def op_s_sub_u32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])

        def run(w):
            x, y = ga(w), gb(w); put(w, x - y); w.scc = 1 if y > x else 0
        return run

    def op_s_add_i32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])

        def run(w):
            x, y = ga(w), gb(w)
            sx = x - (1 << 32) if x & 0x80000000 else x
            sy = y - (1 << 32) if y & 0x80000000 else y
            r = sx + sy; put(w, r); w.scc = 1 if not (-(1 << 31) <= r < (1 << 31)) else 0
        return run

    def op_s_sub_i32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])

        def run(w):
            x, y = ga(w), gb(w)
            sx = x - (1 << 32) if x & 0x80000000 else x
            sy = y - (1 << 32) if y & 0x80000000 else y
            r = sx - sy; put(w, r); w.scc = 1 if not (-(1 << 31) <= r < (1 << 31)) else 0
        return run

    def op_s_mul_i32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])
        return lambda w: put(w, ga(w) * gb(w))

    def op_s_mul_hi_i32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])

        def run(w):
            x, y = ga(w), gb(w)
            sx = x - (1 << 32) if x & 0x80000000 else x
            sy = y - (1 << 32) if y & 0x80000000 else y
            put(w, (sx * sy) >> 32)
        return run

    def op_s_mul_hi_u32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])
        return lambda w: put(w, (ga(w) * gb(w)) >> 32)

    def _sshift(self, a, bits, left):
        put = self.dst32s(a[0]) if bits == 32 else self.dst64s(a[0])
        ga = self.src32(a[1]) if bits == 32 else self.src64s(a[1])
        gb = self.src32(a[2])

        def run(w):
            x, n = ga(w), gb(w) & (bits - 1)
            r = (x << n) & ((1 << bits) - 1) if left else x >> n
            put(w, r); w.scc = 1 if r else 0
        return run

    def op_s_lshl_b32(self, a, mods): return self._sshift(a, 32, True)
    def op_s_lshr_b32(self, a, mods): return self._sshift(a, 32, False)
    def op_s_lshl_b64(self, a, mods): return self._sshift(a, 64, True)
    def op_s_lshr_b64(self, a, mods): return self._sshift(a, 64, False)

    def op_s_ashr_i32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])

        def run(w):
            x = ga(w); sx = x - (1 << 32) if x & 0x80000000 else x
            r = (sx >> (gb(w) & 31)) & 0xffffffff; put(w, r); w.scc = 1 if r else 0
        return run

    def _sminmax_i32(self, a, pick_max):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])

        def run(w):
            x, y = ga(w), gb(w)
            sx = x - (1 << 32) if x & 0x80000000 else x
            sy = y - (1 << 32) if y & 0x80000000 else y
            first = sx > sy if pick_max else sx < sy
            put(w, x if first else y); w.scc = 1 if first else 0
        return run

    def op_s_max_i32(self, a, mods): return self._sminmax_i32(a, True)
    def op_s_min_i32(self, a, mods): return self._sminmax_i32(a, False)

    def op_s_min_u32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])

        def run(w):
            x, y = ga(w), gb(w); put(w, min(x, y)); w.scc = 1 if x < y else 0
        return run

    def op_s_max_u32(self, a, mods):
        put, ga, gb = self.dst32s(a[0]), self.src32(a[1]), self.src32(a[2])

        def run(w):
            x, y = ga(w), gb(w); put(w, max(x, y)); w.scc = 1 if x > y else 0
        return run

    def op_s_getpc_b64(self, a, mods):
        put = self.dst64s(a[0])
        return lambda w: put(w, 0)          # only used as the base of symbol@rel32 pairs, which carry the whole address here

    def op_s_setreg_imm32_b32(self, a, mods):
        m = re.fullmatch(r"hwreg\(HW_REG_MODE,\s*(\d+),\s*(\d+)\)", a[0])
        if not m:
            raise Unsupported("s_setreg of " + a[0])
        off, size, val = int(m.group(1)), int(m.group(2)), int(a[1], 0)
        if off <= 4 and off + size >= 6:          # covers bits 4-5: single-precision denormal mode
            mode = (val >> (4 - off)) & 3
            return lambda w: setattr(w, "flush", mode == 0)
        return lambda w: None

    def op_s_waitcnt(self, a, mods): return lambda w: None
    def op_s_nop(self, a, mods): return lambda w: None
    def op_s_sleep(self, a, mods): return lambda w: None

    def op_s_barrier(self, a, mods):
        def run(w): w.at_barrier = True
        return run

    def op_s_endpgm(self, a, mods):
        def run(w): w.done = True
        return run

    def _branch(self, a, cond):
        target = self.k.labels[a[0]]

        def run(w):
            if cond(w): return target
        return run

    def op_s_branch(self, a, mods): return self._branch(a, lambda w: True)
    def op_s_cbranch_scc0(self, a, mods): return self._branch(a, lambda w: w.scc == 0)
    def op_s_cbranch_scc1(self, a, mods): return self._branch(a, lambda w: w.scc == 1)
    def op_s_cbranch_vccz(self, a, mods): return self._branch(a, lambda w: w.vcc == 0)
    def op_s_cbranch_vccnz(self, a, mods): return self._branch(a, lambda w: w.vcc != 0)
    def op_s_cbranch_execz(self, a, mods): return self._branch(a, lambda w: w.exec == 0)
    def op_s_cbranch_execnz(self, a, mods): return self._branch(a, lambda w: w.exec != 0)

    # ---- SMEM ----
    def _sload(self, a, n):
        r = self._reg(a[0]); first = r[1]
        base = self.src64s(a[1])
        off = self.src32(a[2]) if len(a) > 2 else (lambda w: 0)
        lib = self.lib

        def run(w):
            addr = base(w) + off(w)
            if not lib.emu_known(addr, 4 * n):
                raise StrayAccess("scalar load of %d bytes at 0x%x" % (4 * n, addr))
            vals = struct.unpack("<%dI" % n, ctypes.string_at(addr, 4 * n))
            for i, v in enumerate(vals):
                w.S[first + i] = v
        return run

    def op_s_load_dword(self, a, mods): return self._sload(a, 1)
    def op_s_load_dwordx2(self, a, mods): return self._sload(a, 2)
    def op_s_load_dwordx4(self, a, mods): return self._sload(a, 4)
    def op_s_load_dwordx8(self, a, mods): return self._sload(a, 8)
    def op_s_load_dwordx16(self, a, mods): return self._sload(a, 16)

    # ---- VALU ----
    def op_v_mov_b32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcu(a[1])
        return lambda w: self.vwrite(w, d, g(w))

    def op_v_mov_b64(self, a, mods):
        d, g = self._reg(a[0])[1], self.src64v(a[1])
        return lambda w: self.vwrite64(w, d, g(w))

    def op_v_not_b32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcu(a[1])
        return lambda w: self.vwrite(w, d, ~g(w))

    def op_v_bfrev_b32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcu(a[1])

        def run(w):
            x = g(w).copy()
            x = ((x >> U32(1)) & U32(0x55555555)) | ((x & U32(0x55555555)) << U32(1))
            x = ((x >> U32(2)) & U32(0x33333333)) | ((x & U32(0x33333333)) << U32(2))
            x = ((x >> U32(4)) & U32(0x0f0f0f0f)) | ((x & U32(0x0f0f0f0f)) << U32(4))
            x = ((x >> U32(8)) & U32(0x00ff00ff)) | ((x & U32(0x00ff00ff)) << U32(8))
            x = (x >> U32(16)) | (x << U32(16))
            self.vwrite(w, d, x)
        return run

    def _fma(self, w, x, y, z):
        if w.flush:
            x, y, z = _flush(x), _flush(y), _flush(z)
        x, y, z = (np.ascontiguousarray(v, dtype=F32) for v in (x, y, z))
        out = np.empty(64, F32)
        self.lib.emu_vfma(x.ctypes.data, y.ctypes.data, z.ctypes.data, out.ctypes.data, 64)
        return _flush(out) if w.flush else out

    def op_v_fma_f32(self, a, mods):
        d = self._reg(a[0])[1]
        gs = [self.srcf(t) for t in a[1:4]]
        if "clamp" in mods or "mul" in mods or "div" in mods:
            raise Unsupported("output modifier on v_fma_f32")
        return lambda w: self.vwrite(w, d, self._fma(w, gs[0](w), gs[1](w), gs[2](w)).view(U32))

    def op_v_fmac_f32(self, a, mods):
        d = self._reg(a[0])[1]
        ga, gb = self.srcf(a[1]), self.srcf(a[2])
        return lambda w: self.vwrite(w, d, self._fma(w, ga(w), gb(w), w.V[d].view(F32)).view(U32))

    def op_v_rcp_f32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcf(a[1])

        def run(w):
            x = _flush(g(w))
            with np.errstate(all="ignore"):
                r = (F32(1) / x).astype(F32)
            self.vwrite(w, d, _flush(r).view(U32))
        return run

    def op_v_fmamk_f32(self, a, mods):          # D = S0 * K + S1
        d = self._reg(a[0])[1]
        g0, gk, g1 = self.srcf(a[1]), self.srcf(a[2]), self.srcf(a[3])
        return lambda w: self.vwrite(w, d, self._fma(w, g0(w), gk(w), g1(w)).view(U32))

    def op_v_fmaak_f32(self, a, mods):          # D = S0 * S1 + K
        d = self._reg(a[0])[1]
        g0, g1, gk = self.srcf(a[1]), self.srcf(a[2]), self.srcf(a[3])
        return lambda w: self.vwrite(w, d, self._fma(w, g0(w), g1(w), gk(w)).view(U32))

    def op_v_exp_f32(self, a, mods):            # 2^x; the hardware's result is within one ulp of this one
        d, g = self._reg(a[0])[1], self.srcf(a[1])

        def run(w):
            with np.errstate(all="ignore"):
                r = np.exp2(_flush(g(w)).astype(np.float64)).astype(F32)
            self.vwrite(w, d, _flush(r).view(U32))
        return run

    def op_v_log_f32(self, a, mods):            # log2 x; likewise
        d, g = self._reg(a[0])[1], self.srcf(a[1])

        def run(w):
            with np.errstate(all="ignore"):
                r = np.log2(_flush(g(w)).astype(np.float64)).astype(F32)
            self.vwrite(w, d, _flush(r).view(U32))
        return run

    def op_v_ldexp_f32(self, a, mods):
        d, g0, g1 = self._reg(a[0])[1], self.srcf(a[1]), self.srcu(a[2])

        def run(w):
            x = g0(w)
            if w.flush: x = _flush(x)
            with np.errstate(all="ignore"):
                r = np.ldexp(x.astype(np.float64), np.clip(g1(w).view(I32), -400, 400)).astype(F32)
            self.vwrite(w, d, (_flush(r) if w.flush else r).view(U32))
        return run

    def op_v_frexp_mant_f32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcf(a[1])

        def run(w):
            x = g(w)
            if w.flush: x = _flush(x)
            with np.errstate(all="ignore"):
                m, _ = np.frexp(x)
            self.vwrite(w, d, np.where(np.isfinite(x), m, x).astype(F32).view(U32))      # infinities and NaNs pass through
        return run

    def op_v_frexp_exp_i32_f32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcf(a[1])

        def run(w):
            x = g(w)
            if w.flush: x = _flush(x)
            with np.errstate(all="ignore"):
                _, e = np.frexp(x)
            self.vwrite(w, d, np.where(np.isfinite(x) & (x != 0), e, 0).astype(I32).view(U32))
        return run

    def op_v_cvt_f64_f32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcf(a[1])

        def run(w):
            x = g(w)
            if w.flush: x = _flush(x)
            with np.errstate(all="ignore"):
                self.vwrite64(w, d, x.astype(np.float64).view(U64))
        return run

    def op_v_frexp_exp_i32_f64(self, a, mods):
        d, g = self._reg(a[0])[1], self.src64v(a[1])

        def run(w):
            x = g(w).view(np.float64)
            with np.errstate(all="ignore"):
                _, e = np.frexp(x)
            self.vwrite(w, d, np.where(np.isfinite(x) & (x != 0), e, 0).astype(I32).view(U32))
        return run

    def op_v_mad_i64_i32(self, a, mods):
        d = self._reg(a[0])[1]
        put = self.lanemask_dst(a[1])
        g0, g1, g2 = self.srcu(a[2]), self.srcu(a[3]), self.src64v(a[4])

        def run(w):
            p = g0(w).view(I32).astype(np.int64) * g1(w).view(I32).astype(np.int64)
            with np.errstate(over="ignore"):
                r = p + g2(w).view(np.int64)
            put(w, 0)          # (the carry-out is not used by these kernels; a use would name the register, which reads as zero)
            self.vwrite64(w, d, r.view(U64))
        return run

    def op_v_bitop3_b16(self, a, mods):
        d = self._reg(a[0])[1]
        gs = [self.srcu(t) for t in a[1:4]]
        table = int(mods["bitop3"], 0)

        def run(w):
            x, y, z = (g(w) for g in gs)
            r = np.zeros(64, U32)
            for i in range(8):
                if table >> i & 1:
                    r |= (x if i & 4 else ~x) & (y if i & 2 else ~y) & (z if i & 1 else ~z)
            self.vwrite(w, d, (w.V[d] & U32(0xffff0000)) | (r & U32(0xffff)))      # the low half is written, the high half kept
        return run

    def op_v_rcp_iflag_f32(self, a, mods): return self.op_v_rcp_f32(a, mods)

    def op_v_rsq_f32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcf(a[1])

        def run(w):
            with np.errstate(all="ignore"):
                x = _flush(g(w)).astype(np.float64)
                r = (1.0 / np.sqrt(x)).astype(F32)
            self.vwrite(w, d, _flush(r).view(U32))
        return run

    def op_v_sqrt_f32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcf(a[1])

        def run(w):
            x = _flush(g(w))
            with np.errstate(all="ignore"):
                r = np.sqrt(x).astype(F32)
            self.vwrite(w, d, _flush(r).view(U32))
        return run

    def op_v_cvt_f32_u32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcu(a[1])
        return lambda w: self.vwrite(w, d, g(w).astype(F32).view(U32))

    def op_v_cvt_f32_i32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcu(a[1])
        return lambda w: self.vwrite(w, d, g(w).view(I32).astype(F32).view(U32))

    def op_v_cvt_u32_f32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcf(a[1])

        def run(w):
            with np.errstate(all="ignore"):
                x = g(w).astype(np.float64)
                x = np.where(np.isnan(x), 0.0, np.clip(np.trunc(x), 0.0, 4294967295.0))
            self.vwrite(w, d, x.astype(U64).astype(U32))
        return run

    def op_v_cvt_i32_f32(self, a, mods):
        d, g = self._reg(a[0])[1], self.srcf(a[1])

        def run(w):
            with np.errstate(all="ignore"):
                x = g(w).astype(np.float64)
                x = np.where(np.isnan(x), 0.0, np.clip(np.trunc(x), -2147483648.0, 2147483647.0))
            self.vwrite(w, d, x.astype(np.int64).astype(I32).view(U32))
        return run

    def _cvt_ubyte(self, a, n):
        d, g = self._reg(a[0])[1], self.srcu(a[1])
        return lambda w: self.vwrite(w, d, ((g(w) >> U32(8 * n)) & U32(0xff)).astype(F32).view(U32))

    def op_v_cvt_f32_ubyte0(self, a, mods): return self._cvt_ubyte(a, 0)
    def op_v_cvt_f32_ubyte1(self, a, mods): return self._cvt_ubyte(a, 1)
    def op_v_cvt_f32_ubyte2(self, a, mods): return self._cvt_ubyte(a, 2)
    def op_v_cvt_f32_ubyte3(self, a, mods): return self._cvt_ubyte(a, 3)

    def op_v_cndmask_b32(self, a, mods):
        d = self._reg(a[0])[1]
        # sources may carry float modifiers in the e64 form
        ga = self.srcf(a[1]) if a[1][0] in "-|" else self.srcu(a[1])
        gb = self.srcf(a[2]) if a[2][0] in "-|" else self.srcu(a[2])
        gm = self.src64s(a[3]) if len(a) > 3 else (lambda w: w.vcc)

        def run(w):
            sel = mask_to_bools(gm(w))
            x, y = ga(w), gb(w)
            self.vwrite(w, d, np.where(sel, y.view(U32), x.view(U32)))
        return run

    def op_v_cmp_class_f32(self, a, mods):
        put, gx, gm = self.lanemask_dst(a[0]), self.srcu(a[1]), self.srcu(a[2])

        def run(w):
            bits_, m = gx(w), gm(w)
            f = bits_.view(F32)
            sign = (bits_ >> U32(31)) != 0
            exp = (bits_ >> U32(23)) & U32(0xff)
            man = bits_ & U32(0x7fffff)
            nan = (exp == 255) & (man != 0)
            snan = nan & ((man & U32(0x400000)) == 0)
            inf = (exp == 255) & (man == 0)
            zero = (exp == 0) & (man == 0)
            den = (exp == 0) & (man != 0)
            norm = ~nan & ~inf & ~zero & ~den
            cls = [snan, nan & ~snan, inf & sign, norm & sign, den & sign, zero & sign, zero & ~sign, den & ~sign, norm & ~sign, inf & ~sign]
            hit = np.zeros(64, bool)
            for i, c in enumerate(cls):
                hit |= c & (((m >> U32(i)) & U32(1)) != 0)
            put(w, bools_to_mask(hit) & w.exec)
        return run

    def op_v_div_scale_f32(self, a, mods):
        d = self._reg(a[0])[1]
        put = self.lanemask_dst(a[1])
        g0, g1, g2 = self.srcf(a[2]), self.srcf(a[3]), self.srcf(a[4])

        def run(w):
            s0, s1, s2 = g0(w), g1(w), g2(w)
            # only the plain case is modelled: operands whose quotient needs no scaling pass through and the flag stays clear
            e1 = ((s1.view(U32) >> U32(23)) & U32(0xff)).astype(np.int64)
            e2 = ((s2.view(U32) >> U32(23)) & U32(0xff)).astype(np.int64)
            plain = (e1 > 0) & (e1 < 255) & (e2 > 23) & (e2 < 255) & (np.abs(e2 - e1) < 96) & (e1 < 253) & ((e2 - e1) > -125) & (e1 > 2)
            special = ~plain & w.em
            # zeros, infinities and NaNs are settled by v_div_fixup whatever is computed in between
            settled = (e1 == 0) | (e1 == 255) | (e2 == 0) | (e2 == 255)
            if (special & ~settled).any():
                raise Unsupported("v_div_scale_f32 on operands that need scaling")
            self.vwrite(w, d, s0.view(U32))
            put(w, 0)
        return run

    def op_v_div_fmas_f32(self, a, mods):
        d = self._reg(a[0])[1]
        gs = [self.srcf(t) for t in a[1:4]]

        def run(w):
            if w.vcc & w.exec:
                raise Unsupported("v_div_fmas_f32 with the scale flag set")
            # (the division sequence runs with denormals on: s_setreg around it)
            self.vwrite(w, d, self._fma(w, gs[0](w), gs[1](w), gs[2](w)).view(U32))
        return run

    def op_v_div_fixup_f32(self, a, mods):
        d = self._reg(a[0])[1]
        gq, gd, gn = self.srcf(a[1]), self.srcf(a[2]), self.srcf(a[3])

        def run(w):
            q, den, num = gq(w), gd(w), gn(w)
            if w.flush:
                den, num = _flush(den), _flush(num)
            sign = np.signbit(den) ^ np.signbit(num)
            with np.errstate(all="ignore"):
                r = np.where(sign, -np.abs(q), np.abs(q)).astype(F32)
                inf = np.where(sign, F32(-np.inf), F32(np.inf))
                zero = np.where(sign, F32(-0.0), F32(0.0))
                r = np.where(np.isinf(den) | (num == 0), zero, r)
                r = np.where((den == 0) | np.isinf(num), inf, r)
                r = np.where(((den == 0) & (num == 0)) | (np.isinf(den) & np.isinf(num)), F32(np.nan), r)
                r = np.where(np.isnan(den), den, r)
                r = np.where(np.isnan(num), num, r)
            r = r.astype(F32)
            if w.flush:
                r = _flush(r)
            self.vwrite(w, d, r.view(U32))
        return run

    def op_v_lshl_add_u64(self, a, mods):
        d = self._reg(a[0])[1]
        g0, g1, g2 = self.src64v(a[1]), self.srcu(a[2]), self.src64v(a[3])
        return lambda w: self.vwrite64(w, d, (g0(w) << (g1(w) & U32(63)).astype(U64)) + g2(w))

    def op_v_lshlrev_b64(self, a, mods):
        d = self._reg(a[0])[1]
        g0, g1 = self.srcu(a[1]), self.src64v(a[2])
        return lambda w: self.vwrite64(w, d, g1(w) << (g0(w) & U32(63)).astype(U64))

    def op_v_lshrrev_b64(self, a, mods):
        d = self._reg(a[0])[1]
        g0, g1 = self.srcu(a[1]), self.src64v(a[2])
        return lambda w: self.vwrite64(w, d, g1(w) >> (g0(w) & U32(63)).astype(U64))

    def op_v_mad_u64_u32(self, a, mods):
        d = self._reg(a[0])[1]
        put = self.lanemask_dst(a[1])
        g0, g1, g2 = self.srcu(a[2]), self.srcu(a[3]), self.src64v(a[4])

        def run(w):
            p = g0(w).astype(U64) * g1(w).astype(U64)
            with np.errstate(over="ignore"):
                r = p + g2(w)
            put(w, bools_to_mask(r < p) & w.exec)
            self.vwrite64(w, d, r)
        return run

    def _addc(self, a, sub, rev=False, carry_in=True):
        d = self._reg(a[0])[1]
        put = self.lanemask_dst(a[1])
        g0, g1 = self.srcu(a[2]), self.srcu(a[3])
        gc = self.src64s(a[4]) if len(a) > 4 else None

        def run(w):
            x, y = g0(w).astype(np.int64), g1(w).astype(np.int64)
            if rev: x, y = y, x
            c = mask_to_bools(gc(w)).astype(np.int64) if gc else 0
            r = x - y - c if sub else x + y + c
            put(w, bools_to_mask((r < 0) | (r > 0xffffffff)) & w.exec)
            self.vwrite(w, d, (r & 0xffffffff).astype(U32))
        return run

    def op_v_add_co_u32(self, a, mods): return self._addc(a, False)
    def op_v_addc_co_u32(self, a, mods): return self._addc(a, False)
    def op_v_sub_co_u32(self, a, mods): return self._addc(a, True)
    def op_v_subrev_co_u32(self, a, mods): return self._addc(a, True, rev=True)
    def op_v_subb_co_u32(self, a, mods): return self._addc(a, True)
    def op_v_subbrev_co_u32(self, a, mods): return self._addc(a, True, rev=True)

    def op_v_bitop3_b32(self, a, mods):
        d = self._reg(a[0])[1]
        gs = [self.srcu(t) for t in a[1:4]]
        table = int(mods["bitop3"], 0)

        def run(w):
            x, y, z = (g(w) for g in gs)
            r = np.zeros(64, U32)
            for i in range(8):
                if table >> i & 1:
                    r |= (x if i & 4 else ~x) & (y if i & 2 else ~y) & (z if i & 1 else ~z)
            self.vwrite(w, d, r)
        return run

    def op_v_mbcnt_lo_u32_b32(self, a, mods):
        d = self._reg(a[0])[1]
        g0, g1 = self.srcu(a[1]), self.srcu(a[2])

        def run(w):
            m = g0(w).astype(U64)
            below = np.where(LANES >= 32, U64(0xffffffff), (U64(1) << np.minimum(LANES, U64(31))) - U64(1))
            x = m & below
            cnt = np.array([bin(int(v)).count("1") for v in x], dtype=U32)
            self.vwrite(w, d, cnt + g1(w))
        return run

    def op_v_mbcnt_hi_u32_b32(self, a, mods):
        d = self._reg(a[0])[1]
        g0, g1 = self.srcu(a[1]), self.srcu(a[2])

        def run(w):
            m = g0(w).astype(U64)
            k = np.where(LANES > 32, LANES - U64(32), U64(0))
            below = (U64(1) << k) - U64(1)
            x = m & below
            cnt = np.array([bin(int(v)).count("1") for v in x], dtype=U32)
            self.vwrite(w, d, cnt + g1(w))
        return run

    def op_v_readfirstlane_b32(self, a, mods):
        put = self.dst32s(a[0]); n = self._reg(a[1])[1]

        def run(w):
            lane = (w.exec & -w.exec).bit_length() - 1 if w.exec else 0
            put(w, int(w.V[n][lane]))
        return run

    def op_v_readlane_b32(self, a, mods):
        put = self.dst32s(a[0]); n = self._reg(a[1])[1]; gl = self.src32(a[2])
        return lambda w: put(w, int(w.V[n][gl(w) & 63]))

    def op_v_writelane_b32(self, a, mods):
        n = self._reg(a[0])[1]; gv = self.src32(a[1]); gl = self.src32(a[2])

        def run(w): w.V[n][gl(w) & 63] = gv(w) & 0xffffffff
        return run

    def op_v_swap_b32(self, a, mods):
        x, y = self._reg(a[0])[1], self._reg(a[1])[1]

        def run(w):
            t = w.V[x].copy()
            self.vwrite(w, x, w.V[y].copy())
            self.vwrite(w, y, t)
        return run

    # ---- LDS ----
    def _lds_index(self, w, addr, extra, ndw):
        a = (addr.astype(np.int64) + extra) & 0xffffffff      # (32-bit address arithmetic: the compiler uses a "negative" base with an offset)
        ok = w.em & (a >= 0) & (a + 4 * ndw <= self.lds_bytes)
        self.lds_out_of_range += int((w.em & ~ok).sum())
        if self.lds_debug is not None and (w.em & ~ok).any():
            self.lds_debug.append((w.pc, a[w.em & ~ok][:4].tolist()))
        if (a[w.em] & 3).any():
            raise Unsupported("unaligned LDS access")
        return a >> 2, ok

    def _ds_read(self, a, mods, ndw):
        d = self._reg(a[0])[1]; ga = self.srcu(a[1]); off = int(mods.get("offset", "0"), 0)

        def run(w):
            idx, ok = self._lds_index(w, ga(w), off, ndw)
            safe = np.where(ok, idx, 0)
            for k in range(ndw):
                val = np.where(ok, self.lds[safe + k], U32(0))
                self.vwrite(w, d + k, val)
        return run

    def op_ds_read_b32(self, a, mods): return self._ds_read(a, mods, 1)
    def op_ds_read_b64(self, a, mods): return self._ds_read(a, mods, 2)
    def op_ds_read_b96(self, a, mods): return self._ds_read(a, mods, 3)
    def op_ds_read_b128(self, a, mods): return self._ds_read(a, mods, 4)

    def op_ds_read2_b32(self, a, mods):
        d = self._reg(a[0])[1]; ga = self.srcu(a[1])
        o0, o1 = int(mods.get("offset0", "0"), 0) * 4, int(mods.get("offset1", "0"), 0) * 4

        def run(w):
            for k, off in enumerate((o0, o1)):
                idx, ok = self._lds_index(w, ga(w), off, 1)
                self.vwrite(w, d + k, np.where(ok, self.lds[np.where(ok, idx, 0)], U32(0)))
        return run

    def _ds_write(self, a, mods, ndw):
        ga = self.srcu(a[0]); s = self._reg(a[1])[1]; off = int(mods.get("offset", "0"), 0)

        def run(w):
            idx, ok = self._lds_index(w, ga(w), off, ndw)
            for k in range(ndw):
                self.lds[idx[ok] + k] = w.V[s + k][ok]
        return run

    def op_ds_write_b32(self, a, mods): return self._ds_write(a, mods, 1)
    def op_ds_write_b64(self, a, mods): return self._ds_write(a, mods, 2)
    def op_ds_write_b96(self, a, mods): return self._ds_write(a, mods, 3)
    def op_ds_write_b128(self, a, mods): return self._ds_write(a, mods, 4)

    def _ds_write2(self, a, mods, ndw, stride):
        ga = self.srcu(a[0]); s0, s1 = self._reg(a[1])[1], self._reg(a[2])[1]
        o0, o1 = int(mods.get("offset0", "0"), 0) * stride, int(mods.get("offset1", "0"), 0) * stride

        def run(w):
            for s, off in ((s0, o0), (s1, o1)):
                idx, ok = self._lds_index(w, ga(w), off, ndw)
                for k in range(ndw):
                    self.lds[idx[ok] + k] = w.V[s + k][ok]
        return run

    def op_ds_write2_b32(self, a, mods): return self._ds_write2(a, mods, 1, 4)
    def op_ds_write2_b64(self, a, mods): return self._ds_write2(a, mods, 2, 8)
    def op_ds_write2st64_b32(self, a, mods): return self._ds_write2(a, mods, 1, 4 * 64)
    def op_ds_write2st64_b64(self, a, mods): return self._ds_write2(a, mods, 2, 8 * 64)

    def op_ds_bpermute_b32(self, a, mods):
        d = self._reg(a[0])[1]; ga = self.srcu(a[1]); s = self._reg(a[2])[1]
        off = int(mods.get("offset", "0"), 0)

        def run(w):
            src = ((ga(w) + U32(off)) >> U32(2)) & U32(63)
            val = np.where(w.em[src], w.V[s][src], U32(0))       # a lane that is switched off contributes 0
            self.vwrite(w, d, val)
        return run

    # ---- global / scratch memory ----
    def _addr(self, a_addr, a_base, mods):
        """address = 64-bit VGPR pair, or SGPR pair + 32-bit VGPR offset; plus the immediate offset."""
        off = int(mods.get("offset", "0"), 0)
        r = self._reg(a_addr)
        if a_base == "off":
            n = r[1]
            return lambda w: (w.V[n].astype(U64) | (w.V[n + 1].astype(U64) << U64(32))) + U64(off & FULL)
        gb = self.src64s(a_base); n = r[1]
        return lambda w: w.V[n].astype(U64) + U64((gb(w) + off) & FULL)

    def _stray(self, w, what, ins, addr, bad):
        lanes = [l for l in range(64) if bad >> l & 1]
        raise StrayAccess("%s outside every allocation: lanes %s, first address 0x%x; %s (assembly line %d)" %
                          (what, lanes[:8], int(addr[lanes[0]]), ins.text, ins.line))

    def _gload(self, a, mods, ndw, ins):
        d = self._reg(a[0])[1]
        ga = self._addr(a[1], a[2], mods)
        lib = self.lib

        def run(w):
            if not w.exec: return
            addr = np.ascontiguousarray(ga(w), dtype=U64)
            out = np.zeros((ndw, 64), U32)
            bad = lib.emu_gather(addr.ctypes.data, w.exec, ndw, out.ctypes.data)
            if bad: self._stray(w, "load", ins, addr, bad)
            for k in range(ndw):
                self.vwrite(w, d + k, out[k])
        return run

    def op_global_load_dword(self, a, mods, sdwa, ins): return self._gload(a, mods, 1, ins)
    def op_global_load_dwordx2(self, a, mods, sdwa, ins): return self._gload(a, mods, 2, ins)
    def op_global_load_dwordx3(self, a, mods, sdwa, ins): return self._gload(a, mods, 3, ins)
    def op_global_load_dwordx4(self, a, mods, sdwa, ins): return self._gload(a, mods, 4, ins)

    def _gstore(self, a, mods, ndw, ins):
        s = self._reg(a[1])[1]
        ga = self._addr(a[0], a[2], mods)
        lib = self.lib

        def run(w):
            if not w.exec: return
            addr = np.ascontiguousarray(ga(w), dtype=U64)
            data = np.ascontiguousarray(w.V[s:s + ndw])
            bad = lib.emu_scatter(addr.ctypes.data, w.exec, ndw, data.ctypes.data)
            if bad: self._stray(w, "store", ins, addr, bad)
        return run

    # (flat loads of these kernels address global memory: no operand for a base, the pointer is the whole address)
    def op_flat_load_dword(self, a, mods, sdwa, ins): return self._gload(a + ["off"], mods, 1, ins)
    def op_flat_load_dwordx2(self, a, mods, sdwa, ins): return self._gload(a + ["off"], mods, 2, ins)
    def op_flat_load_dwordx4(self, a, mods, sdwa, ins): return self._gload(a + ["off"], mods, 4, ins)

    def _gstore_small(self, a, mods, nbytes, ins):
        s = self._reg(a[1])[1]
        ga = self._addr(a[0], a[2], mods)
        lib = self.lib

        def run(w):
            if not w.exec: return
            addr = ga(w)
            for l in range(64):
                if not w.em[l]: continue
                p = int(addr[l])
                if not lib.emu_known(p, nbytes):
                    self._stray(w, "store", ins, addr, 1 << l)
                ctypes.memmove(p, int(w.V[s][l]).to_bytes(4, "little")[:nbytes], nbytes)
        return run

    def op_global_store_byte(self, a, mods, sdwa, ins): return self._gstore_small(a, mods, 1, ins)
    def op_global_store_short(self, a, mods, sdwa, ins): return self._gstore_small(a, mods, 2, ins)

    def op_global_store_dword(self, a, mods, sdwa, ins): return self._gstore(a, mods, 1, ins)
    def op_global_store_dwordx2(self, a, mods, sdwa, ins): return self._gstore(a, mods, 2, ins)
    def op_global_store_dwordx3(self, a, mods, sdwa, ins): return self._gstore(a, mods, 3, ins)
    def op_global_store_dwordx4(self, a, mods, sdwa, ins): return self._gstore(a, mods, 4, ins)

    def op_global_atomic_add(self, a, mods, sdwa, ins):
        ret = len(a) == 4
        d = self._reg(a[0])[1] if ret else None
        a_addr, a_data, a_base = (a[1], a[2], a[3]) if ret else (a[0], a[1], a[2])
        s = self._reg(a_data)[1]
        ga = self._addr(a_addr, a_base, mods)
        lib = self.lib

        def run(w):
            if not w.exec: return
            addr = np.ascontiguousarray(ga(w), dtype=U64)
            data = np.ascontiguousarray(w.V[s]); before = np.zeros(64, U32)
            bad = lib.emu_atomic_add_u32(addr.ctypes.data, w.exec, data.ctypes.data, before.ctypes.data)
            if bad: self._stray(w, "atomic", ins, addr, bad)
            if ret: self.vwrite(w, d, before)
        return run

    def _scratch(self, a, mods, ndw, load):
        off = int(mods.get("offset", "0"), 0)
        if load:
            d = self._reg(a[0])[1]; a_v, a_s = a[1], a[2]
        else:
            a_v, d, a_s = a[0], self._reg(a[1])[1], a[2]
        if a_v != "off" or a_s != "off":
            raise Unsupported("scratch access with a register address")

        def run(w):
            base = off >> 2
            if off & 3 or base + ndw > w.scratch.shape[1]:
                raise StrayAccess("scratch access at offset %d beyond the kernel's %d bytes" % (off, self.scratch_bytes))
            for k in range(ndw):
                if load:
                    self.vwrite(w, d + k, w.scratch[:, base + k])
                else:
                    np.copyto(w.scratch[:, base + k], w.V[d + k], where=w.em)
        return run

    def op_scratch_load_dword(self, a, mods): return self._scratch(a, mods, 1, True)
    def op_scratch_load_dwordx2(self, a, mods): return self._scratch(a, mods, 2, True)
    def op_scratch_load_dwordx3(self, a, mods): return self._scratch(a, mods, 3, True)
    def op_scratch_load_dwordx4(self, a, mods): return self._scratch(a, mods, 4, True)
    def op_scratch_store_dword(self, a, mods): return self._scratch(a, mods, 1, False)
    def op_scratch_store_dwordx2(self, a, mods): return self._scratch(a, mods, 2, False)
    def op_scratch_store_dwordx3(self, a, mods): return self._scratch(a, mods, 3, False)
    def op_scratch_store_dwordx4(self, a, mods): return self._scratch(a, mods, 4, False)

    # ---- running ----
    def _kernarg(self, args, grid, block):
        size = int(self.k.desc.get("kernarg_size", len(args)))
        buf = bytearray(max(size, len(args)) + 64)      # (the compiler may read a whole dwordx8 where fewer words are arguments)
        buf[:len(args)] = args
        for arg in self.k.meta[".args"]:
            kind, off = arg[".value_kind"], arg[".offset"]
            val = {"hidden_block_count_x": grid[0], "hidden_block_count_y": grid[1], "hidden_block_count_z": grid[2],
                   "hidden_group_size_x": block[0], "hidden_group_size_y": block[1], "hidden_group_size_z": block[2],
                   "hidden_remainder_x": 0, "hidden_remainder_y": 0, "hidden_remainder_z": 0,
                   "hidden_global_offset_x": 0, "hidden_global_offset_y": 0, "hidden_global_offset_z": 0,
                   "hidden_grid_dims": 1}.get(kind)
            if val is not None:
                buf[off:off + arg[".size"]] = int(val).to_bytes(arg[".size"], "little")
        return buf

    def launch(self, grid, block, args, max_instructions=None, time_slice=997):
        """grid, block: 3-tuples; args: the explicit kernel arguments as the device lays them out. The waves of a work-group
        take turns of `time_slice` instructions (they share LDS and race for the tile queues, as on the machine);
        work-groups run one after the other."""
        threads = block[0] * block[1] * block[2]
        karg = self._kernarg(bytes(args), grid, block)
        kbuf = ctypes.create_string_buffer(bytes(karg), len(karg))
        self.lib.emu_register_range(ctypes.addressof(kbuf), len(karg))
        d = self.k.desc
        try:
            for g in range(grid[0] * grid[1] * grid[2]):
                gx, gy, gz = g % grid[0], g // grid[0] % grid[1], g // (grid[0] * grid[1])
                self.lds = np.zeros(self.lds_bytes // 4 + 16, U32)
                if self.garbage is not None:
                    self.lds[:] = self.garbage.integers(0, 1 << 32, size=self.lds.size, dtype=np.uint64).astype(U32)
                waves = []
                for wi in range((threads + 63) // 64):
                    w = Wave(self.nv)
                    if self.garbage is not None:
                        w.V[:] = self.garbage.integers(0, 1 << 32, size=w.V.shape, dtype=np.uint64).astype(U32)
                        w.S = [int(x) for x in self.garbage.integers(0, 1 << 32, size=128, dtype=np.uint64)]
                        w.vcc = int(self.garbage.integers(0, 1 << 63, dtype=np.uint64)); w.scc = int(self.garbage.integers(0, 2)); w.m0 = w.S[127]
                    s = 0
                    if d.get("user_sgpr_private_segment_buffer"): s += 4
                    if d.get("user_sgpr_dispatch_ptr"): s += 2
                    if d.get("user_sgpr_queue_ptr"): s += 2
                    if d.get("user_sgpr_kernarg_segment_ptr"):
                        a = ctypes.addressof(kbuf); w.S[s] = a & 0xffffffff; w.S[s + 1] = a >> 32; s += 2
                    if d.get("user_sgpr_dispatch_id"): s += 2
                    if d.get("user_sgpr_private_segment_size"): s += 1
                    if d.get("user_sgpr_kernarg_preload_length"): raise Unsupported("kernel-argument preload")
                    if s != d.get("user_sgpr_count", s): raise Unsupported("user SGPR layout")
                    if d.get("system_sgpr_workgroup_id_x"): w.S[s] = gx; s += 1
                    if d.get("system_sgpr_workgroup_id_y"): w.S[s] = gy; s += 1
                    if d.get("system_sgpr_workgroup_id_z"): w.S[s] = gz; s += 1
                    tid = np.arange(64, dtype=U32) + U32(64 * wi)
                    live = tid < threads
                    # the work-item id arrives packed in v0: x in bits 0-9, y in 10-19, z in 20-29
                    tx, ty, tz = tid % U32(block[0]), tid // U32(block[0]) % U32(block[1]), tid // U32(block[0] * block[1])
                    w.V[0][:] = np.where(live, tx | (ty << U32(10)) | (tz << U32(20)), 0)
                    w.set_exec(bools_to_mask(live))
                    w.flush = d.get("float_denorm_mode_32", 0) == 0
                    w.scratch = np.full((64, max(self.scratch_bytes // 4, 1)), 0xdeadbeef, U32)
                    if self.garbage is not None:
                        w.scratch[:] = self.garbage.integers(0, 1 << 32, size=w.scratch.shape, dtype=np.uint64).astype(U32)
                    waves.append(w)
                pending = list(waves)
                while pending:
                    for w in pending:
                        self._run_wave(w, max_instructions, time_slice)
                    if all(w.at_barrier or w.done for w in pending):
                        for w in pending:
                            w.at_barrier = False
                    # (a wave that waits at the barrier does not run until every other wave has arrived or ended)
                    pending = [w for w in pending if not w.done]
        finally:
            self.lib.emu_forget_range(ctypes.addressof(kbuf))

    def _run_wave(self, w, max_instructions, time_slice):
        code = self.compiled
        n = len(code)
        pc = w.pc
        count = 0
        trace = self.trace
        while not w.done and not w.at_barrier and count < time_slice:
            if pc >= n:
                raise Unsupported("ran past the end of the kernel")
            if trace is not None:
                trace(self, w, pc)
            if self.check_waits:
                self._check_wait(w, pc)
            if self.histogram is not None:
                self.histogram[pc] += 1
            nxt = code[pc](w)
            pc = pc + 1 if nxt is None else nxt
            count += 1
            if max_instructions and self.executed + count > max_instructions:
                w.pc = pc
                self.executed += count
                raise RuntimeError("instruction budget of %d exceeded at assembly line %d" % (max_instructions, self.k.instrs[min(pc, n - 1)].line))
        w.pc = pc
        self.executed += count
