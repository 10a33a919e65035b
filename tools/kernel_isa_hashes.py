#!/usr/bin/env python3
"""Fingerprints of the tracer kernels' instruction streams, from hipcc's gfx950 assembly of trace_kernels.hip: per kernel the
number of instructions and a hash over them with labels renumbered and names left out. Two builds whose fingerprints are equal
execute the same instructions; a change of source that leaves a kernel's fingerprint alone has not touched that kernel.

    python3 tools/kernel_isa_hashes.py [--write profiles/r03_kernel_isa_hashes.json] [-DRT_...=v ...]

profiles/r03_kernel_isa_hashes.json holds the fingerprints of the seven one-ray kernels of the build that passed the GPU
tests and was profiled (r03a); tests/test_isa_interpreter.py compares the current build with it."""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu", "isa"))


def toolchain():
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
    return " | ".join(l.strip() for l in out.splitlines() if "version" in l.lower())


def fingerprints(asm_path):
    import gcn_interp as gi
    kernels, _ = gi.parse_assembly(asm_path)
    out = {}
    for name, k in kernels.items():
        m = re.search(r"\d+(trace_kernelI(?:Lb[01]E)+)E", name)
        if not m:
            continue
        flags = re.findall(r"Lb([01])E", m.group(1))
        # position of a label = the instruction it precedes: hashing (index, text) pairs with labels replaced by their targets
        target = {lab: at for lab, at in k.labels.items()}
        text = []
        for ins in k.instrs:
            line = ins.text
            for lab in re.findall(r"\.LBB\d+_\d+", line):
                line = line.replace(lab, "@%d" % target[lab])
            text.append(line)
        key = "trace_kernel<%s>" % ", ".join("true" if f == "1" else "false" for f in flags)
        out[key] = {"instructions": len(text), "sha1": hashlib.sha1("\n".join(text).encode()).hexdigest()[:16]}
    return out


def main():
    from harness import assemble
    defines = tuple(a for a in sys.argv[1:] if a.startswith("-D"))
    tag = "trace_kernels" + "".join(c if c.isalnum() else "_" for c in "".join(defines))
    fp = fingerprints(assemble(extra=defines, tag=tag))
    for k in sorted(fp):
        print("%-60s %5d instructions  %s" % (k, fp[k]["instructions"], fp[k]["sha1"]))
    if "--write" in sys.argv:
        path = sys.argv[sys.argv.index("--write") + 1]
        one_ray = {k: v for k, v in fp.items() if k.endswith(", false>")}
        with open(path, "w") as f:
            json.dump({"toolchain": toolchain(), "what": "instruction streams of the one-ray tracer kernels (last template argument false)",
                       "kernels": one_ray}, f, indent=1, sort_keys=True)
        print("wrote", path)


if __name__ == "__main__":
    main()
