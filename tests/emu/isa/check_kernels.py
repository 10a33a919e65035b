#!/usr/bin/env python3
"""Runs the tracer's kernels from the compiler's gfx950 assembly (gcn_interp) on small scenes and compares with the oracle:
the check of a kernel build below source level that needs no GPU. For the tree it lies in, with the library's flags plus any
-D options given (kernel variants):

    python3 tests/emu/isa/check_kernels.py [-DRT_...=v ...] [--large] [--seed 7]

Per case: pixels differing from the oracle (must be 0), instructions executed, LDS accesses out of range, and at the end the
instructions that named a register whose memory result had not been waited for (must be none). Registers, LDS and scratch
start with random bits. Exit status 1 on any difference, stray address or missing wait."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), HERE]

import raytracing_amd as ra  # noqa: E402
from emulation import Emulated  # noqa: E402
from raytracing_amd import scenes  # noqa: E402
from rt_oracle import Oracle  # noqa: E402
from util import differing_pixels, make_scene, setup  # noqa: E402
from harness import FromAssembly, assemble  # noqa: E402


def main():
    defines = tuple(a for a in sys.argv[1:] if a.startswith("-D"))
    large = "--large" in sys.argv
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 7
    tag = "trace_kernels" + "".join(c if c.isalnum() else "_" for c in "".join(defines))
    asm = assemble(extra=defines, tag=tag)
    noise, sky, seeds = scenes.noise_texture(64, 48), scenes.sky_cubemap(32), scenes.frame_seeds(8)
    cases = [("lightbox", {}, 40, 24), ("sponza_like", {"target_tris": 6000}, 45, 27)] if large else [("lightbox", {}, 24, 16), ("sponza_like", {"target_tris": 6000}, 27, 13)]
    bad = 0
    with Emulated(devices=1, cus=2):
        for name, kw, W, H in cases:
            sc = make_scene(name, **kw)
            g, o = ra.Renderer(W, H), Oracle(W, H)
            setup(g, sc, noise, sky)
            setup(o, sc, noise, sky)
            with FromAssembly(asm, kernels=("trace_kernel", "prepare_frames_kernel", "prepare_rays_kernel"), garbage=seed, check_waits=True) as run:
                for bl, frames, split, count in ((1, 1, False, False), (0, 1, False, False), (2, 3, False, False), (2, 3, True, False), (8, 2, False, False), (4, 2, False, True)):
                    for r in (g, o):
                        r.set_bounce_limit(bl); r.clear(); r.counters(reset=True)
                    g.set_counting(count)
                    t, n0 = time.time(), len(run.launches)
                    if split:
                        for k in range(frames):
                            g.render(k, seeds[k:k + 1]); run.check()
                    else:
                        g.render(0, seeds[:frames]); run.check()
                    o.render(0, seeds[:frames])
                    d = differing_pixels(g.read_accum(), o.read_accum())
                    cg, co = g.counters(reset=True), o.counters(reset=True)
                    same = all(cg[k] == co[k] for k in co) if count else cg["rays"] == co["rays"]
                    tr = [l for l in run.launches[n0:] if "trace_kernel" in l[0]]
                    print("%-12s %dx%d bounce limit %d, %d frame(s)%s%s: %d pixels differ, counters %s; %s, %d instructions, %d LDS accesses out of range (%.0f s)" % (
                        name, W, H, bl, frames, " one per launch" if split else "", " counting" if count else "", d, "equal" if same else "DIFFER",
                        tr[0][0].strip("()"), sum(l[2] for l in tr), sum(l[3] for l in tr), time.time() - t), flush=True)
                    bad += (d != 0) + (not same) + sum(l[3] for l in tr)
                g.set_counting(False)
                ids = bool((g.primary_ids(seeds[0]) == o.primary_ids(seeds[0])).all()); run.check()
                print("%-12s primary ids %s" % (name, "equal" if ids else "DIFFER"))
                v = run.wait_violations()
                print("%-12s instructions that named a register still in flight: %d %s" % (name, len(v), v[:4]))
                bad += (not ids) + len(v)
            g.close()
    print("ok" if bad == 0 else "FAILED")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
