#!/usr/bin/env python3
"""Counts what a kernel build executes, from its gfx950 assembly (gcn_interp): wave-instructions by kind and per ray, on a
fixed small workload. No GPU and no timing: the numbers are counts, exact for this workload, and comparable between builds
of the same tree (kernel variants: give -D options).

    python3 tests/emu/isa/instruction_mix.py [-DRT_...=v ...]
"""
import os
import sys

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
    tag = "trace_kernels" + "".join(c if c.isalnum() else "_" for c in "".join(defines))
    asm = assemble(extra=defines, tag=tag)
    noise, sky, seeds = scenes.noise_texture(64, 48), scenes.sky_cubemap(32), scenes.frame_seeds(4)
    W, H, bl = 64, 40, 8
    sc = make_scene("sponza_like", target_tris=6000)
    with Emulated(devices=1, cus=2):          # eight waves for 40 tiles
        g, o = ra.Renderer(W, H), Oracle(W, H)
        setup(g, sc, noise, sky, bounce_limit=bl)
        setup(o, sc, noise, sky, bounce_limit=bl)
        with FromAssembly(asm) as run:
            g.render(0, seeds[:1]); run.check()          # first launch: tiles in their natural order
            for m in run.machines.values():
                m.histogram = [0] * len(m.k.instrs)
            g.counters(reset=True)
            g.render(1, seeds[1:2]); run.check()
            rays = g.counters(reset=True)["rays"]
            o.render(0, seeds[:2])
            assert differing_pixels(g.read_accum(), o.read_accum()) == 0
            mix = {}
            for m in run.machines.values():
                for k, v in m.instruction_mix().items():
                    mix[k] = mix.get(k, 0) + v
        g.close()
    total = sum(mix.values())
    print("sponza-like (6 000 triangles), %dx%d, bounce limit %d, second frame, %d rays, options: %s" % (W, H, bl, rays, " ".join(defines) or "(none)"))
    print("wave-instructions %d = %.1f per ray" % (total, total / rays))
    for k in ("valu", "salu", "branch", "vmem", "lds", "scratch", "smem", "wait"):
        print("  %-8s %9d  %6.2f per ray  %5.1f %%" % (k, mix.get(k, 0), mix.get(k, 0) / rays, 100.0 * mix.get(k, 0) / total))


if __name__ == "__main__":
    main()
