#!/usr/bin/env python3
"""Random small configurations with the library's kernels executed from their gfx950 assembly (gcn_interp) inside the host
emulation, against the oracle, bit for bit: random scene, image size, bounce limit, frames per launch and launch split, rays
per pixel (1 = the reference's setting, or 2..5), full counters on or off, one or two shards. Registers, LDS and scratch
start with random bits, every global address is checked against the runtime's allocations, and every use of a register
whose memory result has not been waited for counts as a failure. No GPU; a case takes 5 to 30 seconds.

    python3 tests/emu/isa/fuzz_kernels.py [--cases N] [--seed S] [--rays one|several|both] [-DRT_...=v ...]
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), HERE]

import numpy as np  # noqa: E402

import raytracing_amd as ra  # noqa: E402
from emulation import Emulated  # noqa: E402
from raytracing_amd import scenes, sharding  # noqa: E402
from rt_oracle import Oracle  # noqa: E402
from util import differing_pixels, make_scene, setup  # noqa: E402
from harness import FromAssembly, assemble  # noqa: E402

GENS = [("lightbox", {}), ("sponza_like", {"target_tris": 3000}), ("dragon_like", {"target_tris": 3000}), ("teapot_like", {"target_tris": 3000})]
KERNELS = ("trace_kernel", "prepare_frames_kernel", "prepare_rays_kernel", "untile_kernel", "retile_kernel", "reduce_tally_kernel", "order_keys_kernel")
COUNTERS = ("rays", "node_pops", "boxes_tested", "tris_tested", "noise_fetches", "sky_fetches", "hits")


def run(cases, seed, rays_mode="both", defines=(), log=print):
    rng = np.random.default_rng(seed)
    noise, sky = scenes.noise_texture(37, 23, seed=5), scenes.sky_cubemap(17, seed=9)
    tag = "trace_kernels" + "".join(c if c.isalnum() else "_" for c in "".join(defines))
    asm = assemble(extra=tuple(defines), tag=tag)
    bad = 0
    t_all = time.time()
    with Emulated(devices=1, cus=2):
        for case in range(cases):
            name, kw = GENS[rng.integers(len(GENS))]
            W, H = int(rng.integers(9, 36)), int(rng.integers(5, 22))
            bl = int(rng.choice([0, 1, 2, 4, 8, 16]))
            frames = int(rng.integers(1, 4))
            several = {"one": False, "several": True}.get(rays_mode, bool(rng.integers(2)))
            rays = int(rng.integers(2, 6)) if several else 1
            count = bool(rng.integers(2))
            world = int(rng.choice([1, 1, 2]))
            seeds = scenes.frame_seeds(frames + 2)[int(rng.integers(0, 3)):][:frames]
            sc = make_scene(name, **kw)
            o = Oracle(W, H)
            setup(o, sc, noise, sky, bounce_limit=bl)
            o.set_ray_count(rays)
            o.render(0, seeds)
            want = o.read_accum()
            t0 = time.time()
            note = ""
            with FromAssembly(asm, kernels=KERNELS, garbage=seed * 1000 + case, check_waits=True) as run_:
                per = sharding.shard_floats(W, H, world)
                flat = np.zeros(world * per, dtype=np.float32)
                got = None
                for rank in range(world):
                    g = ra.Renderer(W, H)
                    setup(g, sc, noise, sky, bounce_limit=bl)
                    g.set_ray_count(rays)
                    g.set_counting(count)
                    if world > 1:
                        g.set_shard(rank, world)
                    split = int(rng.integers(1, frames + 1))        # frames in the first launch; the rest one by one
                    g.render(0, seeds[:split]); run_.check()
                    for k in range(split, frames):
                        g.render(k, seeds[k:k + 1]); run_.check()
                    if world == 1:
                        got = g.read_accum(); run_.check()
                        if count:
                            cg, co = g.counters(), o.counters()
                            run_.check()
                            wrong = [k for k in COUNTERS if cg[k] != co[k]]
                            if wrong:
                                note += " counters differ: " + ",".join(wrong)
                    else:
                        part = np.empty(per, dtype=np.float32)
                        run_.share(part)
                        g.export_shard(part.ctypes.data); g.sync(); run_.check()
                        flat[rank * per:(rank + 1) * per] = part
                    g.close()
                if world > 1:
                    got = sharding.assemble_numpy(flat, W, H, world)
                waits = len(run_.wait_violations())
                lds = sum(l[3] for l in run_.launches)
                launched = len(run_.launches)
            n = differing_pixels(got, want)
            if n:
                note += " DIFFERS in %d pixels" % n
            if waits:
                note += " %d uses of a register before its wait" % waits
            if lds:
                note += " %d LDS accesses out of range" % lds
            log("case %3d: %-11s %2dx%-2d bl %2d frames %d rays %d counters %d shards %d  %d launches from assembly  %3.0f s  %s" % (
                case, name, W, H, bl, frames, rays, count, world, launched, time.time() - t0, note.strip() or "ok"))
            bad += bool(note)
    log("%d cases, %d bad, %.0f s" % (cases, bad, time.time() - t_all))
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rays", choices=("one", "several", "both"), default="both")
    args, rest = ap.parse_known_args()
    sys.exit(1 if run(args.cases, args.seed, args.rays, [a for a in rest if a.startswith("-D")], log=lambda s: print(s, flush=True)) else 0)
