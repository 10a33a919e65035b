#!/usr/bin/env python3
"""Randomised parity check: GPU (through the C ABI) against the CPU oracle, bit for bit, on random configurations.
  python3 tests/fuzz_parity.py [cases] [seed] [large]
With "large" the images reach 900x500 and up to 12 frames are split over as many launches, which takes the launch
sequence past the points at which the tile order is recomputed.
Each case draws a scene generator and size, an image size (not necessarily a multiple of 8), a camera pose and field of
view, a bounce limit, the number of frames and how they are split over launches, and optionally a shard count.
Test infrastructure: the oracle is only the checker."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import raytracing_amd as ra
from raytracing_amd import scenes, sharding
from rt_oracle import Oracle
from util import differing_pixels, make_scene, setup

GENS = [("lightbox", {}), ("sponza_like", {"target_tris": 6000}), ("sponza_like", {"target_tris": 20000}),
        ("dragon_like", {"target_tris": 8000}), ("teapot_like", {"target_tris": 9000})]


def run(cases, seed, large=False, log=print, host_buffers=False, ray_counts=False):
    """Runs `cases` random configurations; returns how many of them differ from the oracle. host_buffers: shards are
    exported into host arrays instead of torch device tensors (for the host emulation of the library, whose device memory
    is host memory). ray_counts: every case also draws the number of rays per pixel and frame (the shader's RAY_COUNT)
    from 2..9, from a random stream of its own, so that the cases are otherwise the ones the same seed gives without it."""
    rng = np.random.default_rng(seed)
    rng_rays = np.random.default_rng(seed + 7919)
    noises = [scenes.noise_texture(64, 48), scenes.noise_texture(37, 23, seed=5), scenes.noise_texture()]
    skies = [scenes.sky_cubemap(32), scenes.sky_cubemap(17, seed=9)]
    bad = 0
    t0 = time.time()
    for case in range(cases):
        name, kw = GENS[rng.integers(len(GENS))]
        W, H = (int(rng.integers(200, 900)), int(rng.integers(100, 500))) if large else (int(rng.integers(9, 150)), int(rng.integers(5, 90)))
        pos = tuple(float(x) for x in rng.uniform(-2.0, 6.0, 3) * np.array([1.0, 0.5, 0.4]))
        f = rng.normal(size=3); f[2] *= 0.4; f = f / np.linalg.norm(f)
        fov = np.float32(rng.uniform(40.0, 150.0) * 0.01745329251994329576923690768489)
        camera = dict(position=pos, facing=tuple(np.float32(f)), fov=float(fov), viewport_width=float(np.float32(2.0 * math.tan(float(fov) / 2.0))))
        bl = int(rng.choice([0, 1, 2, 3, 4, 5, 8, 16, 32]))
        frames = int(rng.integers(1, 13 if large else 7))
        split = int(rng.integers(1, frames + 1))            # frames in the first launch; the rest one by one
        world = int(rng.choice([1, 1, 2, 3, 5]))
        noise, sky = noises[rng.integers(len(noises))], skies[rng.integers(len(skies))]
        seeds = scenes.frame_seeds(frames + 3)[int(rng.integers(0, 3)):][:frames]
        rays = int(rng_rays.integers(2, 10)) if ray_counts else 1
        sc = make_scene(name, **kw)
        o = Oracle(W, H, threads=16)
        setup(o, sc, noise, sky, camera=camera, bounce_limit=bl)
        o.set_ray_count(rays)
        o.render(0, seeds)
        want = o.read_accum()

        def traced(rank):
            g = ra.Renderer(W, H)
            setup(g, sc, noise, sky, camera=camera, bounce_limit=bl)
            g.set_ray_count(rays)
            if world > 1:
                g.set_shard(rank, world)
            g.render(0, seeds[:split])
            for k in range(split, frames):
                g.render(k, seeds[k:k + 1])
            return g

        if world == 1:
            g = traced(0)
            got = g.read_accum()
            g.close()
        else:
            per = sharding.shard_floats(W, H, world)
            flat = np.zeros(world * per, dtype=np.float32)
            for rank in range(world):
                g = traced(rank)
                if host_buffers:
                    part = np.empty(per, dtype=np.float32)
                    g.export_shard(part.ctypes.data); g.sync()
                    flat[rank * per:(rank + 1) * per] = part
                else:
                    import torch
                    t = torch.empty(per, dtype=torch.float32, device="cuda")
                    g.export_shard(t.data_ptr()); g.sync()
                    flat[rank * per:(rank + 1) * per] = t.cpu().numpy()
                g.close()
            got = sharding.assemble_numpy(flat, W, H, world)
        n = differing_pixels(got, want)
        log("case %2d: %-11s %6s tris  %3dx%-3d bl %2d frames %d (first launch %d) shards %d%s  %s" % (
            case, name, kw.get("target_tris", "-"), W, H, bl, frames, split, world, " rays %d" % rays if ray_counts else "",
            "ok" if n == 0 else "DIFFERS in %d pixels" % n))
        bad += n != 0
    log("%d cases, %d with differences, %.0f s" % (cases, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                      large="large" in sys.argv[3:], ray_counts="rays" in sys.argv[3:]) else 0)
