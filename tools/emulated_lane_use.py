"""Scheduler populations of a kernel build on the host emulation (tests/emu): rounds and lane use per scheduler state on
a Sponza-like scene, with about eight tiles per wave as in the 4K benchmark. Counts, not time: which lanes take part in
which round is decided by ballots, which the emulation reproduces; only the assignment of tiles to waves differs from a
GPU run. Usage: EMU_SO=<path to a librt_emulated.so> python tools/emulated_lane_use.py"""
import os, sys, time
import ctypes
os.environ["RT_EMU_CUS"] = os.environ.get("RT_EMU_CUS", "8"); os.environ["RT_EMU_THREADS"] = "3"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/oracle"):
    sys.path.insert(0, p)
import numpy as np
import raytracing_amd as ra
from raytracing_amd import _lib
_emulated = ctypes.CDLL(os.environ["EMU_SO"])            # put under the Python layer by hand, as the tests do: the package itself refuses it
for _name, (_res, _args) in _lib.SIGNATURES.items():
    getattr(_emulated, _name).restype, getattr(_emulated, _name).argtypes = _res, _args
_lib._lib = _emulated
from raytracing_amd import scenes
from util import make_scene, setup
NOISE = scenes.noise_texture(64, 48); SKY = scenes.sky_cubemap(32); SEEDS = scenes.frame_seeds(8)
W, H = int(os.environ.get("W", 160)), int(os.environ.get("H", 96))
sc = make_scene("sponza_like", target_tris=20000)
g = ra.Renderer(W, H)
setup(g, sc, NOISE, SKY, bounce_limit=8)
for k in range(3): g.render(k, SEEDS[k:k+1])      # warm the tile order
g.set_counting(True); g.counters(reset=True)
t = time.time()
for k in range(3, 6): g.render(k, SEEDS[k:k+1])
c = g.counters(reset=True)
names = ["new", "interior", "leaf", "shade"]
print("rays", c["rays"], "boxes/ray %.2f" % (c["boxes_tested"] / c["rays"]), "%.1fs" % (time.time() - t))
for i, n in enumerate(names):
    r, l = c["sched_rounds"][i], c["sched_lanes"][i]
    print("%-9s rounds %9d  lanes %10d  lane use %.3f" % (n, r, l, l / (64.0 * max(1, r))))
