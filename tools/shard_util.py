import os, sys
sys.path.insert(0, os.getcwd())
import bench
import raytracing_amd as ra
from raytracing_amd import scenes
sc, W, H, bl = bench.load_workload("sponza_like_3840x2160_b8", 16)
seeds = scenes.frame_seeds(12)
for world in (1, 2, 4, 8):
    r = ra.Renderer(W, H)
    bench.setup(r, sc, scenes.noise_texture(), scenes.sky_cubemap(), bl)
    r.set_shard(0, world)
    for k in range(9):
        r.render(k, seeds[k:k + 1])
    r.set_counting(True); r.counters(reset=True)
    for k in range(9, 12):
        r.render(k, seeds[k:k + 1])
    c = r.counters()
    use = [round(l / (64.0 * max(1, n)), 3) for l, n in zip(c["sched_lanes"], c["sched_rounds"])]
    share = [round(x / max(1, sum(c["sched_cycles"])), 3) for x in c["sched_cycles"]]
    print("world %d: lane use new/interior/leaf/feed %s  time share %s  rays/frame %d" % (world, use, share, c["rays"] // 3))
    r.close()
