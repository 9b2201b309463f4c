"""Profiling target: render rank 0's share of the headline frame partitioned over <world> ranks, <reps> times (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
world, reps = int(sys.argv[1]), int(sys.argv[2])
sp = scenes.sponza_like()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    for _ in range(reps):
        st = ctx.render(1920, 1080, sp.camera, mode=2, spp=64, max_bounces=4, tile_size=32, tile_rank=0, tile_world=world)
    print(st["kernel_ms"], st["rays"] / st["kernel_ms"] / 1e3)
