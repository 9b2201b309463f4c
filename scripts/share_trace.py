"""Kernel-trace target: rank 0's 1/8 share of the headline frame, 3 times (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
sp = scenes.sponza_like()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    for rep in range(3):
        st = ctx.render(1920, 1080, sp.camera, mode=2, spp=64, max_bounces=4, tile_size=32, tile_rank=0, tile_world=8)
    print(st["kernel_ms"], st["wall_ms"])
