"""Time of rank 0's share of the headline frame for world sizes 1 and 8 (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
sp = scenes.sponza_like()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    for world, ts in ((1, 128), (8, 32)):
        best = 1e9
        for rep in range(3):
            st = ctx.render(1920, 1080, sp.camera, mode=2, spp=64, max_bounces=4, tile_size=ts, tile_rank=0, tile_world=world)
            best = min(best, st["kernel_ms"])
        print(f"world={world}: {best:.2f} ms  {st['rays']/best/1e3:.0f} Mrays/s", flush=True)
