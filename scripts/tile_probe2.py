"""tile size at a resolution that is a multiple of 128 (no partial tiles) vs 1080 rows (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
sp = scenes.sponza_like()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    for w, h in ((1920, 1080),):
        for ts in (128, 64):
            best = 1e9
            for rep in range(3):
                st = ctx.render(w, h, sp.camera, mode=2, spp=64, max_bounces=4, tile_size=ts)
                best = min(best, st["kernel_ms"])
            print(f"{w}x{h} tile {ts}: {best:.2f} ms {st['rays']/best/1e3:.0f} Mrays/s", flush=True)
