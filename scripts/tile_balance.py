"""Per-rank time of the tile partition on ONE GPU: render rank r of `world` for every r (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sp = scenes.sponza_like()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    full = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4)
    full = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4)
    print(f"full frame: {full['kernel_ms']:.2f} ms, {full['rays']/1e6:.1f} M segments")
    for world in (2, 4, 8):
        for ts in (128, 64, 32):
            ms, rays = [], []
            for r in range(world):
                st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, tile_size=ts, tile_rank=r, tile_world=world)
                ms.append(st["kernel_ms"]); rays.append(st["rays"])
            print(f"world={world} tile={ts:3d}: max={max(ms):.2f} mean={sum(ms)/world:.2f} ideal={full['kernel_ms']/world:.2f} "
                  f"efficiency={full['kernel_ms']/world/max(ms):.3f} rays max/mean={max(rays)/(sum(rays)/world):.3f}", flush=True)
