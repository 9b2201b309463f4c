"""Full-frame time vs tile size on one GPU, plus image identity across tile sizes (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_raytracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sp = scenes.sponza_like()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    ref = None
    for ts in (128, 128, 256, 64, 32, 16, 8):
        st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, tile_size=ts)
        img = ctx.read_rgb32f()
        if ref is None: ref = img
        print(f"tile={ts:3d}: {st['kernel_ms']:.2f} ms  {st['rays']/st['kernel_ms']/1e3:.0f} Mrays/s  identical={np.array_equal(ref.view(np.uint32), img.view(np.uint32))}", flush=True)
