"""Id-space limits of the wavefront pipeline (development aid): 32 lights at 4K (batches clamped by the 32-bit queue
positions) and a frame whose single sample exceeds the limit (megakernel fallback); both against the state-machine kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_raytracer_amd import api, scenes
sc = scenes.random_soup(3000, seed=3, size=0.5, n_spheres=2, n_lights=32)
with api.Context() as ctx:
    ctx.upload_scene(sc)
    for w, h, spp, b in ((3840, 2160, 6, 2), (8192, 6144, 1, 1)):
        st = ctx.render(w, h, sc.camera, mode=2, spp=spp, max_bounces=b); a = ctx.read_rgb32f()
        st2 = ctx.render(w, h, sc.camera, mode=2, spp=spp, max_bounces=b, kernel_sm=True); c = ctx.read_rgb32f()
        print(f"{w}x{h} {spp} spp {b} bounces, 32 lights: default {st['kernel_ms']:.1f} ms, state machine {st2['kernel_ms']:.1f} ms, "
              f"identical={np.array_equal(a.view(np.uint32), c.view(np.uint32))} rays {st['rays']/1e6:.0f}M == {st2['rays']/1e6:.0f}M", flush=True)
