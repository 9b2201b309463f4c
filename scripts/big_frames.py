"""Largest single-GPU shapes of BASELINE.json: full 4K frames (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_raytracer_amd import api, scenes
for name, spp, bounces in (("bistro_like", 64, 4), ("sponza_like", 256, 8)):
    sc = scenes.SCENES[name]()
    with api.Context() as ctx:
        ctx.upload_scene(sc)
        t0 = time.time()
        st = ctx.render(3840, 2160, sc.camera, mode=2, spp=spp, max_bounces=bounces)
        img = ctx.read_rgb32f()
        print(f"{name} 3840x2160 {spp} spp {bounces} bounces: {st['kernel_ms']:.1f} ms, {st['rays']/1e9:.2f} G segments, {st['rays']/st['kernel_ms']/1e3:.0f} Mrays/s, "
              f"finite={np.isfinite(img).all()} mean={img.mean():.4f} wall {time.time()-t0:.1f} s", flush=True)
