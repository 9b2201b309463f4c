"""Host-visible cost of the boundary: scene upload (H2D + BVH build) and frame read-back (D2H) next to the kernel time
(development aid; numbers quoted in DESIGN.md)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_raytracer_amd import api, scenes
sp = scenes.sponza_like()
with api.Context() as ctx:
    t0 = time.perf_counter(); ctx.upload_scene(sp); up = time.perf_counter() - t0
    t0 = time.perf_counter(); ctx.upload_scene(sp); up2 = time.perf_counter() - t0
    print(f"upload_scene (BVH build + H2D of {sp.n_triangles} triangles): first {up*1e3:.1f} ms, again {up2*1e3:.1f} ms")
    for mode, kw in ((1, {}), (2, {"spp": 64, "max_bounces": 4})):
        for rep in range(3):
            t0 = time.perf_counter(); st = ctx.render(1920, 1080, sp.camera, mode=mode, **kw); t1 = time.perf_counter()
            comb = ctx.read_rgba8_combined(); t2 = time.perf_counter()
            rgb = ctx.read_rgb32f(); t3 = time.perf_counter()
        print(f"mode {mode}: kernel {st['kernel_ms']:.3f} ms, rt_render wall {1e3*(t1-t0):.3f} ms, read rgba8 combined (8.3 MB) {1e3*(t2-t1):.3f} ms, "
              f"read rgb32f (24.9 MB) {1e3*(t3-t2):.3f} ms; rays {st['rays']/1e6:.1f} M -> {st['rays']/(t2-t0)/1e6:.0f} Mrays/s incl. rgba8 read-back")
