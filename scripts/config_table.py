"""One line per BASELINE.json config that fits one GPU (development aid; numbers quoted in DESIGN.md)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
rows = [("cornell12", 256, 256, 1, 0, 1, 1), ("cornell12", 1920, 1080, 64, 0, 1, 1), ("sponza_like", 1920, 1080, 16, 4, 1, 1),
        ("sponza_like", 1920, 1080, 64, 4, 1, 1), ("sponza_like", 3840, 2160, 256, 8, 8, 32), ("bistro_like", 3840, 2160, 64, 8, 8, 32),
        ("bistro_like", 1920, 1080, 16, 4, 1, 128)]
last = None
with api.Context() as ctx:
    for name, w, h, spp, bounces, world, ts in rows:
        if name != last:
            sc = scenes.SCENES[name]()
            t0 = time.time(); ctx.upload_scene(sc); up = time.time() - t0
            last = name
        best = None
        for rep in range(2):
            st = ctx.render(w, h, sc.camera, mode=2, spp=spp, max_bounces=bounces, tile_size=ts if world > 1 else 128, tile_rank=0, tile_world=world)
            if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
        print(f"{name} {sc.n_triangles} tris {w}x{h} {spp} spp {bounces} bounces, share 1/{world}: {best['kernel_ms']:.2f} ms, {best['rays']/1e6:.1f} M segments "
              f"(cam {best['primary_rays']/1e6:.1f} cont {best['continuation_rays']/1e6:.1f} shadow {best['shadow_rays']/1e6:.1f}), {best['rays']/best['kernel_ms']/1e3:.0f} Mrays/s; "
              f"bvh nodes {best['bvh_nodes']} depth {best['bvh_depth']}, upload+build {up:.1f} s", flush=True)
