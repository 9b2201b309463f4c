"""Round-3 probe: camera beams on / off on the headline scene: frame time, closest-hit-only frame time, CRC, list statistics."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sp = scenes.sponza_like()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    ctx.prepare()
    for label, kw in (("beams", {}), ("no beams", {"no_beams": True}), ("beams", {}), ("no beams", {"no_beams": True})):
        ms = min(ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, **kw)["kernel_ms"] for _ in range(3))
        crc = zlib.crc32(ctx.read_rgb32f().tobytes())
        ms_ns = min(ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, no_shadows=True, **kw)["kernel_ms"] for _ in range(3))
        st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, no_shadows=True, counters=True, **kw)
        segs = st["rays"]
        print(f"{label}: frame {ms:.2f} ms crc {crc} | closest-only {ms_ns:.2f} ms | visits/seg {st['node_visits']/segs:.2f} tris/seg {st['tri_tests']/segs:.2f}", flush=True)
