"""Render the headline frame with trees from the host SAH build and the device build and compare (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, zlib
from gpu_raytracer_amd import api, scenes
import oracle
sp = scenes.sponza_like()
imgs = {}
for m in ("0", "1", "2"):
    os.environ["RT_BUILD_METHOD"] = m
    with api.Context() as ctx:
        ctx.upload_scene(sp)
        st = ctx.render(1920, 1080, sp.camera, mode=2, spp=64, max_bounces=4, tile_size=32)
        imgs[m] = ctx.read_rgb32f()
        print("method", m, "crc", zlib.crc32(imgs[m].tobytes()) & 0xFFFFFFFF, "rays", st["rays"], st["primary_rays"], st["continuation_rays"], st["shadow_rays"], flush=True)
for a, b in (("0", "2"), ("1", "2"), ("0", "1")):
    d = (imgs[a].view(np.uint32) != imgs[b].view(np.uint32)).any(-1)
    ys, xs = np.nonzero(d)
    print(a, "vs", b, ":", d.sum(), "pixels differ", list(zip(xs[:8].tolist(), ys[:8].tolist())))
d = (imgs["0"].view(np.uint32) != imgs["2"].view(np.uint32)).any(-1)
ys, xs = np.nonzero(d)
packed = oracle.PackedScene(sp)
for x, y in list(zip(xs.tolist(), ys.tolist()))[:4]:
    ref = oracle.render_extended(packed, 1920, 1080, 64, 4, region=(x, y, 1, 1))["rgb"][0, 0]
    print("pixel", x, y, "oracle", ref.view(np.uint32), "sah", imgs["0"][y, x].view(np.uint32), "device", imgs["2"][y, x].view(np.uint32))
