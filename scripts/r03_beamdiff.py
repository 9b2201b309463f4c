import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
sp = scenes.sponza_like()
W, H, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 4
with api.Context() as ctx:
    ctx.upload_scene(sp)
    ctx.render(W, H, sp.camera, mode=2, spp=spp, max_bounces=0, no_shadows=True, kernel_pipeline=True, no_beams=True)
    ref = ctx.read_rgb32f().copy()
    ctx.render(W, H, sp.camera, mode=2, spp=spp, max_bounces=0, no_shadows=True, kernel_pipeline=True, counters=bool(int(os.environ.get('BD_COUNT', '0'))))
    got = ctx.read_rgb32f().copy()
    tiles_x = (W + 127) // 128
    counts = ctx.debug_beams(200000)
    print("blocks", len(counts), "overflow", int((counts == 0xFFFFFFFF).sum()), "list mean", counts[counts != 0xFFFFFFFF].mean(), "max", counts[counts != 0xFFFFFFFF].max(),
          "hist", np.histogram(counts[counts != 0xFFFFFFFF], bins=[0, 1, 8, 16, 32, 64, 96, 129])[0])
    diff = (ref.view(np.uint32) != got.view(np.uint32)).any(-1)
    ys, xs = np.nonzero(diff)
    print("differing pixels", len(ys))
    # block index of a pixel: tile (128) -> 16x16 blocks per tile, block-major inside the tile
    for y, x in list(zip(ys, xs))[:12]:
        ty, tx = y // 128, x // 128
        by, bx = (y % 128) // 8, (x % 128) // 8
        b = (ty * tiles_x + tx) * 256 + by * 16 + bx
        print((x, y), "block", b, "count", counts[b] if b < len(counts) else None, "ref", ref[y, x], "got", got[y, x])
