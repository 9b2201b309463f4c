import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
sp = scenes.sponza_like()
W, H, spp = 1920, 1080, 8
os.environ["RT_WF_PROBE"] = "2"
with api.Context() as ctx:
    ctx.upload_scene(sp)
    st = ctx.render(W, H, sp.camera, mode=2, spp=spp, max_bounces=0, no_shadows=True, kernel_pipeline=True, counters=True)
    d = list(ctx.debug_counters().values())
    # diag[k] = totals[8+k-3]... rt_api maps cnt[9]=t[6], cnt[10]=t[7], cnt[11..15]=t[8..12]; debug_counters returns cnt[8..15]
    print("mismatches", d[1])
    b, prims, ts, nl, p = d[3], d[4], d[5], d[6], d[7]
    print("block", b, "list prim", prims >> 32, "walk prim", prims & 0xFFFFFFFF, "list t", np.uint32(ts >> 32).view(np.float32), "walk t", np.uint32(ts & 0xFFFFFFFF).view(np.float32), "n_list", nl >> 32, "walk slot", nl & 0xFFFFFFFF, "path", p)
