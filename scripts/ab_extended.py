"""A/B of the extended-mode implementations on the headline workload (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_raytracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
only = sys.argv[2] if len(sys.argv) > 2 else None
sp = scenes.sponza_like()
K = {"wf": {}, "sm": {"kernel_sm": True}, "v1": {"kernel_v1": True}}
with api.Context() as ctx:
    ctx.upload_scene(sp)
    imgs = {}
    for rnd in range(2):
        for name, kw in K.items():
            if only and name != only: continue
            st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, **kw)
            imgs[name] = ctx.read_rgb32f()
            print(f"{name} {spp}spp 4b: kernel_ms={st['kernel_ms']:.2f} wall_ms={st['wall_ms']:.2f} rays={st['rays']/1e6:.1f}M Mrays/s={st['rays']/st['kernel_ms']/1e3:.0f}", flush=True)
    if not only:
        print("bit-exact wf==sm==v1:", all(np.array_equal(imgs["wf"].view(np.uint32), imgs[k].view(np.uint32)) for k in ("sm", "v1")))
    for name in ([only] if only else ["wf", "v1"]):
        st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=0, **K[name])
        print(f"{name} {spp}spp 0b: kernel_ms={st['kernel_ms']:.2f} Mrays/s={st['rays']/st['kernel_ms']/1e3:.0f}")
    st = ctx.render(1920, 1080, sp.camera, mode=2, spp=4, max_bounces=4, counters=True, **K[only or "wf"])
    print(f"counters: nodes/seg={st['node_visits']/st['rays']:.1f} tris/seg={st['tri_tests']/st['rays']:.2f} segs={st['rays']/1e6:.1f}M (cam {st['primary_rays']/1e6:.1f} cont {st['continuation_rays']/1e6:.1f} shadow {st['shadow_rays']/1e6:.1f})")
