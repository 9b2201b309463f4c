"""A/B of the extended-mode kernels on the headline workload (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_raytracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sp = scenes.sponza_like()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    imgs = {}
    for rnd in range(2):
        for name, v1 in (("v1", True), ("v2", False)):
            st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, kernel_v1=v1)
            imgs[name] = ctx.read_rgb32f()
            print(f"{name} {spp}spp 4b: kernel_ms={st['kernel_ms']:.2f} rays={st['rays']/1e6:.1f}M Mrays/s={st['rays']/st['kernel_ms']/1e3:.0f}", flush=True)
    print("v1 == v2 bit-exact:", np.array_equal(imgs["v1"].view(np.uint32), imgs["v2"].view(np.uint32)))
    st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=0)
    print(f"v2 {spp}spp 0b: kernel_ms={st['kernel_ms']:.2f} Mrays/s={st['rays']/st['kernel_ms']/1e3:.0f}")
    st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, no_shadows=True)
    print(f"v2 {spp}spp 4b noshadow: kernel_ms={st['kernel_ms']:.2f} Mrays/s={st['rays']/st['kernel_ms']/1e3:.0f}")
    for b, ns in ((4, False), (0, False)):
        st = ctx.render(1920, 1080, sp.camera, mode=2, spp=4, max_bounces=b, counters=True, no_shadows=ns)
        dg = ctx.debug_counters()
        print(f"diag 4spp {b}b: segs={st['rays']/1e6:.1f}M nodes/seg={st['node_visits']/st['rays']:.1f} tris/seg={st['tri_tests']/st['rays']:.2f} "
              f"trans passes/seg={dg['transition_passes']*64/st['rays']:.2f} lanes/pass={dg['transition_lanes']/max(1,dg['transition_passes']):.1f} "
              f"node util={dg['node_lanes']/max(1,dg['node_iters'])/64:.2f} node iters/seg(wave-norm)={dg['node_iters']*64/st['rays']:.1f} "
              f"leaf util={dg['leaf_lanes']/max(1,dg['leaf_iters'])/64:.2f} leaf iters/seg={dg['leaf_iters']*64/st['rays']:.1f} "
              f"cycles: transition {dg['cycles_transition']/(dg['cycles_transition']+dg['cycles_traversal']):.2f}", flush=True)
