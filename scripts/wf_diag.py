"""Wave-level step statistics of the persistent traversal kernels (counting variant; development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sp = scenes.sponza_like()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    for label, kw in (("closest only (no shadows)", {"no_shadows": True}), ("closest + shadow", {})):
        st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, counters=True, **kw)
        d = list(ctx.debug_counters().values())
        segs, nodes, tris = st["rays"], st["node_visits"], st["tri_tests"]
        smax, g16, g24, nsteps, lsteps, llanes, ltrips, refills = d
        print(f"{label}: segs={segs/1e6:.1f}M nodes/seg={nodes/segs:.2f} tris/seg={tris/segs:.2f}")
        print(f"   node steps={nsteps/1e6:.2f}M lanes/step={nodes/max(1,nsteps):.1f} | leaf steps={lsteps/1e6:.2f}M lanes/step={llanes/max(1,lsteps):.1f} "
              f"trips/step={ltrips/max(1,lsteps):.2f} tri lane-tests per trip={tris/max(1,ltrips):.1f} | refills={refills/1e6:.2f}M segs/refill={segs/max(1,refills):.1f} stack max={smax}")
