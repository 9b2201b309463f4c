"""Round-3 probe 1 (development aid): where the closest-hit traversal's visits go on the headline scene.
 - collapse cost (RT_BVH8_COST_TRAVERSE) against visits / triangle tests / time of the closest-hit-only frame (RT_FLAG_NO_SHADOWS:
   the paths are the same, no shadow segments are traced);
 - RT_WF_PROBE=1: every closest-hit segment walked a second time starting from its own hit distance - the visits no ordering can avoid;
 - Cornell 1080p 64 spp primary only: pipeline vs the nested-loop megakernel vs the state machine."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes

def frame(ctx, sc, spp, bounces, reps=3, **kw):
    ms = []
    for _ in range(reps):
        st = ctx.render(1920, 1080, sc.camera, mode=2, spp=spp, max_bounces=bounces, **kw)
        ms.append(st["kernel_ms"])
    return min(ms), st

sp = scenes.sponza_like()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for ct8 in (None, "2", "3", "4", "6", "10"):
    for ml in (None,):
        if ct8: os.environ["RT_BVH8_COST_TRAVERSE"] = ct8
        with api.Context() as ctx:
            t0 = time.perf_counter(); ctx.upload_scene(sp); up = (time.perf_counter() - t0) * 1e3
            b = ctx.debug_check_bvh()
            ms_ns, _ = frame(ctx, sp, spp, 4, no_shadows=True)
            ms_all, _ = frame(ctx, sp, spp, 4)
            st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, counters=True, no_shadows=True)
            d = list(ctx.debug_counters().values())
            segs = st["rays"]
            line = (f"ct8={ct8 or 'default'} nodes={b['nodes']} leaves={b['leaves']} depth={b['depth']} upload={up:.0f}ms | closest-only {ms_ns:.2f} ms, full {ms_all:.2f} ms | "
                    f"visits/seg={st['node_visits']/segs:.2f} tris/seg={st['tri_tests']/segs:.2f} empty visits={d[1]/max(1,st['node_visits']):.3f} entered/visit={d[2]/max(1,st['node_visits']):.2f} "
                    f"lanes/node step={st['node_visits']/max(1,d[3]):.1f} leaf lanes/step={d[5]/max(1,d[4]):.1f} trips={d[6]/max(1,d[4]):.2f}")
            if ct8 is None:
                os.environ["RT_WF_PROBE"] = "1"
                st2 = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, counters=True, no_shadows=True)
                del os.environ["RT_WF_PROBE"]
                line += f" | PROBE second walk: visits/seg={st2['node_visits']/segs:.2f} tris/seg={st2['tri_tests']/segs:.2f}"
                crc = zlib.crc32(ctx.read_rgb32f().tobytes())
                st3 = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=4, no_shadows=True)
                line += f" crc probe {crc} plain {zlib.crc32(ctx.read_rgb32f().tobytes())}"
            print(line, flush=True)
os.environ.pop("RT_BVH8_COST_TRAVERSE", None)

co = scenes.cornell12()
with api.Context() as ctx:
    ctx.upload_scene(co)
    for label, kw in (("pipeline", {}), ("nested megakernel", {"kernel_v1": True}), ("state machine", {"kernel_sm": True})):
        ms, st = frame(ctx, co, 64, 0, reps=4, **kw)
        print(f"cornell 1080p 64spp primary: {label}: {ms:.2f} ms, {st['rays']/1e6:.1f} M segments, crc {zlib.crc32(ctx.read_rgb32f().tobytes())}", flush=True)
