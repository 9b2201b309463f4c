import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
for name in ("sponza_like", "bistro_like"):
    sc = scenes.SCENES[name]()
    with api.Context() as ctx:
        ctx.upload_scene(sc)
        st = ctx.render(1920, 1080, sc.camera, mode=2, spp=2, max_bounces=4, counters=True)
        dg = ctx.debug_counters()
        print(name, "depth", st["bvh_depth"], "nodes", st["bvh_nodes"], "segs %.1fM" % (st["rays"]/1e6), "nodes/seg %.1f" % (st["node_visits"]/st["rays"]), "stack max", dg["transition_passes"], "visits>16: %.4f%%" % (100*dg["transition_lanes"]/st["node_visits"]), ">24: %.5f%%" % (100*dg["node_iters"]/st["node_visits"]), "Mrays/s", st["rays"]/st["kernel_ms"]/1e3)
