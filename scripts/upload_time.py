"""rt_upload_scene time (BVH build + H2D) for the two large scenes (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
for name in ("sponza_like", "bistro_like"):
    sc = scenes.SCENES[name]()
    with api.Context() as ctx:
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); ctx.upload_scene(sc); ts.append(time.perf_counter() - t0)
        print(f"{name}: upload {min(ts)*1e3:.0f} ms (best of 3)", flush=True)
