"""Profiling target for the light-grid shadow stage: the headline frame a few times (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_like"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sc = scenes.SCENES[name]()
with api.Context() as ctx:
    ctx.upload_scene(sc)
    for rep in range(3):
        st = ctx.render(1920, 1080, sc.camera, mode=2, spp=spp, max_bounces=4)
    print(st["kernel_ms"])
