"""bistro-like 1080p 16 spp 4 bounces: time and counters (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
sp = scenes.bistro_like()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    for rep in range(3):
        st = ctx.render(1920, 1080, sp.camera, mode=2, spp=16, max_bounces=4)
    print(f"{st['kernel_ms']:.2f} ms {st['rays']/st['kernel_ms']/1e3:.0f} Mrays/s", flush=True)
    st = ctx.render(1920, 1080, sp.camera, mode=2, spp=4, max_bounces=4, counters=True)
    print(f"nodes/seg={st['node_visits']/st['rays']:.1f} tris/seg={st['tri_tests']/st['rays']:.2f}")
