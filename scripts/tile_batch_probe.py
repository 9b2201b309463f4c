"""tile size x paths-per-batch on the headline frame (development aid)."""
import sys, os, subprocess
here = os.path.dirname(os.path.abspath(__file__))
code = '''
import sys, os
sys.path.insert(0, %r)
from gpu_raytracer_amd import api, scenes
sp = scenes.sponza_like()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    for ts in (128, 64):
        best = 1e9
        for rep in range(3):
            st = ctx.render(1920, 1080, sp.camera, mode=2, spp=64, max_bounces=4, tile_size=ts)
            best = min(best, st["kernel_ms"])
        print("   tile", ts, "%%.2f ms" %% best, flush=True)
''' % os.path.dirname(here)
for target in ("67108864", "75000000", "100000000"):
    print("RT_WF_TARGET_PATHS =", target, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RT_WF_TARGET_PATHS=target))
