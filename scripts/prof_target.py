"""Minimal profiling target: render the headline frame a few times with one kernel (development aid)."""
import sys, os, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
spp, bounces, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
v1 = len(sys.argv) > 4 and sys.argv[4] == "v1"
sm = len(sys.argv) > 4 and sys.argv[4] == "sm"
sp = scenes.SCENES[os.environ.get("RT_PROF_SCENE", "sponza_like")]()
with api.Context() as ctx:
    ctx.upload_scene(sp)
    for _ in range(reps):
        st = ctx.render(1920, 1080, sp.camera, mode=2, spp=spp, max_bounces=bounces, kernel_v1=v1, kernel_sm=sm)
    print(st["kernel_ms"], st["rays"] / st["kernel_ms"] / 1e3, "crc", zlib.crc32(ctx.read_rgb32f().tobytes()))
