import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes
sp = scenes.sponza_like()
W, H = 1920, 1080
px = [(1290, 237), (1236, 443), (684, 502), (539, 513), (706, 237), (810, 52)]
with api.Context() as ctx:
    ctx.upload_scene(sp)
    for spp in range(2, 10):
        ctx.render(W, H, sp.camera, mode=2, spp=spp, max_bounces=0, no_shadows=True, kernel_pipeline=True, no_beams=True)
        ref = ctx.read_rgb32f().copy()
        ctx.render(W, H, sp.camera, mode=2, spp=spp, max_bounces=0, no_shadows=True, kernel_pipeline=True)
        got = ctx.read_rgb32f().copy()
        ctx.render(W, H, sp.camera, mode=2, spp=spp, max_bounces=0, no_shadows=True, kernel_sm=True)
        sm = ctx.read_rgb32f().copy()
        print(spp, [(bool((ref[y, x] != got[y, x]).any()), bool((ref[y, x] != sm[y, x]).any())) for x, y in px], "all diff beams", int((ref != got).any(-1).sum()), "sm", int((ref != sm).any(-1).sum()), flush=True)
