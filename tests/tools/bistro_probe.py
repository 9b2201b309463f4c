import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle
from gpu_raytracer_amd import api, scenes
scene = scenes.bistro_like()
w, h = 96, 54
ref = oracle.render_frame(oracle.PackedScene(scene), w, h, mode=1)
for rep in range(3):
    with api.Context() as ctx:
        ctx.upload_scene(scene)
        st = ctx.render(w, h, scene.camera, mode=1)
        prim, t = ctx.read_hits(); rgb = ctx.read_rgb32f()
    print(os.environ.get("RT_HIP_LIB", "default"), "rep", rep, "prim mismatches", int((prim != ref["prim"]).sum()), "t mismatches", int((t != ref["t"]).sum()), "nodes", st["bvh_nodes"], "depth", st["bvh_depth"], flush=True)
