"""Debug: far-origin parity (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle
from gpu_raytracer_amd import api, scenes
scene = scenes.random_soup(900, seed=12, size=0.5, n_spheres=1, n_lights=2)
w, h = 160, 96
for dist in (2e3, 2e5, 2e6, 2e7):
    cam = scene.camera.copy(); cam["position"] = (0, 0, dist); cam["direction"] = (0, 0, -1); cam["fov"] = 2e-5 * 2e7 / dist
    ref = oracle.render_frame(oracle.PackedScene(scene, use_bvh=False), w, h, camera=cam)
    with api.Context() as ctx:
        ctx.upload_scene(scene)
        ctx.render(w, h, cam, mode=0)
        prim, t = ctx.read_hits()
    ne = prim != ref["prim"]
    print(f"dist={dist:g}: prim mismatches {ne.sum()} t mismatches {(t.view(np.uint32) != ref['t'].view(np.uint32)).sum()} hits ref {(ref['prim']!=0xFFFFFFFF).sum()}")
    ys, xs = np.nonzero(ne)
    for y, x in list(zip(ys, xs))[:4]:
        print("   ", y, x, "gpu", hex(prim[y, x]), t[y, x], "ref", hex(ref["prim"][y, x]), ref["t"][y, x])
