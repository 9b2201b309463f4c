"""Checker: cameras with NaN / infinite / zero components must not hang the kernels (every loop ends when all comparisons
are false) and must give the oracle's image where the reference semantics are defined by the same float rules.
usage: nan_camera.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import oracle
from gpu_raytracer_amd import api, scenes, hostpack as H

scene = scenes.random_soup(800, seed=2, n_spheres=2, n_lights=3)
packed = oracle.PackedScene(scene, use_bvh=False)
nan, inf = float("nan"), float("inf")
cams = {
    "zero direction": H.camera(direction=(0.0, 0.0, 0.0)),
    "nan position": H.camera(position=(nan, 0.0, 0.0)),
    "inf position": H.camera(position=(inf, 0.0, 5.0)),
    "nan fov": H.camera(fov=nan),
    "fov 180": H.camera(fov=180.0),
    "fov 0": H.camera(fov=0.0),
    "up parallel to direction": H.camera(direction=(0.0, 1.0, 0.0), up=(0.0, 1.0, 0.0)),
    "huge position": H.camera(position=(3e38, -3e38, 3e38)),
}
bad = 0
with api.Context() as ctx:
    ctx.upload_scene(scene)
    for name, cam in cams.items():
        for mode in (0, 1):
            ref = oracle.render_frame(packed, 64, 48, camera=cam, mode=mode)
            ctx.render(64, 48, cam, mode=mode)
            ok = np.array_equal(ctx.read_rgba8_combined(), ref["combined"]) and np.array_equal(ctx.read_hits()[0], ref["prim"])
            print(f"{name}: mode {mode} {'ok' if ok else 'MISMATCH'}", flush=True)
            bad += 0 if ok else 1
        ext = oracle.render_extended(packed, 64, 48, 2, 2, camera=cam, frame_seed=1)
        for kw in ({}, {"kernel_sm": True}, {"kernel_v1": True}):
            st = ctx.render(64, 48, cam, mode=2, spp=2, max_bounces=2, frame_seed=1, **kw)
            g = ctx.read_rgb32f()
            ok = np.array_equal(np.isnan(g), np.isnan(ext["rgb"])) and np.array_equal(g[~np.isnan(g)].view(np.uint32), ext["rgb"][~np.isnan(ext["rgb"])].view(np.uint32))
            print(f"{name}: extended {kw} {'ok' if ok else 'MISMATCH'} ({st['rays']} segments)", flush=True)
            bad += 0 if ok else 1
print(f"nan_camera: {bad} mismatches")
sys.exit(1 if bad else 0)
