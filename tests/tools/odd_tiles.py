"""Checker: tile sizes that are not multiples of the 8x8 pixel block (and tiny / huge ones) - every partition of the frame
must reassemble to the whole-frame image, in the reference mode against the oracle and in the extended mode against the
128-pixel tiling.  usage: odd_tiles.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import oracle
from gpu_raytracer_amd import api, scenes

scene = scenes.random_soup(3000, seed=9, n_spheres=2, n_lights=3)
w, h = 333, 211
bad = 0
with api.Context() as ctx:
    ctx.upload_scene(scene)
    ref = oracle.render_frame(oracle.PackedScene(scene, use_bvh=False), w, h, mode=1)
    st0 = ctx.render(w, h, scene.camera, mode=2, spp=3, max_bounces=2, frame_seed=4)
    ext = ctx.read_rgb32f()
    for ts in (1, 3, 7, 9, 20, 50, 100, 127, 129, 200, 333, 1000, 4096):
        for world in (1, 2, 3):
            acc1, acc2, rays1, rays2 = np.zeros_like(ext), np.zeros_like(ext), 0, 0
            ty, tx = np.meshgrid(np.arange(h) // ts, np.arange(w) // ts, indexing="ij")
            for r in range(world):
                own = ((ty * ((w + ts - 1) // ts) + tx) % world) == r
                s1 = ctx.render(w, h, scene.camera, mode=1, tile_size=ts, tile_rank=r, tile_world=world)
                acc1[own] = ctx.read_rgb32f()[own]
                s2 = ctx.render(w, h, scene.camera, mode=2, spp=3, max_bounces=2, frame_seed=4, tile_size=ts, tile_rank=r, tile_world=world)
                acc2[own] = ctx.read_rgb32f()[own]
                rays1 += s1["rays"]
                rays2 += s2["rays"]
            ok = (np.array_equal(acc1.view(np.uint32), ref["rgb"].view(np.uint32)) and rays1 == w * h and
                  np.array_equal(acc2.view(np.uint32), ext.view(np.uint32)) and rays2 == st0["rays"])
            print(f"tile {ts:5d} world {world}: {'ok' if ok else 'MISMATCH'}", flush=True)
            bad += 0 if ok else 1
print(f"odd_tiles: {bad} mismatches")
sys.exit(1 if bad else 0)
