"""Soak, third profile (development aid): the per-light triangle lists of the shadow stage (csrc/shadow_grid.h) against the BVH on random
scenes - soups of every density and triangle size, lights thrown anywhere (inside the geometry, ON vertices / edges / faces of triangles,
far away), directional lights along axes and along triangle planes, spot lights, spheres; each case rendered with the lists and with
RT_FLAG_NO_SHADOW_GRID, bits compared.  usage: soak3.py <seconds>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_raytracer_amd import api, scenes
from gpu_raytracer_amd import hostpack as H
from gpu_raytracer_amd import types as T
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
t_end = time.time() + budget
it = bad = with_grid = answered = shadow = 0
with api.Context() as ctx:
    while time.time() < t_end:
        n = int(rng.choice([12, 200, 3000, 20000, 120000]))
        sc = scenes.random_soup(n, seed=int(rng.integers(1, 10**6)), extent=float(rng.uniform(1.0, 8.0)), size=float(rng.choice([0.02, 0.1, 0.6, 3.0])),
                                n_spheres=int(rng.integers(0, 3)), n_lights=0)
        pos = np.asarray(sc.vertices["position"], np.float64)
        tri = np.stack([sc.triangles["v0_index"], sc.triangles["v1_index"], sc.triangles["v2_index"]], 1)
        lo, hi = pos.min(0), pos.max(0)
        lights = []
        for _ in range(int(rng.integers(1, 9))):
            kind = int(rng.integers(0, 8))
            t = tri[int(rng.integers(0, len(tri)))]
            a, b, c = pos[t[0]], pos[t[1]], pos[t[2]]
            if kind == 0: p = lo + (hi - lo) * rng.uniform(0, 1, 3)                       # anywhere inside the box
            elif kind == 1: p = a                                                         # ON a vertex
            elif kind == 2: p = a + (b - a) * rng.uniform()                               # ON an edge
            elif kind == 3: u, v = sorted(rng.uniform(0, 1, 2)); p = a * u + b * (v - u) + c * (1 - v)  # inside a triangle
            elif kind == 4: p = (lo + hi) / 2 + (hi - lo) * rng.uniform(2, 40) * rng.choice([-1, 1], 3)  # far outside
            elif kind == 5: p = a + np.cross(b - a, c - a) * rng.uniform(-1e-3, 1e-3)     # a hair off a triangle's plane
            if kind <= 5:
                lights.append(H.light_point(tuple(p), (1, 1, 1), float(rng.uniform(0.5, 30))) if rng.uniform() < 0.7 else
                              H.light_spot(tuple(p), tuple(rng.normal(size=3)), (1, 1, 1), float(rng.uniform(1, 30)), 50.0, 0.3, 0.8))
            elif kind == 6: lights.append(H.light_directional(tuple(np.eye(3)[int(rng.integers(0, 3))] * rng.choice([-1, 1])), (1, 1, 1), 0.7))  # along an axis
            else: lights.append(H.light_directional(tuple(b - a if np.linalg.norm(b - a) > 0 else (0, -1, 0)), (1, 1, 1), 0.7))         # along a triangle's edge: in its plane
        sc = scenes.Scene(sc.name, sc.spheres, np.array(lights, dtype=T.LIGHT), sc.vertices, sc.triangles, sc.materials, sc.camera)
        ctx.upload_scene(sc)
        ctx.prepare()
        g = ctx.debug_shadow_grid()
        with_grid += g["lights_with_grid"]
        for _ in range(3):
            w, h = int(rng.integers(16, 400)), int(rng.integers(16, 300))
            spp, bounces, seed = int(rng.integers(1, 12)), int(rng.integers(0, 6)), int(rng.integers(0, 2**31))
            cam = sc.camera
            if rng.uniform() < 0.3:  # from inside the soup
                cam = H.camera(position=tuple(lo + (hi - lo) * rng.uniform(0, 1, 3)), direction=tuple(rng.normal(size=3)))
            st0 = ctx.render(w, h, cam, mode=2, spp=spp, max_bounces=bounces, frame_seed=seed, no_shadow_grid=True)
            ref = ctx.read_rgb32f().copy()
            st1 = ctx.render(w, h, cam, mode=2, spp=spp, max_bounces=bounces, frame_seed=seed, counters=True)
            ok = np.array_equal(ref.view(np.uint32), ctx.read_rgb32f().view(np.uint32)) and st0["rays"] == st1["rays"]
            u = ctx.debug_shadow_grid()
            answered += u["segments_answered"]; shadow += st1["shadow_rays"]
            it += 1
            if not ok:
                bad += 1
                print(f"MISMATCH it={it} tris={n} lights={[(int(l['light_type']), [float(x) for x in l['position']], [float(x) for x in l['direction']]) for l in sc.lights]} {w}x{h} spp={spp} bounces={bounces} seed={seed}", flush=True)
        if it % 60 == 0:
            print(f"{it} cases, {bad} mismatches, {with_grid} grids, {answered / max(1, shadow):.2f} of {shadow / 1e6:.0f} M shadow segments answered by lists, {t_end - time.time():.0f} s left", flush=True)
print(f"done: {it} cases, {bad} mismatches, {with_grid} grids, {answered / max(1, shadow):.2f} of {shadow / 1e6:.0f} M shadow segments answered by lists")
sys.exit(1 if bad else 0)
