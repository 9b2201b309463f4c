"""Round-3 probe 2 (development aid): would splitting the large triangles pay?  The sponza-like scene with every triangle whose longest
edge exceeds a threshold subdivided 1 -> 4 (midpoints; same surfaces, other triangle ids, so only times and counts are comparable):
node visits / triangle tests per closest-hit segment and the closest-hit-only frame time, against the scene as it is."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_raytracer_amd import api, scenes, types as T
import dataclasses

def subdivide(sc, thr, rounds=4):
    v = sc.vertices["position"].astype(np.float32)
    tri = np.stack([sc.triangles["v0_index"], sc.triangles["v1_index"], sc.triangles["v2_index"]], 1).astype(np.int64)
    mat = sc.triangles["material_id"].copy()
    for _ in range(rounds):
        p = v[tri]  # (n,3,3)
        e = np.stack([np.linalg.norm(p[:, 1] - p[:, 0], axis=1), np.linalg.norm(p[:, 2] - p[:, 1], axis=1), np.linalg.norm(p[:, 0] - p[:, 2], axis=1)], 1)
        big = e.max(1) > thr
        if not big.any(): break
        pb = p[big]
        m01, m12, m20 = (pb[:, 0] + pb[:, 1]) * 0.5, (pb[:, 1] + pb[:, 2]) * 0.5, (pb[:, 2] + pb[:, 0]) * 0.5
        base = len(v)
        nb = len(pb)
        v = np.concatenate([v, m01, m12, m20]).astype(np.float32)
        i0, i1, i2 = tri[big, 0], tri[big, 1], tri[big, 2]
        a, b, c = base + np.arange(nb), base + nb + np.arange(nb), base + 2 * nb + np.arange(nb)
        new = np.concatenate([np.stack([i0, a, c], 1), np.stack([a, i1, b], 1), np.stack([c, b, i2], 1), np.stack([a, b, c], 1)])
        tri = np.concatenate([tri[~big], new])
        mat = np.concatenate([mat[~big], np.tile(mat[big], 4)])
    vertices = np.zeros(len(v), dtype=T.VERTEX); vertices["position"] = v
    triangles = np.zeros(len(tri), dtype=T.TRIANGLE)
    triangles["v0_index"], triangles["v1_index"], triangles["v2_index"], triangles["material_id"] = tri[:, 0], tri[:, 1], tri[:, 2], mat
    return dataclasses.replace(sc, vertices=vertices, triangles=triangles, name=f"{sc.name}_sub{thr}")

sp = scenes.sponza_like()
p = sp.vertices["position"][np.stack([sp.triangles["v0_index"], sp.triangles["v1_index"], sp.triangles["v2_index"]], 1)]
e = np.stack([np.linalg.norm(p[:, 1] - p[:, 0], axis=1), np.linalg.norm(p[:, 2] - p[:, 1], axis=1), np.linalg.norm(p[:, 0] - p[:, 2], axis=1)], 1).max(1)
area = 0.5 * np.linalg.norm(np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), axis=1)
print("longest edge percentiles", np.percentile(e, [50, 90, 99, 99.9, 100]).round(3), "area share of tris with edge>1:", (area[e > 1].sum() / area.sum()).round(3), "count", int((e > 1).sum()), flush=True)
spp = 16
for thr in (None, 2.0, 1.0, 0.5, 0.25):
    sc = sp if thr is None else subdivide(sp, thr)
    for method in ("2", "0"):
        os.environ["RT_BUILD_METHOD"] = method
        with api.Context() as ctx:
            ctx.upload_scene(sc)
            b = ctx.debug_check_bvh()
            ms = min(ctx.render(1920, 1080, sc.camera, mode=2, spp=spp, max_bounces=4, no_shadows=True)["kernel_ms"] for _ in range(3))
            st = ctx.render(1920, 1080, sc.camera, mode=2, spp=spp, max_bounces=4, counters=True, no_shadows=True)
            segs = st["rays"]
            print(f"thr={thr} method={method} tris={sc.n_triangles} nodes={b['nodes']} leaves={b['leaves']} depth={b['depth']} | closest-only {ms:.2f} ms | visits/seg={st['node_visits']/segs:.2f} tris/seg={st['tri_tests']/segs:.2f}", flush=True)
os.environ.pop("RT_BUILD_METHOD", None)
