"""Soak for the camera beams (development aid): random cameras - anywhere in and around the scene, any direction, fov 2...170 degrees, tilted
`up` vectors - random resolutions, tile sizes, sample counts; every frame rendered with the beams and with RT_FLAG_NO_BEAMS must carry the
same bits and count the same segments.  usage: soak_beams.py <seconds> [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpu_raytracer_amd import api, scenes, hostpack as H
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31337)
scene_list = [scenes.sponza_like(), scenes.random_soup(40000, seed=6, size=0.4, n_spheres=3, n_lights=3), scenes.cornell12(), scenes.bistro_like(n_triangles=1000000)]
t_end = time.time() + budget
cases = bad = lists = blocks = 0
with api.Context() as ctx:
    while time.time() < t_end:
        sc = scene_list[rng.integers(len(scene_list))]
        ctx.upload_scene(sc)
        p = np.asarray(sc.vertices["position"], np.float64)
        lo, hi = p.min(0), p.max(0)
        for _ in range(10):
            kind = rng.integers(0, 4)
            pos = lo + (hi - lo) * rng.uniform(-0.3, 1.3, 3) if kind else np.asarray(sc.camera["position"], np.float64) + rng.normal(size=3) * 0.2
            if kind == 3: pos = p[rng.integers(len(p))] + rng.normal(size=3) * 1e-3  # a hair off a vertex
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            up = rng.normal(size=3) if rng.uniform() < 0.5 else np.array([0.0, 1.0, 0.0])
            cam = H.camera(position=tuple(pos), direction=tuple(d), up=tuple(up), fov=float(rng.choice([2.0, 20.0, 45.0, 60.0, 90.0, 130.0, 170.0])))
            w, h = int(rng.integers(8, 900)), int(rng.integers(8, 500))
            kw = dict(mode=2, spp=int(rng.integers(1, 9)), max_bounces=int(rng.integers(0, 4)), frame_seed=int(rng.integers(0, 2**31)),
                      tile_size=int(rng.choice([0, 16, 24, 32, 50, 128])), kernel_pipeline=True)
            a = ctx.render(w, h, cam, **kw); ia = ctx.read_rgb32f().view(np.uint32).copy()
            c = ctx.debug_beams(1 << 22)
            b = ctx.render(w, h, cam, no_beams=True, **kw); ib = ctx.read_rgb32f().view(np.uint32)
            ok = np.array_equal(ia, ib) and (a["primary_rays"], a["continuation_rays"], a["shadow_rays"]) == (b["primary_rays"], b["continuation_rays"], b["shadow_rays"])
            cases += 1; blocks += len(c); lists += int((c != 0xFFFFFFFF).sum())
            if not ok:
                bad += 1
                print("MISMATCH", sc.name, w, h, kw, cam, int((ia != ib).any(-1).sum()), "pixels", flush=True)
        print(f"{cases} cases, {bad} mismatches, {lists}/{blocks} blocks with a list, {t_end - time.time():.0f} s left", flush=True)
print(f"done: {cases} cases, {bad} mismatches, {lists}/{blocks} blocks with a list")
