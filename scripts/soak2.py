"""Soak, second profile (development aid): many lights (up to the 32 of the wavefront pipeline and beyond: megakernel
fallback), long paths (up to 20 bounces: the every-8-iterations early exit), large sample counts split over batches
(RT_WF_TARGET_PATHS small), the three extended-mode implementations against each other.  usage: soak2.py <seconds>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_WF_TARGET_PATHS"] = "300000"  # many small batches
import numpy as np
from gpu_raytracer_amd import api, scenes
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
t_end = time.time() + budget
it = bad = 0
with api.Context() as ctx:
    while time.time() < t_end:
        nl = int(rng.choice([1, 3, 8, 31, 32, 33, 40]))
        sc = scenes.random_soup(int(rng.integers(50, 4000)), seed=int(rng.integers(1, 10**6)), size=float(rng.uniform(0.1, 0.9)), n_spheres=int(rng.integers(0, 4)), n_lights=nl)
        ctx.upload_scene(sc)
        for _ in range(4):
            w, h = int(rng.integers(8, 300)), int(rng.integers(8, 200))
            spp, bounces = int(rng.integers(1, 40)), int(rng.integers(0, 21))
            seed = int(rng.integers(0, 2**31))
            imgs = []
            for kw in ({}, {"kernel_sm": True}, {"kernel_v1": True}):
                st = ctx.render(w, h, sc.camera, mode=2, spp=spp, max_bounces=bounces, frame_seed=seed, **kw)
                imgs.append((ctx.read_rgb32f(), st["rays"], st["shadow_rays"]))
            a = imgs[0]
            ok = all(np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and a[1] == b[1] and a[2] == b[2] for b in imgs[1:])
            it += 1
            if not ok:
                bad += 1
                print(f"MISMATCH it={it} tris={sc.n_triangles} lights={nl} {w}x{h} spp={spp} bounces={bounces} seed={seed}", flush=True)
        if it % 40 == 0:
            print(f"{it} cases, {bad} mismatches, {t_end - time.time():.0f} s left", flush=True)
print(f"done: {it} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
