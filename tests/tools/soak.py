"""Soak test (development aid): many renders with random shapes / sample counts / partitions; every render is repeated
and must be bit-identical, tile partitions must reassemble to the full frame, and small cases are compared with the CPU
statement.  Looks for rare races in the queue machinery.  usage: soak.py <seconds> [devices]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, oracle
from gpu_raytracer_amd import api, scenes
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
n_dev = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # > 1: one context over GPU 0 listed that many times
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 12345)
scene_list = [scenes.sponza_like(), scenes.random_soup(5000, seed=4, size=0.3, n_spheres=2, n_lights=5), scenes.cornell12(), scenes.default_scene()]
t_end = time.time() + budget
it = bad = 0
with api.Context((0,) * n_dev) as ctx:
    while time.time() < t_end:
        sc = scene_list[rng.integers(len(scene_list))]
        ctx.upload_scene(sc)
        for _ in range(6):
            w, h = int(rng.integers(1, 700)), int(rng.integers(1, 400))
            spp, bounces = int(rng.integers(1, 12)), int(rng.integers(0, 7))
            seed = int(rng.integers(0, 2**31))
            kw = dict(mode=2, spp=spp, max_bounces=bounces, frame_seed=seed)
            st = ctx.render(w, h, sc.camera, **kw); a = ctx.read_rgb32f()
            st2 = ctx.render(w, h, sc.camera, **kw); b = ctx.read_rgb32f()
            ok = np.array_equal(a.view(np.uint32), b.view(np.uint32)) and st["rays"] == st2["rays"]
            world = int(rng.integers(2, 5)); ts = int(rng.choice([16, 32, 64, 128]))
            full = np.zeros_like(a); rays = 0
            for r in range(world):
                s3 = ctx.render(w, h, sc.camera, tile_size=ts, tile_rank=r, tile_world=world, **kw); part = ctx.read_rgb32f(); rays += s3["rays"]
                ty, tx = np.meshgrid(np.arange(h) // ts, np.arange(w) // ts, indexing="ij")
                own = ((ty * ((w + ts - 1) // ts) + tx) % world) == r
                full[own] = part[own]
            ok = ok and np.array_equal(full.view(np.uint32), a.view(np.uint32)) and rays == st["rays"]
            if sc.n_triangles <= 5000 and w * h * spp <= 60000:
                ref = oracle.render_extended(oracle.PackedScene(sc, use_bvh=False), w, h, spp, bounces, frame_seed=seed)
                ok = ok and np.array_equal(ref["rgb"].view(np.uint32), a.view(np.uint32))
            # reference mode too
            ctx.render(w, h, sc.camera, mode=1); p1, t1 = ctx.read_hits(); ctx.render(w, h, sc.camera, mode=1); p2, t2 = ctx.read_hits()
            ok = ok and np.array_equal(p1, p2) and np.array_equal(t1.view(np.uint32), t2.view(np.uint32))
            it += 1
            if not ok:
                bad += 1
                print(f"MISMATCH it={it} scene={sc.name} {w}x{h} spp={spp} bounces={bounces} seed={seed} world={world} ts={ts}", flush=True)
        if it % 60 == 0:
            print(f"{it} cases, {bad} mismatches, {t_end - time.time():.0f} s left", flush=True)
print(f"done: {it} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
