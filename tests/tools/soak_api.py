"""Soak test of the boundary (development aid): random sequences of C-ABI calls - uploads, whole-frame renders in the
three modes, explicit tile x channel dispatches in random order (some skipped), size changes, reads in random order -
against a CPU model of the three channel textures driven by the oracle.  Looks for ordering bugs between the
asynchronous dispatches, target reallocation, the read-back epilogues and scene replacement.
usage: soak_api.py <seconds> [devices]   devices > 1: one context over GPU 0 listed that many times (the several-device path)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import oracle
from gpu_raytracer_amd import api, scenes

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
n_dev = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 2024)
scene_list = [scenes.default_scene(), scenes.cornell12(), scenes.random_soup(300, seed=9, size=0.5, n_spheres=2, n_lights=2), scenes.empty_scene()]
packs = [oracle.PackedScene(s, use_bvh=False) for s in scene_list]  # brute force: ties resolve to the lowest index
t_end = time.time() + budget
ops = bad = 0
t_print = time.time()


def eq(a, b, label=""):
    same = a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))
    if not same:
        n = int((a.view(np.uint8) != b.view(np.uint8)).sum()) if a.shape == b.shape else -1
        print(f"  differs: {label} shapes {a.shape} {b.shape} bytes {n}", flush=True)
    return same


with api.Context((0,) * n_dev) as ctx:
    si = 0
    ctx.upload_scene(scene_list[si])
    size, model = None, None  # model: [red, green, blue] H x W x 4 uint8, or None when unknown (after an extended render)
    while time.time() < t_end:
        r = rng.random()
        ok = True
        what = ""
        if r < 0.12:  # replace the scene: targets and their content stay
            si = int(rng.integers(len(scene_list)))
            ctx.upload_scene(scene_list[si])
            what = f"upload {scene_list[si].name}"
        elif r < 0.45:  # whole frame, reference semantics
            w, h = (int(rng.integers(1, 300)), int(rng.integers(1, 200))) if size is None or rng.random() < 0.5 else size
            mode = int(rng.integers(0, 2))
            ref = oracle.render_frame(packs[si], w, h, mode=mode)
            ctx.render(w, h, scene_list[si].camera, mode=mode)
            size, model = (w, h), [ref["red"].copy(), ref["green"].copy(), ref["blue"].copy()]
            what = f"render {w}x{h} mode {mode}"
            for k in rng.permutation(4)[: int(rng.integers(1, 5))]:
                if k == 0:
                    p, t = ctx.read_hits()
                    ok = ok and eq(p, ref["prim"], "prim") and eq(t, ref["t"], "t")
                elif k == 1:
                    ok = ok and eq(ctx.read_rgb32f(), ref["rgb"], "rgb")
                elif k == 2:
                    ok = ok and eq(ctx.read_rgba8_combined(), ref["combined"], "combined")
                else:
                    ok = ok and all(eq(a, b, "channel") for a, b in zip(ctx.read_rgba8_channels(), model))
        elif r < 0.55:  # extended mode, small
            w, h = (int(rng.integers(1, 120)), int(rng.integers(1, 80))) if size is None or rng.random() < 0.5 else size
            if w * h <= 12000:
                spp, bounces, seed = int(rng.integers(1, 5)), int(rng.integers(0, 4)), int(rng.integers(0, 2**31))
                ext = oracle.render_extended(packs[si], w, h, spp, bounces, frame_seed=seed)
                kw = [{}, {"kernel_sm": True}, {"kernel_v1": True}][int(rng.integers(3))]
                ctx.render(w, h, scene_list[si].camera, mode=2, spp=spp, max_bounces=bounces, frame_seed=seed, **kw)
                ok = eq(ctx.read_rgb32f(), ext["rgb"], "ext rgb")
                size, model = (w, h), None
                what = f"extended {w}x{h} {spp} spp {bounces} bounces {kw}"
        else:  # a run of explicit dispatches, the reference's process_color_channel
            new_size = False
            if size is None or rng.random() < 0.3:
                cand = (int(rng.integers(1, 400)), int(rng.integers(1, 300)))
                if cand != size:
                    size, model, new_size = cand, None, True
            w, h = size
            if model is None and new_size:
                model = [np.zeros((h, w, 4), np.uint8) for _ in range(3)]  # fresh targets read as zero
            tx, ty = (w + 127) // 128, (h + 127) // 128
            mode = int(rng.integers(0, 2))
            jobs = [(t, c) for t in range(tx * ty) for c in range(3) if rng.random() < 0.8]
            for j in rng.permutation(len(jobs)):
                t, c = jobs[j]
                pc = packs[si].push_constants(w, h, channel=c, mode=mode, tile_offset=((t % tx) * 128, (t // tx) * 128))
                ctx.dispatch_tile(pc)
                if model is not None:
                    oracle.dispatch(packs[si], pc, model[c])
            what = f"{len(jobs)} dispatches {w}x{h} mode {mode} new_size {new_size}"
            if model is not None and jobs:
                if rng.random() < 0.5:
                    ok = all(eq(a, b, "channel") for a, b in zip(ctx.read_rgba8_channels(), model))
                else:
                    comb = np.stack([model[0][..., 0], model[1][..., 1], model[2][..., 2], np.full((h, w), 255, np.uint8)], -1)
                    ok = eq(ctx.read_rgba8_combined(), comb, "comb")
            elif jobs:
                ctx.read_rgba8_combined()  # still exercises the wait
        ops += 1
        if not ok:
            bad += 1
            print("MISMATCH:", what, flush=True)
        if time.time() - t_print > 30:
            t_print = time.time()
            print(f"{ops} operations, {bad} mismatches", flush=True)
print(f"soak_api: {ops} operations, {bad} mismatches")
sys.exit(1 if bad else 0)
