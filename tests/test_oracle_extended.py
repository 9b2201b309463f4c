"""CPU checks of the extended-mode specification (oracle/rt_oracle.cpp ExtKernel).

Extended mode has no reference implementation (parity unpinned); these tests anchor it to the pinned
reference semantics and check its internal consistency:
  * spp = 1, max_bounces = 0, no shadows  ==  mode 1, bit for bit (same ray, same shading)
  * determinism; frame_seed changes the samples
  * more samples converge (N spp vs 4N spp)
  * white furnace: closed white diffuse box with no lights -> radiance stays bounded by the sky term
  * shadow rays only remove light
"""
import numpy as np
import pytest

from gpu_raytracer_amd import hostpack as H
from gpu_raytracer_amd import scenes
from gpu_raytracer_amd import types as T


@pytest.mark.parametrize("name", ["default", "cornell12", "single_triangle", "empty"])
def test_extended_reduces_to_mode1(oracle_mod, name):
    s = scenes.SCENES[name]()
    p = oracle_mod.PackedScene(s)
    a = oracle_mod.render_frame(p, 80, 48, mode=1)
    b = oracle_mod.render_extended(p, 80, 48, 1, 0, flags=oracle_mod.EXT_NO_SHADOWS)
    np.testing.assert_array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
    assert b["segments"] == {"camera": 80 * 48, "continuation": 0, "shadow": 0, "roulette": 0}


def test_extended_is_deterministic_and_seeded(oracle_mod):
    p = oracle_mod.PackedScene(scenes.cornell12())
    a = oracle_mod.render_extended(p, 48, 32, 4, 3, threads=1)
    b = oracle_mod.render_extended(p, 48, 32, 4, 3, threads=4)
    np.testing.assert_array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
    assert a["segments"] == b["segments"]
    c = oracle_mod.render_extended(p, 48, 32, 4, 3, frame_seed=12345)
    assert not np.array_equal(a["rgb"], c["rgb"])
    assert np.isfinite(a["rgb"]).all() and (a["rgb"] >= 0).all()


def test_extended_converges_with_more_samples(oracle_mod):
    p = oracle_mod.PackedScene(scenes.cornell12())
    ref = oracle_mod.render_extended(p, 40, 30, 256, 3)["rgb"]
    e4 = np.abs(oracle_mod.render_extended(p, 40, 30, 4, 3)["rgb"] - ref).mean()
    e16 = np.abs(oracle_mod.render_extended(p, 40, 30, 16, 3)["rgb"] - ref).mean()
    e64 = np.abs(oracle_mod.render_extended(p, 40, 30, 64, 3)["rgb"] - ref).mean()
    assert e64 < e16 < e4 and e64 < 0.6 * e4


def test_segment_accounting(oracle_mod):
    s = scenes.cornell12()
    p = oracle_mod.PackedScene(s)
    r = oracle_mod.render_extended(p, 32, 32, 8, 4)
    seg = r["segments"]
    assert seg["camera"] == 32 * 32 * 8
    assert 0 < seg["continuation"] <= 4 * seg["camera"]
    assert 0 < seg["shadow"] <= len(s.lights) * (seg["camera"] + seg["continuation"])
    assert r["counters"]["rays"] == seg["camera"] + seg["continuation"] + seg["shadow"]
    r0 = oracle_mod.render_extended(p, 32, 32, 8, 4, flags=oracle_mod.EXT_NO_SHADOWS)
    assert r0["segments"]["shadow"] == 0


def test_shadows_only_remove_light(oracle_mod):
    """An occluder between the light and a floor darkens it; nothing gets brighter (1 spp, no bounces:
    same rays, so the comparison is exact per pixel)."""
    floor = [((-3, -1, 0), (3, -1, 0), (3, -1, -6), 0), ((-3, -1, 0), (3, -1, -6), (-3, -1, -6), 0)]
    blocker = [((-0.7, 0.5, -2.3), (0.7, 0.5, -2.3), (0.7, 0.5, -3.7), 0), ((-0.7, 0.5, -2.3), (0.7, 0.5, -3.7), (-0.7, 0.5, -3.7), 0)]
    s = scenes.single_triangle()
    s.vertices, s.triangles = H.legacy_to_indexed(floor + blocker)
    s.lights = np.array([H.light_point((0.0, 2.0, -3.0), (1, 1, 1), 6.0)], dtype=T.LIGHT)
    p = oracle_mod.PackedScene(s)
    lit = oracle_mod.render_extended(p, 96, 64, 1, 0, flags=oracle_mod.EXT_NO_SHADOWS)["rgb"]
    shadowed = oracle_mod.render_extended(p, 96, 64, 1, 0)["rgb"]
    assert (shadowed <= lit).all()
    assert (shadowed < lit - 1e-3).mean() > 0.01  # a visible shadow exists


def test_white_furnace_energy_bound(oracle_mod):
    """Closed box, albedo 1 diffuse walls, no lights, no emission: every path either escapes nowhere (closed)
    or terminates; the only radiance sources are the terminal ambient term (0.1 * albedo) — so the image is
    bounded by 0.1 per channel and, with many bounces, approaches it."""
    c = scenes.cornell12()
    quad = [((-1, -1, 1), (-1, 1, 1), (1, 1, 1), 0), ((-1, -1, 1), (1, 1, 1), (1, -1, 1), 0)]  # close the front, facing -z
    legacy = []
    v = c.vertices["position"]
    for t in c.triangles[:10]:
        legacy.append((tuple(v[t["v0_index"]]), tuple(v[t["v1_index"]]), tuple(v[t["v2_index"]]), 0))
    c.vertices, c.triangles = H.legacy_to_indexed(legacy + quad)
    c.materials = np.array([H.material_diffuse((1.0, 1.0, 1.0))], dtype=T.MATERIAL)
    c.lights = np.zeros(0, T.LIGHT)
    cam = H.camera((0.0, 0.0, 0.5), (0, 0, -1), (0, 1, 0), 60.0)
    p = oracle_mod.PackedScene(c)
    r = oracle_mod.render_extended(p, 24, 24, 32, 6, camera=cam)
    assert r["rgb"].max() <= 0.1 * 20.0 + 1e-4  # russian roulette boosts single samples by at most 1/0.05
    np.testing.assert_allclose(r["rgb"].mean(), 0.1, rtol=0.15)


def test_fast_traversal_flavour_gives_the_same_images(oracle_mod):
    """bench.py times the CPU twice: the reference's traversal (no culling, 32+-triangle chunks above 100k triangles) and
    a decent one (ordered, culled, one triangle per leaf).  The second is not the reference's algorithm, so it must at
    least be shown to produce the same frames.  Comparator: the brute-force path (shader/src/lib.rs:272-296), whose
    visiting order is the triangle order - the reference-format BVH of a <= 100k-triangle scene has an unpinned
    topology and resolves equal-t ties (cornell12's shared edges) by it."""
    from gpu_raytracer_amd import scenes
    for scene, w, h in ((scenes.random_soup(3000, seed=9, size=0.4, n_spheres=2, n_lights=3), 48, 32), (scenes.cornell12(), 40, 40)):
        brute = oracle_mod.PackedScene(scene, use_bvh=False)
        fast = oracle_mod.PackedScene(scene, bvh=oracle_mod.build_bvh(scene.triangles, scene.vertices, per_triangle=True))
        a = oracle_mod.render_extended(brute, w, h, 3, 2, frame_seed=5)
        b1 = oracle_mod.render_frame(brute, w, h, mode=1)
        oracle_mod.set_fast_traversal(True)
        try:
            c = oracle_mod.render_extended(fast, w, h, 3, 2, frame_seed=5)
            d = oracle_mod.render_frame(fast, w, h, mode=1)
        finally:
            oracle_mod.set_fast_traversal(False)
        np.testing.assert_array_equal(a["rgb"].view(np.uint32), c["rgb"].view(np.uint32))
        assert a["segments"] == c["segments"]
        np.testing.assert_array_equal(b1["rgb"].view(np.uint32), d["rgb"].view(np.uint32))
        np.testing.assert_array_equal(b1["prim"], d["prim"])
        assert d["counters"]["tri_tests"] < b1["counters"]["tri_tests"]


def test_extended_region_equals_crop_of_the_frame(oracle_mod):
    """oracle_render_extended_region: a rectangle of a frame is that frame's crop (a pixel's samples depend only on its
    coordinates in the full frame), segment counts add up over a partition of the frame."""
    p = oracle_mod.PackedScene(scenes.cornell12())
    w, h, spp, bounces = 61, 37, 3, 2
    full = oracle_mod.render_extended(p, w, h, spp, bounces, frame_seed=9)
    total = {k: 0 for k in full["segments"]}
    for (x0, y0, rw, rh) in ((0, 0, 20, 37), (20, 0, 41, 10), (20, 10, 41, 27)):
        part = oracle_mod.render_extended(p, w, h, spp, bounces, frame_seed=9, region=(x0, y0, rw, rh))
        np.testing.assert_array_equal(part["rgb"].view(np.uint32), full["rgb"][y0:y0 + rh, x0:x0 + rw].view(np.uint32))
        for k in total:
            total[k] += part["segments"][k]
    assert total == full["segments"]
    with pytest.raises(RuntimeError):
        oracle_mod.render_extended(p, w, h, spp, bounces, region=(50, 0, 20, 5))


def test_reference_bvh_walk_and_brute_force_disagree_at_a_grazed_edge(oracle_mod):
    """Found in round 2 when a second tree builder changed one pixel of the headline frame: at pixel (844, 563) of the
    sponza-like 1920x1080 64-spp 4-bounce frame one segment grazes a triangle that Möller–Trumbore
    (intersection.rs:91-138) accepts but whose leaf box the reference's slab test (intersection.rs:151-164, no margin)
    rejects, so the reference's BVH walk and its own brute-force path (lib.rs:192-211) give different colours.  The HIP
    path's box filter is conservative by construction and reproduces the brute-force answer with every tree
    (tests/test_gpu_device_build.py); this test pins the two oracle values so the finding stays reproducible."""
    sp = scenes.sponza_like()
    region = (844, 563, 1, 1)
    walk = oracle_mod.render_extended(oracle_mod.PackedScene(sp), 1920, 1080, 64, 4, region=region)["rgb"][0, 0]
    brute = oracle_mod.render_extended(oracle_mod.PackedScene(sp, use_bvh=False), 1920, 1080, 64, 4, region=region)["rgb"][0, 0]
    assert walk.view(np.uint32).tolist() == [1066384201, 1057341577, 1044572349]
    assert brute.view(np.uint32).tolist() == [1066366301, 1057316370, 1044521191]
    assert 2e-3 < np.abs(walk - brute).max() < 3e-3  # one sample of 64 changed: ONE outlier pixel of 2,073,600 (the stated gate allows 0.1 % of them)
