"""The per-light triangle lists of the shadow stage (csrc/shadow_grid.h) against the BVH: the same shadow segments must get the same
answer from both, so a frame rendered with the grids (the default) and one rendered with RT_FLAG_NO_SHADOW_GRID carry the same bits -
on ordinary scenes, on the configurations the lists' construction has special cases for (triangles at the light: the near list; a
light on a triangle's plane or edge-on to it; lights inside and far outside the geometry; axis-aligned directional lights; spot
lights; cube-face seams; non-finite vertices; coordinates too large for a grid), and against the CPU oracle.  The counters say that
the lists did answer the segments (a test that silently fell back to the BVH would prove nothing)."""
import os

import numpy as np
import pytest

from gpu_raytracer_amd import hostpack as H
from gpu_raytracer_amd import scenes
from gpu_raytracer_amd import types as T
from test_gpu_adversarial import _grid, _scene

pytestmark = pytest.mark.gpu


def _same_frame(ctx, scene, w, h, spp, bounces, min_answered=0.5, camera=None, expect_grids=None):
    cam = scene.camera if camera is None else camera
    ctx.upload_scene(scene)
    assert ctx.debug_shadow_grid()["lights_with_grid"] == 0 and ctx.stats()["grid_bytes"] == 0  # uploads build no grids (round 3) ...
    ctx.prepare()                                                                               # ... rt_prepare, or the first frame that needs them, does
    info = ctx.debug_shadow_grid()
    assert ctx.stats()["grid_bytes"] == info["bytes"]
    if expect_grids is not None:
        assert info["lights_with_grid"] == expect_grids, (scene.name, info, [ctx.debug_shadow_grid(i) for i in range(len(scene.lights))])
    st = ctx.render(w, h, cam, mode=2, spp=spp, max_bounces=bounces, frame_seed=5, no_shadow_grid=True)
    ref = ctx.read_rgb32f().copy()
    st2 = ctx.render(w, h, cam, mode=2, spp=spp, max_bounces=bounces, frame_seed=5)
    got = ctx.read_rgb32f()
    assert (st["primary_rays"], st["continuation_rays"], st["shadow_rays"]) == (st2["primary_rays"], st2["continuation_rays"], st2["shadow_rays"])
    diff = np.flatnonzero((ref.view(np.uint32) != got.view(np.uint32)).reshape(h * w, -1).any(axis=1))
    assert diff.size == 0, f"{scene.name}: {diff.size} pixels differ between the light grids and the BVH, first at {divmod(int(diff[0]), w)[::-1]}"
    ctx.render(w, h, cam, mode=2, spp=spp, max_bounces=bounces, frame_seed=5, counters=True)
    use = ctx.debug_shadow_grid()
    if info["lights_with_grid"] and st["shadow_rays"]:
        assert use["segments_answered"] >= min_answered * st["shadow_rays"] * info["lights_with_grid"] / max(1, len(scene.lights)) * 0.5, (scene.name, use, st)
    return info, use, st


def test_default_scene_and_cornell(gpu_ctx):
    info, use, st = _same_frame(gpu_ctx, scenes.default_scene(), 320, 200, 4, 3)
    info, use, st = _same_frame(gpu_ctx, scenes.cornell12(), 256, 256, 8, 4, expect_grids=1)
    assert use["segments_answered"] == st["shadow_rays"]  # twelve triangles: no cell is heavy, every segment is answered by its list


def test_sponza_like_every_light_has_a_grid(gpu_ctx):
    sc = scenes.sponza_like()
    info, use, st = _same_frame(gpu_ctx, sc, 640, 360, 4, 4, expect_grids=len(sc.lights))
    assert use["segments_answered"] > 0.9 * st["shadow_rays"]
    assert use["entries_read"] < 4 * use["segments_answered"]  # short lists are the point (2.2 per segment on the headline frame)


def test_cluttered_scene_lists_cut_to_what_a_walk_reads(gpu_ctx):
    """bistro-like: foliage seen end-on makes long lists.  Until the end of round 3 the build refused such grids; now the lists are
    stored cut to the entries a walk can look at (the nearest RT_SG_SORTED_PREFIX of a list, nothing of a list over `heavy`), the lights
    keep their grids, and the lists still answer like the BVH, with a part of the segments handed on."""
    sc = scenes.bistro_like(n_triangles=1200000)
    info, use, st = _same_frame(gpu_ctx, sc, 320, 180, 2, 3, min_answered=0.2, expect_grids=len(sc.lights))
    per_light = [gpu_ctx.debug_shadow_grid(i) for i in range(len(sc.lights))]
    assert max(g["longest"] for g in per_light) > 128                                  # there are lists over `heavy` ...
    block_bytes = sum((6 if g["kind"] == 1 else 1) * g["res"] ** 2 * 128 for g in per_light)
    assert info["bytes"] - block_bytes < 0.9 * 48 * info["entries"]                    # ... and the grids hold fewer entries than were rasterised
    assert 0 < st["shadow_rays"] - use["segments_answered"] < 0.6 * st["shadow_rays"]
    os.environ["RT_SHADOW_GRID_MEAN"] = "1e9"  # (the acceptance rule out of the way: the same lists)
    try:
        _same_frame(gpu_ctx, sc, 320, 180, 2, 3, min_answered=0.2, expect_grids=len(sc.lights))
    finally:
        del os.environ["RT_SHADOW_GRID_MEAN"]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_soups_with_every_kind_of_light(gpu_ctx, seed):
    sc = scenes.random_soup(30000, seed=seed, n_spheres=3, n_lights=6)  # point, directional and spot lights, spheres in the way
    _same_frame(gpu_ctx, sc, 256, 256, 4, 4)


def _box_room(lo, hi):
    """Twelve triangles, normals inward."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    c = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
    quads = [(0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (4, 5, 1, 0), (3, 2, 6, 7)]
    tris = []
    for a, b, cc, d in quads:
        tris += [(c[a], c[b], c[cc]), (c[a], c[cc], c[d])]
    return np.array(tris, np.float32)


def test_lights_in_awkward_places(gpu_ctx):
    """Lights ON geometry (the near list), on a triangle's plane, exactly at a box corner direction (cube-face seams: the room's corners
    lie on the diagonals of a light at its centre), far outside, and axis-aligned directional lights over axis-aligned geometry."""
    room = _box_room((-2, -2, -6), (2, 2, -1))
    floor, ids = _grid(12, 12, z=-5.5, size=3.0)
    blockers = np.array([[(-0.5, -0.5, -3.0), (0.5, -0.5, -3.0), (0.0, 0.5, -3.0)], [(-1.0, 0.2, -4.0), (0.3, 0.1, -4.0), (-0.4, 1.0, -4.2)]], np.float32)
    tris = np.concatenate([room, floor, blockers])
    mats = [0] * len(room) + list(ids) + [1, 3]
    cam = H.camera(position=(0.0, 0.0, -1.2), direction=(0.0, -0.1, -1.0))
    lights = np.array([
        H.light_point((0.0, 0.0, -3.5), (1, 1, 1), 2.0),               # the room's centre: its corners sit on the cube map's diagonals
        H.light_point((0.0, 0.0, -3.0), (1, 0.8, 0.6), 1.0),           # ON the first blocker's plane, inside it
        H.light_point((2.0, 2.0, -1.0), (0.5, 0.7, 1.0), 3.0),         # exactly at a corner of the room (touches three walls)
        H.light_point((40.0, 55.0, 30.0), (1, 1, 1), 900.0),           # far outside
        H.light_directional((0.0, -1.0, 0.0), (1, 1, 1), 0.5),         # axis-aligned
        H.light_directional((0.0, 0.0, -1.0), (1, 1, 1), 0.5),         # along the camera axis, parallel to four walls
        H.light_spot((0.0, 1.9, -3.0), (0.0, -1.0, 0.0), (1, 1, 1), 3.0, 20.0, 0.3, 0.6),
    ], dtype=T.LIGHT)
    sc = _scene("awkward lights", tris, mats, lights=lights, camera=cam)
    info, use, st = _same_frame(gpu_ctx, sc, 256, 256, 8, 4, camera=cam)
    near = [gpu_ctx.debug_shadow_grid(i)["near"] for i in range(len(lights))]
    assert near[1] >= 1 and near[2] >= 3, near  # the triangles those lights sit on are in their near lists


def test_non_finite_vertices_and_huge_coordinates(gpu_ctx, oracle_mod):
    """Triangles with NaN / infinite vertices are never hit and never rasterised; a scene whose coordinates are too large for the
    segments' offsets to mean anything gets no grid at all.  Both render like the oracle."""
    floor, ids = _grid(8, 8, z=-4.0, size=4.0)
    bad = np.array([[(np.nan, 0, -3), (1, 0, -3), (0, 1, -3)], [(0, 0, -2.5), (np.inf, 0, -2.5), (0, 1, -2.5)], [(0, 0, -2), (1, 0, -2), (0, -np.inf, -2)]], np.float32)
    blocker = np.array([[(-0.6, -0.6, -3.2), (0.6, -0.6, -3.2), (0.0, 0.7, -3.2)]], np.float32)
    sc = _scene("non-finite", np.concatenate([floor, bad, blocker]), list(ids) + [0, 1, 2, 3])
    _same_frame(gpu_ctx, sc, 128, 128, 4, 3)
    packed = oracle_mod.PackedScene(sc, use_bvh=False)
    ext = oracle_mod.render_extended(packed, 64, 64, 2, 2, camera=sc.camera, frame_seed=3)
    gpu_ctx.render(64, 64, sc.camera, mode=2, spp=2, max_bounces=2, frame_seed=3)
    np.testing.assert_array_equal(gpu_ctx.read_rgb32f().view(np.uint32), ext["rgb"].view(np.uint32))

    far = np.concatenate([floor, blocker]) + np.float32(2.0e7)
    cam = H.camera(position=(2.0e7, 2.0e7, 2.0e7))
    lights = np.array([H.light_point((2.0e7 + 2, 2.0e7 + 3, 2.0e7 + 4), (1, 1, 1), 1.5), H.light_directional((0.3, -1.0, -0.2), (0.6, 0.7, 1.0), 0.8)], dtype=T.LIGHT)
    sc = _scene("2e7 away", far, list(ids) + [3], lights=lights, camera=cam)
    _same_frame(gpu_ctx, sc, 96, 96, 2, 2, camera=cam, expect_grids=0)


def test_a_ground_plane_of_two_triangles_under_a_light(gpu_ctx):
    """Two triangles that fill a whole cube face (a hundred thousand cells each) plus clutter above them."""
    ground = np.array([[(-50, -2, -50), (50, -2, -50), (50, -2, 50)], [(-50, -2, -50), (50, -2, 50), (-50, -2, 50)]], np.float32)
    soup = scenes.random_soup(20000, seed=9, n_lights=0)
    pos = np.asarray(soup.vertices["position"])
    idx = np.stack([soup.triangles["v0_index"], soup.triangles["v1_index"], soup.triangles["v2_index"]], 1)
    tris = np.concatenate([ground, pos[idx]])
    lights = np.array([H.light_point((0.0, 6.0, -3.0), (1, 1, 1), 30.0), H.light_directional((0.2, -1.0, 0.1), (1, 1, 1), 0.7)], dtype=T.LIGHT)
    sc = _scene("ground plane", tris, [0, 0] + [int(m) for m in soup.triangles["material_id"] % 4], lights=lights)
    _same_frame(gpu_ctx, sc, 256, 256, 4, 3, min_answered=0.0)  # (the clutter is dense: most of its cells are left to the BVH)


def test_full_size_frames_agree(gpu_ctx):
    """The headline scene's frame at 4 spp (125 M segments): lists and BVH walk give equal bits, and the lists answer more than 95 % of its
    shadow segments.  (The bistro-like scene with forced grids is in test_cluttered_scene_refuses_its_grids_and_forced_grids_still_agree; the 64-spp frame checksum is pinned in
    tests/test_gpu_baseline_configs.py.)"""
    sc = scenes.sponza_like()
    info, use, st = _same_frame(gpu_ctx, sc, 1920, 1080, 4, 4, expect_grids=len(sc.lights))
    assert use["segments_answered"] > 0.95 * st["shadow_rays"]


def test_one_lane_and_two_lanes_give_the_same_frame(gpu_ctx):
    """The pipeline runs its batches alternately on two streams (default) or all on one (RT_WF_LANES=1): same pixel sums, added in the
    same order."""
    sc = scenes.sponza_like()
    gpu_ctx.upload_scene(sc)
    frames = {}
    for lanes in ("2", "1", "2"):
        os.environ["RT_WF_LANES"] = lanes
        try:
            for spp in (1, 2, 5):
                st = gpu_ctx.render(480, 270, sc.camera, mode=2, spp=spp, max_bounces=4, frame_seed=3)
                img = gpu_ctx.read_rgb32f().copy()
                key = (spp,)
                if key in frames:
                    np.testing.assert_array_equal(frames[key][0].view(np.uint32), img.view(np.uint32), err_msg=f"lanes {lanes} spp {spp}")
                    assert frames[key][1] == (st["rays"], st["shadow_rays"])
                else:
                    frames[key] = (img, (st["rays"], st["shadow_rays"]))
        finally:
            del os.environ["RT_WF_LANES"]


def test_slivers_and_needles(gpu_ctx):
    """Long thin triangles (aspect 1e4 and more) and needles between the lights and a floor: the closest-point arithmetic behind a
    triangle's key is least exact there, and a key must never exceed the triangle's true distance from the light."""
    rng = np.random.default_rng(31)
    n = 3000
    a = rng.uniform(-3, 3, (n, 3)) + np.array([0, 0, -5.0])
    long_dir = rng.normal(size=(n, 3))
    long_dir /= np.linalg.norm(long_dir, axis=1, keepdims=True)
    side = np.cross(long_dir, rng.normal(size=(n, 3)))
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    length = rng.choice([0.5, 3.0, 12.0], n)[:, None]
    width = rng.choice([1e-2, 1e-3, 1e-4], n)[:, None]
    slivers = np.stack([a, a + long_dir * length, a + long_dir * length * rng.uniform(0, 1, (n, 1)) + side * width], 1)
    floor, ids = _grid(6, 6, z=-9.0, size=14.0)
    tris = np.concatenate([floor, slivers.astype(np.float32)])
    lights = np.array([H.light_point((0.0, 0.0, -5.0), (1, 1, 1), 8.0), H.light_point((2.5, 2.0, -2.0), (1, 0.9, 0.8), 8.0),
                       H.light_point(tuple(a[0] + long_dir[0] * 0.3), (1, 1, 1), 3.0),  # ON a needle
                       H.light_directional((0.1, -0.2, -1.0), (1, 1, 1), 0.8)], dtype=T.LIGHT)
    sc = _scene("slivers", tris, list(ids) + [int(i) % 4 for i in range(n)], lights=lights)
    _same_frame(gpu_ctx, sc, 320, 240, 8, 4, min_answered=0.0)


def test_a_fan_of_needles_takes_the_lists_through_every_regime_of_the_walk(gpu_ctx, oracle_mod):
    """300 thin blades fanned out around the axis under a point light, over a floor: a cell on the axis has all 300 in its list, cells
    further out ever fewer, and a segment from the floor misses nearly all of them - so the frame has lists decided inside the cell's own
    block (<= 2 entries), lists walked on by parked segments (the wave's LDS list, several passes), walks that reach the limit
    (RT_WF_GRID_WALK) and cells over `heavy`, both handed on to the BVH.  Same bits as the BVH walk, and as the oracle."""
    n = 300
    ang = np.linspace(0.0, 2 * np.pi, n, endpoint=False)
    z = -1.0 + 2.5 * (np.arange(n) * 37 % n) / n  # heights shuffled, so that distance order and angle order differ
    r0, r1, half = 0.0, 3.5, 0.004
    c, s = np.cos(ang), np.sin(ang)
    a = np.stack([r0 * c, r0 * s, z], 1)
    b = np.stack([r1 * c - half * s, r1 * s + half * c, z], 1)
    d = np.stack([r1 * c + half * s, r1 * s - half * c, z], 1)
    blades = np.stack([a, b, d], 1).astype(np.float32)
    floor, ids = _grid(12, 12, z=-3.0, size=9.0)
    tris = np.concatenate([floor, blades])
    lights = np.array([H.light_point((0.0, 0.0, 4.0), (1, 1, 1), 30.0), H.light_directional((0.0, 0.0, -1.0), (1, 1, 1), 0.6)], dtype=T.LIGHT)
    cam = H.camera(position=(0.0, -6.5, 3.0), direction=(0.0, 6.5, -6.0), up=(0.0, 0.0, 1.0), fov=55.0)
    sc = _scene("fan", tris, list(ids) + [int(i) % 4 for i in range(n)], lights=lights, camera=cam)
    os.environ["RT_SHADOW_GRID_MEAN"] = "1e9"  # (the long lists on the axis must not make the build refuse the grids)
    try:
        info, use, st = _same_frame(gpu_ctx, sc, 480, 320, 8, 3, min_answered=0.3, camera=cam, expect_grids=2)
        per_light = [gpu_ctx.debug_shadow_grid(i) for i in range(2)]
        assert max(g["longest"] for g in per_light) > 128, per_light                           # cells over `heavy` exist ...
        assert 0 < st["shadow_rays"] - use["segments_answered"] < 0.5 * st["shadow_rays"]       # ... and segments were handed on, and most were not
        assert use["entries_read"] > 3.0 * use["segments_answered"]                             # walks well beyond a cell's block
        packed = oracle_mod.PackedScene(sc, use_bvh=False)
        ext = oracle_mod.render_extended(packed, 96, 64, 2, 2, camera=cam, frame_seed=9)
        gpu_ctx.render(96, 64, cam, mode=2, spp=2, max_bounces=2, frame_seed=9)
        np.testing.assert_array_equal(gpu_ctx.read_rgb32f().view(np.uint32), ext["rgb"].view(np.uint32))
    finally:
        del os.environ["RT_SHADOW_GRID_MEAN"]
