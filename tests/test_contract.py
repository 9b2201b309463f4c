"""Contract tests: struct layout and the reference's own host-side unit tests, restated.

Reference tests mirrored here (shared/src/lib.rs:1328-1456):
  test_triangle_bounding_box :1368, test_aabb_union :1385, test_aabb_center :1395,
  test_aabb_surface_area :1402, test_bvh_node_leaf :1410, test_bvh_node_internal :1422,
  test_push_constants_with_metadata :1434; plus K7 (pack_flags / pack_tile_size round trips).
"""
import os
import subprocess

import numpy as np
import pytest

from gpu_raytracer_amd import hostpack as H
from gpu_raytracer_amd import types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("compiler,lang", [("gcc", "c"), ("g++", "c++")])
def test_headers_compile_and_static_asserts_hold(compiler, lang):
    """include/rt_shared.h pins every size/offset with static asserts; both headers are valid C and C++."""
    src = '#include "rt_hip.h"\nint main(void){return 0;}\n'
    subprocess.run([compiler, "-x", lang, "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-"],
                   input=src.encode(), check=True)


def test_numpy_dtype_sizes_match_header():
    for name, size in T.EXPECTED_SIZES.items():
        assert getattr(T, name).itemsize == size
    assert T.PUSH_CONSTANTS.fields["packed_flags"][1] == 120
    assert T.PUSH_CONSTANTS.fields["frame_seed"][1] == 124
    assert T.PUSH_CONSTANTS.fields["metadata_offsets"][1] == 80
    assert T.MATERIAL.fields["ior_transmission_f16"][1] == 28
    assert T.LIGHT.fields["direction"][1] == 32
    assert T.BVH_NODE.fields["left_child"][1] == 32


def test_render_params_and_stats_match_c_layout():
    src = ('#include <stdio.h>\n#include "rt_hip.h"\nint main(void){printf("%zu %zu %zu %zu\\n", sizeof(rt_render_params),'
           ' offsetof(rt_render_params, width), sizeof(rt_stats), offsetof(rt_stats, kernel_ms));return 0;}\n')
    exe = "/tmp/_rt_layout_probe"
    subprocess.run(["gcc", "-x", "c", "-I", os.path.join(ROOT, "include"), "-o", exe, "-"], input=src.encode(), check=True)
    vals = list(map(int, subprocess.check_output([exe]).split()))
    assert vals == [T.RENDER_PARAMS.itemsize, T.RENDER_PARAMS.fields["width"][1], T.STATS.itemsize, T.STATS.fields["kernel_ms"][1]]


def _aabb_union(a, b):
    return np.minimum(a[0], b[0]), np.maximum(a[1], b[1])


def test_triangle_bounding_box():  # shared/src/lib.rs:1368-1382
    v = np.array([[0, 0, 0], [1, 0, 0], [0.5, 1, 0]], np.float32)
    assert v.min(0).tolist() == [0, 0, 0] and v.max(0).tolist() == [1, 1, 0]


def test_aabb_union_center_surface_area():  # :1385-1407
    mn, mx = _aabb_union((np.zeros(3), np.ones(3)), (np.full(3, 0.5), np.full(3, 2.0)))
    assert mn.tolist() == [0, 0, 0] and mx.tolist() == [2, 2, 2]
    lo, hi = np.zeros(3, np.float32), np.array([2, 4, 6], np.float32)
    assert ((lo + hi) * np.float32(0.5)).tolist() == [1, 2, 3]
    d = np.array([2, 3, 4], np.float32)
    assert 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]) == 52.0


def test_bvh_node_leaf_and_internal_sentinels(oracle_mod):  # :1410-1431, via the builder restatement
    verts = np.zeros(3, T.VERTEX)
    verts["position"] = [[0, 0, 0], [1, 0, 0], [0.5, 1, 0]]
    tris = np.array([(0, 1, 2, 0)], T.TRIANGLE)
    nodes, idx = oracle_mod.build_bvh(tris, verts)
    assert nodes[0]["left_child"] == 0xFFFFFFFF and nodes[0]["right_child"] == 0xFFFFFFFF
    assert nodes[0]["triangle_start"] == 0 and nodes[0]["triangle_count"] == 1
    verts2 = np.zeros(6, T.VERTEX)
    verts2["position"] = [[0, 0, 0], [1, 0, 0], [0.5, 1, 0], [2, 0, 0], [3, 0, 0], [2.5, 1, 0]]
    nodes, idx = oracle_mod.build_bvh(np.array([(0, 1, 2, 0), (3, 4, 5, 1)], T.TRIANGLE), verts2)
    root = nodes[0]
    assert root["left_child"] == 1 and root["right_child"] == 2  # pre-order: left = parent + 1
    assert root["triangle_start"] == 0 and root["triangle_count"] == 0


def test_push_constants_with_metadata():  # :1434-1455
    off = np.zeros((), T.SCENE_METADATA_OFFSETS)
    for k, v in zip(off.dtype.names, (0, 10, 500, 2, 600, 50, 1000, 100, 1500, 20)):
        off[k] = v
    pc = H.push_constants((1920.0, 1080.0), H.camera(), 20, 5, (0, 0), (128, 128), (15, 8), 100, off, 0)
    assert pc["metadata_offsets"]["bvh_nodes_count"] == 50
    assert pc["triangle_count"] == 20
    assert pc["metadata_offsets"]["spheres_count"] == 10 and pc["metadata_offsets"]["lights_count"] == 2
    assert int(pc["packed_flags"]) & 0xFF == 0
    # PushConstants::new: pack_flags(channel, 0, 4, 0)  (shared/src/lib.rs:1088)
    assert int(pc["packed_flags"]) == (4 << 16)
    assert pc.tobytes()[120:124] == np.uint32(4 << 16).tobytes()


def test_k7_pack_round_trips():
    for w, h in [(128, 128), (128, 56), (1, 65535), (70000, 3)]:
        p = H.pack_tile_size(w, h)
        assert (p & 0xFFFF, p >> 16) == (min(w, 65535), min(h, 65535))
    for ch, cur, mx, mode in [(0, 0, 4, 0), (2, 3, 7, 1), (255, 255, 255, 255), (256, 1, 1, 1)]:
        f = H.pack_flags(ch, cur, mx, mode)
        assert (f & 0xFF, (f >> 8) & 0xFF, (f >> 16) & 0xFF, (f >> 24) & 0xFF) == (ch & 0xFF, cur & 0xFF, mx & 0xFF, mode & 0xFF)


def test_tile_helper():  # TileHelper, shared/src/lib.rs:1187-1203
    assert H.tile_count(1920, 1080) == (15, 9)
    assert H.tile_count(3840, 2160) == (30, 17)
    assert H.tile_count(256, 256) == (2, 2)
    assert H.tiles_per_frame(135) == 4 and H.tiles_per_frame(510) == 7
    assert H.tiles_per_frame(16) == 16 and H.tiles_per_frame(17) == 2 and H.tiles_per_frame(5000) == 1 and H.tiles_per_frame(0) == 1


def test_material_f16_packing_matches_numpy_half(oracle_mod):
    """Material::new packs metallic lo16 / roughness hi16, ior lo16 / transmission hi16 (RNE)."""
    m = H.material_new((0.8, 0.3, 0.3), 1.0, 0.1, (0, 0, 0), 1.5, 0.9)
    mr, it = int(m["metallic_roughness_f16"]), int(m["ior_transmission_f16"])
    assert mr & 0xFFFF == 0x3C00  # 1.0
    assert mr >> 16 == int(np.float16(0.1).view(np.uint16))
    assert it & 0xFFFF == 0x3E00  # 1.5
    assert it >> 16 == int(np.float16(0.9).view(np.uint16))
    lib = oracle_mod.lib()
    rng = np.random.default_rng(7)
    vals = np.concatenate([rng.standard_normal(2000).astype(np.float32) * 10, rng.random(2000).astype(np.float32),
                           np.array([0, -0.0, 1e-8, 6e-8, 6.1e-5, 65504, 65520, 1e6, np.inf, -np.inf, 2.98e-8, 8.9407e-8], np.float32)])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16)
    for v, w in zip(vals, want):
        assert lib.oracle_f32_to_f16(float(v)) == int(w.view(np.uint16)), v
    allh = np.arange(0, 65536, 7, dtype=np.uint16)
    for h in allh:
        f = np.float32(lib.oracle_f16_to_f32(int(h)))
        w = np.uint16(h).view(np.float16).astype(np.float32)
        assert (np.isnan(f) and np.isnan(w)) or f.tobytes() == w.tobytes()
