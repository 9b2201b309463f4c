"""ctypes loader for the CPU oracle (liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; nothing in gpu_raytracer_amd/ does.  It takes the numpy arrays of a
gpu_raytracer_amd.scenes.Scene, packs them exactly as the reference host would
(binding 1 metadata buffer, 3 triangle buffers, materials, push constants) and
runs the restated kernel on them.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from gpu_raytracer_amd import hostpack as H
from gpu_raytracer_amd import types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


class Bindings(C.Structure):
    _fields_ = [("scene_metadata", C.c_void_p), ("scene_metadata_len", C.c_uint64),
                ("triangles", C.c_void_p * 3), ("triangles_len", C.c_uint64 * 3),
                ("materials", C.c_void_p), ("materials_len", C.c_uint64)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "node_visits", "tri_tests", "sphere_tests", "stack_drops", "oob_reads")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("rt_oracle.cpp", "rt_oracle_bvh.cpp", "rt_oracle.h")]
    stale = os.path.exists(_LIB_PATH) and any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_dispatch.restype = C.c_int
        _lib.oracle_render_frame.restype = C.c_int
        _lib.oracle_render_extended.restype = C.c_int
        _lib.oracle_render_extended_region.restype = C.c_int
        _lib.oracle_build_bvh.restype = C.c_int
        _lib.oracle_f32_to_f16.restype = C.c_uint16
        _lib.oracle_f32_to_f16.argtypes = [C.c_float]
        _lib.oracle_f16_to_f32.restype = C.c_float
        _lib.oracle_f16_to_f32.argtypes = [C.c_uint16]
    return _lib


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None and a.size else C.c_void_p(0)


def set_fast_traversal(on):
    """Baseline flavour (ii): ordered, distance-culled traversal instead of the reference's (same image)."""
    lib().oracle_set_fast_traversal(C.c_int(1 if on else 0))


def build_bvh(triangles, vertices, per_triangle=False):
    """BvhBuilder::build restatement -> (nodes, triangle_indices).  per_triangle: one triangle per leaf at every size
    (baseline flavour ii; the reference chunks above 100,000 triangles)."""
    l = lib()
    fn = l.oracle_build_bvh_per_triangle if per_triangle else l.oracle_build_bvh
    tris = np.ascontiguousarray(triangles)
    verts = np.ascontiguousarray(vertices)
    nn, ni = C.c_uint32(0), C.c_uint32(0)
    rc = fn(_ptr(tris), C.c_uint32(len(tris)), _ptr(verts), C.c_uint32(len(verts)),
            C.c_void_p(0), C.byref(nn), C.c_void_p(0), C.byref(ni))
    if rc != 0:
        raise ValueError(f"oracle_build_bvh failed: {rc}")
    nodes = np.zeros(nn.value, dtype=T.BVH_NODE)
    idx = np.zeros(ni.value, dtype=np.uint32)
    rc = fn(_ptr(tris), C.c_uint32(len(tris)), _ptr(verts), C.c_uint32(len(verts)),
            _ptr(nodes), C.byref(nn), _ptr(idx), C.byref(ni))
    if rc != 0:
        raise ValueError(f"oracle_build_bvh failed: {rc}")
    return nodes, idx


class PackedScene:
    """The reference's GPU-side view of a scene: bindings 1-5 + the offsets for the push constants."""

    def __init__(self, scene, use_bvh=True, materials_capacity=None, triangles_per_buffer=T.REF_TRIANGLES_PER_BUFFER,
                 bvh=None):
        self.scene = scene
        if bvh is not None:
            self.nodes, self.tri_indices = bvh
        elif use_bvh:
            self.nodes, self.tri_indices = build_bvh(scene.triangles, scene.vertices)
        else:  # bvh_nodes_count == 0 -> brute-force path (shader/src/lib.rs:192-211)
            self.nodes, self.tri_indices = np.zeros(0, T.BVH_NODE), np.zeros(0, np.uint32)
        self.metadata, self.offsets = H.pack_scene_metadata(scene.spheres, scene.lights, self.nodes,
                                                            self.tri_indices, scene.vertices)
        self.triangles_per_buffer = triangles_per_buffer
        self.tri_bufs = H.split_triangles(scene.triangles, triangles_per_buffer)
        mats = np.ascontiguousarray(scene.materials)
        if materials_capacity is not None and materials_capacity > len(mats):
            # the reference binds a buffer of capacity >= 64 elements (src/buffers.rs:74, 101-106); stale part is zero
            mats = np.concatenate([mats, np.zeros(materials_capacity - len(mats), dtype=T.MATERIAL)])
        self.materials = mats
        self.bindings = Bindings()
        self.bindings.scene_metadata = _ptr(self.metadata)
        self.bindings.scene_metadata_len = self.metadata.size
        for i in range(3):
            self.bindings.triangles[i] = self.tri_bufs[i].ctypes.data if self.tri_bufs[i].size else None
            self.bindings.triangles_len[i] = len(self.tri_bufs[i])
        self.bindings.materials = _ptr(self.materials)
        self.bindings.materials_len = len(self.materials)

    def push_constants(self, width, height, camera=None, channel=0, mode=0, cur_bounce=0, max_bounce=4, frame_seed=0,
                       tile_offset=(0, 0), tile_size=None, tile=T.TILE_SIZE):
        cam = self.scene.camera if camera is None else camera
        if tile_size is None:
            tile_size = (min(tile, width - tile_offset[0]), min(tile, height - tile_offset[1]))
        return H.push_constants((width, height), cam, len(self.scene.triangles), len(self.scene.materials),
                                tile_offset, tile_size, H.tile_count(width, height, tile), self.triangles_per_buffer,
                                self.offsets, channel, mode, cur_bounce, max_bounce, frame_seed)


def dispatch(packed, pc, image):
    """One main_cs dispatch (one tile, one channel) into `image` (H x W x 4 uint8)."""
    c = Counters()
    h, w = image.shape[:2]
    pcb = np.ascontiguousarray(pc)
    rc = lib().oracle_dispatch(C.byref(packed.bindings), _ptr(pcb), _ptr(image), C.c_uint32(w), C.c_uint32(h), C.byref(c))
    if rc != 0:
        raise RuntimeError(f"oracle_dispatch failed: {rc}")
    return c.as_dict()


def render_frame(packed, width, height, camera=None, mode=0, cur_bounce=0, max_bounce=4, frame_seed=0,
                 tile_size=T.TILE_SIZE, threads=None, faithful3=False, want_rgba8=True):
    """Whole frame through the tile x channel loop.  Returns dict(rgb, red, green, blue, combined, prim, t, counters)."""
    threads = threads or os.cpu_count() or 1
    pc = np.ascontiguousarray(packed.push_constants(width, height, camera, 0, mode, cur_bounce, max_bounce, frame_seed,
                                                    tile=tile_size))
    rgb = np.zeros((height, width, 3), np.float32)
    chans = [np.zeros((height, width, 4), np.uint8) for _ in range(3)] if want_rgba8 else [None] * 3
    prim = np.zeros((height, width), np.uint32)
    ts = np.zeros((height, width), np.float32)
    c = Counters()
    rc = lib().oracle_render_frame(C.byref(packed.bindings), _ptr(pc), C.c_uint32(tile_size), C.c_int(threads),
                                   C.c_int(1 if faithful3 else 0), _ptr(chans[0]), _ptr(chans[1]), _ptr(chans[2]),
                                   _ptr(rgb), _ptr(prim), _ptr(ts), C.byref(c))
    if rc != 0:
        raise RuntimeError(f"oracle_render_frame failed: {rc}")
    out = {"rgb": rgb, "prim": prim, "t": ts, "counters": c.as_dict()}
    if want_rgba8:
        out["red"], out["green"], out["blue"] = chans
        comb = np.zeros((height, width, 4), np.uint8)  # main_fs, shader/src/lib.rs:383-388
        comb[..., 0], comb[..., 1], comb[..., 2], comb[..., 3] = chans[0][..., 0], chans[1][..., 1], chans[2][..., 2], 255
        out["combined"] = comb
    return out


EXT_NO_SHADOWS = 2


def render_extended(packed, width, height, spp, max_bounces, camera=None, frame_seed=0, flags=0, threads=None, region=None):
    """Extended mode (the build's own path tracer, CPU statement).  Returns dict(rgb, segments, counters).
    region = (x0, y0, w, h): only that rectangle of the width x height frame (rgb has the rectangle's shape)."""
    threads = threads or os.cpu_count() or 1
    pc = np.ascontiguousarray(packed.push_constants(width, height, camera, 0, 1, 0, max_bounces & 0xFF, frame_seed))
    x0, y0, rw, rh = region if region is not None else (0, 0, width, height)
    rgb = np.zeros((rh, rw, 3), np.float32)
    seg = (C.c_uint64 * 4)()
    c = Counters()
    rc = lib().oracle_render_extended_region(C.byref(packed.bindings), _ptr(pc), C.c_uint32(spp), C.c_uint32(max_bounces),
                                             C.c_uint32(flags), C.c_int(threads), C.c_uint32(x0), C.c_uint32(y0), C.c_uint32(rw),
                                             C.c_uint32(rh), _ptr(rgb), seg, C.byref(c))
    if rc != 0:
        raise RuntimeError(f"oracle_render_extended failed: {rc}")
    return {"rgb": rgb, "counters": c.as_dict(),
            "segments": {"camera": int(seg[0]), "continuation": int(seg[1]), "shadow": int(seg[2]), "roulette": int(seg[3])}}
