"""ctypes binding of librt_hip.so (the C ABI in include/rt_hip.h).

This is the stub a host-language binding would look like (see INTEGRATION.md for the
Rust `extern "C"` version).  There is no fallback: if the library is missing, or no HIP
device is present, the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIB", os.path.join(_HERE, "librt_hip.so"))  # RT_HIP_LIB: development override (kernel variants)

RT_OK = 0
ERRORS = {-1: "RT_ERR_BAD_ARG", -2: "RT_ERR_OOM", -3: "RT_ERR_HIP", -4: "RT_ERR_NOT_UPLOADED", -5: "RT_ERR_INTERNAL"}
MODE_LEGACY, MODE_WAVEFRONT, MODE_EXTENDED = 0, 1, 2
FLAG_COUNTERS = 1
FLAG_NO_SHADOWS = 2
FLAG_KERNEL_V1 = 4
FLAG_KERNEL_SM = 8
FLAG_NO_SHADOW_GRID = 16
FLAG_KERNEL_PIPELINE = 32
FLAG_NO_BEAMS = 64
FLAG_STAGE_TIMES = 128
PREPARE_SHADOW_GRIDS = 1
PREPARE_QUALITY_TREE = 2
EXTENDED_AVAILABLE = True

# every symbol include/rt_hip.h declares
ABI_SYMBOLS = [
    "rt_create", "rt_upload_scene", "rt_upload_scene_packed", "rt_upload_textures", "rt_prepare", "rt_render", "rt_dispatch_tile",
    "rt_read_rgb32f", "rt_read_rgba8_channels", "rt_read_rgba8_combined", "rt_read_hits",
    "rt_get_stats", "rt_last_error", "rt_destroy", "rt_version",
]


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


_lib = None


def load():
    """Load librt_hip.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m gpu_raytracer_amd.build` "
                              "(there is no CPU fallback for the hot path)")
        lib = C.CDLL(LIB_PATH)
        lib.rt_last_error.restype = C.c_char_p
        lib.rt_last_error.argtypes = [C.c_void_p]
        lib.rt_version.restype = C.c_char_p
        lib.rt_destroy.restype = None
        lib.rt_destroy.argtypes = [C.c_void_p]
        for name in ABI_SYMBOLS:
            if "RT_HIP_LIB" in os.environ and not hasattr(lib, name):
                continue  # an older build under A/B (development only): entry points added since are simply absent
            fn = getattr(lib, name)
            if name not in ("rt_last_error", "rt_version", "rt_destroy"):
                fn.restype = C.c_int
        _lib = lib
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None and a.size else C.c_void_p(0)


class Context:
    """One rt_ctx.  Mirrors the life cycle RenderState/BufferManager/ComputeRenderer have in the reference."""

    def __init__(self, device_ids=(0,)):
        self.lib = load()
        self._h = C.c_void_p(0)
        ids = (C.c_int * len(device_ids))(*device_ids)
        rc = self.lib.rt_create(C.byref(self._h), ids, C.c_int(len(device_ids)))
        if rc != RT_OK:
            raise RtError(rc, self.lib.rt_last_error(None).decode())
        self.width = self.height = 0

    def _check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, self.lib.rt_last_error(self._h).decode())

    def close(self):
        if self._h:
            self.lib.rt_destroy(self._h)
            self._h = C.c_void_p(0)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- uploads ---------------------------------------------------------------------
    def upload_scene(self, scene, ref_nodes=None, ref_tri_indices=None):
        keep = [np.ascontiguousarray(a) for a in (scene.spheres, scene.lights, scene.vertices, scene.triangles, scene.materials)]
        sp, li, ve, tr, ma = keep
        rn = np.ascontiguousarray(ref_nodes) if ref_nodes is not None else None
        ri = np.ascontiguousarray(ref_tri_indices, dtype=np.uint32) if ref_tri_indices is not None else None
        self._check(self.lib.rt_upload_scene(
            self._h, _p(sp), C.c_uint32(len(sp)), _p(li), C.c_uint32(len(li)), _p(ve), C.c_uint32(len(ve)),
            _p(tr), C.c_uint32(len(tr)), _p(ma), C.c_uint32(len(ma)),
            _p(rn), C.c_uint32(0 if rn is None else len(rn)), _p(ri), C.c_uint32(0 if ri is None else len(ri))))

    def upload_scene_packed(self, metadata, offsets, tri_bufs, triangles_per_buffer, materials):
        md = np.ascontiguousarray(metadata, dtype=np.uint32)
        off = np.ascontiguousarray(offsets)
        bufs = [np.ascontiguousarray(b) for b in tri_bufs]
        ptrs = (C.c_void_p * 3)(*[b.ctypes.data if b.size else None for b in bufs])
        counts = (C.c_uint32 * 3)(*[len(b) for b in bufs])
        ma = np.ascontiguousarray(materials)
        self._check(self.lib.rt_upload_scene_packed(
            self._h, _p(md), C.c_size_t(md.size), _p(off), ptrs, counts, C.c_uint32(triangles_per_buffer),
            _p(ma), C.c_uint32(len(ma))))

    def upload_textures(self, textures, texture_data):
        """Bindings 6-7 (TextureInfo[] + texture bytes): validated and recorded, never sampled (as in the reference)."""
        ti = np.ascontiguousarray(textures, dtype=T.TEXTURE_INFO)
        td = np.ascontiguousarray(texture_data, dtype=np.uint8)
        self._check(self.lib.rt_upload_textures(self._h, _p(ti), C.c_uint32(len(ti)), _p(td), C.c_size_t(td.size)))

    def prepare(self, what=PREPARE_SHADOW_GRIDS):
        """rt_prepare: build ahead of time what the first extended-mode frame would otherwise build (PREPARE_SHADOW_GRIDS: the light grids), and /
        or rebuild the tree with the host builder for a scene that stays (PREPARE_QUALITY_TREE)."""
        self._check(self.lib.rt_prepare(self._h, C.c_uint32(what)))

    # -- rendering -------------------------------------------------------------------
    def render(self, width, height, camera, mode=MODE_LEGACY, spp=1, max_bounces=4, frame_seed=0, tile_size=0,
               tile_rank=0, tile_world=1, counters=False, no_shadows=False, kernel_v1=False, kernel_sm=False, no_shadow_grid=False, kernel_pipeline=False, no_beams=False, stage_times=False):
        p = np.zeros((), dtype=T.RENDER_PARAMS)
        p["camera"] = camera
        p["width"], p["height"], p["spp"], p["max_bounces"], p["mode"] = width, height, spp, max_bounces, mode
        p["frame_seed"], p["tile_size"], p["tile_rank"], p["tile_world"] = frame_seed, tile_size, tile_rank, tile_world
        p["flags"] = (FLAG_COUNTERS if counters else 0) | (FLAG_NO_SHADOWS if no_shadows else 0) | (FLAG_KERNEL_V1 if kernel_v1 else 0) | (FLAG_KERNEL_SM if kernel_sm else 0) | (FLAG_NO_SHADOW_GRID if no_shadow_grid else 0) | (FLAG_KERNEL_PIPELINE if kernel_pipeline else 0) | (FLAG_NO_BEAMS if no_beams else 0) | (FLAG_STAGE_TIMES if stage_times else 0)
        self._check(self.lib.rt_render(self._h, _p(p)))
        self.width, self.height = width, height
        return self.stats()

    def dispatch_tile(self, pc):
        pcb = np.ascontiguousarray(pc)
        self._check(self.lib.rt_dispatch_tile(self._h, _p(pcb)))
        res = pcb["resolution"].reshape(-1)
        self.width, self.height = int(res[0]), int(res[1])

    # -- read-back -------------------------------------------------------------------
    def read_rgb32f(self):
        out = np.zeros((self.height, self.width, 3), np.float32)
        self._check(self.lib.rt_read_rgb32f(self._h, _p(out), C.c_size_t(out.size)))
        return out

    def read_rgba8_channels(self):
        outs = [np.zeros((self.height, self.width, 4), np.uint8) for _ in range(3)]
        self._check(self.lib.rt_read_rgba8_channels(self._h, _p(outs[0]), _p(outs[1]), _p(outs[2]), C.c_size_t(outs[0].size)))
        return outs

    def read_rgba8_combined(self):
        out = np.zeros((self.height, self.width, 4), np.uint8)
        self._check(self.lib.rt_read_rgba8_combined(self._h, _p(out), C.c_size_t(out.size)))
        return out

    def read_hits(self):
        prim = np.zeros((self.height, self.width), np.uint32)
        t = np.zeros((self.height, self.width), np.float32)
        self._check(self.lib.rt_read_hits(self._h, _p(prim), _p(t), C.c_size_t(prim.size)))
        return prim, t

    def debug_beams(self, n_blocks):
        """Development aid: per owned 8x8 pixel block the length of its camera-beam triangle list (0xFFFFFFFF: none)."""
        out = np.zeros(n_blocks, np.uint32)
        m = self.lib.rt_debug_beams(self._h, _p(out), C.c_uint32(n_blocks))
        if m < 0:
            raise RtError(m, "rt_debug_beams")
        return out[:m]

    def debug_stage_times(self):
        """(sum of the k_wf_shadow_grid launch durations in ms, launches) of the last frame rendered with stage_times=True."""
        out = (C.c_double * 2)()
        self._check(self.lib.rt_debug_stage_times(self._h, out))
        return float(out[0]), int(out[1])

    def debug_counters(self):
        out = (C.c_ulonglong * 8)()
        self._check(self.lib.rt_debug_counters(self._h, out))
        names = ("transition_passes", "transition_lanes", "node_iters", "node_lanes", "leaf_iters", "leaf_lanes", "cycles_transition", "cycles_traversal")
        return dict(zip(names, [int(v) for v in out]))

    def debug_shadow_grid(self, light=None):
        """Development aid: the light grids (csrc/shadow_grid.h) of the first device.  light=None: totals and the last counted frame's use."""
        out = (C.c_ulonglong * 8)()
        self._check(self.lib.rt_debug_shadow_grid(self._h, C.c_uint32(0xFFFFFFFF if light is None else light), out))
        names = ("lights_with_grid", "entries", "bytes", "segments_answered", "entries_read") if light is None else ("kind", "res", "entries", "near", "longest", "heavy_cells", "filled_cells")
        return dict(zip(names, [int(v) for v in out]))

    def debug_check_bvh(self):
        """Development aid: validate the tree this context holds on its first device the way the kernels decode it.
        Returns dict(failures, nodes, leaves, depth, real_depth, placed_once, method) - method 0 host SAH, 1 host PLOC, 2 device build."""
        out = (C.c_uint32 * 8)()
        fails = self.lib.rt_debug_check_bvh(self._h, out)
        names = ("nodes", "leaves", "depth", "real_depth", "placed_once", "method", "nodes_hash", "tris_hash")
        d = dict(zip(names, [int(v) for v in out]))
        d["failures"] = int(fails)
        return d

    def stats(self):
        st = np.zeros((), dtype=T.STATS)
        self._check(self.lib.rt_get_stats(self._h, _p(st)))
        return {k: st[k].item() for k in st.dtype.names if not k.startswith("_")}


def version():
    return load().rt_version().decode()
