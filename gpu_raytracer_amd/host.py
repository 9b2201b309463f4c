"""ctypes access to the C++ host mirror (include/rt_host.h): the reference-named host functions
(Material::new, PushConstants::new, TileHelper, SceneBuilder::build_default_scene, BvhBuilder::build,
BufferManager packing, ComputeRenderer::run_compute loop) implemented in csrc/host/raytracer_host.hpp."""
import ctypes as C

import numpy as np

from . import api
from . import types as T

HOST_SYMBOLS = ["rt_host_material_new", "rt_host_light_new", "rt_host_push_constants_new", "rt_host_tile_count",
                "rt_host_tiles_per_frame", "rt_host_default_scene", "rt_host_bvh_build", "rt_host_pack_scene_metadata",
                "rt_host_render_progressive", "rt_host_branchless_float_if_nonnan", "rt_host_branchless_float_if",
                "rt_host_branchless_u32_if", "rt_host_bvh_triangle", "rt_host_triangle_aabb"]


def _lib():
    lib = api.load()
    lib.rt_host_tiles_per_frame.restype = C.c_uint32
    for n in ("rt_host_default_scene", "rt_host_bvh_build", "rt_host_pack_scene_metadata", "rt_host_render_progressive"):
        getattr(lib, n).restype = C.c_int
    for n in ("rt_host_material_new", "rt_host_light_new", "rt_host_push_constants_new", "rt_host_tile_count"):
        getattr(lib, n).restype = None
    return lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None and a.size else C.c_void_p(0)


def material_new(albedo, metallic, roughness, emission, ior, transmission):
    out = np.zeros(1, T.MATERIAL)
    _lib().rt_host_material_new(_p(out), _f3(albedo), C.c_float(metallic), C.c_float(roughness), _f3(emission), C.c_float(ior),
                                C.c_float(transmission))
    return out[0]


def light_new(light_type, position=(0, 0, 0), direction=(0, 0, 0), color=(1, 1, 1), intensity=1.0, rng=np.inf, inner=0.0, outer=0.0):
    out = np.zeros(1, T.LIGHT)
    _lib().rt_host_light_new(_p(out), C.c_uint32(light_type), _f3(position), _f3(direction), _f3(color), C.c_float(intensity),
                             C.c_float(rng), C.c_float(inner), C.c_float(outer))
    return out[0]


def push_constants_new(resolution, camera, triangle_count, material_count, tile_offset, tile_size, total_tiles, triangles_per_buffer,
                       offsets, channel, mode=0, cur_bounce=0, max_bounce=4, frame_seed=0):
    out = np.zeros(1, T.PUSH_CONSTANTS)
    cam = np.ascontiguousarray(camera)
    off = np.ascontiguousarray(offsets)
    u2 = lambda v: (C.c_uint32 * 2)(*[int(x) for x in v])
    _lib().rt_host_push_constants_new(_p(out), (C.c_float * 2)(*[float(x) for x in resolution]), _p(cam), C.c_uint32(triangle_count),
                                      C.c_uint32(material_count), u2(tile_offset), u2(tile_size), u2(total_tiles),
                                      C.c_uint32(triangles_per_buffer), _p(off), C.c_uint32(channel), C.c_uint32(mode),
                                      C.c_uint32(cur_bounce), C.c_uint32(max_bounce), C.c_uint32(frame_seed))
    return out[0]


def branchless_float_if_nonnan(condition, if_true, if_false):
    lib = _lib()
    lib.rt_host_branchless_float_if_nonnan.restype = C.c_float
    return lib.rt_host_branchless_float_if_nonnan(C.c_int(bool(condition)), C.c_float(if_true), C.c_float(if_false))


def branchless_float_if(condition, if_true, if_false):
    """branchless_float_if!(cond, a, b) -> (value, valid)  (shared/src/lib.rs:1295-1313)"""
    lib = _lib()
    lib.rt_host_branchless_float_if.restype = C.c_float
    valid = C.c_int(0)
    v = lib.rt_host_branchless_float_if(C.c_int(bool(condition)), C.c_float(if_true), C.c_float(if_false), C.byref(valid))
    return v, bool(valid.value)


def branchless_u32_if(condition, if_true, if_false):
    lib = _lib()
    lib.rt_host_branchless_u32_if.restype = C.c_uint32
    return lib.rt_host_branchless_u32_if(C.c_int(bool(condition)), C.c_uint32(if_true), C.c_uint32(if_false))


def bvh_triangle(triangle, vertices):
    """BvhTriangle::centroid + BvhTriangleWithVertices::aabb (src/bvh.rs:27-55) -> (centroid[3], aabb record)"""
    lib = _lib()
    lib.rt_host_bvh_triangle.restype = C.c_int
    tri = np.ascontiguousarray(triangle, dtype=T.TRIANGLE).reshape(())
    ve = np.ascontiguousarray(vertices, dtype=T.VERTEX)
    c = np.zeros(3, np.float32)
    box = np.zeros((), dtype=T.AABB)
    rc = lib.rt_host_bvh_triangle(C.c_void_p(tri.ctypes.data), _p(ve), C.c_uint32(len(ve)), _p(c), C.c_void_p(box.ctypes.data))
    if rc != 0:
        raise ValueError(f"rt_host_bvh_triangle failed: {rc}")
    return c, box


def triangle_aabb(triangle, vertices):
    """BvhBuilder::triangle_aabb (src/bvh.rs:272-275) -> aabb record"""
    lib = _lib()
    lib.rt_host_triangle_aabb.restype = C.c_int
    tri = np.ascontiguousarray(triangle, dtype=T.TRIANGLE).reshape(())
    ve = np.ascontiguousarray(vertices, dtype=T.VERTEX)
    box = np.zeros((), dtype=T.AABB)
    rc = lib.rt_host_triangle_aabb(C.c_void_p(tri.ctypes.data), _p(ve), C.c_uint32(len(ve)), C.c_void_p(box.ctypes.data))
    if rc != 0:
        raise ValueError(f"rt_host_triangle_aabb failed: {rc}")
    return box


def tile_count(width, height, tile_size=T.TILE_SIZE):
    tx, ty = C.c_uint32(0), C.c_uint32(0)
    _lib().rt_host_tile_count(C.c_uint32(width), C.c_uint32(height), C.c_uint32(tile_size), C.byref(tx), C.byref(ty))
    return tx.value, ty.value


def tiles_per_frame(total):
    return int(_lib().rt_host_tiles_per_frame(C.c_uint32(total)))


def default_scene():
    sp, tr, ve = np.zeros(6, T.SPHERE), np.zeros(2, T.TRIANGLE), np.zeros(6, T.VERTEX)
    ma, li, cam = np.zeros(4, T.MATERIAL), np.zeros(1, T.LIGHT), np.zeros(1, T.CAMERA)
    n = [C.c_uint32(len(a)) for a in (sp, tr, ve, ma, li)]
    rc = _lib().rt_host_default_scene(_p(sp), C.byref(n[0]), _p(tr), C.byref(n[1]), _p(ve), C.byref(n[2]), _p(ma), C.byref(n[3]),
                                      _p(li), C.byref(n[4]), _p(cam))
    if rc != 0:
        raise RuntimeError("rt_host_default_scene failed")
    return sp[:n[0].value], tr[:n[1].value], ve[:n[2].value], ma[:n[3].value], li[:n[4].value], cam[0]


def bvh_build(triangles, vertices):
    tris, verts = np.ascontiguousarray(triangles), np.ascontiguousarray(vertices)
    nn, ni = C.c_uint32(0), C.c_uint32(0)
    lib = _lib()
    rc = lib.rt_host_bvh_build(_p(tris), C.c_uint32(len(tris)), _p(verts), C.c_uint32(len(verts)), C.c_void_p(0), C.byref(nn), C.c_void_p(0), C.byref(ni))
    if rc != 0:
        raise api.RtError(rc, "rt_host_bvh_build")
    nodes, idx = np.zeros(nn.value, T.BVH_NODE), np.zeros(ni.value, np.uint32)
    rc = lib.rt_host_bvh_build(_p(tris), C.c_uint32(len(tris)), _p(verts), C.c_uint32(len(verts)), _p(nodes), C.byref(nn), _p(idx), C.byref(ni))
    if rc != 0:
        raise api.RtError(rc, "rt_host_bvh_build")
    return nodes, idx


def pack_scene_metadata(spheres, lights, nodes, tri_indices, vertices):
    arrs = [np.ascontiguousarray(a) for a in (spheres, lights, nodes)] + [np.ascontiguousarray(tri_indices, dtype=np.uint32),
                                                                           np.ascontiguousarray(vertices)]
    words = 5 * len(arrs[0]) + 13 * len(arrs[1]) + 12 * len(arrs[2]) + len(arrs[3]) + 3 * len(arrs[4])
    out = np.zeros(words, np.uint32)
    off = np.zeros(1, T.SCENE_METADATA_OFFSETS)
    rc = _lib().rt_host_pack_scene_metadata(_p(arrs[0]), C.c_uint32(len(arrs[0])), _p(arrs[1]), C.c_uint32(len(arrs[1])), _p(arrs[2]),
                                            C.c_uint32(len(arrs[2])), _p(arrs[3]), C.c_uint32(len(arrs[3])), _p(arrs[4]),
                                            C.c_uint32(len(arrs[4])), _p(out), C.c_size_t(words), _p(off))
    if rc != 0:
        raise api.RtError(rc, "rt_host_pack_scene_metadata")
    return out, off[0]


def render_progressive(ctx, scene, width, height, camera=None):
    """The reference's frame loop (BvhBuilder::build -> BufferManager -> ComputeRenderer::run_compute) on an api.Context."""
    arrs = [np.ascontiguousarray(a) for a in (scene.spheres, scene.lights, scene.vertices, scene.triangles, scene.materials)]
    cam = np.ascontiguousarray(scene.camera if camera is None else camera)
    nd, nc = C.c_uint32(0), C.c_uint32(0)
    rc = _lib().rt_host_render_progressive(ctx._h, _p(arrs[0]), C.c_uint32(len(arrs[0])), _p(arrs[1]), C.c_uint32(len(arrs[1])),
                                           _p(arrs[2]), C.c_uint32(len(arrs[2])), _p(arrs[3]), C.c_uint32(len(arrs[3])),
                                           _p(arrs[4]), C.c_uint32(len(arrs[4])), _p(cam), C.c_uint32(width), C.c_uint32(height),
                                           C.byref(nd), C.byref(nc))
    ctx._check(rc)
    ctx.width, ctx.height = width, height
    return nd.value, nc.value


# ---- row N1: glTF loading -----------------------------------------------------------------------------
HOST_SYMBOLS += ["rt_host_camera_rotate", "rt_host_camera_move"]


def camera_rotate(camera, delta_x, delta_y):
    """CameraController::rotate_camera (src/input.rs:49-76) on a copy of `camera`."""
    cam = np.array(camera, dtype=T.CAMERA).copy()
    lib = _lib()
    lib.rt_host_camera_rotate.restype = None
    lib.rt_host_camera_rotate(_p(cam), C.c_double(delta_x), C.c_double(delta_y))
    return cam


def camera_move(camera, forward, right):
    """CameraController::move_camera (src/input.rs:79-97) on a copy of `camera`."""
    cam = np.array(camera, dtype=T.CAMERA).copy()
    lib = _lib()
    lib.rt_host_camera_move.restype = None
    lib.rt_host_camera_move(_p(cam), C.c_float(forward), C.c_float(right))
    return cam


HOST_SYMBOLS += ["rt_host_gltf_load", "rt_host_gltf_load_glb", "rt_host_scene_counts", "rt_host_scene_copy", "rt_host_scene_free",
                 "rt_host_write_ppm", "rt_host_write_png", "rt_host_write_exr", "rt_host_progressive_timing"]
GLTF_ERRORS = {-11: "IoError", -12: "GltfError", -13: "ValidationError"}


class GltfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{GLTF_ERRORS.get(code, code)}: {msg}")
        self.code = code


def _scene_from_handle(lib, h, name):
    from . import hostpack as H
    from .scenes import Scene
    counts = (C.c_uint32 * 6)()
    lib.rt_host_scene_counts(h, counts)
    sp, li, ve = np.zeros(counts[0], T.SPHERE), np.zeros(counts[1], T.LIGHT), np.zeros(counts[2], T.VERTEX)
    tr, ma, ca = np.zeros(counts[3], T.TRIANGLE), np.zeros(counts[4], T.MATERIAL), np.zeros(counts[5], T.CAMERA)
    lib.rt_host_scene_copy(h, _p(sp), _p(li), _p(ve), _p(tr), _p(ma), _p(ca))
    lib.rt_host_scene_free(h)
    # SceneState::load_from_gltf (src/scene.rs:43-69): first camera of the file, else Camera::new
    cam = ca[0].copy() if len(ca) else H.camera()
    return Scene(name, sp, li, ve, tr, ma, cam, {"cameras": ca})


def load_gltf(path, scene_index=-1):
    """GltfLoader::load_from_path + extract_scene -> Scene (camera = first camera or Camera::new)."""
    lib = _lib()
    lib.rt_host_gltf_load.restype = C.c_int
    lib.rt_host_scene_free.restype = None
    lib.rt_host_scene_free.argtypes = [C.c_void_p]
    lib.rt_host_scene_counts.restype = None
    lib.rt_host_scene_counts.argtypes = [C.c_void_p, C.c_void_p]
    lib.rt_host_scene_copy.argtypes = [C.c_void_p] * 7
    h = C.c_void_p(0)
    err = C.create_string_buffer(512)
    rc = lib.rt_host_gltf_load(str(path).encode(), C.c_int(scene_index), C.byref(h), err, C.c_size_t(512))
    if rc != 0:
        raise GltfError(rc, err.value.decode())
    return _scene_from_handle(lib, h, str(path))


def load_glb(data, scene_index=-1):
    lib = _lib()
    lib.rt_host_gltf_load_glb.restype = C.c_int
    lib.rt_host_scene_free.restype = None
    lib.rt_host_scene_free.argtypes = [C.c_void_p]
    lib.rt_host_scene_counts.restype = None
    lib.rt_host_scene_counts.argtypes = [C.c_void_p, C.c_void_p]
    lib.rt_host_scene_copy.argtypes = [C.c_void_p] * 7
    h = C.c_void_p(0)
    err = C.create_string_buffer(512)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rc = lib.rt_host_gltf_load_glb(buf, C.c_size_t(len(data)), C.c_int(scene_index), C.byref(h), err, C.c_size_t(512))
    if rc != 0:
        raise GltfError(rc, err.value.decode())
    return _scene_from_handle(lib, h, "glb")


# ---- rows N3 / N4 ------------------------------------------------------------------------------------
def write_image(path, rgba8):
    """rgba8 (H x W x 4 uint8) to .png / .ppm, or the float image (H x W x 3 float32) to .exr."""
    if str(path).lower().endswith(".exr"):
        img = np.ascontiguousarray(rgba8, dtype=np.float32)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("an .exr takes the H x W x 3 float image")
        fn = _lib().rt_host_write_exr
        fn.restype = C.c_int
        rc = fn(str(path).encode(), _p(img), C.c_uint32(img.shape[1]), C.c_uint32(img.shape[0]))
        if rc != 0:
            raise OSError(f"cannot write {path} ({rc})")
        return
    img = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w = img.shape[:2]
    fn = _lib().rt_host_write_png if str(path).lower().endswith(".png") else _lib().rt_host_write_ppm
    fn.restype = C.c_int
    rc = fn(str(path).encode(), _p(img), C.c_uint32(w), C.c_uint32(h))
    if rc != 0:
        raise OSError(f"cannot write {path} ({rc})")


def progressive_timing():
    out = (C.c_double * 7)()
    fn = _lib().rt_host_progressive_timing
    fn.restype = None
    fn(out)
    return dict(zip(("total_ms", "calls", "tiles", "tiles_per_s", "p50_ms", "p95_ms", "p99_ms"), [float(v) for v in out]))
