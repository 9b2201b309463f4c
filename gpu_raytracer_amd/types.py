"""numpy mirrors of the contract structs in include/rt_shared.h.

Reference: shared/src/lib.rs:38-227 (#[repr(C)] Pod types).  These dtypes are
what the test/bench harness uses to build inputs for both the HIP library and
the oracle; sizes are asserted against the C header's static asserts.
"""
import numpy as np

f32 = np.float32
u32 = np.uint32

CAMERA = np.dtype([("position", f32, 3), ("direction", f32, 3), ("up", f32, 3), ("fov", f32)])
MATERIAL = np.dtype([
    ("albedo", f32, 3), ("metallic_roughness_f16", u32),
    ("emission", f32, 3), ("ior_transmission_f16", u32),
    ("specular_factor", f32), ("specular_color", f32, 3),
    ("attenuation_distance", f32), ("attenuation_color", f32, 3),
    ("thickness_factor", f32), ("diffuse_factor", f32, 3),
    ("glossiness_factor", f32), ("material_type", u32),
    ("texture_indices", u32, 8), ("_padding", f32, 2),
])
LIGHT = np.dtype([
    ("position", f32, 3), ("light_type", u32), ("color", f32, 3), ("intensity", f32),
    ("direction", f32, 3), ("range_packed", u32), ("cone_angles_packed", u32),
])
TEXTURE_INFO = np.dtype([("width", u32), ("height", u32), ("format", u32), ("mip_levels", u32),
                         ("offset", u32), ("size", u32), ("_padding", u32, 2)])
SPHERE = np.dtype([("center", f32, 3), ("radius", f32), ("material_id", u32)])
VERTEX = np.dtype([("position", f32, 3)])
TRIANGLE = np.dtype([("v0_index", u32), ("v1_index", u32), ("v2_index", u32), ("material_id", u32)])
AABB = np.dtype([("min", f32, 3), ("_padding0", f32), ("max", f32, 3), ("_padding1", f32)])
BVH_NODE = np.dtype([("bounds", AABB), ("left_child", u32), ("right_child", u32),
                     ("triangle_start", u32), ("triangle_count", u32)])
WAVEFRONT_RAY = np.dtype([
    ("origin", f32, 3), ("ray_type", u32), ("direction", f32, 3), ("bounce_depth", u32),
    ("throughput", f32, 3), ("medium_ior", f32), ("pixel_coord", u32, 2), ("inv_pdf", f32),
    ("t_min", f32), ("t_max", f32), ("wavelength_channel", u32), ("active", u32),
])
WAVEFRONT_COUNTERS = np.dtype([
    ("total_rays_generated", u32), ("rays_per_bounce", u32, 8), ("active_bounce_depths", u32),
    ("max_bounce_depth", u32), ("frame_seed", u32), ("_padding", u32, 3),
])
SCENE_METADATA_OFFSETS = np.dtype([
    ("spheres_offset", u32), ("spheres_count", u32), ("lights_offset", u32), ("lights_count", u32),
    ("bvh_nodes_offset", u32), ("bvh_nodes_count", u32),
    ("triangle_indices_offset", u32), ("triangle_indices_count", u32),
    ("vertices_offset", u32), ("vertices_count", u32),
])
PUSH_CONSTANTS = np.dtype([
    ("resolution", f32, 2), ("camera", CAMERA), ("triangle_count", u32), ("material_count", u32),
    ("tile_offset", u32, 2), ("tile_size_packed", u32), ("total_tiles", u32, 2),
    ("triangles_per_buffer", u32), ("metadata_offsets", SCENE_METADATA_OFFSETS),
    ("packed_flags", u32), ("frame_seed", u32),
])

# rt_hip.h
RENDER_PARAMS = np.dtype([
    ("camera", CAMERA), ("width", u32), ("height", u32), ("spp", u32), ("max_bounces", u32),
    ("mode", u32), ("frame_seed", u32), ("tile_size", u32), ("tile_rank", u32), ("tile_world", u32),
    ("flags", u32),
])
STATS = np.dtype([
    ("rays", np.uint64), ("primary_rays", np.uint64), ("continuation_rays", np.uint64), ("shadow_rays", np.uint64),
    ("pixels", np.uint64),
    ("node_visits", np.uint64), ("tri_tests", np.uint64),
    ("kernel_ms", np.float64), ("wall_ms", np.float64),
    ("node_bytes", np.uint64), ("tri_bytes", np.uint64), ("scene_bytes", np.uint64),
    ("bvh_nodes", u32), ("bvh_depth", u32), ("n_devices", u32), ("flags", u32),
    ("texture_bytes", np.uint64), ("n_textures", u32), ("tree_build", u32),
    ("grid_bytes", np.uint64), ("grid_build_ms", np.float64),
])
STAT_MEGAKERNEL_FALLBACK = 1
STAT_SINGLE_PASS = 2
PREPARE_SHADOW_GRIDS = 1
PREPARE_QUALITY_TREE = 2
MAX_BOUNCES = 255

EXPECTED_SIZES = {
    "CAMERA": 40, "MATERIAL": 128, "LIGHT": 52, "TEXTURE_INFO": 32, "SPHERE": 20, "VERTEX": 12,
    "TRIANGLE": 16, "AABB": 32, "BVH_NODE": 48, "WAVEFRONT_RAY": 76, "WAVEFRONT_COUNTERS": 60,
    "SCENE_METADATA_OFFSETS": 40, "PUSH_CONSTANTS": 128,
}
for _name, _size in EXPECTED_SIZES.items():
    assert globals()[_name].itemsize == _size, (_name, globals()[_name].itemsize, _size)

TILE_SIZE = 128
MIN_RAY_DISTANCE = np.float32(0.00001)
INVALID_INDEX = 0xFFFFFFFF
REF_TRIANGLES_PER_BUFFER = 8388608  # src/buffers.rs:49-53
