/*
 * rt_shared.h — the host<->device data contract of the ray-casting hot path.
 *
 * C mirrors of the reference's `shared` crate #[repr(C)] Pod types
 * (reference: shared/src/lib.rs:38-227).  A Rust host can pass its own
 * `Vec<T>` buffers to the C ABI in rt_hip.h unchanged: every struct here has
 * the same size, field order and field offsets as its Rust counterpart, which
 * the static asserts at the bottom of this file pin.
 *
 * All fields are little-endian and 4-byte aligned; there is no implicit
 * padding anywhere.
 */
#ifndef RT_SHARED_H
#define RT_SHARED_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RaytracerConfig constants — shared/src/lib.rs:20-35 */
#define RT_TILE_SIZE 128u              /* RaytracerConfig::TILE_SIZE            :21 */
#define RT_THREAD_GROUP_X 16u          /* RaytracerConfig::THREAD_GROUP_SIZE    :22 */
#define RT_THREAD_GROUP_Y 16u
#define RT_MIN_RAY_DISTANCE 0.00001f   /* RaytracerConfig::MIN_RAY_DISTANCE     :27 */
#define RT_MAX_PUSH_CONSTANT_SIZE 128u /* RaytracerConfig::MAX_PUSH_CONSTANT_SIZE :30 */
#define RT_INVALID_INDEX 0xFFFFFFFFu   /* leaf / "no child" sentinel, :157-158, :809-810 */
/* src/buffers.rs:49-53: 128 MiB / sizeof(Triangle) */
#define RT_REF_TRIANGLES_PER_BUFFER 8388608u

/* shared/src/lib.rs:38-45 */
typedef struct rt_camera {
    float position[3];
    float direction[3];
    float up[3];
    float fov; /* vertical, degrees */
} rt_camera;

/* shared/src/lib.rs:49-66 */
typedef struct rt_material {
    float albedo[3];
    uint32_t metallic_roughness_f16; /* metallic lo16, roughness hi16 (IEEE half) */
    float emission[3];
    uint32_t ior_transmission_f16;   /* ior lo16, transmission hi16 (IEEE half) */
    float specular_factor;
    float specular_color[3];
    float attenuation_distance;
    float attenuation_color[3];
    float thickness_factor;
    float diffuse_factor[3];
    float glossiness_factor;
    uint32_t material_type; /* 0 metallic-roughness, 1 specular-glossiness */
    uint32_t texture_indices[8];
    float _padding[2];
} rt_material;

/* shared/src/lib.rs:70-82 (52 bytes = 13 words; the in-source 44/48 comments are wrong) */
typedef struct rt_light {
    float position[3];
    uint32_t light_type; /* 0 directional, 1 point, 2 spot */
    float color[3];
    float intensity;
    float direction[3];
    uint32_t range_packed;       /* f16 in lo16; never read by the kernel */
    uint32_t cone_angles_packed; /* inner lo16, outer hi16; never read by the kernel */
} rt_light;

/* shared/src/lib.rs:85-95 */
typedef struct rt_texture_info {
    uint32_t width, height, format, mip_levels, offset, size;
    uint32_t _padding[2];
} rt_texture_info;

/* shared/src/lib.rs:99-106 */
typedef struct rt_sphere {
    float center[3];
    float radius;
    uint32_t material_id;
} rt_sphere;

/* shared/src/lib.rs:110-115 */
typedef struct rt_vertex {
    float position[3];
} rt_vertex;

/* shared/src/lib.rs:119-127 */
typedef struct rt_triangle {
    uint32_t v0_index, v1_index, v2_index;
    uint32_t material_id;
} rt_triangle;

/* shared/src/lib.rs:143-150 */
typedef struct rt_aabb {
    float min[3];
    float _padding0;
    float max[3];
    float _padding1;
} rt_aabb;

/* shared/src/lib.rs:153-161; leaf <=> left_child == 0xFFFFFFFF (shader/src/bvh.rs:60) */
typedef struct rt_bvh_node {
    rt_aabb bounds;
    uint32_t left_child;
    uint32_t right_child;
    uint32_t triangle_start;
    uint32_t triangle_count;
} rt_bvh_node;

/* shared/src/lib.rs:165-181 */
typedef struct rt_wavefront_ray {
    float origin[3];
    uint32_t ray_type; /* 0 camera, 1 reflection, 2 transmission, 3 shadow */
    float direction[3];
    uint32_t bounce_depth;
    float throughput[3];
    float medium_ior;
    uint32_t pixel_coord[2];
    float inv_pdf;
    float t_min;
    float t_max;
    uint32_t wavelength_channel;
    uint32_t active;
} rt_wavefront_ray;

/* shared/src/lib.rs:185-194 */
typedef struct rt_wavefront_counters {
    uint32_t total_rays_generated;
    uint32_t rays_per_bounce[8];
    uint32_t active_bounce_depths;
    uint32_t max_bounce_depth;
    uint32_t frame_seed;
    uint32_t _padding[3];
} rt_wavefront_counters;

/* shared/src/lib.rs:197-210 — offsets in u32 words into the combined metadata buffer */
typedef struct rt_scene_metadata_offsets {
    uint32_t spheres_offset, spheres_count;
    uint32_t lights_offset, lights_count;
    uint32_t bvh_nodes_offset, bvh_nodes_count;
    uint32_t triangle_indices_offset, triangle_indices_count;
    uint32_t vertices_offset, vertices_count;
} rt_scene_metadata_offsets;

/* shared/src/lib.rs:213-227 */
typedef struct rt_push_constants {
    float resolution[2];
    rt_camera camera;
    uint32_t triangle_count;
    uint32_t material_count;
    uint32_t tile_offset[2];
    uint32_t tile_size_packed; /* width lo16, height hi16 (:1138-1150) */
    uint32_t total_tiles[2];
    uint32_t triangles_per_buffer;
    rt_scene_metadata_offsets metadata_offsets;
    uint32_t packed_flags; /* channel 0-7, cur bounce 8-15, max bounce 16-23, mode 24-31 (:1154-1179) */
    uint32_t frame_seed;
} rt_push_constants;

/* Strides (u32 words) of the sections of binding 1, shader/src/scene_access.rs:34-178 */
#define RT_SPHERE_WORDS 5u
#define RT_LIGHT_WORDS 13u
#define RT_BVH_NODE_WORDS 12u
#define RT_VERTEX_WORDS 3u

#ifdef __cplusplus
} /* extern "C" */
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

RT_STATIC_ASSERT(sizeof(rt_camera) == 40, "Camera is 40 B");
RT_STATIC_ASSERT(offsetof(rt_camera, direction) == 12 && offsetof(rt_camera, up) == 24 &&
                     offsetof(rt_camera, fov) == 36, "Camera offsets");
RT_STATIC_ASSERT(sizeof(rt_material) == 128, "Material is 128 B");
RT_STATIC_ASSERT(offsetof(rt_material, metallic_roughness_f16) == 12 && offsetof(rt_material, emission) == 16 &&
                     offsetof(rt_material, ior_transmission_f16) == 28 && offsetof(rt_material, specular_factor) == 32 &&
                     offsetof(rt_material, specular_color) == 36 && offsetof(rt_material, attenuation_distance) == 48 &&
                     offsetof(rt_material, attenuation_color) == 52 && offsetof(rt_material, thickness_factor) == 64 &&
                     offsetof(rt_material, diffuse_factor) == 68 && offsetof(rt_material, glossiness_factor) == 80 &&
                     offsetof(rt_material, material_type) == 84 && offsetof(rt_material, texture_indices) == 88 &&
                     offsetof(rt_material, _padding) == 120, "Material offsets");
RT_STATIC_ASSERT(sizeof(rt_light) == 52, "Light is 52 B");
RT_STATIC_ASSERT(offsetof(rt_light, light_type) == 12 && offsetof(rt_light, color) == 16 &&
                     offsetof(rt_light, intensity) == 28 && offsetof(rt_light, direction) == 32 &&
                     offsetof(rt_light, range_packed) == 44 && offsetof(rt_light, cone_angles_packed) == 48, "Light offsets");
RT_STATIC_ASSERT(sizeof(rt_texture_info) == 32, "TextureInfo is 32 B");
RT_STATIC_ASSERT(sizeof(rt_sphere) == 20, "Sphere is 20 B");
RT_STATIC_ASSERT(sizeof(rt_vertex) == 12, "Vertex is 12 B");
RT_STATIC_ASSERT(sizeof(rt_triangle) == 16, "Triangle is 16 B");
RT_STATIC_ASSERT(sizeof(rt_aabb) == 32 && offsetof(rt_aabb, max) == 16, "Aabb is 32 B");
RT_STATIC_ASSERT(sizeof(rt_bvh_node) == 48, "BvhNode is 48 B");
RT_STATIC_ASSERT(offsetof(rt_bvh_node, left_child) == 32 && offsetof(rt_bvh_node, right_child) == 36 &&
                     offsetof(rt_bvh_node, triangle_start) == 40 && offsetof(rt_bvh_node, triangle_count) == 44, "BvhNode offsets");
RT_STATIC_ASSERT(sizeof(rt_wavefront_ray) == 76, "WavefrontRay is 76 B");
RT_STATIC_ASSERT(offsetof(rt_wavefront_ray, active) == 72, "WavefrontRay.active is word 18");
RT_STATIC_ASSERT(sizeof(rt_wavefront_counters) == 60, "WavefrontCounters is 60 B");
RT_STATIC_ASSERT(sizeof(rt_scene_metadata_offsets) == 40, "SceneMetadataOffsets is 40 B");
RT_STATIC_ASSERT(sizeof(rt_push_constants) == 128, "PushConstants is 128 B");
RT_STATIC_ASSERT(offsetof(rt_push_constants, camera) == 8 && offsetof(rt_push_constants, triangle_count) == 48 &&
                     offsetof(rt_push_constants, material_count) == 52 && offsetof(rt_push_constants, tile_offset) == 56 &&
                     offsetof(rt_push_constants, tile_size_packed) == 64 && offsetof(rt_push_constants, total_tiles) == 68 &&
                     offsetof(rt_push_constants, triangles_per_buffer) == 76 &&
                     offsetof(rt_push_constants, metadata_offsets) == 80 && offsetof(rt_push_constants, packed_flags) == 120 &&
                     offsetof(rt_push_constants, frame_seed) == 124, "PushConstants offsets");

#endif /* RT_SHARED_H */
