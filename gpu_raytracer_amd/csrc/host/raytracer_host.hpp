// raytracer_host.hpp — C++ host side above the C ABI, mirroring the reference's host API for
// the hot path (same names, argument meaning and error behaviour), so that a host written
// against the reference's `shared` crate and `src/{buffers,compute,bvh,scene}.rs` maps 1:1:
//
//   shared/src/lib.rs   Camera, Material, Light, Sphere, Vertex, Triangle, TriangleLegacy, Aabb,
//                       BvhNode, WavefrontRay, WavefrontCounters, SceneMetadataOffsets,
//                       PushConstants, TileHelper, SceneBuilder, RaytracerConfig
//   src/bvh.rs          BvhBuilder::build -> BvhResult (reference-format BVH, row N2 of SURVEY §8f)
//   src/scene.rs        SceneState
//   src/buffers.rs      BufferManager::update_* (packing of bindings 1-5)  -> rt_upload_scene_packed
//   src/compute.rs      ComputeRenderer::run_compute (tile x channel loop)  -> rt_dispatch_tile
//
// The Pod structs themselves are the C structs of include/rt_shared.h; the free functions in
// the per-type namespaces below are the reference's associated functions.
#ifndef RT_RAYTRACER_HOST_HPP
#define RT_RAYTRACER_HOST_HPP

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../../include/rt_hip.h"
#include "../half.h"

namespace raytracer {

// shared/src/lib.rs:18-35
struct RaytracerConfig {
    static constexpr uint32_t TILE_SIZE = 128;
    static constexpr uint32_t THREAD_GROUP_SIZE_X = 16, THREAD_GROUP_SIZE_Y = 16;
    static constexpr size_t DEFAULT_MAX_SPHERES = 64, DEFAULT_MAX_TRIANGLES = 64;
    static constexpr float CAMERA_MOVE_SPEED = 0.1f, CAMERA_ROTATE_SENSITIVITY = 0.005f, CAMERA_PITCH_CLAMP = 0.99f;
    static constexpr float MIN_RAY_DISTANCE = 0.00001f;
    static constexpr uint32_t MAX_PUSH_CONSTANT_SIZE = 128;
};
static_assert(sizeof(rt_push_constants) <= RaytracerConfig::MAX_PUSH_CONSTANT_SIZE, "src/main.rs:223-225");

using Camera = rt_camera;
using Material = rt_material;
using Light = rt_light;
using Sphere = rt_sphere;
using Vertex = rt_vertex;
using Triangle = rt_triangle;
using Aabb = rt_aabb;
using BvhNode = rt_bvh_node;
using WavefrontRay = rt_wavefront_ray;
using WavefrontCounters = rt_wavefront_counters;
using SceneMetadataOffsets = rt_scene_metadata_offsets;
using PushConstants = rt_push_constants;

namespace camera {
inline Camera new_() { return Camera{{0.0f, 0.0f, 5.0f}, {0.0f, 0.0f, -1.0f}, {0.0f, 1.0f, 0.0f}, 45.0f}; } // :231-238
} // namespace camera

// CameraController — src/input.rs:46-97: the arithmetic behind mouse-drag rotation and key movement (the event plumbing
// of InputState is window-system code and stays out).  Used for scripted fly-throughs: the same camera path as the
// interactive application would produce from the same deltas.
namespace camera_controller {
// mouse delta in pixels -> yaw about +Y, then the y component pushed by the vertical delta and clamped, then normalised
inline void rotate_camera(Camera& cam, double delta_x, double delta_y) { // :49-76
    const float yaw = (float)delta_x * RaytracerConfig::CAMERA_ROTATE_SENSITIVITY;
    const float pitch = (float)delta_y * RaytracerConfig::CAMERA_ROTATE_SENSITIVITY;
    const float c = std::cos(yaw), s = std::sin(yaw);
    const float x = cam.direction[0], z = cam.direction[2];
    cam.direction[0] = x * c - z * s;
    cam.direction[2] = x * s + z * c;
    const float y = cam.direction[1] - pitch;
    cam.direction[1] = std::max(-RaytracerConfig::CAMERA_PITCH_CLAMP, std::min(RaytracerConfig::CAMERA_PITCH_CLAMP, y));
    const float len = std::sqrt(cam.direction[0] * cam.direction[0] + cam.direction[1] * cam.direction[1] + cam.direction[2] * cam.direction[2]);
    if (len > 0.0f)
        for (float& v : cam.direction) v /= len;
}
// forward / right in units of CAMERA_MOVE_SPEED; "right" is direction x up (not normalised, as in the reference)
inline void move_camera(Camera& cam, float forward, float right) { // :79-97
    const float speed = RaytracerConfig::CAMERA_MOVE_SPEED;
    for (int a = 0; a < 3; a++) cam.position[a] += cam.direction[a] * forward * speed;
    const float* d = cam.direction;
    const float* u = cam.up;
    const float side[3] = {d[1] * u[2] - d[2] * u[1], d[2] * u[0] - d[0] * u[2], d[0] * u[1] - d[1] * u[0]};
    for (int a = 0; a < 3; a++) cam.position[a] += side[a] * right * speed;
}
} // namespace camera_controller

namespace material {
inline uint16_t f32_to_f16_u16(float v) { return rt::f32_to_f16_bits(v); } // :250-252 (half::f16::from_f32, RNE)
// Material::new :255-291
inline Material new_(const float albedo[3], float metallic, float roughness, const float emission[3], float ior, float transmission) {
    Material m;
    std::memset(&m, 0, sizeof m);
    std::memcpy(m.albedo, albedo, 12);
    m.metallic_roughness_f16 = (uint32_t)f32_to_f16_u16(metallic) | ((uint32_t)f32_to_f16_u16(roughness) << 16);
    std::memcpy(m.emission, emission, 12);
    m.ior_transmission_f16 = (uint32_t)f32_to_f16_u16(ior) | ((uint32_t)f32_to_f16_u16(transmission) << 16);
    m.specular_factor = 1.0f;
    m.specular_color[0] = m.specular_color[1] = m.specular_color[2] = 1.0f;
    m.attenuation_distance = std::numeric_limits<float>::infinity();
    m.attenuation_color[0] = m.attenuation_color[1] = m.attenuation_color[2] = 1.0f;
    m.thickness_factor = 0.0f;
    std::memcpy(m.diffuse_factor, albedo, 12);
    m.glossiness_factor = 1.0f - roughness;
    m.material_type = 0;
    for (int i = 0; i < 8; i++) m.texture_indices[i] = 0xFFFFFFFFu;
    return m;
}
inline Material diffuse(const float a[3]) { const float z[3] = {0, 0, 0}; return new_(a, 0.0f, 1.0f, z, 1.5f, 0.0f); }               // :315-317
inline Material metallic(const float a[3], float roughness) { const float z[3] = {0, 0, 0}; return new_(a, 1.0f, roughness, z, 1.5f, 0.0f); } // :320-322
inline Material glass(const float a[3], float ior, float transmission) { const float z[3] = {0, 0, 0}; return new_(a, 0.0f, 0.0f, z, ior, transmission); } // :325-327
inline Material emissive(const float a[3], const float e[3]) { return new_(a, 0.0f, 1.0f, e, 1.5f, 0.0f); }                           // :330-332
inline Material specular_glossiness(const float diffuse_[3], const float specular[3], float glossiness) {                             // :335-346
    const float z[3] = {0, 0, 0};
    Material m = new_(diffuse_, 0.0f, 1.0f - glossiness, z, 1.5f, 0.0f);
    m.material_type = 1;
    std::memcpy(m.diffuse_factor, diffuse_, 12);
    std::memcpy(m.specular_color, specular, 12);
    m.glossiness_factor = glossiness;
    return m;
}
inline Material with_volume(Material m, float thickness, float attenuation_distance, const float attenuation_color[3]) { // :349-354
    m.thickness_factor = thickness;
    m.attenuation_distance = attenuation_distance;
    std::memcpy(m.attenuation_color, attenuation_color, 12);
    return m;
}
inline Material with_specular(Material m, float factor, const float color[3]) { // :357-361
    m.specular_factor = factor;
    std::memcpy(m.specular_color, color, 12);
    return m;
}
inline void set_metallic(Material& m, float v) { m.metallic_roughness_f16 = (m.metallic_roughness_f16 & 0xFFFF0000u) | f32_to_f16_u16(v); }            // :371-375
inline void set_roughness(Material& m, float v) { m.metallic_roughness_f16 = (m.metallic_roughness_f16 & 0x0000FFFFu) | ((uint32_t)f32_to_f16_u16(v) << 16); } // :379-383
inline void set_ior(Material& m, float v) { m.ior_transmission_f16 = (m.ior_transmission_f16 & 0xFFFF0000u) | f32_to_f16_u16(v); }                      // :387-391
inline void set_transmission(Material& m, float v) { m.ior_transmission_f16 = (m.ior_transmission_f16 & 0x0000FFFFu) | ((uint32_t)f32_to_f16_u16(v) << 16); } // :395-399
inline void unpack_metallic_roughness(const Material& m, float* metallic_, float* roughness) { // :403-422
    *metallic_ = rt::f16_bits_to_f32((uint16_t)(m.metallic_roughness_f16 & 0xFFFF));
    *roughness = rt::f16_bits_to_f32((uint16_t)(m.metallic_roughness_f16 >> 16));
}
inline void unpack_ior_transmission(const Material& m, float* ior, float* transmission) { // :426-444
    *ior = rt::f16_bits_to_f32((uint16_t)(m.ior_transmission_f16 & 0xFFFF));
    *transmission = rt::f16_bits_to_f32((uint16_t)(m.ior_transmission_f16 >> 16));
}
} // namespace material

namespace light {
inline uint32_t pack_range(float r) { return rt::f32_to_f16_bits(r); }                                                                    // :483-486
inline uint32_t pack_cone_angles(float i, float o) { return (uint32_t)rt::f32_to_f16_bits(i) | ((uint32_t)rt::f32_to_f16_bits(o) << 16); } // :490-494
inline Light make(const float pos[3], uint32_t type, const float color[3], float intensity, const float dir[3], float range, float inner, float outer) {
    Light l;
    std::memcpy(l.position, pos, 12);
    l.light_type = type;
    std::memcpy(l.color, color, 12);
    l.intensity = intensity;
    std::memcpy(l.direction, dir, 12);
    l.range_packed = pack_range(range);
    l.cone_angles_packed = pack_cone_angles(inner, outer);
    return l;
}
inline Light directional(const float direction[3], const float color[3], float intensity) { // :497-522
    const float z[3] = {0, 0, 0};
    return make(z, 0, color, intensity, direction, std::numeric_limits<float>::infinity(), 0.0f, 0.0f);
}
inline Light point(const float position[3], const float color[3], float intensity, float range) { // :525-550
    const float z[3] = {0, 0, 0};
    return make(position, 1, color, intensity, z, range, 0.0f, 0.0f);
}
inline Light spot(const float position[3], const float direction[3], const float color[3], float intensity, float range, float inner, float outer) { // :553-586
    return make(position, 2, color, intensity, direction, range, inner, outer);
}
inline float unpack_range(const Light& l) { return rt::f16_bits_to_f32((uint16_t)(l.range_packed & 0xFFFF)); } // :589-602
} // namespace light

namespace aabb {
inline Aabb new_(const float mn[3], const float mx[3]) { return Aabb{{mn[0], mn[1], mn[2]}, 0.0f, {mx[0], mx[1], mx[2]}, 0.0f}; } // :753-760
inline Aabb empty() { // :763-768
    const float inf = std::numeric_limits<float>::infinity();
    const float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    return new_(mn, mx);
}
inline Aabb union_(const Aabb& a, const Aabb& b) { // :771-784
    float mn[3], mx[3];
    for (int i = 0; i < 3; i++) {
        mn[i] = std::fmin(a.min[i], b.min[i]);
        mx[i] = std::fmax(a.max[i], b.max[i]);
    }
    return new_(mn, mx);
}
inline void center(const Aabb& a, float out[3]) { for (int i = 0; i < 3; i++) out[i] = (a.min[i] + a.max[i]) * 0.5f; } // :787-793
inline float surface_area(const Aabb& a) { // :796-801
    float dx = a.max[0] - a.min[0], dy = a.max[1] - a.min[1], dz = a.max[2] - a.min[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}
} // namespace aabb

namespace triangle {
inline Triangle new_indexed(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t material_id) { return Triangle{v0, v1, v2, material_id}; } // :661-668
inline Aabb bounding_box(const Triangle& t, const Vertex* vertices) { // :671-685
    const float *a = vertices[t.v0_index].position, *b = vertices[t.v1_index].position, *c = vertices[t.v2_index].position;
    float mn[3], mx[3];
    for (int i = 0; i < 3; i++) {
        mn[i] = std::fmin(std::fmin(a[i], b[i]), c[i]);
        mx[i] = std::fmax(std::fmax(a[i], b[i]), c[i]);
    }
    return aabb::new_(mn, mx);
}
} // namespace triangle

// branchless_float_if! / branchless_u32_if! (shared/src/lib.rs:1293-1326).  The shader never uses them; they are part of
// the `shared` crate's surface and its unit tests pin their NaN behaviour.  f32::min returns the non-NaN operand = fminf.
namespace branchless {
inline float float_if_nonnan(bool condition, float if_true, float if_false) { // the `@nonnan` arm, :1314-1316
    return (float)(uint32_t)condition * if_true + (float)(((uint32_t)condition) ^ 1u) * if_false;
}
struct FloatIf {
    float value;
    bool valid;
};
inline FloatIf float_if(bool condition, float if_true, float if_false) { // :1295-1313 with max_val = f32::MAX, max_minus_one = f32::MAX - 1.0, lt
    const float max_val = 3.402823466e+38f, max_minus_one = max_val - 1.0f;
    const float actual_if_true = std::fmin(if_true, max_val); // a NaN operand becomes max_val
    const float actual_if_false = std::fmin(if_false, max_val);
    const float true_contrib = float_if_nonnan(actual_if_true < max_minus_one, actual_if_true, actual_if_false);
    const float false_contrib = float_if_nonnan(actual_if_false < max_minus_one, actual_if_false, actual_if_true);
    const float res = float_if_nonnan(condition, true_contrib, false_contrib);
    return FloatIf{res, res < max_minus_one};
}
inline uint32_t u32_if(bool condition, uint32_t if_true, uint32_t if_false) { // :1319-1326
    return if_true ^ ((if_true ^ if_false) & ((0u + (1u * (uint32_t)condition)) - 1u));
}
} // namespace branchless

// BvhTriangle (src/bvh.rs:16-38): the builder's per-triangle wrapper; centroid in the reference's operation order
struct BvhTriangle {
    Triangle triangle;
    size_t node_index;
};
namespace bvh_triangle {
inline BvhTriangle new_(const Triangle& t, size_t node_index) { return BvhTriangle{t, node_index}; } // :22-24
inline void centroid(const BvhTriangle& b, const Vertex* vertices, float out[3]) {                    // :27-37
    const float *v0 = vertices[b.triangle.v0_index].position, *v1 = vertices[b.triangle.v1_index].position, *v2 = vertices[b.triangle.v2_index].position;
    for (int i = 0; i < 3; i++) out[i] = (v0[i] + v1[i] + v2[i]) / 3.0f;
}
inline Aabb aabb(const BvhTriangle& b, const Vertex* vertices) { return triangle::bounding_box(b.triangle, vertices); } // BvhTriangleWithVertices::aabb, :47-55
} // namespace bvh_triangle

struct TriangleLegacy { // :133-140
    float v0[3];
    uint32_t material_id;
    float v1[3];
    uint32_t _padding0;
    float v2[3];
};
namespace triangle_legacy {
inline TriangleLegacy new_(const float v0[3], const float v1[3], const float v2[3], uint32_t material_id) { // :690-698
    TriangleLegacy t;
    std::memcpy(t.v0, v0, 12);
    t.material_id = material_id;
    std::memcpy(t.v1, v1, 12);
    t._padding0 = 0;
    std::memcpy(t.v2, v2, 12);
    return t;
}
// to_indexed :715-748: first-seen order, `==` on f32 (so -0.0 == 0.0 and NaN != NaN)
inline void to_indexed(const std::vector<TriangleLegacy>& tris, std::vector<Vertex>& vertices, std::vector<Triangle>& indexed) {
    vertices.clear();
    indexed.clear();
    auto find_or_add = [&](const float p[3]) -> uint32_t {
        for (size_t i = 0; i < vertices.size(); i++)
            if (vertices[i].position[0] == p[0] && vertices[i].position[1] == p[1] && vertices[i].position[2] == p[2]) return (uint32_t)i;
        vertices.push_back(Vertex{{p[0], p[1], p[2]}});
        return (uint32_t)vertices.size() - 1;
    };
    for (const auto& t : tris) {
        uint32_t a = find_or_add(t.v0), b = find_or_add(t.v1), c = find_or_add(t.v2);
        indexed.push_back(triangle::new_indexed(a, b, c, t.material_id));
    }
}
} // namespace triangle_legacy

namespace bvh_node {
inline BvhNode leaf(const Aabb& b, uint32_t start, uint32_t count) { return BvhNode{b, 0xFFFFFFFFu, 0xFFFFFFFFu, start, count}; } // :806-814
inline BvhNode internal(const Aabb& b, uint32_t l, uint32_t r) { return BvhNode{b, l, r, 0, 0}; }                                  // :817-825
inline bool is_leaf(const BvhNode& n) { return n.left_child == 0xFFFFFFFFu && n.right_child == 0xFFFFFFFFu; }                      // :828-830
} // namespace bvh_node

namespace wavefront_ray {
// WavefrontRay::new :835-859
inline WavefrontRay new_(const float origin[3], const float direction[3], uint32_t ray_type, uint32_t bounce_depth, const float throughput[3],
                         float medium_ior, const uint32_t pixel_coord[2], uint32_t wavelength_channel) {
    WavefrontRay r;
    std::memcpy(r.origin, origin, 12);
    r.ray_type = ray_type;
    std::memcpy(r.direction, direction, 12);
    r.bounce_depth = bounce_depth;
    std::memcpy(r.throughput, throughput, 12);
    r.medium_ior = medium_ior;
    r.pixel_coord[0] = pixel_coord[0];
    r.pixel_coord[1] = pixel_coord[1];
    r.inv_pdf = 1.0f;
    r.t_min = 0.001f;
    r.t_max = std::numeric_limits<float>::max();
    r.wavelength_channel = wavelength_channel;
    r.active = 1;
    return r;
}
inline WavefrontRay camera_ray(const float o[3], const float d[3], const uint32_t px[2], uint32_t channel) { // :862-878
    const float one[3] = {1.0f, 1.0f, 1.0f};
    return new_(o, d, 0, 0, one, 1.0f, px, channel);
}
inline WavefrontRay shadow_ray(const float o[3], const float d[3], float t_max, const uint32_t px[2], uint32_t channel) { // :935-956
    const float one[3] = {1.0f, 1.0f, 1.0f};
    WavefrontRay r = new_(o, d, 3, 0, one, 1.0f, px, channel);
    r.t_max = t_max;
    return r;
}
inline void deactivate(WavefrontRay& r) { r.active = 0; }              // :959-961
inline bool is_active(const WavefrontRay& r) { return r.active != 0; } // :964-966
inline void apply_russian_roulette(WavefrontRay& r, float continuation_probability, float rng_value) { // :969-978
    if (rng_value > continuation_probability) {
        deactivate(r);
    } else {
        r.throughput[0] /= continuation_probability;
        r.throughput[1] /= continuation_probability;
        r.throughput[2] /= continuation_probability;
    }
}
} // namespace wavefront_ray

namespace wavefront_counters {
inline WavefrontCounters new_(uint32_t max_bounce_depth, uint32_t frame_seed) { // :983-992
    WavefrontCounters c;
    std::memset(&c, 0, sizeof c);
    c.max_bounce_depth = max_bounce_depth;
    c.frame_seed = frame_seed;
    return c;
}
inline void add_rays(WavefrontCounters& c, uint32_t depth, uint32_t count) { // :1003-1009
    if (depth < 8) {
        c.rays_per_bounce[depth] += count;
        c.total_rays_generated += count;
        c.active_bounce_depths |= 1u << depth;
    }
}
} // namespace wavefront_counters

namespace push_constants {
inline uint32_t pack_tile_size(uint32_t w, uint32_t h) { return (std::min(w, 65535u) & 0xFFFFu) | ((std::min(h, 65535u) & 0xFFFFu) << 16); } // :1138-1142
inline void unpack_tile_size(const PushConstants& p, uint32_t* w, uint32_t* h) { *w = p.tile_size_packed & 0xFFFF; *h = (p.tile_size_packed >> 16) & 0xFFFF; } // :1146-1150
inline uint32_t pack_flags(uint32_t channel, uint32_t cur, uint32_t max, uint32_t mode) { // :1154-1159
    return (channel & 0xFF) | ((cur & 0xFF) << 8) | ((max & 0xFF) << 16) | ((mode & 0xFF) << 24);
}
inline uint32_t color_channel(const PushConstants& p) { return p.packed_flags & 0xFF; }                 // :1162-1164
inline uint32_t current_bounce_depth(const PushConstants& p) { return (p.packed_flags >> 8) & 0xFF; }   // :1167-1169
inline uint32_t max_bounce_depth(const PushConstants& p) { return (p.packed_flags >> 16) & 0xFF; }      // :1172-1174
inline uint32_t wavefront_mode(const PushConstants& p) { return (p.packed_flags >> 24) & 0xFF; }        // :1177-1179
// PushConstants::new :1076-1102
inline PushConstants new_(const float resolution[2], const Camera& cam, uint32_t triangle_count, uint32_t material_count,
                          const uint32_t tile_offset[2], const uint32_t tile_size[2], const uint32_t total_tiles[2],
                          uint32_t triangles_per_buffer, const SceneMetadataOffsets& offsets, uint32_t channel) {
    PushConstants p;
    p.resolution[0] = resolution[0];
    p.resolution[1] = resolution[1];
    p.camera = cam;
    p.triangle_count = triangle_count;
    p.material_count = material_count;
    p.tile_offset[0] = tile_offset[0];
    p.tile_offset[1] = tile_offset[1];
    p.tile_size_packed = pack_tile_size(tile_size[0], tile_size[1]);
    p.total_tiles[0] = total_tiles[0];
    p.total_tiles[1] = total_tiles[1];
    p.triangles_per_buffer = triangles_per_buffer;
    p.metadata_offsets = offsets;
    p.packed_flags = pack_flags(channel, 0, 4, 0);
    p.frame_seed = 0;
    return p;
}
// PushConstants::new_wavefront :1105-1134
inline PushConstants new_wavefront(const float resolution[2], const Camera& cam, uint32_t triangle_count, uint32_t material_count,
                                   const uint32_t tile_offset[2], const uint32_t tile_size[2], const uint32_t total_tiles[2],
                                   uint32_t triangles_per_buffer, const SceneMetadataOffsets& offsets, uint32_t channel,
                                   uint32_t current_bounce, uint32_t max_bounce, uint32_t frame_seed) {
    PushConstants p = new_(resolution, cam, triangle_count, material_count, tile_offset, tile_size, total_tiles, triangles_per_buffer, offsets, channel);
    p.packed_flags = pack_flags(channel, current_bounce, max_bounce, 1);
    p.frame_seed = frame_seed;
    return p;
}
} // namespace push_constants

struct TileHelper { // :1183-1204
    static void calculate_tile_count(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t* tx, uint32_t* ty) {
        *tx = (width + tile_size - 1) / tile_size;
        *ty = (height + tile_size - 1) / tile_size;
    }
    static uint32_t calculate_tiles_per_frame(uint32_t total) {
        uint32_t v;
        if (total <= 16) v = total;
        else if (total <= 64) v = total / 8;
        else if (total <= 256) v = total / 32;
        else if (total <= 1024) v = total / 64;
        else v = 1;
        return std::max(v, 1u);
    }
};

struct SceneBuilder { // :1208-1291
    static void build_default_scene(std::vector<Sphere>& spheres, std::vector<Triangle>& triangles, std::vector<Vertex>& vertices,
                                    std::vector<Material>& materials, std::vector<Light>& lights) {
        const float red[3] = {0.8f, 0.3f, 0.3f}, yellow[3] = {0.8f, 0.8f, 0.2f}, blue[3] = {0.2f, 0.3f, 0.8f}, white[3] = {1.0f, 1.0f, 1.0f};
        const float glow[3] = {0.5f, 0.5f, 1.0f};
        materials = {material::diffuse(red), material::metallic(yellow, 0.1f), material::glass(blue, 1.5f, 0.9f), material::emissive(white, glow)};
        spheres = {Sphere{{0.0f, 0.0f, -1.0f}, 0.5f, 0},  Sphere{{-1.0f, 0.0f, -1.0f}, 0.5f, 1}, Sphere{{1.0f, 0.0f, -1.0f}, 0.5f, 2},
                   Sphere{{2.0f, 0.0f, -3.0f}, 0.5f, 2},  Sphere{{-2.0f, 0.0f, -4.0f}, 0.5f, 1}, Sphere{{-1.0f, 2.0f, -5.0f}, 0.5f, 3}};
        const float a0[3] = {0.0f, 1.0f, -2.0f}, a1[3] = {-0.5f, 0.0f, -2.0f}, a2[3] = {0.5f, 0.0f, -2.0f};
        const float b0[3] = {1.5f, 0.5f, -3.0f}, b1[3] = {1.0f, -0.5f, -3.0f}, b2[3] = {2.0f, -0.5f, -3.0f};
        std::vector<TriangleLegacy> legacy = {triangle_legacy::new_(a0, a1, a2, 0), triangle_legacy::new_(b0, b1, b2, 1)};
        triangle_legacy::to_indexed(legacy, vertices, triangles);
        const float lp[3] = {5.0f, 7.0f, 4.0f};
        lights = {light::point(lp, white, 1.0f, std::numeric_limits<float>::infinity())};
    }
};

// ------------------------------------------------------------------------------------------
// BvhBuilder — src/bvh.rs:88-374.  Produces the REFERENCE-format BVH (48-byte nodes, root first):
//   empty -> one empty leaf (:105-114); > 100,000 triangles -> build_chunked (:154-247), exact;
//   otherwise one triangle per leaf, pre-order (left = parent + 1, :278-374) over our own
//   binned-SAH topology (the reference takes it from the un-vendored `bvh` crate: unpinned).
// The HIP kernels do not consume this structure (they use the library's own layout); it exists
// so a host that keeps src/scene.rs's flow still gets `bvh_nodes` / `triangle_indices`.
// ------------------------------------------------------------------------------------------
struct BvhResult {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> triangle_indices;
};

class BvhBuilder {
  public:
    static BvhResult build(const std::vector<Triangle>& triangles, const std::vector<Vertex>& vertices) {
        if (triangles.empty()) {
            BvhResult r;
            r.nodes.push_back(bvh_node::leaf(aabb::empty(), 0, 0));
            return r;
        }
        return triangles.size() > 100000 ? build_chunked(triangles, vertices) : build_standard(triangles, vertices);
    }
    // BvhBuilder::triangle_aabb, src/bvh.rs:272-275 (private there, exercised by its own test :510-523): the box build_chunked unions per triangle
    static Aabb triangle_aabb(const Triangle& t, const Vertex* vertices) { return triangle::bounding_box(t, vertices); }

  private:
    static BvhResult build_chunked(const std::vector<Triangle>& triangles, const std::vector<Vertex>& vertices) {
        BvhResult r;
        const size_t per_leaf = std::max<size_t>(triangles.size() / 10000, 32);
        std::vector<BvhNode> level;
        for (size_t base = 0; base < triangles.size(); base += per_leaf) {
            const size_t len = std::min(per_leaf, triangles.size() - base);
            Aabb box = aabb::empty();
            for (size_t i = 0; i < len; i++) box = aabb::union_(box, triangle_aabb(triangles[base + i], vertices.data()));
            const uint32_t start = (uint32_t)r.triangle_indices.size();
            for (size_t i = 0; i < len; i++) r.triangle_indices.push_back((uint32_t)(base + i));
            level.push_back(bvh_node::leaf(box, start, (uint32_t)len));
        }
        if (level.size() == 1) {
            r.nodes = level;
            return r;
        }
        // build_simple_top_level_bvh :192-247: pair neighbours bottom-up, children stored before parents,
        // then reverse so the root is node 0 and remap child indices.
        std::vector<BvhNode> store;
        while (level.size() > 1) {
            std::vector<BvhNode> parents;
            for (size_t i = 0; i < level.size(); i += 2) {
                const uint32_t first = (uint32_t)store.size();
                store.push_back(level[i]);
                if (i + 1 < level.size()) {
                    store.push_back(level[i + 1]);
                    parents.push_back(bvh_node::internal(aabb::union_(level[i].bounds, level[i + 1].bounds), first, first + 1));
                } else {
                    parents.push_back(bvh_node::internal(level[i].bounds, first, 0xFFFFFFFFu));
                }
            }
            level.swap(parents);
        }
        store.push_back(level[0]);
        std::reverse(store.begin(), store.end());
        const uint32_t last = (uint32_t)store.size() - 1;
        for (auto& n : store) {
            if (bvh_node::is_leaf(n)) continue;
            if (n.left_child != 0xFFFFFFFFu) n.left_child = last - n.left_child;
            if (n.right_child != 0xFFFFFFFFu) n.right_child = last - n.right_child;
        }
        r.nodes.swap(store);
        return r;
    }

    struct Prim {
        Aabb box;
        float c[3];
        uint32_t id;
    };

    static uint32_t emit(std::vector<Prim>& prims, size_t lo, size_t hi, BvhResult& out) {
        const uint32_t me = (uint32_t)out.nodes.size();
        if (hi - lo == 1) {
            const uint32_t start = (uint32_t)out.triangle_indices.size();
            out.triangle_indices.push_back(prims[lo].id);
            out.nodes.push_back(bvh_node::leaf(prims[lo].box, start, 1));
            return me;
        }
        Aabb box = aabb::empty();
        float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = lo; i < hi; i++) {
            box = aabb::union_(box, prims[i].box);
            for (int a = 0; a < 3; a++) {
                cmin[a] = std::fmin(cmin[a], prims[i].c[a]);
                cmax[a] = std::fmax(cmax[a], prims[i].c[a]);
            }
        }
        // binned SAH over the widest centroid axis (8 bins), object median when degenerate
        int axis = 0;
        for (int a = 1; a < 3; a++)
            if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
        size_t mid = lo + (hi - lo) / 2;
        const float ext = cmax[axis] - cmin[axis];
        bool split_done = false;
        if (ext > 0.0f && hi - lo > 4) {
            constexpr int B = 8;
            Aabb bb[B];
            size_t bc[B] = {0};
            for (auto& b : bb) b = aabb::empty();
            const float scale = (float)B / ext;
            auto bin_of = [&](const Prim& p) { return std::min(B - 1, std::max(0, (int)((p.c[axis] - cmin[axis]) * scale))); };
            for (size_t i = lo; i < hi; i++) {
                const int b = bin_of(prims[i]);
                bb[b] = aabb::union_(bb[b], prims[i].box);
                bc[b]++;
            }
            float best = INFINITY;
            int best_b = -1;
            for (int s = 0; s < B - 1; s++) {
                Aabb l = aabb::empty(), r = aabb::empty();
                size_t nl = 0, nr = 0;
                for (int b = 0; b <= s; b++) { l = aabb::union_(l, bb[b]); nl += bc[b]; }
                for (int b = s + 1; b < B; b++) { r = aabb::union_(r, bb[b]); nr += bc[b]; }
                if (!nl || !nr) continue;
                const float cost = aabb::surface_area(l) * (float)nl + aabb::surface_area(r) * (float)nr;
                if (cost < best) { best = cost; best_b = s; }
            }
            if (best_b >= 0) {
                auto it = std::partition(prims.begin() + (long)lo, prims.begin() + (long)hi, [&](const Prim& p) { return bin_of(p) <= best_b; });
                mid = (size_t)(it - prims.begin());
                split_done = mid > lo && mid < hi;
            }
        }
        if (!split_done) {
            mid = lo + (hi - lo) / 2;
            std::nth_element(prims.begin() + (long)lo, prims.begin() + (long)mid, prims.begin() + (long)hi, [&](const Prim& a, const Prim& b) {
                return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.id < b.id);
            });
        }
        out.nodes.push_back(bvh_node::internal(box, 0, 0));
        const uint32_t l = emit(prims, lo, mid, out);
        const uint32_t r = emit(prims, mid, hi, out);
        out.nodes[me] = bvh_node::internal(box, l, r);
        return me;
    }

    static BvhResult build_standard(const std::vector<Triangle>& triangles, const std::vector<Vertex>& vertices) {
        std::vector<Prim> prims(triangles.size());
        for (size_t i = 0; i < triangles.size(); i++) {
            prims[i].box = triangle::bounding_box(triangles[i], vertices.data());
            aabb::center(prims[i].box, prims[i].c);
            prims[i].id = (uint32_t)i;
        }
        BvhResult r;
        r.nodes.reserve(2 * triangles.size());
        emit(prims, 0, prims.size(), r);
        return r;
    }
};

// ------------------------------------------------------------------------------------------
// SceneState — src/scene.rs:8-119 (the host Vecs the uploads read from)
// ------------------------------------------------------------------------------------------
struct SceneState {
    std::vector<Sphere> spheres;
    std::vector<Triangle> triangles;
    std::vector<Vertex> vertices;
    std::vector<Material> materials;
    std::vector<Light> lights;
    std::vector<BvhNode> bvh_nodes;
    std::vector<uint32_t> triangle_indices;
    Camera camera;

    static SceneState new_() { // :20-40: default scene + its BVH
        SceneState s;
        SceneBuilder::build_default_scene(s.spheres, s.triangles, s.vertices, s.materials, s.lights);
        s.camera = camera::new_();
        s.rebuild_bvh();
        return s;
    }
    void rebuild_bvh() { // :122-127
        BvhResult r = BvhBuilder::build(triangles, vertices);
        bvh_nodes.swap(r.nodes);
        triangle_indices.swap(r.triangle_indices);
    }
};

// ------------------------------------------------------------------------------------------
// BufferManager — src/buffers.rs.  Packs binding 1 exactly as update_scene_metadata does
// (:213-268) and splits triangles as update_triangles does (:274-336), then hands the bytes to
// rt_upload_scene_packed.  Growth policy and dirty flags are GPU-allocation details of wgpu and
// are not reproduced; `needs_update` semantics are: upload when marked dirty or counts changed.
// ------------------------------------------------------------------------------------------
class BufferManager {
  public:
    uint32_t triangles_per_buffer = RT_REF_TRIANGLES_PER_BUFFER; // max_triangles_per_buffer :49-53
    bool scene_metadata_dirty = true, triangles_dirty = true, materials_dirty = true;

    static SceneMetadataOffsets pack_scene_metadata(const std::vector<Sphere>& spheres, const std::vector<Light>& lights,
                                                    const std::vector<BvhNode>& nodes, const std::vector<uint32_t>& tri_indices,
                                                    const std::vector<Vertex>& vertices, std::vector<uint32_t>& combined) {
        const size_t ws = spheres.size() * sizeof(Sphere) / 4, wl = lights.size() * sizeof(Light) / 4, wn = nodes.size() * sizeof(BvhNode) / 4,
                     wi = tri_indices.size(), wv = vertices.size() * sizeof(Vertex) / 4;
        combined.resize(ws + wl + wn + wi + wv);
        uint32_t* p = combined.data();
        if (ws) std::memcpy(p, spheres.data(), ws * 4);
        if (wl) std::memcpy(p + ws, lights.data(), wl * 4);
        if (wn) std::memcpy(p + ws + wl, nodes.data(), wn * 4);
        if (wi) std::memcpy(p + ws + wl + wn, tri_indices.data(), wi * 4);
        if (wv) std::memcpy(p + ws + wl + wn + wi, vertices.data(), wv * 4);
        return SceneMetadataOffsets{0, (uint32_t)spheres.size(), (uint32_t)ws, (uint32_t)lights.size(), (uint32_t)(ws + wl), (uint32_t)nodes.size(),
                                    (uint32_t)(ws + wl + wn), (uint32_t)tri_indices.size(), (uint32_t)(ws + wl + wn + wi), (uint32_t)vertices.size()};
    }

    bool needs_update() const { return scene_metadata_dirty || triangles_dirty || materials_dirty; } // :473-501
    void mark_all_dirty() { scene_metadata_dirty = triangles_dirty = materials_dirty = true; }

    // update_scene_metadata + update_triangles + update_materials in one upload; returns RT_OK or an rt error code
    int update(rt_ctx* ctx, const SceneState& scene, SceneMetadataOffsets* offsets_out) {
        std::vector<uint32_t> combined;
        offsets_ = pack_scene_metadata(scene.spheres, scene.lights, scene.bvh_nodes, scene.triangle_indices, scene.vertices, combined);
        if (offsets_out) *offsets_out = offsets_;
        bool counts_changed = scene.triangles.size() != last_triangles_ || scene.materials.size() != last_materials_ || combined.size() != last_words_;
        if (!needs_update() && !counts_changed) return RT_OK;
        const rt_triangle* bufs[3] = {nullptr, nullptr, nullptr};
        uint32_t counts[3] = {0, 0, 0};
        for (size_t b = 0; b < 3; b++) {
            const size_t start = b * (size_t)triangles_per_buffer;
            if (start >= scene.triangles.size()) break;
            bufs[b] = scene.triangles.data() + start;
            counts[b] = (uint32_t)std::min<size_t>(triangles_per_buffer, scene.triangles.size() - start);
        }
        int rc = rt_upload_scene_packed(ctx, combined.data(), combined.size(), &offsets_, bufs, counts, triangles_per_buffer,
                                        scene.materials.data(), (uint32_t)scene.materials.size());
        if (rc != RT_OK) return rc;
        scene_metadata_dirty = triangles_dirty = materials_dirty = false;
        last_triangles_ = scene.triangles.size();
        last_materials_ = scene.materials.size();
        last_words_ = combined.size();
        return RT_OK;
    }
    const SceneMetadataOffsets& offsets() const { return offsets_; }

  private:
    SceneMetadataOffsets offsets_{};
    size_t last_triangles_ = (size_t)-1, last_materials_ = (size_t)-1, last_words_ = (size_t)-1;
};

// ------------------------------------------------------------------------------------------
// ProgressiveState + ComputeRenderer — src/renderer.rs:821-855, src/compute.rs:12-251.
// run_compute processes `tiles_per_frame` tiles per call (TileHelper policy), each tile as three
// colour-channel dispatches with PushConstants::new — exactly the reference's dispatch sequence,
// each dispatch being one rt_dispatch_tile.
// ------------------------------------------------------------------------------------------
struct ProgressiveState {
    uint32_t width = 0, height = 0, tiles_x = 0, tiles_y = 0, current_tile = 0, tiles_per_frame = 1;
    bool needs_recompute = true, is_progressive_rendering = false;
    void resize(uint32_t w, uint32_t h) {
        width = w;
        height = h;
        TileHelper::calculate_tile_count(w, h, RaytracerConfig::TILE_SIZE, &tiles_x, &tiles_y);
        tiles_per_frame = TileHelper::calculate_tiles_per_frame(tiles_x * tiles_y);
        needs_recompute = true;
        is_progressive_rendering = false;
        current_tile = 0;
    }
};

struct ComputeRenderer {
    // Returns RT_OK, or the failing call's error code.  *done is set when the image is complete.
    static int run_compute(rt_ctx* ctx, BufferManager& buffers, const SceneState& scene, ProgressiveState& progressive, bool* done) {
        if (done) *done = false;
        if (progressive.needs_recompute && !progressive.is_progressive_rendering) { // handle_progressive_rendering_setup :53-82
            progressive.is_progressive_rendering = true;
            progressive.current_tile = 0;
            progressive.needs_recompute = false;
        }
        if (!progressive.is_progressive_rendering) {
            if (done) *done = true;
            return RT_OK;
        }
        const uint32_t total = progressive.tiles_x * progressive.tiles_y;
        if (progressive.current_tile >= total) { // check_rendering_completion :85-100
            progressive.is_progressive_rendering = false;
            if (done) *done = true;
            return RT_OK;
        }
        const uint32_t tiles_this_frame = std::min(progressive.tiles_per_frame, total - progressive.current_tile); // :103-106
        SceneMetadataOffsets offsets;
        int rc = buffers.update(ctx, scene, &offsets); // update_buffers_and_bind_groups :109-134
        if (rc != RT_OK) return rc;
        for (uint32_t i = 0; i < tiles_this_frame; i++) { // execute_compute_pass :137-166 / process_tile :169-191
            const uint32_t tile = progressive.current_tile + i;
            const uint32_t tx = tile % progressive.tiles_x, ty = tile / progressive.tiles_x; // calculate_tile_dimensions :194-209
            const uint32_t off[2] = {tx * RaytracerConfig::TILE_SIZE, ty * RaytracerConfig::TILE_SIZE};
            const uint32_t size[2] = {std::min(RaytracerConfig::TILE_SIZE, progressive.width - off[0]),
                                      std::min(RaytracerConfig::TILE_SIZE, progressive.height - off[1])};
            const uint32_t total_tiles[2] = {progressive.tiles_x, progressive.tiles_y};
            const float res[2] = {(float)progressive.width, (float)progressive.height};
            for (uint32_t channel = 0; channel < 3; channel++) { // process_color_channel :212-251
                PushConstants pc = push_constants::new_(res, scene.camera, (uint32_t)scene.triangles.size(), (uint32_t)scene.materials.size(), off,
                                                        size, total_tiles, buffers.triangles_per_buffer, offsets, channel);
                rc = rt_dispatch_tile(ctx, &pc);
                if (rc != RT_OK) return rc;
            }
        }
        progressive.current_tile += tiles_this_frame;
        if (progressive.current_tile >= total) {
            progressive.is_progressive_rendering = false;
            if (done) *done = true;
        }
        return RT_OK;
    }
};

} // namespace raytracer
#endif
