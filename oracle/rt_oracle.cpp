// rt_oracle.cpp — CPU oracle: scalar restatement of the reference kernel `main_cs`.
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle.h): loaded by tests/, smoke() and the
// cpu_baseline leg of bench.py; never by the product library.
//
// Every function cites the reference file:line it follows.  Arithmetic is kept
// in the reference's operation order (glam scalar-math: dot = x*x + y*y + z*z
// left to right; normalize = v * (1 / sqrt(dot))), compiled with
// -ffp-contract=off so no multiply-add is fused behind our back.  Rust
// f32::min/max are NaN-suppressing: fminf/fmaxf everywhere, never a<b?a:b.
//
// Parity pinning: no reference vectors exist for this path;
// pinned by the hand-derived KATs in tests/test_oracle_kat.py.

#include "rt_oracle.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

namespace {

struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
// glam Vec3::dot (scalar-math): (x*x) + (y*y) + (z*z)
inline float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
// glam Vec3::cross
inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline float length(V3 a) { return sqrtf(dot(a, a)); }
// glam Vec3::normalize = self * self.length_recip(), length_recip = 1.0 / length
inline V3 normalize(V3 a) { return a * (1.0f / length(a)); }
inline V3 vmin(V3 a, V3 b) { return v3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return v3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }

inline float bits_f32(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline uint32_t f32_bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

// IEEE binary16 <-> binary32.  f16_to_f32 is exact (UnpackHalf2x16); f32_to_f16
// rounds to nearest even (PackHalf2x16's rounding is implementation defined; RNE
// is what `half::f16::from_f32` does on the host, shared/src/lib.rs:250-252).
float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h >> 15) << 31;
    uint32_t exp = (h >> 10) & 0x1F;
    uint32_t frac = h & 0x3FF;
    if (exp == 0) {
        if (frac == 0) return bits_f32(sign);
        // subnormal: value = frac * 2^-24
        float v = (float)frac * (1.0f / 16777216.0f);
        return sign ? -v : v;
    }
    if (exp == 31) return bits_f32(sign | 0x7F800000u | (frac << 13));
    return bits_f32(sign | ((exp + 112) << 23) | (frac << 13));
}
uint16_t f32_to_f16(float v) {
    uint32_t u = f32_bits(v);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t absu = u & 0x7FFFFFFFu;
    if (absu >= 0x7F800000u) {                       // inf / nan
        return (uint16_t)(sign | 0x7C00u | ((absu > 0x7F800000u) ? (0x200u | ((absu >> 13) & 0x3FFu)) : 0));
    }
    if (absu >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u); // rounds to >= 65520 -> inf
    if (absu < 0x33000001u) return (uint16_t)sign;              // < 2^-25 (or == 2^-25 ties to even 0)
    int32_t e = (int32_t)(absu >> 23) - 127;
    uint32_t m = (absu & 0x7FFFFFu) | 0x800000u;
    uint32_t shift, half_exp;
    if (e < -14) { // subnormal half
        shift = (uint32_t)(13 + (-14 - e));
        half_exp = 0;
    } else {
        shift = 13;
        half_exp = (uint32_t)(e + 15);
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    // q includes the implicit bit for normals: (half_exp << 10) + (q - 0x400) ; carries propagate into exp
    uint32_t out = (half_exp == 0) ? q : ((half_exp << 10) + (q - 0x400u));
    return (uint16_t)(sign | out);
}

struct Counters {
    uint64_t rays = 0, node_visits = 0, tri_tests = 0, sphere_tests = 0, stack_drops = 0, oob_reads = 0;
    void add_to(oracle_counters* o) const {
        if (!o) return;
        o->rays += rays;
        o->node_visits += node_visits;
        o->tri_tests += tri_tests;
        o->sphere_tests += sphere_tests;
        o->stack_drops += stack_drops;
        o->oob_reads += oob_reads;
    }
};

// ---------------------------------------------------------------------------------
// SceneAccessor — shader/src/scene_access.rs:5-178.  Decodes binding 1 by u32 offsets.
// Out-of-range reads return 0 (storage-buffer robustness) and are counted.
// ---------------------------------------------------------------------------------
struct SceneAccessor {
    const uint32_t* md;
    uint64_t md_len;
    const rt_push_constants* pc;
    Counters* c;

    uint32_t word(uint64_t i) const {
        if (i >= md_len) {
            c->oob_reads++;
            return 0;
        }
        return md[i];
    }
    float wordf(uint64_t i) const { return bits_f32(word(i)); }

    uint32_t sphere_count() const { return pc->metadata_offsets.spheres_count; }      // :20-22
    uint32_t light_count() const { return pc->metadata_offsets.lights_count; }        // :25-27
    uint32_t bvh_node_count() const { return pc->metadata_offsets.bvh_nodes_count; }  // :30-32

    uint64_t sphere_base(uint32_t i) const { return (uint64_t)pc->metadata_offsets.spheres_offset + (uint64_t)i * RT_SPHERE_WORDS; }
    V3 sphere_center(uint32_t i) const { uint64_t o = sphere_base(i); return v3(wordf(o), wordf(o + 1), wordf(o + 2)); } // :34-43
    float sphere_radius(uint32_t i) const { return wordf(sphere_base(i) + 3); }       // :46-50
    uint32_t sphere_material(uint32_t i) const { return word(sphere_base(i) + 4); }   // :53-57

    uint64_t light_base(uint32_t i) const { return (uint64_t)pc->metadata_offsets.lights_offset + (uint64_t)i * RT_LIGHT_WORDS; }
    V3 light_position(uint32_t i) const { uint64_t o = light_base(i); return v3(wordf(o), wordf(o + 1), wordf(o + 2)); }       // :60-69
    uint32_t light_type(uint32_t i) const { return word(light_base(i) + 3); }                                                   // :72-76
    V3 light_color(uint32_t i) const { uint64_t o = light_base(i); return v3(wordf(o + 4), wordf(o + 5), wordf(o + 6)); }       // :79-88
    float light_intensity(uint32_t i) const { return wordf(light_base(i) + 7); }                                                // :91-95
    V3 light_direction(uint32_t i) const { uint64_t o = light_base(i); return v3(wordf(o + 8), wordf(o + 9), wordf(o + 10)); }  // :98-107

    uint64_t node_base(uint32_t i) const { return (uint64_t)pc->metadata_offsets.bvh_nodes_offset + (uint64_t)i * RT_BVH_NODE_WORDS; }
    V3 node_min(uint32_t i) const { uint64_t o = node_base(i); return v3(wordf(o), wordf(o + 1), wordf(o + 2)); }               // :110-119
    V3 node_max(uint32_t i) const { uint64_t o = node_base(i); return v3(wordf(o + 4), wordf(o + 5), wordf(o + 6)); }           // :122-131
    uint32_t node_left(uint32_t i) const { return word(node_base(i) + 8); }                                                     // :134-138
    uint32_t node_right(uint32_t i) const { return word(node_base(i) + 9); }                                                    // :141-145
    uint32_t node_tri_start(uint32_t i) const { return word(node_base(i) + 10); }                                               // :148-152
    uint32_t node_tri_count(uint32_t i) const { return word(node_base(i) + 11); }                                               // :155-159

    uint32_t triangle_index(uint32_t i) const { return word((uint64_t)pc->metadata_offsets.triangle_indices_offset + i); }      // :162-166
    V3 vertex_position(uint32_t i) const {                                                                                      // :169-178
        uint64_t o = (uint64_t)pc->metadata_offsets.vertices_offset + (uint64_t)i * RT_VERTEX_WORDS;
        return v3(wordf(o), wordf(o + 1), wordf(o + 2));
    }
};

// PushConstants accessors — shared/src/lib.rs:1146-1179
inline uint32_t pc_channel(const rt_push_constants* pc) { return pc->packed_flags & 0xFF; }
inline uint32_t pc_cur_bounce(const rt_push_constants* pc) { return (pc->packed_flags >> 8) & 0xFF; }
inline uint32_t pc_max_bounce(const rt_push_constants* pc) { return (pc->packed_flags >> 16) & 0xFF; }
inline uint32_t pc_mode(const rt_push_constants* pc) { return (pc->packed_flags >> 24) & 0xFF; }
inline void pc_tile_size(const rt_push_constants* pc, uint32_t* w, uint32_t* h) {
    *w = pc->tile_size_packed & 0xFFFF;
    *h = (pc->tile_size_packed >> 16) & 0xFFFF;
}
// Rust `f32 as u32` saturates and maps NaN to 0.
inline uint32_t f32_as_u32(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

struct Ray {
    V3 origin, direction;
};
inline V3 ray_at(const Ray& r, float t) { return r.origin + r.direction * t; } // shader/src/ray.rs:56-58

struct Intersection { // shader/src/intersection.rs:10-15
    float t;
    V3 point, normal;
    uint32_t material_id;
    uint32_t prim_id; // not in the reference: original triangle index / 0x80000000|sphere, for index parity
};
struct IntersectionResult { // :19-22
    bool hit;
    Intersection is;
};
inline IntersectionResult miss() { // :26-30
    return IntersectionResult{false, Intersection{3.402823466e+38f, v3(0, 0, 0), v3(0, 0, 0), 0xFFFFFFFFu, 0xFFFFFFFFu}};
}

// Ray::from_screen_coordinates — shader/src/ray.rs:22-53 (+ Ray::new :14-19 normalises again)
Ray ray_from_screen(uint32_t px, uint32_t py, const rt_push_constants* pc) {
    uint32_t width = f32_as_u32(pc->resolution[0]);
    uint32_t height = f32_as_u32(pc->resolution[1]);
    float u = ((float)px + 0.5f) / (float)width;
    float v = ((float)py + 0.5f) / (float)height;
    float aspect = (float)width / (float)height;
    float fov_scale = tanf(pc->camera.fov * 0.5f * 3.14159265358979323846f / 180.0f);
    float cx = (u * 2.0f - 1.0f) * aspect * fov_scale;
    float cy = (1.0f - v * 2.0f) * fov_scale;
    V3 fwd = v3(pc->camera.direction[0], pc->camera.direction[1], pc->camera.direction[2]);
    V3 up = v3(pc->camera.up[0], pc->camera.up[1], pc->camera.up[2]);
    V3 right = cross(fwd, up);
    V3 true_up = cross(right, fwd);
    V3 dir = fwd + right * cx + true_up * cy;
    V3 dirn = normalize(dir);
    V3 origin = v3(pc->camera.position[0], pc->camera.position[1], pc->camera.position[2]);
    return Ray{origin, normalize(dirn)};
}

// generate_camera_ray — shader/src/wavefront.rs:75-112 (resolution used as f32, one normalise;
// Ray::from_wavefront_ray :36-41 does not re-normalise)
Ray ray_from_screen_wavefront(uint32_t px, uint32_t py, const rt_push_constants* pc) {
    float u = ((float)px + 0.5f) / pc->resolution[0];
    float v = ((float)py + 0.5f) / pc->resolution[1];
    float aspect = pc->resolution[0] / pc->resolution[1];
    float fov_scale = tanf(pc->camera.fov * 0.5f * 3.14159265358979323846f / 180.0f);
    float cx = (u * 2.0f - 1.0f) * aspect * fov_scale;
    float cy = (1.0f - v * 2.0f) * fov_scale;
    V3 fwd = v3(pc->camera.direction[0], pc->camera.direction[1], pc->camera.direction[2]);
    V3 up = v3(pc->camera.up[0], pc->camera.up[1], pc->camera.up[2]);
    V3 right = cross(fwd, up);
    V3 true_up = cross(right, fwd);
    V3 dir = fwd + right * cx + true_up * cy;
    V3 origin = v3(pc->camera.position[0], pc->camera.position[1], pc->camera.position[2]);
    return Ray{origin, normalize(dir)};
}

// test_sphere_intersection — shader/src/intersection.rs:52-87
IntersectionResult test_sphere(const Ray& ray, const SceneAccessor& sa, uint32_t idx, float max_t) {
    sa.c->sphere_tests++;
    V3 center = sa.sphere_center(idx);
    float radius = sa.sphere_radius(idx);
    V3 oc = ray.origin - center;
    float a = dot(ray.direction, ray.direction);
    float b = 2.0f * dot(oc, ray.direction);
    float c = dot(oc, oc) - radius * radius;
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return miss();
    float sq = sqrtf(disc);
    float t1 = (-b - sq) / (2.0f * a);
    float t2 = (-b + sq) / (2.0f * a);
    float t = (t1 > RT_MIN_RAY_DISTANCE) ? t1 : t2;
    if (t > RT_MIN_RAY_DISTANCE && t < max_t) {
        V3 p = ray_at(ray, t);
        V3 n = normalize(p - center);
        return IntersectionResult{true, Intersection{t, p, n, sa.sphere_material(idx), 0x80000000u | idx}};
    }
    return miss();
}

// test_triangle_intersection_direct — shader/src/intersection.rs:91-138 (Möller–Trumbore)
IntersectionResult test_triangle(const Ray& ray, V3 v0, V3 v1, V3 v2, uint32_t material_id, uint32_t prim_id, float max_t) {
    V3 e1 = v1 - v0;
    V3 e2 = v2 - v0;
    V3 h = cross(ray.direction, e2);
    float a = dot(e1, h);
    if (fabsf(a) < RT_MIN_RAY_DISTANCE) return miss();
    float f = 1.0f / a;
    V3 s = ray.origin - v0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return miss();
    V3 q = cross(s, e1);
    float v = f * dot(ray.direction, q);
    if (v < 0.0f || u + v > 1.0f) return miss();
    float t = f * dot(e2, q);
    if (t > RT_MIN_RAY_DISTANCE && t < max_t) {
        V3 p = ray_at(ray, t);
        V3 n = normalize(cross(e1, e2));
        return IntersectionResult{true, Intersection{t, p, n, material_id, prim_id}};
    }
    return miss();
}

// ray_aabb_intersect — shader/src/intersection.rs:151-164
bool ray_aabb(V3 o, V3 d, V3 bmin, V3 bmax) {
    V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    V3 t1 = (bmin - o) * inv;
    V3 t2 = (bmax - o) * inv;
    V3 tmin = vmin(t1, t2);
    V3 tmax = vmax(t1, t2);
    float tmin_max = fmaxf(fmaxf(tmin.x, tmin.y), tmin.z);
    float tmax_min = fminf(fminf(tmax.x, tmax.y), tmax.z);
    return tmax_min >= 0.0f && tmin_max <= tmax_min;
}

struct Kernel {
    const oracle_bindings* b;
    const rt_push_constants* pc;
    SceneAccessor sa;
    Counters* c;
    bool fast = false; // baseline flavour (ii), see traverse_fast

    // TriangleAccessor::get_triangle_vertices_direct — shader/src/triangle_access.rs:18-60
    bool get_triangle(uint32_t index, V3* v0, V3* v1, V3* v2, uint32_t* mat) const {
        if (pc->triangles_per_buffer == 0) return false; // division by zero is UB on the device; treat as invalid
        uint32_t buffer_index = index / pc->triangles_per_buffer;
        uint64_t local = index % pc->triangles_per_buffer;
        if (buffer_index < 3 && local < b->triangles_len[buffer_index] && b->triangles[buffer_index]) {
            const rt_triangle& t = b->triangles[buffer_index][local];
            *v0 = sa.vertex_position(t.v0_index);
            *v1 = sa.vertex_position(t.v1_index);
            *v2 = sa.vertex_position(t.v2_index);
            *mat = t.material_id;
            return true;
        }
        return false;
    }

    // test_leaf_triangles — shader/src/bvh.rs:91-133
    IntersectionResult test_leaf(const Ray& ray, uint32_t node, float max_t) const {
        uint32_t start = sa.node_tri_start(node);
        uint32_t count = sa.node_tri_count(node);
        IntersectionResult result = miss();
        float closest_t = max_t;
        for (uint32_t i = 0; i < count; i++) {
            if (start + i >= pc->metadata_offsets.triangle_indices_count) break;
            uint32_t tri = sa.triangle_index(start + i);
            V3 v0, v1, v2;
            uint32_t mat;
            if (get_triangle(tri, &v0, &v1, &v2, &mat)) {
                c->tri_tests++;
                IntersectionResult r = test_triangle(ray, v0, v1, v2, mat, tri, closest_t);
                if (r.hit) {
                    closest_t = r.is.t;
                    result = r;
                }
            }
        }
        return result;
    }

    // BvhTraverser::traverse_and_intersect — shader/src/bvh.rs:18-88
    IntersectionResult traverse(const Ray& ray, float max_t) const {
        if (fast) return traverse_fast(ray, max_t, false);
        if (sa.bvh_node_count() == 0) return miss();
        IntersectionResult result = miss();
        float closest_t = max_t;
        uint32_t stack[64];
        for (int i = 0; i < 64; i++) stack[i] = 0xFFFFFFFFu;
        int sp = 0;
        stack[0] = 0;
        sp += 1;
        while (sp > 0) {
            sp -= 1;
            uint32_t node = stack[sp];
            if (node == 0xFFFFFFFFu || node >= sa.bvh_node_count()) continue;
            c->node_visits++;
            V3 bmin = sa.node_min(node);
            V3 bmax = sa.node_max(node);
            if (!ray_aabb(ray.origin, ray.direction, bmin, bmax)) continue;
            uint32_t left = sa.node_left(node);
            uint32_t right = sa.node_right(node);
            if (left == 0xFFFFFFFFu) {
                IntersectionResult leaf = test_leaf(ray, node, closest_t);
                if (leaf.hit) {
                    closest_t = leaf.is.t;
                    result = leaf;
                }
            } else {
                if (right != 0xFFFFFFFFu) {
                    if (sp < 63) stack[sp++] = right;
                    else c->stack_drops++;
                }
                if (left != 0xFFFFFFFFu) {
                    if (sp < 63) stack[sp++] = left;
                    else c->stack_drops++;
                }
            }
        }
        return result;
    }

    // ---- Baseline flavour (ii) of SURVEY 8(d): NOT the reference's algorithm.  The same node format walked the way
    // a CPU ray tracer would: both children slab-tested at the parent against the closest hit so far, near child
    // first, far child pushed.  Used only for the second CPU timing of bench.py ("what the host cores can do with a
    // decent traversal") and checked to give the same image: equal t resolves to the lower triangle index, which
    // is what the reference's own visiting order yields on its chunked BVH and on the brute-force path.
    static bool slab_entry(V3 o, V3 inv, V3 bmin, V3 bmax, float limit, float* entry) {
        V3 t1 = (bmin - o) * inv;
        V3 t2 = (bmax - o) * inv;
        V3 tmin = vmin(t1, t2);
        V3 tmax = vmax(t1, t2);
        float tmin_max = fmaxf(fmaxf(tmin.x, tmin.y), tmin.z);
        float tmax_min = fminf(fminf(tmax.x, tmax.y), tmax.z);
        if (!(tmax_min >= 0.0f && tmin_max <= tmax_min)) return false;
        float e = fmaxf(tmin_max, 0.0f);
        if (e > limit) return false; // <=: candidates at exactly the closest distance are still visited (tie rule)
        *entry = e;
        return true;
    }
    IntersectionResult traverse_fast(const Ray& ray, float max_t, bool any_hit) const {
        const uint32_t n_nodes = sa.bvh_node_count();
        if (n_nodes == 0) return miss();
        const V3 o = ray.origin, d = ray.direction;
        const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        IntersectionResult result = miss();
        float closest_t = max_t, e;
        uint32_t stack[128];
        int sp = 0;
        uint32_t node = 0;
        c->node_visits++;
        if (!slab_entry(o, inv, sa.node_min(0), sa.node_max(0), closest_t, &e)) return result;
        for (;;) {
            const uint32_t left = sa.node_left(node), right = sa.node_right(node);
            bool descend = false;
            if (left == 0xFFFFFFFFu) {
                const uint32_t start = sa.node_tri_start(node), count = sa.node_tri_count(node);
                for (uint32_t i = 0; i < count; i++) {
                    if (start + i >= pc->metadata_offsets.triangle_indices_count) break;
                    const uint32_t tri = sa.triangle_index(start + i);
                    V3 v0, v1, v2;
                    uint32_t mat;
                    if (!get_triangle(tri, &v0, &v1, &v2, &mat)) continue;
                    c->tri_tests++;
                    IntersectionResult r = test_triangle(ray, v0, v1, v2, mat, tri, 3.402823466e+38f);
                    if (!r.hit) continue;
                    if (r.is.t < closest_t || (result.hit && r.is.t == closest_t && tri < result.is.prim_id)) {
                        closest_t = r.is.t;
                        result = r;
                        if (any_hit) return result;
                    }
                }
            } else {
                float el = 0.0f, er = 0.0f;
                const bool hl = left < n_nodes && slab_entry(o, inv, sa.node_min(left), sa.node_max(left), closest_t, &el);
                const bool hr = right != 0xFFFFFFFFu && right < n_nodes && slab_entry(o, inv, sa.node_min(right), sa.node_max(right), closest_t, &er);
                c->node_visits += 2;
                if (hl && hr) {
                    const bool left_first = el <= er;
                    if (sp < 128) stack[sp++] = left_first ? right : left;
                    node = left_first ? left : right;
                    descend = true;
                } else if (hl || hr) {
                    node = hl ? left : right;
                    descend = true;
                }
            }
            if (descend) continue;
            // next subtree whose box is still within reach (the closest hit may have moved since it was pushed)
            bool found = false;
            while (sp > 0) {
                node = stack[--sp];
                if (slab_entry(o, inv, sa.node_min(node), sa.node_max(node), closest_t, &e)) {
                    found = true;
                    break;
                }
            }
            if (!found) break;
        }
        return result;
    }
    // any primitive strictly inside (MIN_RAY_DISTANCE, tmax)?  Same answer as `find_closest(ray).t < tmax`.
    bool any_hit(const Ray& ray, float tmax) const {
        c->rays++;
        for (uint32_t i = 0; i < sa.sphere_count(); i++)
            if (test_sphere(ray, sa, i, tmax).hit) return true;
        if (pc->metadata_offsets.bvh_nodes_count > 0) return traverse_fast(ray, tmax, true).hit;
        return brute_force(ray, tmax).hit;
    }

    // test_all_triangles_brute_force — shader/src/lib.rs:272-296
    IntersectionResult brute_force(const Ray& ray, float max_t) const {
        IntersectionResult result = miss();
        float closest_t = max_t;
        for (uint32_t i = 0; i < pc->triangle_count; i++) {
            V3 v0, v1, v2;
            uint32_t mat;
            if (get_triangle(i, &v0, &v1, &v2, &mat)) {
                c->tri_tests++;
                IntersectionResult r = test_triangle(ray, v0, v1, v2, mat, i, closest_t);
                if (r.hit) {
                    closest_t = r.is.t;
                    result = r;
                }
            }
        }
        return result;
    }

    // test_sphere_intersections — shader/src/lib.rs:252-269
    IntersectionResult test_spheres(const Ray& ray, float max_t) const {
        IntersectionResult result = miss();
        float closest_t = max_t;
        for (uint32_t i = 0; i < sa.sphere_count(); i++) {
            IntersectionResult r = test_sphere(ray, sa, i, closest_t);
            if (r.hit) {
                closest_t = r.is.t;
                result = r;
            }
        }
        return result;
    }

    // find_closest_intersection — shader/src/lib.rs:174-249 (dup wavefront.rs:214-289)
    IntersectionResult find_closest(const Ray& ray) const {
        c->rays++;
        float closest_t = 3.402823466e+38f - 2.0f; // f32::MAX - 2.0 == f32::MAX
        IntersectionResult sphere = test_spheres(ray, closest_t);
        if (sphere.hit) closest_t = sphere.is.t;
        IntersectionResult tri = (pc->metadata_offsets.bvh_nodes_count > 0) ? traverse(ray, closest_t) : brute_force(ray, closest_t);
        // the branchless_u32_if selection of :214-248, written out
        uint32_t sphere_closer = (sphere.is.t < tri.is.t) ? 1u : 0u;
        uint32_t both_hit = (tri.hit && sphere.hit) ? 1u : 0u;
        uint32_t sphere_only = (sphere.hit && !tri.hit) ? 1u : 0u;
        uint32_t triangle_only = (tri.hit && !sphere.hit) ? 1u : 0u;
        uint32_t use_sphere = both_hit * sphere_closer + sphere_only;
        uint32_t use_triangle = both_hit * (1u - sphere_closer) + triangle_only;
        if (use_sphere != 0) return sphere;
        if (use_triangle != 0) return tri;
        return miss();
    }

    // ---- MaterialEvaluator — shader/src/material.rs:16-83 ----
    struct Mat {
        V3 albedo, emission;
        float metallic, ior, transmission;
    };
    Mat material(uint32_t id) const {
        const rt_material& m = b->materials[id];
        Mat o;
        o.albedo = v3(m.albedo[0], m.albedo[1], m.albedo[2]);         // :16-18
        o.emission = v3(m.emission[0], m.emission[1], m.emission[2]); // :21-23
        o.metallic = f16_to_f32((uint16_t)(m.metallic_roughness_f16 & 0xFFFF)); // :26-28
        o.ior = f16_to_f32((uint16_t)(m.ior_transmission_f16 & 0xFFFF));        // :36-38
        o.transmission = f16_to_f32((uint16_t)(m.ior_transmission_f16 >> 16));  // :61-63
        return o;
    }
    static V3 evaluate_brdf(const Mat& m, float light_intensity) { // :76-83
        V3 diffuse = m.albedo / 3.14159265358979323846f;           // :71-73
        float is_metallic = (m.metallic > 0.5f) ? 1.0f : 0.0f;     // :66-68
        V3 metallic_contrib = m.albedo * light_intensity * 0.5f;
        V3 dielectric_contrib = diffuse * light_intensity;
        return metallic_contrib * is_metallic + dielectric_contrib * (1.0f - is_metallic);
    }
    static float ior_for_channel(const Mat& m, uint32_t channel) { // :42-58
        static const float table[4] = {-0.018f, 0.0f, 0.035f, 0.0f};
        uint32_t safe = channel < 3 ? channel : 3;
        return m.ior + table[safe];
    }

    // ---- LightingCalculator — shader/src/lighting.rs:20-139 ----
    V3 light_contribution(const Intersection& is, const Mat& m, uint32_t li) const { // :50-94
        float index_valid = (li < pc->metadata_offsets.lights_count) ? 1.0f : 0.0f;
        V3 lpos = sa.light_position(li);
        uint32_t ltype = sa.light_type(li);
        V3 lcol = sa.light_color(li);
        float lint = sa.light_intensity(li);
        V3 ldir = sa.light_direction(li);
        // calculate_directional_light :97-110
        V3 dir_light_dir = -normalize(ldir);
        float dir_intensity = fmaxf(dot(is.normal, dir_light_dir), 0.0f) * lint;
        // calculate_point_spot_light :113-139
        V3 to_light = lpos - is.point;
        float distance = length(to_light);
        V3 pld = normalize(to_light);
        float att32 = 1.0f / (1.0f + distance * distance * 0.01f);
        float att = f16_to_f32(f32_to_f16(att32));
        float point_intensity = fmaxf(dot(is.normal, pld), 0.0f) * lint * att;
        float spot_factor = fmaxf(dot(-normalize(ldir), pld), 0.0f);
        float spot_intensity = point_intensity * spot_factor;
        // :80-86
        float is_dir = (ltype == 0) ? 1.0f : 0.0f;
        float is_point = (ltype == 1) ? 1.0f : 0.0f;
        float is_spot = (ltype == 2) ? 1.0f : 0.0f;
        float final_i = dir_intensity * is_dir + point_intensity * is_point + spot_intensity * is_spot;
        V3 brdf = evaluate_brdf(m, final_i);
        float valid = ((final_i > 0.0f) ? 1.0f : 0.0f) * index_valid;
        return brdf * lcol * valid;
    }
    V3 calculate_lighting(const Intersection& is, const Mat& m) const { // :20-47
        V3 total = v3(0, 0, 0);
        total = total + m.albedo * 0.1f;
        for (uint32_t li = 0; li < sa.light_count(); li++) total = total + light_contribution(is, m, li);
        return total + m.emission;
    }

    // calculate_shading — shader/src/lib.rs:300-338 (== calculate_wavefront_shading
    // wavefront.rs:168-211 with throughput (1,1,1))
    V3 calculate_shading(const Intersection& is, uint32_t channel, bool wavefront) const {
        if ((uint64_t)is.material_id >= b->materials_len) return v3(1.0f, 0.0f, 1.0f);
        Mat m = material(is.material_id);
        V3 lighting = calculate_lighting(is, m);
        float tf = fminf(fmaxf(m.transmission, 0.0f), 1.0f);
        V3 out;
        if (tf > 0.0f) {
            float wl_ior = ior_for_channel(m, channel);
            float disp = (wl_ior - 1.0f) / (m.ior - 1.0f);
            V3 transmitted = v3(0.2f, 0.2f, 0.3f) * disp;
            out = lighting * (1.0f - tf) + transmitted * tf;
        } else {
            out = lighting;
        }
        if (wavefront) out = out * v3(1.0f, 1.0f, 1.0f); // throughput of a camera ray, shared/src/lib.rs:873
        return out;
    }

    // The colour main_cs computes before filter_color_by_channel, for one pixel and one
    // channel's push constants.  `hit` is the trace result (shared between channels
    // when the caller fuses the three passes).
    V3 shade(const IntersectionResult& hit, uint32_t channel) const {
        if (pc_mode(pc) != 0) {
            // run_wavefront_raytracing — shader/src/lib.rs:92-149: only the pass whose
            // current_bounce_depth lies in 0..=max_bounce_depth contributes.
            if (pc_cur_bounce(pc) > pc_max_bounce(pc)) return v3(0, 0, 0);
            // process_wavefront_ray — wavefront.rs:116-165
            if (!hit.hit) return v3(0.1f, 0.2f, 0.3f) * v3(1.0f, 1.0f, 1.0f);
            return calculate_shading(hit.is, channel, true);
        }
        if (hit.hit) return calculate_shading(hit.is, channel, false); // lib.rs:74-78
        return v3(0, 0, 0);
    }

    IntersectionResult trace_pixel(uint32_t px, uint32_t py) const {
        if (pc_mode(pc) != 0) {
            if (pc_cur_bounce(pc) > pc_max_bounce(pc)) return miss(); // no ray is processed at all
            return find_closest(ray_from_screen_wavefront(px, py, pc));
        }
        return find_closest(ray_from_screen(px, py, pc));
    }
};

// filter_color_by_channel — shader/src/lib.rs:342-349
inline V3 filter_channel(V3 c, uint32_t channel) {
    switch (channel) {
        case 0: return v3(c.x, 0.0f, 0.0f);
        case 1: return v3(0.0f, c.y, 0.0f);
        case 2: return v3(0.0f, 0.0f, c.z);
        default: return c;
    }
}

// Rgba8Unorm store: clamp to [0,1], scale, round to nearest (ties away from the even
// side are driver-defined; we round half up), NaN -> 0.
inline uint8_t unorm8(float v) {
    if (!(v > 0.0f)) return 0;
    if (v >= 1.0f) return 255;
    return (uint8_t)(int)floorf(v * 255.0f + 0.5f);
}
inline void image_write(uint8_t* img, uint32_t w, uint32_t x, uint32_t y, V3 c) {
    uint8_t* p = img + ((size_t)y * w + x) * 4;
    p[0] = unorm8(c.x);
    p[1] = unorm8(c.y);
    p[2] = unorm8(c.z);
    p[3] = 255;
}

// is_pixel_in_bounds — shader/src/lib.rs:152-163
inline bool in_bounds(uint32_t idx, uint32_t idy, const rt_push_constants* pc) {
    uint32_t px = pc->tile_offset[0] + idx;
    uint32_t py = pc->tile_offset[1] + idy;
    uint32_t width = f32_as_u32(pc->resolution[0]);
    uint32_t height = f32_as_u32(pc->resolution[1]);
    uint32_t tw, th;
    pc_tile_size(pc, &tw, &th);
    return idx < tw && idy < th && px < width && py < height;
}

std::atomic<int> g_fast_traversal{0};

Kernel make_kernel(const oracle_bindings* b, const rt_push_constants* pc, Counters* c) {
    Kernel k;
    k.fast = g_fast_traversal.load() != 0;
    k.b = b;
    k.pc = pc;
    k.c = c;
    k.sa = SceneAccessor{b->scene_metadata, b->scene_metadata_len, pc, c};
    return k;
}


// =====================================================================================
// Extended mode (RT_MODE_EXTENDED) — the build's own path tracer, stated on the CPU.
// No reference implementation exists; the pieces it is built from are the reference's:
// SimpleRng (shader/src/wavefront.rs:46-72), the pixel seed (shader/src/lib.rs:103-105),
// generate_camera_ray (wavefront.rs:75-112), find_closest_intersection, the lighting /
// BRDF of lighting.rs + material.rs, WavefrontRay's ray types and epsilon
// (shared/src/lib.rs:833-956) and apply_russian_roulette (shared/src/lib.rs:969-978).
// =====================================================================================
struct SimpleRng { // wavefront.rs:46-72
    uint32_t seed;
    uint32_t next_u32() {
        seed = seed * 1664525u + 1013904223u;
        return seed;
    }
    float next_f32() { return (float)(next_u32() >> 8) / 16777216.0f; }
};

// pixel_seed of lib.rs:103-105, sample index mixed in, scrambled once so neighbouring
// pixels / samples do not start on correlated LCG states.
inline SimpleRng rng_for(uint32_t pixel_seed, uint32_t sample) {
    uint32_t h = pixel_seed + sample * 0x9E3779B9u;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return SimpleRng{h};
}

// sin/cos of 2*pi*u, u in [0,1), as explicit fmaf polynomials so the CPU and the GPU produce
// the same bits (libm and the device math library do not agree on sinf/cosf).
inline void sincos_2pi(float u, float* s_out, float* c_out) {
    float f4 = u * 4.0f;
    float qf = floorf(f4);
    int q = (int)qf;
    float x = (f4 - qf) * 1.57079632679489661923f;
    float x2 = x * x;
    float sp = fmaf(x2, -2.50521083854417187751e-8f, 2.75573192239858906526e-6f);
    sp = fmaf(x2, sp, -1.98412698412698412698e-4f);
    sp = fmaf(x2, sp, 8.33333333333333333333e-3f);
    sp = fmaf(x2, sp, -1.66666666666666666667e-1f);
    sp = fmaf(x2, sp, 1.0f);
    float sn = x * sp;
    float cp = fmaf(x2, 2.08767569878680989792e-9f, -2.75573192239858906526e-7f);
    cp = fmaf(x2, cp, 2.48015873015873015873e-5f);
    cp = fmaf(x2, cp, -1.38888888888888888889e-3f);
    cp = fmaf(x2, cp, 4.16666666666666666667e-2f);
    cp = fmaf(x2, cp, -0.5f);
    float cs = fmaf(x2, cp, 1.0f);
    switch (q & 3) {
        case 0: *s_out = sn; *c_out = cs; break;
        case 1: *s_out = cs; *c_out = -sn; break;
        case 2: *s_out = -sn; *c_out = -cs; break;
        default: *s_out = -cs; *c_out = sn; break;
    }
}

// uniform direction on the unit sphere from two uniforms
inline V3 unit_vector(float u1, float u2) {
    float z = 1.0f - 2.0f * u1;
    float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float sn, cs;
    sincos_2pi(u2, &sn, &cs);
    return v3(r * cs, r * sn, z);
}

struct ExtCounts {
    uint64_t camera = 0, continuation = 0, shadow = 0, roulette = 0;
};

#define EXT_EPS 0.001f /* WavefrontRay::t_min (shared/src/lib.rs:854), used as the origin offset */

struct ExtKernel {
    Kernel k; // pc: mode 1 push constants of the frame
    uint32_t max_bounces, flags;

    // Direct lighting at a vertex: the reference's calculate_lighting (lighting.rs:20-47) with every
    // light's contribution gated by a shadow ray (unless ORACLE_EXT_NO_SHADOWS).  `ambient` adds the
    // reference's 0.1 * albedo term (only at the terminal vertex, where no indirect light follows).
    V3 direct(const Intersection& is, const Kernel::Mat& m, bool ambient, ExtCounts& ec) const {
        V3 total = v3(0, 0, 0);
        if (ambient) total = total + m.albedo * 0.1f;
        for (uint32_t li = 0; li < k.sa.light_count(); li++) {
            V3 contrib = k.light_contribution(is, m, li);
            if (!(contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f)) {
                total = total + contrib;
                continue;
            }
            if (!(flags & ORACLE_EXT_NO_SHADOWS)) {
                uint32_t ltype = k.sa.light_type(li);
                V3 dir;
                float tmax;
                if (ltype == 0) {
                    dir = -normalize(k.sa.light_direction(li));
                    tmax = 3.402823466e+38f;
                } else {
                    V3 to_light = k.sa.light_position(li) - is.point;
                    tmax = length(to_light);
                    dir = normalize(to_light);
                }
                Ray sr{is.point + is.normal * EXT_EPS, dir};
                ec.shadow++;
                if (k.fast) {
                    if (k.any_hit(sr, tmax)) continue; // occluded (baseline flavour ii: same answer, found sooner)
                } else {
                    IntersectionResult occ = k.find_closest(sr);
                    if (occ.hit && occ.is.t < tmax) continue; // occluded
                }
            }
            total = total + contrib;
        }
        return total + m.emission;
    }

    V3 trace_path(uint32_t px, uint32_t py, uint32_t sample, uint32_t spp, ExtCounts& ec) const {
        const rt_push_constants* pc = k.pc;
        uint32_t width = f32_as_u32(pc->resolution[0]);
        SimpleRng rng = rng_for(pc->frame_seed + px + py * width, sample);
        float jx = 0.5f, jy = 0.5f;
        if (spp > 1) {
            jx = rng.next_f32();
            jy = rng.next_f32();
        }
        // generate_camera_ray (wavefront.rs:75-112) through the jittered position
        Ray ray;
        {
            float u = ((float)px + jx) / pc->resolution[0];
            float v = ((float)py + jy) / pc->resolution[1];
            float aspect = pc->resolution[0] / pc->resolution[1];
            float fov_scale = tanf(pc->camera.fov * 0.5f * 3.14159265358979323846f / 180.0f);
            float cx = (u * 2.0f - 1.0f) * aspect * fov_scale;
            float cy = (1.0f - v * 2.0f) * fov_scale;
            V3 fwd = v3(pc->camera.direction[0], pc->camera.direction[1], pc->camera.direction[2]);
            V3 up = v3(pc->camera.up[0], pc->camera.up[1], pc->camera.up[2]);
            V3 right = cross(fwd, up);
            V3 true_up = cross(right, fwd);
            ray.origin = v3(pc->camera.position[0], pc->camera.position[1], pc->camera.position[2]);
            ray.direction = normalize(fwd + right * cx + true_up * cy);
        }
        V3 radiance = v3(0, 0, 0);
        V3 throughput = v3(1.0f, 1.0f, 1.0f); // WavefrontRay::camera_ray, shared/src/lib.rs:873
        uint32_t channel = 3;                 // hero wavelength channel not chosen yet
        for (uint32_t depth = 0;; depth++) {
            if (depth == 0) ec.camera++; else ec.continuation++;
            IntersectionResult hit = k.find_closest(ray);
            if (!hit.hit) { // process_wavefront_ray, wavefront.rs:146-151
                radiance = radiance + v3(0.1f, 0.2f, 0.3f) * throughput;
                break;
            }
            const Intersection& is = hit.is;
            if ((uint64_t)is.material_id >= k.b->materials_len) { // magenta, wavefront.rs:177-179
                radiance = radiance + v3(1.0f, 0.0f, 1.0f) * throughput;
                break;
            }
            Kernel::Mat m = k.material(is.material_id);
            float roughness = f16_to_f32((uint16_t)(k.b->materials[is.material_id].metallic_roughness_f16 >> 16));
            float tf = fminf(fmaxf(m.transmission, 0.0f), 1.0f);
            bool terminal = depth >= max_bounces;
            V3 lighting = direct(is, m, terminal, ec);
            if (terminal) {
                // the reference's calculate_shading (lib.rs:300-338), all three channel passes at once
                V3 out = lighting;
                if (tf > 0.0f) {
                    float dr = (Kernel::ior_for_channel(m, 0) - 1.0f) / (m.ior - 1.0f);
                    float dg = (Kernel::ior_for_channel(m, 1) - 1.0f) / (m.ior - 1.0f);
                    float db = (Kernel::ior_for_channel(m, 2) - 1.0f) / (m.ior - 1.0f);
                    float keep = 1.0f - tf;
                    out = v3(lighting.x * keep + (0.2f * dr) * tf, lighting.y * keep + (0.2f * dg) * tf,
                             lighting.z * keep + (0.3f * db) * tf);
                }
                radiance = radiance + out * throughput;
                break;
            }
            // the reflective share of this vertex
            radiance = radiance + (lighting * (1.0f - tf)) * throughput;

            // --- continuation (what generate_continuation_rays, wavefront.rs:340-355, only sketches) ---
            V3 d = ray.direction;
            bool front = dot(is.normal, d) < 0.0f;
            V3 nf = front ? is.normal : -is.normal;
            bool transmit = false;
            if (tf > 0.0f) transmit = rng.next_f32() < tf;
            V3 ndir, norigin;
            if (transmit) { // ray_type 2
                if (channel == 3) {
                    uint32_t c = (uint32_t)(rng.next_f32() * 3.0f);
                    channel = c < 2 ? c : 2;
                    throughput = v3(channel == 0 ? throughput.x * 3.0f : 0.0f, channel == 1 ? throughput.y * 3.0f : 0.0f,
                                    channel == 2 ? throughput.z * 3.0f : 0.0f);
                }
                float ior_c = Kernel::ior_for_channel(m, channel);
                float eta = front ? (1.0f / ior_c) : ior_c;
                float cos_i = -dot(nf, d);
                float sin2_t = eta * eta * (1.0f - cos_i * cos_i);
                if (sin2_t > 1.0f) { // total internal reflection
                    ndir = d - nf * (2.0f * dot(d, nf));
                    norigin = is.point + nf * EXT_EPS;
                } else {
                    float cos_t = sqrtf(1.0f - sin2_t);
                    ndir = d * eta + nf * (eta * cos_i - cos_t);
                    norigin = is.point - nf * EXT_EPS;
                }
                ndir = normalize(ndir);
                throughput = throughput * m.albedo;
            } else if (m.metallic > 0.5f) { // ray_type 1, rough mirror
                float u1 = rng.next_f32(), u2 = rng.next_f32();
                V3 r = d - nf * (2.0f * dot(d, nf));
                ndir = normalize(r + unit_vector(u1, u2) * roughness);
                if (!(dot(ndir, nf) > 0.0f)) break; // scattered into the surface: absorbed
                norigin = is.point + nf * EXT_EPS;
                throughput = throughput * m.albedo;
            } else { // ray_type 1, cosine-weighted diffuse
                float u1 = rng.next_f32(), u2 = rng.next_f32();
                V3 w = nf + unit_vector(u1, u2);
                if (dot(w, w) < 1e-12f) w = nf;
                ndir = normalize(w);
                norigin = is.point + nf * EXT_EPS;
                throughput = throughput * m.albedo;
            }
            // apply_russian_roulette (shared/src/lib.rs:969-978) from the third vertex on
            if (depth >= 2) {
                float p = fminf(fmaxf(fmaxf(fmaxf(throughput.x, throughput.y), throughput.z), 0.05f), 1.0f);
                if (rng.next_f32() > p) {
                    ec.roulette++;
                    break;
                }
                throughput = v3(throughput.x / p, throughput.y / p, throughput.z / p);
            }
            ray = Ray{norigin, ndir};
        }
        return radiance;
    }
};

} // namespace

extern "C" {

uint16_t oracle_f32_to_f16(float v) { return f32_to_f16(v); }
float oracle_f16_to_f32(uint16_t h) { return f16_to_f32(h); }

int oracle_dispatch(const oracle_bindings* b, const rt_push_constants* pc, uint8_t* image, uint32_t img_w, uint32_t img_h,
                    oracle_counters* counters) {
    if (!b || !pc || !image) return -1;
    Counters c;
    Kernel k = make_kernel(b, pc, &c);
    uint32_t tw, th;
    pc_tile_size(pc, &tw, &th);
    // dispatch_workgroups(ceil(tw/16), ceil(th/16), 1) x threads(16,16) — src/compute.rs:248-250
    uint32_t gx = ((tw + RT_THREAD_GROUP_X - 1) / RT_THREAD_GROUP_X) * RT_THREAD_GROUP_X;
    uint32_t gy = ((th + RT_THREAD_GROUP_Y - 1) / RT_THREAD_GROUP_Y) * RT_THREAD_GROUP_Y;
    uint32_t channel = pc_channel(pc);
    for (uint32_t idy = 0; idy < gy; idy++) {
        for (uint32_t idx = 0; idx < gx; idx++) {
            if (!in_bounds(idx, idy, pc)) continue; // main_cs :39-41
            uint32_t px = pc->tile_offset[0] + idx, py = pc->tile_offset[1] + idy;
            if (px >= img_w || py >= img_h) continue; // image store out of range is discarded
            IntersectionResult hit = k.trace_pixel(px, py);
            V3 col = filter_channel(k.shade(hit, channel), channel);
            image_write(image, img_w, px, py, col);
        }
    }
    c.add_to(counters);
    return 0;
}

int oracle_render_frame(const oracle_bindings* b, const rt_push_constants* base_pc, uint32_t tile_size, int threads, int faithful3,
                        uint8_t* red, uint8_t* green, uint8_t* blue, float* rgb32f, uint32_t* prim_ids, float* ts,
                        oracle_counters* counters) {
    if (!b || !base_pc) return -1;
    if (tile_size == 0) tile_size = RT_TILE_SIZE;
    uint32_t width = f32_as_u32(base_pc->resolution[0]);
    uint32_t height = f32_as_u32(base_pc->resolution[1]);
    if (width == 0 || height == 0) return -1;
    // TileHelper::calculate_tile_count — shared/src/lib.rs:1187-1191
    uint32_t tiles_x = (width + tile_size - 1) / tile_size;
    uint32_t tiles_y = (height + tile_size - 1) / tile_size;
    uint32_t total = tiles_x * tiles_y;
    if (threads < 1) threads = 1;
    std::atomic<uint32_t> next(0);
    std::vector<Counters> per_thread((size_t)threads);
    uint8_t* imgs[3] = {red, green, blue};

    auto worker = [&](int tid) {
        Counters& c = per_thread[(size_t)tid];
        for (;;) {
            uint32_t tile = next.fetch_add(1);
            if (tile >= total) break;
            // calculate_tile_dimensions — src/compute.rs:194-209
            uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
            uint32_t ox = tx * tile_size, oy = ty * tile_size;
            uint32_t tw = std::min(tile_size, width - ox), th = std::min(tile_size, height - oy);
            rt_push_constants pcs[3];
            for (uint32_t ch = 0; ch < 3; ch++) { // process_tile — src/compute.rs:184-190
                pcs[ch] = *base_pc;
                pcs[ch].tile_offset[0] = ox;
                pcs[ch].tile_offset[1] = oy;
                pcs[ch].tile_size_packed = (std::min(tw, 65535u) & 0xFFFF) | ((std::min(th, 65535u) & 0xFFFF) << 16);
                pcs[ch].total_tiles[0] = tiles_x;
                pcs[ch].total_tiles[1] = tiles_y;
                pcs[ch].packed_flags = (base_pc->packed_flags & 0xFFFFFF00u) | ch;
            }
            Kernel ks[3] = {make_kernel(b, &pcs[0], &c), make_kernel(b, &pcs[1], &c), make_kernel(b, &pcs[2], &c)};
            for (uint32_t idy = 0; idy < th; idy++) {
                for (uint32_t idx = 0; idx < tw; idx++) {
                    uint32_t px = ox + idx, py = oy + idy;
                    size_t pix = (size_t)py * width + px;
                    V3 final_rgb = v3(0, 0, 0);
                    IntersectionResult hit0 = miss();
                    for (uint32_t ch = 0; ch < 3; ch++) {
                        IntersectionResult hit = (ch == 0 || faithful3) ? ks[ch].trace_pixel(px, py) : hit0;
                        if (ch == 0) hit0 = hit;
                        V3 col = filter_channel(ks[ch].shade(hit, ch), ch);
                        if (imgs[ch]) image_write(imgs[ch], width, px, py, col);
                        if (ch == 0) final_rgb.x = col.x;
                        if (ch == 1) final_rgb.y = col.y;
                        if (ch == 2) final_rgb.z = col.z;
                    }
                    if (rgb32f) {
                        rgb32f[pix * 3 + 0] = final_rgb.x;
                        rgb32f[pix * 3 + 1] = final_rgb.y;
                        rgb32f[pix * 3 + 2] = final_rgb.z;
                    }
                    if (prim_ids) prim_ids[pix] = hit0.hit ? hit0.is.prim_id : 0xFFFFFFFFu;
                    if (ts) ts[pix] = hit0.hit ? hit0.is.t : 3.402823466e+38f;
                }
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
    for (auto& c : per_thread) c.add_to(counters);
    return 0;
}

int oracle_render_extended_region(const oracle_bindings* b, const rt_push_constants* base_pc, uint32_t spp, uint32_t max_bounces,
                                  uint32_t flags, int threads, uint32_t x0, uint32_t y0, uint32_t rw, uint32_t rh, float* rgb32f,
                                  uint64_t* segments, oracle_counters* counters) {
    if (!b || !base_pc || !rgb32f || spp == 0) return -1;
    uint32_t width = f32_as_u32(base_pc->resolution[0]);
    uint32_t height = f32_as_u32(base_pc->resolution[1]);
    if (width == 0 || height == 0) return -1;
    if (x0 > width || y0 > height || rw > width - x0 || rh > height - y0) return -1;
    if (threads < 1) threads = 1;
    rt_push_constants pc = *base_pc;
    pc.packed_flags = (base_pc->packed_flags & 0x00FFFFFFu) | (1u << 24);
    std::atomic<uint32_t> next(0);
    std::vector<Counters> per_thread((size_t)threads);
    std::vector<ExtCounts> per_thread_ec((size_t)threads);
    auto worker = [&](int tid) {
        ExtKernel ek;
        ek.k = make_kernel(b, &pc, &per_thread[(size_t)tid]);
        ek.max_bounces = max_bounces;
        ek.flags = flags;
        ExtCounts& ec = per_thread_ec[(size_t)tid];
        for (;;) {
            uint32_t row = next.fetch_add(1);
            if (row >= rh) break;
            for (uint32_t x = 0; x < rw; x++) {
                // a pixel's samples depend only on its own coordinates in the FULL frame (pixel seed, lib.rs:103-105)
                V3 sum = v3(0, 0, 0);
                for (uint32_t s = 0; s < spp; s++) sum = sum + ek.trace_path(x0 + x, y0 + row, s, spp, ec);
                float n = (float)spp;
                size_t pix = (size_t)row * rw + x;
                rgb32f[pix * 3 + 0] = sum.x / n;
                rgb32f[pix * 3 + 1] = sum.y / n;
                rgb32f[pix * 3 + 2] = sum.z / n;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
    for (auto& c : per_thread) c.add_to(counters);
    if (segments) {
        segments[0] = segments[1] = segments[2] = segments[3] = 0;
        for (auto& e : per_thread_ec) {
            segments[0] += e.camera;
            segments[1] += e.continuation;
            segments[2] += e.shadow;
            segments[3] += e.roulette;
        }
    }
    return 0;
}

int oracle_render_extended(const oracle_bindings* b, const rt_push_constants* base_pc, uint32_t spp, uint32_t max_bounces,
                           uint32_t flags, int threads, float* rgb32f, uint64_t* segments, oracle_counters* counters) {
    if (!base_pc) return -1;
    return oracle_render_extended_region(b, base_pc, spp, max_bounces, flags, threads, 0, 0, f32_as_u32(base_pc->resolution[0]),
                                         f32_as_u32(base_pc->resolution[1]), rgb32f, segments, counters);
}

} // extern "C"

// Baseline flavour (ii) switch (process-wide): ordered, distance-culled traversal instead of the reference's.
extern "C" void oracle_set_fast_traversal(int on) { g_fast_traversal.store(on ? 1 : 0); }
