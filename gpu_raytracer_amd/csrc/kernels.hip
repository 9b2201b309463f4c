// kernels.hip — gfx950 kernels of the ray-casting hot path.
//
// Replaces the reference's rust-gpu kernel `main_cs` (shader/src/lib.rs:25-89) and
// everything it calls (ray.rs, bvh.rs, intersection.rs, lighting.rs, material.rs,
// wavefront.rs).  One wavefront (64 lanes) = one 8x8 pixel block; each lane owns a
// pixel.  All three colour channels are produced in ONE pass: the trace and the
// lighting are channel independent, only the transmission term of
// shader/src/lib.rs:323-337 depends on the channel, so the reference's three
// dispatches per tile (src/compute.rs:184-190) collapse into one.
//
// Arithmetic that decides WHICH primitive is hit (ray generation, Möller–Trumbore,
// sphere quadratic) and the shading keep the reference's f32 operation order; the file
// is compiled with -ffp-contract=off so nothing is fused implicitly.  The slab test is
// the reference's own test applied to our boxes plus culling by the closest hit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "kernels.h"

#define RT_MIN_RAY_DISTANCE 0.00001f
#define RT_F32_MAX 3.402823466e+38f
#define RT_PI 3.14159265358979323846f
#define WAVE 64

namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
// glam scalar-math order: (x*x + y*y) + z*z
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ float length(V3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 normalize(V3 a) { return a * (1.0f / length(a)); }

// IEEE half round trip, round-to-nearest-even (shader/src/lighting.rs:125-127)
__device__ __forceinline__ float f16_round_trip(float v) {
    _Float16 h = (_Float16)v; // v_cvt_f16_f32, RNE in the default mode
    return (float)h;
}

struct Hit {
    float t;
    uint32_t prim; // RT_PRIM_MISS, RT_PRIM_SPHERE_FLAG | i, or original triangle index
    uint32_t slot; // index into DevScene::tris for triangle hits
};

struct Counts {
    uint32_t nodes, tris;
};

// ------------------------------------------------------------------------------------
// Ray generation.  Ray::from_screen_coordinates (shader/src/ray.rs:22-53) for mode 0,
// generate_camera_ray (shader/src/wavefront.rs:75-112) for mode 1.  fov_scale / aspect /
// right / true_up are per-frame constants computed on the host in the same order.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void camera_ray(const DevCamera& cam, float sx, float sy, bool wavefront, V3& o, V3& d) {
    float u = sx / cam.width_f;
    float v = sy / cam.height_f;
    float cx = (u * 2.0f - 1.0f) * cam.aspect * cam.fov_scale;
    float cy = (1.0f - v * 2.0f) * cam.fov_scale;
    V3 dir = ld3(cam.forward) + ld3(cam.right) * cx + ld3(cam.true_up) * cy;
    V3 n = normalize(dir);
    o = ld3(cam.origin);
    d = wavefront ? n : normalize(n); // Ray::new normalises a second time (ray.rs:14-19)
}

// ------------------------------------------------------------------------------------
// Spheres: test_sphere_intersections (shader/src/lib.rs:252-269) +
// test_sphere_intersection (shader/src/intersection.rs:52-87).  Linear, wave-uniform loop.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void test_spheres(const DevScene& sc, V3 o, V3 d, Hit& hit) {
    for (uint32_t i = 0; i < sc.n_spheres; i++) {
        const DevSphere& s = sc.spheres[i];
        V3 oc = o - ld3(s.center);
        float a = dot(d, d);
        float b = 2.0f * dot(oc, d);
        float c = dot(oc, oc) - s.radius * s.radius;
        float disc = b * b - 4.0f * a * c;
        if (disc < 0.0f) continue;
        float sq = sqrtf(disc);
        float t1 = (-b - sq) / (2.0f * a);
        float t2 = (-b + sq) / (2.0f * a);
        float t = (t1 > RT_MIN_RAY_DISTANCE) ? t1 : t2;
        if (t > RT_MIN_RAY_DISTANCE && t < hit.t) {
            hit.t = t;
            hit.prim = RT_PRIM_SPHERE_FLAG | i;
            hit.slot = i;
        }
    }
}

// ------------------------------------------------------------------------------------
// Möller–Trumbore, test_triangle_intersection_direct (shader/src/intersection.rs:91-138),
// on a pre-gathered DevTri.  Accept 1e-5 < t < closest (strict).  The reference keeps the
// first triangle found among equal t; its visiting order is its own BVH's.  Here equal t
// is resolved toward the LOWER original triangle index, which is what the reference's
// chunked BVH (> 100k triangles, mesh-order leaves visited left to right,
// src/bvh.rs:154-247) and its brute-force path (shader/src/lib.rs:283) do, and makes the
// result independent of our own topology.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void test_triangle(const DevTri* __restrict__ tris, uint32_t slot, V3 o, V3 d, Hit& hit) {
    const float4* p = reinterpret_cast<const float4*>(tris + slot);
    float4 q0 = p[0], q1 = p[1], q2 = p[2];
    V3 v0 = v3(q0.x, q0.y, q0.z);
    V3 e1 = v3(q0.w, q1.x, q1.y);
    V3 e2 = v3(q1.z, q1.w, q2.x);
    V3 h = cross(d, e2);
    float a = dot(e1, h);
    if (fabsf(a) < RT_MIN_RAY_DISTANCE) return;
    float f = 1.0f / a;
    V3 s = o - v0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return;
    V3 q = cross(s, e1);
    float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return;
    float t = f * dot(e2, q);
    uint32_t prim = __float_as_uint(q2.z);
    if (t > RT_MIN_RAY_DISTANCE && (t < hit.t || (t == hit.t && prim < hit.prim))) {
        hit.t = t;
        hit.prim = prim;
        hit.slot = slot;
    }
}

// ------------------------------------------------------------------------------------
// BVH traversal.  Replaces BvhTraverser::traverse_and_intersect (shader/src/bvh.rs:18-88) and
// ray_aabb_intersect (shader/src/intersection.rs:151-164): per-lane depth-first walk over the
// 4-wide quantised tree of device_layout.h.  A visit fetches one 48-byte node (3 x dwordx4),
// slab-tests its (up to) four children, continues with the nearest hit child and pushes the others
// far-to-near.  Children whose entry distance exceeds the closest hit are skipped (the reference's
// slab test ignores the closest hit and so visits a superset: result-neutral).
//
// The boxes are FILTERS only — which primitive is hit is decided by the reference's
// Möller–Trumbore arithmetic — so they may be conservative but must never be too small.  The
// quantised planes are rounded outward by the builder; the float evaluation here is widened by a
// bound on its own rounding error: with a = scale/d, b = (org - o)/d the plane distances are
// fma(q, a, b), whose absolute error is below 2^-22 * (|org - o| + 255 * scale) / |d| (one rounding
// each in org - o, a, b, and the fma): near planes are moved back and far planes forward by that.
//
// The stack lives in LDS, lane-interleaved (entry k of lane l at stack[k * 64 + l]): ds_read /
// ds_write_b32 with consecutive lanes on consecutive banks.  A visit pushes at most 3 entries, the
// launch provides 3 * depth + 1 entries per lane (DevScene::stack_entries), so it cannot overflow.
// ------------------------------------------------------------------------------------
struct FilterRay { // per-segment constants of the box filter
    V3 o, inv;     // inv = 1/d with |d| clamped away from zero (a filter may do that; the triangle test uses the real d)
};
__device__ __forceinline__ FilterRay make_filter_ray(V3 o, V3 d) {
    FilterRay f;
    f.o = o;
    float dx = fabsf(d.x) < 1e-30f ? copysignf(1e-30f, d.x) : d.x;
    float dy = fabsf(d.y) < 1e-30f ? copysignf(1e-30f, d.y) : d.y;
    float dz = fabsf(d.z) < 1e-30f ? copysignf(1e-30f, d.z) : d.z;
    f.inv = v3(1.0f / dx, 1.0f / dy, 1.0f / dz);
    return f;
}

#define RT_SWAP_IF(cond, ta, tb, ra, rb) \
    {                                    \
        const bool c_ = (cond);          \
        const float tt_ = c_ ? tb : ta;  \
        tb = c_ ? ta : tb;               \
        ta = tt_;                        \
        const uint32_t rr_ = c_ ? rb : ra; \
        rb = c_ ? ra : rb;               \
        ra = rr_;                        \
    }

// One visit of inner node `cur`.  Updates cur / the stack; returns false when the walk is exhausted.
template <bool COUNT>
__device__ __forceinline__ bool visit_node4(const uint4* __restrict__ nodes, uint32_t stack_entries, const FilterRay& fr, float closest_t,
                                            uint32_t* __restrict__ stack, int& sp, uint32_t& cur, Counts& cnt) {
    const uint4* n = nodes + (size_t)cur * 3;
    const uint4 w0 = n[0], w1 = n[1], w2 = n[2];
    if (COUNT) cnt.nodes++;
    const float scx = __uint_as_float((w0.w & 0xFFu) << 23), scy = __uint_as_float(((w0.w >> 8) & 0xFFu) << 23),
                scz = __uint_as_float(((w0.w >> 16) & 0xFFu) << 23);
    const float dox = __uint_as_float(w0.x) - fr.o.x, doy = __uint_as_float(w0.y) - fr.o.y, doz = __uint_as_float(w0.z) - fr.o.z;
    const float ax = scx * fr.inv.x, ay = scy * fr.inv.y, az = scz * fr.inv.z;
    const float bx = dox * fr.inv.x, by = doy * fr.inv.y, bz = doz * fr.inv.z;
    const float ex = (fabsf(dox) + 255.0f * scx) * 2.4e-7f * fabsf(fr.inv.x);
    const float ey = (fabsf(doy) + 255.0f * scy) * 2.4e-7f * fabsf(fr.inv.y);
    const float ez = (fabsf(doz) + 255.0f * scz) * 2.4e-7f * fabsf(fr.inv.z);
    const float bnx = bx - ex, bfx = bx + ex, bny = by - ey, bfy = by + ey, bnz = bz - ez, bfz = bz + ez;
    // entry planes are the lower ones along axes the ray travels in +, the upper ones otherwise
    const bool px = fr.inv.x >= 0.0f, py = fr.inv.y >= 0.0f, pz = fr.inv.z >= 0.0f;
    const uint32_t nxw = px ? w1.z : w2.y, fxw = px ? w2.y : w1.z; // w1.z qlo_x, w2.y qhi_x
    const uint32_t nyw = py ? w1.w : w2.z, fyw = py ? w2.z : w1.w; // w1.w qlo_y, w2.z qhi_y
    const uint32_t nzw = pz ? w2.x : w2.w, fzw = pz ? w2.w : w2.x; // w2.x qlo_z, w2.w qhi_z
    const float limit = closest_t * 1.0000153f; // culling with slack, so equal-t candidates are still visited
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float tnx = fmaf((float)((nxw >> (8 * i)) & 0xFFu), ax, bnx), tfx = fmaf((float)((fxw >> (8 * i)) & 0xFFu), ax, bfx);
        const float tny = fmaf((float)((nyw >> (8 * i)) & 0xFFu), ay, bny), tfy = fmaf((float)((fyw >> (8 * i)) & 0xFFu), ay, bfy);
        const float tnz = fmaf((float)((nzw >> (8 * i)) & 0xFFu), az, bnz), tfz = fmaf((float)((fzw >> (8 * i)) & 0xFFu), az, bfz);
        const float tmin = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
        const float tmax = fminf(fminf(tfx, tfy), fminf(tfz, limit));
        t[i] = (tmin <= tmax * 1.000001f) ? tmin : RT_F32_MAX; // absent children are inverted boxes: never hit
    }
    // child references: inner children first (node_base + i), then leaves (consecutive triangle runs)
    const uint32_t node_base = w1.x & RT_DEV_NODE_BASE_MASK, n_inner = (w1.x >> 26) & 7u;
    const uint32_t cb = w0.w >> 24; // (count - 1) of leaf child j in bits 2j..2j+1
    const uint32_t c0 = (cb & 3u) + 1u, c1 = ((cb >> 2) & 3u) + 1u, c2 = ((cb >> 4) & 3u) + 1u, c3 = ((cb >> 6) & 3u) + 1u;
    const uint32_t l0 = RT_DEV_LEAF_FLAG | (c0 << RT_DEV_LEAF_COUNT_SHIFT) | w1.y;
    const uint32_t l1 = RT_DEV_LEAF_FLAG | (c1 << RT_DEV_LEAF_COUNT_SHIFT) | (w1.y + c0);
    const uint32_t l2 = RT_DEV_LEAF_FLAG | (c2 << RT_DEV_LEAF_COUNT_SHIFT) | (w1.y + c0 + c1);
    const uint32_t l3 = RT_DEV_LEAF_FLAG | (c3 << RT_DEV_LEAF_COUNT_SHIFT) | (w1.y + c0 + c1 + c2);
    uint32_t r0, r1, r2, r3; // leaf j is child n_inner + j
    r0 = n_inner > 0u ? node_base : l0;
    r1 = n_inner > 1u ? node_base + 1u : (n_inner == 1u ? l0 : l1);
    r2 = n_inner > 2u ? node_base + 2u : (n_inner == 2u ? l0 : (n_inner == 1u ? l1 : l2));
    r3 = n_inner > 3u ? node_base + 3u : (n_inner == 3u ? l0 : (n_inner == 2u ? l1 : (n_inner == 1u ? l2 : l3)));
    float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
    // sort the four (distance, reference) pairs by distance: misses (F32_MAX) end up last
    RT_SWAP_IF(t1 < t0, t0, t1, r0, r1)
    RT_SWAP_IF(t3 < t2, t2, t3, r2, r3)
    RT_SWAP_IF(t2 < t0, t0, t2, r0, r2)
    RT_SWAP_IF(t3 < t1, t1, t3, r1, r3)
    RT_SWAP_IF(t2 < t1, t1, t2, r1, r2)
    if (t0 == RT_F32_MAX) { // nothing hit: continue with the stack
        if (sp == 0) return false;
        sp--;
        cur = stack[sp * WAVE];
        return true;
    }
    cur = r0;
    // push the other hits far-to-near without branching: always store three words (the launch provides three
    // spare entries), advance the pointer by the number of real ones
    const int extra = (t1 != RT_F32_MAX ? 1 : 0) + (t2 != RT_F32_MAX ? 1 : 0) + (t3 != RT_F32_MAX ? 1 : 0);
    const uint32_t e0 = extra == 3 ? r3 : (extra == 2 ? r2 : r1);
    const uint32_t e1 = extra == 3 ? r2 : r1;
    stack[sp * WAVE] = e0;
    stack[(sp + 1) * WAVE] = e1;
    stack[(sp + 2) * WAVE] = r1;
    sp += extra;
    (void)stack_entries;
    return true;
}

// ANY_HIT (shadow segments of the extended mode): return at the first accepted triangle.
template <bool COUNT, bool ANY_HIT>
__device__ __forceinline__ void traverse(const DevScene& sc, V3 o, V3 d, uint32_t* __restrict__ stack, Hit& hit, Counts& cnt) {
    if (sc.n_tris == 0) return;
    const FilterRay fr = make_filter_ray(o, d);
    const uint4* __restrict__ nodes = reinterpret_cast<const uint4*>(sc.nodes);
    uint32_t cur = sc.root_ref;
    int sp = 0;
    for (;;) {
        if (!(cur & RT_DEV_LEAF_FLAG)) {
            if (!visit_node4<COUNT>(nodes, sc.stack_entries, fr, hit.t, stack, sp, cur, cnt)) break;
            continue;
        }
        uint32_t start = cur & RT_DEV_LEAF_START_MASK;
        uint32_t count = (cur >> RT_DEV_LEAF_COUNT_SHIFT) & 0xFu;
        for (uint32_t i = 0; i < count; i++) {
            if (COUNT) cnt.tris++;
            test_triangle(sc.tris, start + i, o, d, hit);
            if (ANY_HIT && hit.prim != RT_PRIM_MISS) return;
        }
        if (sp == 0) break;
        sp--;
        cur = stack[sp * WAVE];
    }
}

// find_closest_intersection (shader/src/lib.rs:174-249): spheres first, then triangles with
// max_t = sphere t; a triangle is only accepted strictly closer, so it wins when both hit.
template <bool COUNT>
__device__ __forceinline__ Hit find_closest(const DevScene& sc, V3 o, V3 d, uint32_t* stack, Counts& cnt) {
    Hit hit;
    hit.t = RT_F32_MAX; // f32::MAX - 2.0 == f32::MAX
    hit.prim = RT_PRIM_MISS;
    hit.slot = 0;
    test_spheres(sc, o, d, hit);
    traverse<COUNT, false>(sc, o, d, stack, hit, cnt);
    return hit;
}

// ------------------------------------------------------------------------------------
// Shading: calculate_shading (shader/src/lib.rs:300-338), LightingCalculator
// (shader/src/lighting.rs:20-139), MaterialEvaluator (shader/src/material.rs:16-83).
// Returns the colour of all three channel passes at once: component c is what the
// channel-c dispatch would have kept (filter_color_by_channel, lib.rs:342-349).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ V3 evaluate_brdf(const DevMaterial& m, float intensity) {
    V3 albedo = ld3(m.albedo);
    V3 diffuse = albedo / RT_PI;
    float is_metallic = (m.metallic > 0.5f) ? 1.0f : 0.0f;
    V3 metallic_contrib = albedo * intensity * 0.5f;
    V3 dielectric_contrib = diffuse * intensity;
    return metallic_contrib * is_metallic + dielectric_contrib * (1.0f - is_metallic);
}

// calculate_light_contribution (lighting.rs:50-94) for one light.  Also returns the direction and
// length of the segment toward the light (used by the extended mode's shadow rays).
__device__ __forceinline__ V3 light_contribution(const DevLight& L, const DevMaterial& m, V3 point, V3 normal, V3& to_light_dir,
                                                 float& to_light_dist) {
    V3 ldir = ld3(L.direction);
    V3 dir_light_dir = -normalize(ldir); // lighting.rs:103
    float dir_intensity = fmaxf(dot(normal, dir_light_dir), 0.0f) * L.intensity;
    V3 to_light = ld3(L.position) - point; // lighting.rs:120-122
    float distance = length(to_light);
    V3 pld = normalize(to_light);
    float att = f16_round_trip(1.0f / (1.0f + distance * distance * 0.01f)); // :125-127
    float point_intensity = fmaxf(dot(normal, pld), 0.0f) * L.intensity * att;
    float spot_factor = fmaxf(dot(-normalize(ldir), pld), 0.0f); // :132
    float spot_intensity = point_intensity * spot_factor;
    float is_dir = (L.light_type == 0) ? 1.0f : 0.0f;
    float is_point = (L.light_type == 1) ? 1.0f : 0.0f;
    float is_spot = (L.light_type == 2) ? 1.0f : 0.0f;
    float final_i = dir_intensity * is_dir + point_intensity * is_point + spot_intensity * is_spot;
    V3 brdf = evaluate_brdf(m, final_i);
    float valid = (final_i > 0.0f) ? 1.0f : 0.0f; // index_valid is always 1 inside the loop
    to_light_dir = (L.light_type == 0) ? dir_light_dir : pld;
    to_light_dist = (L.light_type == 0) ? RT_F32_MAX : distance;
    return brdf * ld3(L.color) * valid;
}

__device__ __forceinline__ V3 calculate_lighting(const DevScene& sc, const DevMaterial& m, V3 point, V3 normal) {
    V3 total = v3(0.0f, 0.0f, 0.0f);
    total = total + ld3(m.albedo) * 0.1f; // ambient, lighting.rs:30-31
    for (uint32_t li = 0; li < sc.n_lights; li++) {
        V3 sd;
        float st;
        total = total + light_contribution(sc.lights[li], m, point, normal, sd, st);
    }
    return total + ld3(m.emission);
}

// hit -> (point, geometric normal, material id)
__device__ __forceinline__ void hit_geometry(const DevScene& sc, const Hit& hit, V3 o, V3 d, V3& point, V3& normal, uint32_t& material_id) {
    point = o + d * hit.t; // Ray::at, ray.rs:56-58
    if (hit.prim & RT_PRIM_SPHERE_FLAG) {
        const DevSphere& s = sc.spheres[hit.slot];
        normal = normalize(point - ld3(s.center)); // intersection.rs:80
        material_id = s.material_id;
    } else {
        const float4* p = reinterpret_cast<const float4*>(sc.tris + hit.slot);
        float4 q0 = p[0], q1 = p[1], q2 = p[2];
        V3 e1 = v3(q0.w, q1.x, q1.y);
        V3 e2 = v3(q1.z, q1.w, q2.x);
        normal = normalize(cross(e1, e2)); // geometric, winding dependent, never flipped (intersection.rs:132)
        material_id = __float_as_uint(q2.y);
    }
}

// The transmission mix of calculate_shading (lib.rs:323-337) for all three channel passes at once.
__device__ __forceinline__ V3 transmission_mix(const DevMaterial& m, V3 lighting, float tf) {
    // per channel c: ior_c = ior + {-0.018, 0, +0.035}[c]; disp = (ior_c - 1) / (ior - 1)   (material.rs:42-58, lib.rs:326-334)
    float disp_r = ((m.ior + -0.018f) - 1.0f) / (m.ior - 1.0f);
    float disp_g = ((m.ior + 0.0f) - 1.0f) / (m.ior - 1.0f);
    float disp_b = ((m.ior + 0.035f) - 1.0f) / (m.ior - 1.0f);
    float keep = 1.0f - tf;
    return v3(lighting.x * keep + (0.2f * disp_r) * tf, lighting.y * keep + (0.2f * disp_g) * tf,
              lighting.z * keep + (0.3f * disp_b) * tf);
}

__device__ __forceinline__ V3 shade_hit(const DevScene& sc, const Hit& hit, V3 o, V3 d) {
    V3 point, normal;
    uint32_t material_id;
    hit_geometry(sc, hit, o, d, point, normal, material_id);
    if (material_id >= sc.n_materials) return v3(1.0f, 0.0f, 1.0f); // magenta, lib.rs:307-309
    const DevMaterial m = sc.materials[material_id];
    V3 lighting = calculate_lighting(sc, m, point, normal);
    float tf = fminf(fmaxf(m.transmission, 0.0f), 1.0f); // lib.rs:323
    if (tf > 0.0f) return transmission_mix(m, lighting, tf);
    return lighting;
}

// Rgba8Unorm store conversion: clamp, scale, round half up; NaN -> 0.
__device__ __forceinline__ uint32_t unorm8(float v) {
    if (!(v > 0.0f)) return 0u;
    if (v >= 1.0f) return 255u;
    return (uint32_t)floorf(v * 255.0f + 0.5f);
}

// Block -> pixel mapping shared by the kernels: one wave per 8x8 block of an owned tile.
struct PixelCoord {
    uint32_t x, y;
    bool valid;
};
__device__ __forceinline__ PixelCoord block_pixel(const DevFrame& fr) {
    const uint32_t lane = threadIdx.x;
    uint32_t ox, oy, tw, th, blk;
    const uint32_t bpt_x = (fr.tile_size + 7u) >> 3;
    if (fr.single_tile) {
        ox = fr.tile_off_x; oy = fr.tile_off_y; tw = fr.tile_w; th = fr.tile_h;
        blk = blockIdx.x;
    } else {
        const uint32_t bpt = bpt_x * bpt_x;
        uint32_t k = blockIdx.x / bpt;
        blk = blockIdx.x - k * bpt;
        uint32_t tile = fr.tile_first + k * fr.tile_stride;
        uint32_t ty = tile / fr.tiles_x, tx = tile - ty * fr.tiles_x;
        ox = tx * fr.tile_size; oy = ty * fr.tile_size;
        tw = min(fr.tile_size, fr.width - ox); // calculate_tile_dimensions, src/compute.rs:194-209
        th = min(fr.tile_size, fr.height - oy);
    }
    uint32_t by = blk / bpt_x, bx = blk - by * bpt_x;
    uint32_t idx = bx * 8u + (lane & 7u), idy = by * 8u + (lane >> 3);
    PixelCoord pc;
    pc.x = ox + idx;
    pc.y = oy + idy;
    // is_pixel_in_bounds, shader/src/lib.rs:152-163
    pc.valid = idx < tw && idy < th && pc.x < fr.width && pc.y < fr.height;
    return pc;
}

// ------------------------------------------------------------------------------------
// k_render_reference: modes 0 and 1 (reference semantics): one pixel-centre ray per pixel,
// closest hit, direct lighting; miss = black (mode 0, lib.rs:77) or sky (mode 1,
// wavefront.rs:146-151).  Mode 1 only produces colour when current_bounce <= max_bounce
// (lib.rs:117-121), otherwise (0,0,0).
// ------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(WAVE) void k_render_reference(DevScene sc, DevFrame fr, DevTargets tg) {
    extern __shared__ uint32_t s_stack[]; // DevScene::stack_entries * 64 words
    PixelCoord px = block_pixel(fr);
    if (!px.valid) return;
    uint32_t* stack = s_stack + threadIdx.x;
    Counts cnt = {0u, 0u};
    const bool wavefront = fr.mode != 0;
    V3 color = v3(0.0f, 0.0f, 0.0f);
    Hit hit;
    hit.t = RT_F32_MAX;
    hit.prim = RT_PRIM_MISS;
    hit.slot = 0;
    const bool traced = !wavefront || fr.cur_bounce <= fr.max_bounce;
    if (traced) {
        V3 o, d;
        camera_ray(fr.cam, (float)px.x + 0.5f, (float)px.y + 0.5f, wavefront, o, d);
        hit = find_closest<COUNT>(sc, o, d, stack, cnt);
        if (hit.prim != RT_PRIM_MISS) color = shade_hit(sc, hit, o, d);
        else if (wavefront) color = v3(0.1f, 0.2f, 0.3f);
    }
    const size_t pix = (size_t)px.y * fr.width + px.x;
    if (tg.rgba32f) reinterpret_cast<float4*>(tg.rgba32f)[pix] = make_float4(color.x, color.y, color.z, 1.0f);
    if (tg.prim_id) tg.prim_id[pix] = hit.prim;
    if (tg.hit_t) tg.hit_t[pix] = hit.t;
    // channel texture c keeps only component c (filter_color_by_channel) and alpha 1
    if ((fr.channel_mask & 1u) && tg.chan[0]) reinterpret_cast<uint32_t*>(tg.chan[0])[pix] = unorm8(color.x) | 0xFF000000u;
    if ((fr.channel_mask & 2u) && tg.chan[1]) reinterpret_cast<uint32_t*>(tg.chan[1])[pix] = (unorm8(color.y) << 8) | 0xFF000000u;
    if ((fr.channel_mask & 4u) && tg.chan[2]) reinterpret_cast<uint32_t*>(tg.chan[2])[pix] = (unorm8(color.z) << 16) | 0xFF000000u;
    if (COUNT && tg.counters) {
        atomicAdd(&tg.counters[0], traced ? 1ull : 0ull);
        atomicAdd(&tg.counters[1], (unsigned long long)cnt.nodes);
        atomicAdd(&tg.counters[2], (unsigned long long)cnt.tris);
        atomicAdd(&tg.counters[3], traced ? 1ull : 0ull);
    }
}


// ====================================================================================
// Extended mode (RT_MODE_EXTENDED): jittered samples, shadow rays and real bounces, built on
// the reference's declared-but-stub wavefront API (SimpleRng wavefront.rs:46-72, pixel seed
// lib.rs:103-105, generate_camera_ray wavefront.rs:75-112, WavefrontRay types and epsilon
// shared/src/lib.rs:833-956, apply_russian_roulette shared/src/lib.rs:969-978).  The rules are
// stated in DESIGN.md "Extended mode" (the test suite holds an executable CPU statement of them);
// this is the same arithmetic in the same order.
// ====================================================================================
struct SimpleRng {
    uint32_t seed;
    __device__ __forceinline__ uint32_t next_u32() {
        seed = seed * 1664525u + 1013904223u;
        return seed;
    }
    __device__ __forceinline__ float next_f32() { return (float)(next_u32() >> 8) / 16777216.0f; }
};

__device__ __forceinline__ SimpleRng rng_for(uint32_t pixel_seed, uint32_t sample) {
    uint32_t h = pixel_seed + sample * 0x9E3779B9u;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return SimpleRng{h};
}

// sin/cos(2*pi*u) as explicit-fma polynomials: identical bits on the host oracle and here.
__device__ __forceinline__ void sincos_2pi(float u, float& s_out, float& c_out) {
    float f4 = u * 4.0f;
    float qf = floorf(f4);
    int q = (int)qf;
    float x = (f4 - qf) * 1.57079632679489661923f;
    float x2 = x * x;
    float sp = __builtin_fmaf(x2, -2.50521083854417187751e-8f, 2.75573192239858906526e-6f);
    sp = __builtin_fmaf(x2, sp, -1.98412698412698412698e-4f);
    sp = __builtin_fmaf(x2, sp, 8.33333333333333333333e-3f);
    sp = __builtin_fmaf(x2, sp, -1.66666666666666666667e-1f);
    sp = __builtin_fmaf(x2, sp, 1.0f);
    float sn = x * sp;
    float cp = __builtin_fmaf(x2, 2.08767569878680989792e-9f, -2.75573192239858906526e-7f);
    cp = __builtin_fmaf(x2, cp, 2.48015873015873015873e-5f);
    cp = __builtin_fmaf(x2, cp, -1.38888888888888888889e-3f);
    cp = __builtin_fmaf(x2, cp, 4.16666666666666666667e-2f);
    cp = __builtin_fmaf(x2, cp, -0.5f);
    float cs = __builtin_fmaf(x2, cp, 1.0f);
    switch (q & 3) {
        case 0: s_out = sn; c_out = cs; break;
        case 1: s_out = cs; c_out = -sn; break;
        case 2: s_out = -sn; c_out = -cs; break;
        default: s_out = -cs; c_out = sn; break;
    }
}

__device__ __forceinline__ V3 unit_vector(float u1, float u2) {
    float z = 1.0f - 2.0f * u1;
    float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float sn, cs;
    sincos_2pi(u2, sn, cs);
    return v3(r * cs, r * sn, z);
}

#define EXT_EPS 0.001f /* WavefrontRay::t_min (shared/src/lib.rs:854) as the origin offset */

struct SegCounts {
    uint32_t camera, continuation, shadow;
};

// any hit in (1e-5, tmax)?  Spheres first, then the BVH with early exit.
template <bool COUNT>
__device__ __forceinline__ bool occluded(const DevScene& sc, V3 o, V3 d, float tmax, uint32_t* stack, Counts& cnt) {
    Hit h;
    h.t = tmax;
    h.prim = RT_PRIM_MISS;
    h.slot = 0;
    test_spheres(sc, o, d, h);
    if (h.prim != RT_PRIM_MISS) return true;
    traverse<COUNT, true>(sc, o, d, stack, h, cnt);
    return h.prim != RT_PRIM_MISS;
}

template <bool COUNT>
__device__ __forceinline__ V3 ext_direct(const DevScene& sc, const DevMaterial& m, V3 point, V3 normal, bool ambient, bool shadows,
                                         uint32_t* stack, Counts& cnt, SegCounts& seg) {
    V3 total = v3(0.0f, 0.0f, 0.0f);
    if (ambient) total = total + ld3(m.albedo) * 0.1f;
    for (uint32_t li = 0; li < sc.n_lights; li++) {
        V3 sdir;
        float sdist;
        V3 contrib = light_contribution(sc.lights[li], m, point, normal, sdir, sdist);
        if (!(contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f)) {
            total = total + contrib;
            continue;
        }
        if (shadows) {
            seg.shadow++;
            if (occluded<COUNT>(sc, point + normal * EXT_EPS, sdir, sdist, stack, cnt)) continue;
        }
        total = total + contrib;
    }
    return total + ld3(m.emission);
}

template <bool COUNT>
__device__ __forceinline__ V3 ext_trace_path(const DevScene& sc, const DevFrame& fr, uint32_t px, uint32_t py, uint32_t sample,
                                             uint32_t* stack, Counts& cnt, SegCounts& seg) {
    SimpleRng rng = rng_for(fr.frame_seed + px + py * fr.width, sample);
    float jx = 0.5f, jy = 0.5f;
    if (fr.spp > 1) {
        jx = rng.next_f32();
        jy = rng.next_f32();
    }
    V3 o, d;
    camera_ray(fr.cam, (float)px + jx, (float)py + jy, true, o, d);
    V3 radiance = v3(0.0f, 0.0f, 0.0f);
    V3 throughput = v3(1.0f, 1.0f, 1.0f);
    uint32_t channel = 3;
    const bool shadows = (fr.flags & 2u) == 0;
    for (uint32_t depth = 0;; depth++) {
        if (depth == 0) seg.camera++; else seg.continuation++;
        Hit hit = find_closest<COUNT>(sc, o, d, stack, cnt);
        if (hit.prim == RT_PRIM_MISS) {
            radiance = radiance + v3(0.1f, 0.2f, 0.3f) * throughput;
            break;
        }
        V3 point, normal;
        uint32_t material_id;
        hit_geometry(sc, hit, o, d, point, normal, material_id);
        if (material_id >= sc.n_materials) {
            radiance = radiance + v3(1.0f, 0.0f, 1.0f) * throughput;
            break;
        }
        const DevMaterial m = sc.materials[material_id];
        float tf = fminf(fmaxf(m.transmission, 0.0f), 1.0f);
        bool terminal = depth >= fr.max_bounce;
        V3 lighting = ext_direct<COUNT>(sc, m, point, normal, terminal, shadows, stack, cnt, seg);
        if (terminal) {
            V3 out = lighting;
            if (tf > 0.0f) out = transmission_mix(m, lighting, tf);
            radiance = radiance + out * throughput;
            break;
        }
        radiance = radiance + (lighting * (1.0f - tf)) * throughput;

        bool front = dot(normal, d) < 0.0f;
        V3 nf = front ? normal : -normal;
        bool transmit = false;
        if (tf > 0.0f) transmit = rng.next_f32() < tf;
        V3 ndir, norigin;
        V3 albedo = ld3(m.albedo);
        if (transmit) {
            if (channel == 3) {
                uint32_t c = (uint32_t)(rng.next_f32() * 3.0f);
                channel = c < 2 ? c : 2;
                throughput = v3(channel == 0 ? throughput.x * 3.0f : 0.0f, channel == 1 ? throughput.y * 3.0f : 0.0f,
                                channel == 2 ? throughput.z * 3.0f : 0.0f);
            }
            float offs = channel == 0 ? -0.018f : (channel == 1 ? 0.0f : 0.035f); // material.rs:47-52
            float ior_c = m.ior + offs;
            float eta = front ? (1.0f / ior_c) : ior_c;
            float cos_i = -dot(nf, d);
            float sin2_t = eta * eta * (1.0f - cos_i * cos_i);
            if (sin2_t > 1.0f) {
                ndir = d - nf * (2.0f * dot(d, nf));
                norigin = point + nf * EXT_EPS;
            } else {
                float cos_t = sqrtf(1.0f - sin2_t);
                ndir = d * eta + nf * (eta * cos_i - cos_t);
                norigin = point - nf * EXT_EPS;
            }
            ndir = normalize(ndir);
            throughput = throughput * albedo;
        } else if (m.metallic > 0.5f) {
            float u1 = rng.next_f32(), u2 = rng.next_f32();
            V3 r = d - nf * (2.0f * dot(d, nf));
            ndir = normalize(r + unit_vector(u1, u2) * m.roughness);
            if (!(dot(ndir, nf) > 0.0f)) break;
            norigin = point + nf * EXT_EPS;
            throughput = throughput * albedo;
        } else {
            float u1 = rng.next_f32(), u2 = rng.next_f32();
            V3 w = nf + unit_vector(u1, u2);
            if (dot(w, w) < 1e-12f) w = nf;
            ndir = normalize(w);
            norigin = point + nf * EXT_EPS;
            throughput = throughput * albedo;
        }
        if (depth >= 2) {
            float p = fminf(fmaxf(fmaxf(fmaxf(throughput.x, throughput.y), throughput.z), 0.05f), 1.0f);
            if (rng.next_f32() > p) break;
            throughput = v3(throughput.x / p, throughput.y / p, throughput.z / p);
        }
        o = norigin;
        d = ndir;
    }
    return radiance;
}

__device__ __forceinline__ unsigned long long wave_sum(uint32_t v) {
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, WAVE);
    return s;
}

template <bool COUNT>
__global__ __launch_bounds__(WAVE) void k_render_extended(DevScene sc, DevFrame fr, DevTargets tg) {
    extern __shared__ uint32_t s_stack[]; // DevScene::stack_entries * 64 words
    PixelCoord px = block_pixel(fr);
    uint32_t* stack = s_stack + threadIdx.x;
    Counts cnt = {0u, 0u};
    SegCounts seg = {0u, 0u, 0u};
    if (px.valid) {
        V3 sum = v3(0.0f, 0.0f, 0.0f);
        for (uint32_t s = 0; s < fr.spp; s++) sum = sum + ext_trace_path<COUNT>(sc, fr, px.x, px.y, s, stack, cnt, seg);
        float n = (float)fr.spp;
        V3 color = v3(sum.x / n, sum.y / n, sum.z / n);
        const size_t pix = (size_t)px.y * fr.width + px.x;
        if (tg.rgba32f) reinterpret_cast<float4*>(tg.rgba32f)[pix] = make_float4(color.x, color.y, color.z, 1.0f);
        if (tg.chan[0]) reinterpret_cast<uint32_t*>(tg.chan[0])[pix] = unorm8(color.x) | 0xFF000000u;
        if (tg.chan[1]) reinterpret_cast<uint32_t*>(tg.chan[1])[pix] = (unorm8(color.y) << 8) | 0xFF000000u;
        if (tg.chan[2]) reinterpret_cast<uint32_t*>(tg.chan[2])[pix] = (unorm8(color.z) << 16) | 0xFF000000u;
    }
    // segment counts are part of the result (rt_stats.rays): one atomic per wave and counter
    unsigned long long c0 = wave_sum(seg.camera), c1 = wave_sum(seg.continuation), c2 = wave_sum(seg.shadow);
    unsigned long long n0 = COUNT ? wave_sum(cnt.nodes) : 0ull, n1 = COUNT ? wave_sum(cnt.tris) : 0ull;
    if (threadIdx.x == 0 && tg.counters) {
        atomicAdd(&tg.counters[0], c0 + c1 + c2);
        atomicAdd(&tg.counters[3], c0);
        atomicAdd(&tg.counters[4], c1);
        atomicAdd(&tg.counters[5], c2);
        if (COUNT) {
            atomicAdd(&tg.counters[1], n0);
            atomicAdd(&tg.counters[2], n1);
        }
    }
}


// ====================================================================================
// k_render_extended_sm — extended mode as a per-lane STATE MACHINE (v2).
//
// v1 (k_render_extended) nests the loops samples -> path vertices -> lights -> traversal, so a
// wave runs every traversal in lockstep and lanes whose segment ends early idle until the
// slowest lane is done (measured: ~17 % lane utilisation, profiles/r01_extended_*_v1).  Here
// every lane carries its own path state and the wave alternates between two phases:
//   traversal phase  - one BVH step at a time for every lane that has a segment in flight
//                      (closest-hit and any-hit shadow segments mixed); a lane whose segment
//                      finishes parks;
//   transition phase - entered when no lane is traversing or RT_SM_PARK_THRESHOLD lanes are
//                      parked (__ballot + popcount): parked lanes consume their result (shade,
//                      next light, next bounce, next sample) and set up their next segment.
// Per-lane arithmetic and its order are exactly those of v1 / the CPU statement; only the
// interleaving across lanes changes, so results stay bit-identical.
// ====================================================================================
#ifndef RT_SM_MIN_WAVES
#define RT_SM_MIN_WAVES 4
#endif
#ifndef RT_SM_SPECULATIVE
#define RT_SM_SPECULATIVE 0
#endif
#ifndef RT_SM_LEAF_THRESHOLD
#define RT_SM_LEAF_THRESHOLD 24
#endif
#define REF_NONE RT_DEV_LEAF_FLAG /* an empty leaf reference: "nothing to visit" */
#ifndef RT_SM_PARK_THRESHOLD
#define RT_SM_PARK_THRESHOLD 8
#endif

enum : uint32_t { ST_NEW_SAMPLE = 0, ST_CLOSEST_DONE = 1, ST_SHADOW_DONE = 2, ST_LIGHTS = 3, ST_TRAVERSING = 4, ST_DONE = 5 };

template <bool COUNT>
__global__ __launch_bounds__(WAVE, RT_SM_MIN_WAVES) void k_render_extended_sm(DevScene sc, DevFrame fr, DevTargets tg) {
    extern __shared__ uint32_t s_stack[]; // DevScene::stack_entries * 64 words
    const PixelCoord px = block_pixel(fr);
    uint32_t* __restrict__ stack = s_stack + threadIdx.x;
    const uint4* __restrict__ nodes = reinterpret_cast<const uint4*>(sc.nodes);
    Counts cnt = {0u, 0u};
    SegCounts seg = {0u, 0u, 0u};
    const bool shadows = (fr.flags & 2u) == 0;

    // ---- per-lane path state ----
    uint32_t state = px.valid ? ST_NEW_SAMPLE : ST_DONE;
    uint32_t sample = 0, depth = 0, channel = 3, li = 0, material_id = 0;
    SimpleRng rng = {0u};
    V3 sum = v3(0.0f, 0.0f, 0.0f), radiance = sum, throughput = sum, lighting = sum, pending = sum;
    V3 point = sum, normal = sum, din = sum; // vertex position, geometric normal, incoming direction
    bool terminal = false;
    // ---- per-lane segment (traversal) state ----
    V3 o = sum, d = sum;
    FilterRay fray;
    fray.o = sum;
    fray.inv = sum;
    Hit hit;
    hit.t = RT_F32_MAX;
    hit.prim = RT_PRIM_MISS;
    hit.slot = 0;
    uint32_t cur = REF_NONE, pleaf = REF_NONE;
    int sp = 0;
    bool anyhit = false;

    // start a segment: spheres are tested right away (wave-uniform loop), the BVH walk is deferred to the traversal phase
    auto begin_segment = [&](V3 so, V3 sd, float tmax, bool any) {
        o = so;
        d = sd;
        fray = make_filter_ray(so, sd);
        hit.t = tmax;
        hit.prim = RT_PRIM_MISS;
        hit.slot = 0;
        anyhit = any;
        test_spheres(sc, so, sd, hit);
        sp = 0;
        cur = sc.root_ref;
        pleaf = REF_NONE;
        const bool finished = sc.n_tris == 0 || (any && hit.prim != RT_PRIM_MISS);
        state = finished ? (any ? ST_SHADOW_DONE : ST_CLOSEST_DONE) : ST_TRAVERSING;
    };

    // diagnostics of the counting variant (wave-level, lane 0 accumulates): [8] transition passes, [9] lanes served,
    // [10] node iterations, [11] lanes active in them, [12] leaf iterations, [13] lanes active, [14]/[15] cycles in
    // transition / traversal phases
    unsigned long long dg_tp = 0, dg_tl = 0, dg_ni = 0, dg_nl = 0, dg_li = 0, dg_ll = 0, dg_ct = 0, dg_cv = 0;
    for (;;) {
        // =========================== transition phase ===========================
        unsigned long long t_begin = 0;
        if (COUNT) {
            unsigned long long need = __ballot(state != ST_TRAVERSING && state != ST_DONE);
            if (need) {
                dg_tp++;
                dg_tl += __popcll(need);
            }
            t_begin = __builtin_readcyclecounter();
        }
        while (state != ST_TRAVERSING && state != ST_DONE) {
            if (state == ST_NEW_SAMPLE) {
                if (sample >= fr.spp) {
                    float n = (float)fr.spp;
                    V3 color = v3(sum.x / n, sum.y / n, sum.z / n);
                    const size_t pix = (size_t)px.y * fr.width + px.x;
                    if (tg.rgba32f) reinterpret_cast<float4*>(tg.rgba32f)[pix] = make_float4(color.x, color.y, color.z, 1.0f);
                    if (tg.chan[0]) reinterpret_cast<uint32_t*>(tg.chan[0])[pix] = unorm8(color.x) | 0xFF000000u;
                    if (tg.chan[1]) reinterpret_cast<uint32_t*>(tg.chan[1])[pix] = (unorm8(color.y) << 8) | 0xFF000000u;
                    if (tg.chan[2]) reinterpret_cast<uint32_t*>(tg.chan[2])[pix] = (unorm8(color.z) << 16) | 0xFF000000u;
                    state = ST_DONE;
                    break;
                }
                rng = rng_for(fr.frame_seed + px.x + px.y * fr.width, sample);
                float jx = 0.5f, jy = 0.5f;
                if (fr.spp > 1) {
                    jx = rng.next_f32();
                    jy = rng.next_f32();
                }
                V3 co, cd;
                camera_ray(fr.cam, (float)px.x + jx, (float)px.y + jy, true, co, cd);
                radiance = v3(0.0f, 0.0f, 0.0f);
                throughput = v3(1.0f, 1.0f, 1.0f);
                channel = 3;
                depth = 0;
                seg.camera++;
                begin_segment(co, cd, RT_F32_MAX, false);
                continue;
            }
            bool end_sample = false;
            if (state == ST_CLOSEST_DONE) {
                if (hit.prim == RT_PRIM_MISS) {
                    radiance = radiance + v3(0.1f, 0.2f, 0.3f) * throughput;
                    end_sample = true;
                } else {
                    hit_geometry(sc, hit, o, d, point, normal, material_id);
                    din = d;
                    if (material_id >= sc.n_materials) {
                        radiance = radiance + v3(1.0f, 0.0f, 1.0f) * throughput;
                        end_sample = true;
                    } else {
                        terminal = depth >= fr.max_bounce;
                        lighting = v3(0.0f, 0.0f, 0.0f);
                        if (terminal) lighting = lighting + ld3(sc.materials[material_id].albedo) * 0.1f;
                        li = 0;
                        state = ST_LIGHTS;
                    }
                }
            } else if (state == ST_SHADOW_DONE) {
                if (hit.prim == RT_PRIM_MISS) lighting = lighting + pending;
                li++;
                state = ST_LIGHTS;
            }
            if (!end_sample && state == ST_LIGHTS) {
                const DevMaterial m = sc.materials[material_id];
                bool launched = false;
                while (li < sc.n_lights) {
                    V3 sdir;
                    float sdist;
                    V3 contrib = light_contribution(sc.lights[li], m, point, normal, sdir, sdist);
                    if ((contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f) && shadows) {
                        pending = contrib;
                        seg.shadow++;
                        launched = true;
                        begin_segment(point + normal * EXT_EPS, sdir, sdist, true); // overwrites o/d: the incoming direction lives in `din`
                        break;
                    }
                    lighting = lighting + contrib;
                    li++;
                }
                if (launched) continue;
                // ---- all lights done: finish the vertex ----
                lighting = lighting + ld3(m.emission);
                float tf = fminf(fmaxf(m.transmission, 0.0f), 1.0f);
                if (terminal) {
                    V3 out = lighting;
                    if (tf > 0.0f) out = transmission_mix(m, lighting, tf);
                    radiance = radiance + out * throughput;
                    end_sample = true;
                } else {
                    radiance = radiance + (lighting * (1.0f - tf)) * throughput;
                    bool front = dot(normal, din) < 0.0f;
                    V3 nf = front ? normal : -normal;
                    bool transmit = false;
                    if (tf > 0.0f) transmit = rng.next_f32() < tf;
                    V3 ndir, norigin;
                    V3 albedo = ld3(m.albedo);
                    bool absorbed = false;
                    if (transmit) {
                        if (channel == 3) {
                            uint32_t c = (uint32_t)(rng.next_f32() * 3.0f);
                            channel = c < 2 ? c : 2;
                            throughput = v3(channel == 0 ? throughput.x * 3.0f : 0.0f, channel == 1 ? throughput.y * 3.0f : 0.0f,
                                            channel == 2 ? throughput.z * 3.0f : 0.0f);
                        }
                        float offs = channel == 0 ? -0.018f : (channel == 1 ? 0.0f : 0.035f);
                        float ior_c = m.ior + offs;
                        float eta = front ? (1.0f / ior_c) : ior_c;
                        float cos_i = -dot(nf, din);
                        float sin2_t = eta * eta * (1.0f - cos_i * cos_i);
                        if (sin2_t > 1.0f) {
                            ndir = din - nf * (2.0f * dot(din, nf));
                            norigin = point + nf * EXT_EPS;
                        } else {
                            float cos_t = sqrtf(1.0f - sin2_t);
                            ndir = din * eta + nf * (eta * cos_i - cos_t);
                            norigin = point - nf * EXT_EPS;
                        }
                        ndir = normalize(ndir);
                        throughput = throughput * albedo;
                    } else if (m.metallic > 0.5f) {
                        float u1 = rng.next_f32(), u2 = rng.next_f32();
                        V3 r = din - nf * (2.0f * dot(din, nf));
                        ndir = normalize(r + unit_vector(u1, u2) * m.roughness);
                        absorbed = !(dot(ndir, nf) > 0.0f);
                        norigin = point + nf * EXT_EPS;
                        if (!absorbed) throughput = throughput * albedo;
                    } else {
                        float u1 = rng.next_f32(), u2 = rng.next_f32();
                        V3 w = nf + unit_vector(u1, u2);
                        if (dot(w, w) < 1e-12f) w = nf;
                        ndir = normalize(w);
                        norigin = point + nf * EXT_EPS;
                        throughput = throughput * albedo;
                    }
                    if (!absorbed && depth >= 2) {
                        float p = fminf(fmaxf(fmaxf(fmaxf(throughput.x, throughput.y), throughput.z), 0.05f), 1.0f);
                        if (rng.next_f32() > p) absorbed = true;
                        else throughput = v3(throughput.x / p, throughput.y / p, throughput.z / p);
                    }
                    if (absorbed) {
                        end_sample = true;
                    } else {
                        depth++;
                        seg.continuation++;
                        begin_segment(norigin, ndir, RT_F32_MAX, false);
                    }
                }
            }
            if (end_sample) {
                sum = sum + radiance;
                sample++;
                state = ST_NEW_SAMPLE;
            }
        }
        if (COUNT) {
            unsigned long long t_mid = __builtin_readcyclecounter();
            dg_ct += t_mid - t_begin;
            t_begin = t_mid;
        }
        // =========================== anything left? ===========================
        if (__ballot(state == ST_TRAVERSING) == 0ull) {
            if (__ballot(state != ST_DONE) == 0ull) break;
            continue;
        }
        // =========================== traversal phase ===========================
#if RT_SM_SPECULATIVE
        // Speculative descent with one postponed leaf per lane: a lane that reaches a leaf parks the leaf
        // reference in `pleaf` and keeps walking (pops its next subtree) instead of waiting for the other lanes
        // to reach theirs; the triangle tests run for many lanes at once when RT_SM_LEAF_THRESHOLD lanes hold a
        // postponed leaf or no lane can take a node step.  Closest hit (and "any hit" existence) do not depend
        // on the order in which leaves are tested - ties go to the lower triangle index - so this is
        // result-neutral; its only cost is that culling sees the closest hit a little later.
        for (;;) {
            const bool trav = state == ST_TRAVERSING;
            if (trav && (cur & RT_DEV_LEAF_FLAG) && cur != REF_NONE && pleaf == REF_NONE) {
                pleaf = cur;
                if (sp > 0) {
                    sp--;
                    cur = stack[sp * WAVE];
                } else {
                    cur = REF_NONE;
                }
            }
            const bool can_node = trav && !(cur & RT_DEV_LEAF_FLAG);
            const bool has_leaf = trav && pleaf != REF_NONE;
            const unsigned long long m_node = __ballot(can_node);
            const unsigned long long m_leaf = __ballot(has_leaf);
            if (m_node != 0ull && __popcll(m_leaf) < RT_SM_LEAF_THRESHOLD) {
                if (COUNT) {
                    dg_ni++;
                    dg_nl += __popcll(m_node);
                }
                if (can_node) {
                    if (!visit_node4<COUNT>(nodes, sc.stack_entries, fray, hit.t, stack, sp, cur, cnt)) cur = REF_NONE;
                }
            } else if (m_leaf != 0ull) {
                if (COUNT) {
                    dg_li++;
                    dg_ll += __popcll(m_leaf);
                }
                if (has_leaf) {
                    uint32_t start = pleaf & RT_DEV_LEAF_START_MASK;
                    uint32_t count = (pleaf >> RT_DEV_LEAF_COUNT_SHIFT) & 0xFu;
                    pleaf = REF_NONE;
                    for (uint32_t i = 0; i < count; i++) {
                        if (COUNT) cnt.tris++;
                        test_triangle(sc.tris, start + i, o, d, hit);
                        if (anyhit && hit.prim != RT_PRIM_MISS) break;
                    }
                    if (anyhit && hit.prim != RT_PRIM_MISS) state = ST_SHADOW_DONE;
                }
            }
            if (state == ST_TRAVERSING && cur == REF_NONE && pleaf == REF_NONE) state = anyhit ? ST_SHADOW_DONE : ST_CLOSEST_DONE;
            const unsigned long long still = __ballot(state == ST_TRAVERSING);
            const unsigned long long parked = __ballot(state != ST_TRAVERSING && state != ST_DONE);
            if (still == 0ull || __popcll(parked) >= RT_SM_PARK_THRESHOLD) break;
        }
#else
        // while-while: all lanes that stand on an inner node step until none does, then the lanes that reached
        // a leaf test its triangles.  (A speculative variant that postpones one leaf per lane and keeps
        // descending raised lane utilisation of the node steps from 26 % to 48 % but was not faster: the
        // kernel is bound by vector-memory instructions per segment, not by idle lanes — DESIGN.md §4.)
        for (;;) {
            for (;;) {
                const bool want = state == ST_TRAVERSING && !(cur & RT_DEV_LEAF_FLAG);
                const unsigned long long wmask = __ballot(want);
                if (wmask == 0ull) break;
                if (COUNT) {
                    dg_ni++;
                    dg_nl += __popcll(wmask);
                }
                if (want) {
                    if (!visit_node4<COUNT>(nodes, sc.stack_entries, fray, hit.t, stack, sp, cur, cnt))
                        state = anyhit ? ST_SHADOW_DONE : ST_CLOSEST_DONE;
                }
            }
            if (COUNT) {
                unsigned long long lm = __ballot(state == ST_TRAVERSING);
                if (lm) {
                    dg_li++;
                    dg_ll += __popcll(lm);
                }
            }
            if (state == ST_TRAVERSING) { // cur is a leaf reference here
                uint32_t start = cur & RT_DEV_LEAF_START_MASK;
                uint32_t count = (cur >> RT_DEV_LEAF_COUNT_SHIFT) & 0xFu;
                bool stop = false;
                for (uint32_t i = 0; i < count; i++) {
                    if (COUNT) cnt.tris++;
                    test_triangle(sc.tris, start + i, o, d, hit);
                    if (anyhit && hit.prim != RT_PRIM_MISS) {
                        stop = true;
                        break;
                    }
                }
                if (stop) {
                    state = ST_SHADOW_DONE;
                } else if (sp > 0) {
                    sp--;
                    cur = stack[sp * WAVE];
                } else {
                    state = anyhit ? ST_SHADOW_DONE : ST_CLOSEST_DONE;
                }
            }
            const unsigned long long still = __ballot(state == ST_TRAVERSING);
            const unsigned long long parked = __ballot(state != ST_TRAVERSING && state != ST_DONE);
            if (still == 0ull || __popcll(parked) >= RT_SM_PARK_THRESHOLD) break;
        }
#endif
        if (COUNT) dg_cv += __builtin_readcyclecounter() - t_begin;
    }
    unsigned long long c0 = wave_sum(seg.camera), c1 = wave_sum(seg.continuation), c2 = wave_sum(seg.shadow);
    unsigned long long n0 = COUNT ? wave_sum(cnt.nodes) : 0ull, n1 = COUNT ? wave_sum(cnt.tris) : 0ull;
    if (threadIdx.x == 0 && tg.counters) {
        atomicAdd(&tg.counters[0], c0 + c1 + c2);
        atomicAdd(&tg.counters[3], c0);
        atomicAdd(&tg.counters[4], c1);
        atomicAdd(&tg.counters[5], c2);
        if (COUNT) {
            atomicAdd(&tg.counters[1], n0);
            atomicAdd(&tg.counters[2], n1);
            atomicAdd(&tg.counters[8], dg_tp);
            atomicAdd(&tg.counters[9], dg_tl);
            atomicAdd(&tg.counters[10], dg_ni);
            atomicAdd(&tg.counters[11], dg_nl);
            atomicAdd(&tg.counters[12], dg_li);
            atomicAdd(&tg.counters[13], dg_ll);
            atomicAdd(&tg.counters[14], dg_ct);
            atomicAdd(&tg.counters[15], dg_cv);
        }
    }
}

} // namespace

namespace rt {

static size_t lds_bytes(const DevScene& sc) { return (size_t)(sc.stack_entries + 3u) * WAVE * sizeof(uint32_t); } // +3: visit_node4 stores three words unconditionally

uint32_t blocks_per_tile(uint32_t tile_size) {
    uint32_t b = (tile_size + 7u) >> 3;
    return b * b;
}

hipError_t launch_render_reference(const DevScene& sc, const DevFrame& fr, const DevTargets& tg, bool counters, hipStream_t stream) {
    uint32_t n_tiles = fr.single_tile ? 1u : fr.n_owned_tiles;
    if (n_tiles == 0) return hipSuccess;
    dim3 grid(n_tiles * blocks_per_tile(fr.tile_size)), block(WAVE);
    if (counters)
        hipLaunchKernelGGL(k_render_reference<true>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
    else
        hipLaunchKernelGGL(k_render_reference<false>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
    return hipGetLastError();
}

hipError_t launch_render_extended(const DevScene& sc, const DevFrame& fr, const DevTargets& tg, bool counters, hipStream_t stream) {
    uint32_t n_tiles = fr.n_owned_tiles;
    if (n_tiles == 0) return hipSuccess;
    dim3 grid(n_tiles * blocks_per_tile(fr.tile_size)), block(WAVE);
    const bool v1 = (fr.flags & 4u) != 0; // RT_FLAG_KERNEL_V1: the nested-loop kernel, kept for A/B runs
    if (v1) {
        if (counters)
            hipLaunchKernelGGL(k_render_extended<true>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
        else
            hipLaunchKernelGGL(k_render_extended<false>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
    } else {
        if (counters)
            hipLaunchKernelGGL(k_render_extended_sm<true>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
        else
            hipLaunchKernelGGL(k_render_extended_sm<false>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
    }
    return hipGetLastError();
}

} // namespace rt
