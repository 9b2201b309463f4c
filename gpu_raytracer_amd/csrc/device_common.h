// device_common.h — device-side building blocks shared by kernels.hip (megakernels) and wavefront.hip
// (queue-based pipeline): vector math in the reference's operation order, ray generation, sphere and
// Möller–Trumbore tests, the quantised 4-wide BVH visit, the reference's shading, and the extended mode's
// RNG / sampling helpers.  Everything is __forceinline__; both translation units are compiled with
// -ffp-contract=off so the arithmetic that decides hits and colours is identical in all kernels.
#ifndef RT_DEVICE_COMMON_H
#define RT_DEVICE_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"

namespace rtdev {

#define RT_MIN_RAY_DISTANCE 0.00001f
#define RT_F32_MAX 3.402823466e+38f
#define RT_PI 3.14159265358979323846f
#define WAVE 64


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
// Returns the record's leaf_count word (the run length when `slot` is the first triangle of a leaf).
__device__ __forceinline__ uint32_t test_triangle(const DevTri* __restrict__ tris, uint32_t slot, V3 o, V3 d, Hit& hit) {
    const float4* p = reinterpret_cast<const float4*>(tris + slot);
    float4 q0 = p[0], q1 = p[1], q2 = p[2];
    // one 48-byte record = three 16-byte loads issued together (otherwise the compiler splits them by first use
    // and sinks the later words behind the early-outs: more, and dependent, fetches)
    asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w), "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w));
    const uint32_t leaf_count = __float_as_uint(q2.w);
    V3 v0 = v3(q0.x, q0.y, q0.z);
    V3 e1 = v3(q0.w, q1.x, q1.y);
    V3 e2 = v3(q1.z, q1.w, q2.x);
    V3 h = cross(d, e2);
    float a = dot(e1, h);
    if (fabsf(a) < RT_MIN_RAY_DISTANCE) return leaf_count;
    float f = 1.0f / a;
    V3 s = o - v0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return leaf_count;
    V3 q = cross(s, e1);
    float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return leaf_count;
    float t = f * dot(e2, q);
    uint32_t prim = __float_as_uint(q2.z);
    // equal t: the lower index wins among TRIANGLES only; against a sphere hit or the segment's own limit (prim = MISS, shadow
    // segments) the comparison stays strict, as in find_closest_intersection (lib.rs:214-248: the sphere is kept unless the
    // triangle is strictly closer)
    if (t > RT_MIN_RAY_DISTANCE && (t < hit.t || (t == hit.t && prim < hit.prim && hit.prim < RT_PRIM_SPHERE_FLAG))) {
        hit.t = t;
        hit.prim = prim;
        hit.slot = slot;
    }
    return leaf_count;
}

// All triangles of the leaf `ref`.  ANY_HIT: stop at the first accepted one and return true.
template <bool COUNT, bool ANY_HIT>
__device__ __forceinline__ bool test_leaf(const DevTri* __restrict__ tris, uint32_t ref, V3 o, V3 d, Hit& hit, Counts& cnt) {
    const uint32_t start = ref & RT_DEV_LEAF_START_MASK;
    uint32_t n = 1;
    for (uint32_t i = 0; i < n; i++) {
        if (COUNT) cnt.tris++;
        const uint32_t lc = test_triangle(tris, start + i, o, d, hit);
        if (i == 0) n = lc;
        if (ANY_HIT && hit.prim != RT_PRIM_MISS) return true;
    }
    return false;
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
// Möller–Trumbore arithmetic — so they may be conservative but must never be too small, and "too
// small" is judged against what the FLOAT triangle test accepts, not against exact geometry: a ray
// that grazes an edge can be accepted by the triangle test while missing the exact box by rounding.
// The quantised planes are rounded outward by the builder; the float evaluation here is widened:
// with a = scale/d, b = (org - o)/d the plane distances are fma(q, a, b), whose own absolute error
// is below 2^-22 * (|org - o| + 255 * scale) / |d| (one rounding each in org - o, a, b, 1/d and the
// fma).  Near planes are moved back and far planes forward by RT_FILTER_SLACK (1e-6, four times that
// bound) times the same magnitude; the margin is there for the triangle test's own rounding.  All parity
// tests (up to 3.8 M triangles, bit-exact against brute-force / mesh-order oracles) pass with it.
//
// The stack lives in LDS, lane-interleaved (entry k of lane l at stack[k * 64 + l]): ds_read /
// ds_write_b32 with consecutive lanes on consecutive banks.  A visit pushes at most 3 entries, the
// launch provides 3 * depth + 1 entries per lane (DevScene::stack_entries), so it cannot overflow.
// ------------------------------------------------------------------------------------
#ifndef RT_FILTER_SLACK
#define RT_FILTER_SLACK 1.0e-6f
#endif
#ifndef RT_EXPERIMENT_NO_WIDENING
#define RT_EXPERIMENT_NO_WIDENING 0 /* measurement only */
#endif
#ifndef RT_FILTER_RCP
#define RT_FILTER_RCP 1 /* round 2: -1 % on the headline frame with both trees; the filter's error bound is derived below */
#endif
struct FilterRay { // per-segment constants of the box filter
    V3 o, inv;     // inv = 1/d with |d| clamped away from zero (a filter may do that; the triangle test uses the real d)
};
__device__ __forceinline__ FilterRay make_filter_ray(V3 o, V3 d) {
    FilterRay f;
    f.o = o;
    float dx = fabsf(d.x) < 1e-30f ? copysignf(1e-30f, d.x) : d.x;
    float dy = fabsf(d.y) < 1e-30f ? copysignf(1e-30f, d.y) : d.y;
    float dz = fabsf(d.z) < 1e-30f ? copysignf(1e-30f, d.z) : d.z;
#if RT_FILTER_RCP
    // v_rcp_f32 (1 ulp) instead of three IEEE divisions (~11 instructions each): the filter's error bound grows
    // from 2^-22 to 2^-22 + 2^-23 relative, still far inside RT_FILTER_SLACK
    f.inv = v3(__builtin_amdgcn_rcpf(dx), __builtin_amdgcn_rcpf(dy), __builtin_amdgcn_rcpf(dz));
#else
    f.inv = v3(1.0f / dx, 1.0f / dy, 1.0f / dz);
#endif
    return f;
}

// Per-lane stack access.  LDS_N == 0: every entry lives in LDS (lane-interleaved).  LDS_N > 0: entries below LDS_N
// live in LDS, deeper ones in a global overflow area (also lane-interleaved): measured on the sponza-like and
// bistro-like scenes the stack never exceeds 20 entries and is deeper than 16 in < 0.001 % of the visits, so a
// short LDS part costs ~4 KB per wave instead of (3 * depth + 4) * 256 B and more than doubles the waves per CU.
// Callers test wave-wide whether any lane is near the LDS limit and take the branch-free LDS-only path if none is.
template <int LDS_N>
__device__ __forceinline__ void stack_store(uint32_t* __restrict__ lds, uint32_t* __restrict__ ovf, int k, uint32_t v) {
    if (LDS_N == 0 || k < LDS_N) lds[k * WAVE] = v;
    else ovf[(k - LDS_N) * WAVE] = v;
}
template <int LDS_N>
__device__ __forceinline__ uint32_t stack_load(const uint32_t* __restrict__ lds, const uint32_t* __restrict__ ovf, int k) {
    if (LDS_N == 0 || k < LDS_N) return lds[k * WAVE];
    return ovf[(k - LDS_N) * WAVE];
}
// pop for a whole wave: `sp` already decremented in the lanes where `want` holds
template <int LDS_N>
__device__ __forceinline__ uint32_t stack_pop(const uint32_t* __restrict__ lds, const uint32_t* __restrict__ ovf, int sp) {
    if (LDS_N == 0 || __ballot(sp >= LDS_N) == 0ull) return lds[sp * WAVE];
    return stack_load<LDS_N>(lds, ovf, sp);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define RT_KEY_MASK 0x7FFFFFFCu /* entry distance (non-negative float bits) with the two low mantissa bits replaced by the child slot */
#define RT_KEY_MISS 0x7F7FFFFCu /* F32_MAX & RT_KEY_MASK: keys at or above it are children the ray does not enter */
#define RT_KEY_SORT2(a, b)            \
    {                                 \
        const uint32_t lo_ = min(a, b); \
        b = max(a, b);                \
        a = lo_;                      \
    }

// One visit of inner node `cur`.  Updates cur / the stack; returns false when the walk is exhausted.
template <bool COUNT, int LDS_N = 0>
__device__ __forceinline__ bool visit_node4(const uint4* __restrict__ nodes, const FilterRay& fr, float closest_t, uint32_t* __restrict__ stack,
                                            int& sp, uint32_t& cur, Counts& cnt, uint32_t* __restrict__ ovf = nullptr) {
    const uint4* n = nodes + (size_t)cur * 3;
    const uint4 w0 = n[0], w1 = n[1], w2 = n[2];
    if (COUNT) cnt.nodes++;
    // grid scale 2^k per axis, k a signed byte: a = scale / d as one ldexp of the ray's reciprocal (exact)
    const float ax = ldexpf(fr.inv.x, (int)(int8_t)(w0.w & 0xFFu)), ay = ldexpf(fr.inv.y, (int)(int8_t)((w0.w >> 8) & 0xFFu)),
                az = ldexpf(fr.inv.z, (int)(int8_t)((w0.w >> 16) & 0xFFu));
    const float dox = __uint_as_float(w0.x) - fr.o.x, doy = __uint_as_float(w0.y) - fr.o.y, doz = __uint_as_float(w0.z) - fr.o.z;
    const float bx = dox * fr.inv.x, by = doy * fr.inv.y, bz = doz * fr.inv.z;
    // widening per axis: RT_FILTER_SLACK * (|b| + 255 |a|), see above.  It stays per axis: a ray almost parallel to an
    // axis has an enormous |b| (and error bound) on that axis only; one bound for the whole interval test
    // (tmin - 2E <= tmax, E summed over the axes) is 5 instructions cheaper and 26 % slower because the degenerate axis
    // inflates every box.  Without any widening the frame is 3.7 % faster (RT_EXPERIMENT_NO_WIDENING; not conservative).
    const float ex = fmaf(255.0f, fabsf(ax), fabsf(bx)) * RT_FILTER_SLACK;
    const float ey = fmaf(255.0f, fabsf(ay), fabsf(by)) * RT_FILTER_SLACK;
    const float ez = fmaf(255.0f, fabsf(az), fabsf(bz)) * RT_FILTER_SLACK;
    // (near, far) pairs (b - e, b + e) as one packed fma each; then one packed fma per child and axis
#if RT_EXPERIMENT_NO_WIDENING
    const f32x2 bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
    (void)ex; (void)ey; (void)ez;
#else
    const f32x2 pm = {-1.0f, 1.0f};
    const f32x2 bx2 = __builtin_elementwise_fma((f32x2){ex, ex}, pm, (f32x2){bx, bx});
    const f32x2 by2 = __builtin_elementwise_fma((f32x2){ey, ey}, pm, (f32x2){by, by});
    const f32x2 bz2 = __builtin_elementwise_fma((f32x2){ez, ez}, pm, (f32x2){bz, bz});
#endif
    const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
    // entry planes are the lower ones along axes the ray travels in +, the upper ones otherwise
    const bool px = fr.inv.x >= 0.0f, py = fr.inv.y >= 0.0f, pz = fr.inv.z >= 0.0f;
    const uint32_t nxw = px ? w1.z : w2.y, fxw = px ? w2.y : w1.z; // w1.z qlo_x, w2.y qhi_x
    const uint32_t nyw = py ? w1.w : w2.z, fyw = py ? w2.z : w1.w; // w1.w qlo_y, w2.z qhi_y
    const uint32_t nzw = pz ? w2.x : w2.w, fzw = pz ? w2.w : w2.x; // w2.x qlo_z, w2.w qhi_z
    const float limit = closest_t * 1.0000153f; // culling with slack, so equal-t candidates are still visited
    uint32_t k[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f32x2 qx = {(float)((nxw >> (8 * i)) & 0xFFu), (float)((fxw >> (8 * i)) & 0xFFu)};
        const f32x2 qy = {(float)((nyw >> (8 * i)) & 0xFFu), (float)((fyw >> (8 * i)) & 0xFFu)};
        const f32x2 qz = {(float)((nzw >> (8 * i)) & 0xFFu), (float)((fzw >> (8 * i)) & 0xFFu)};
        const f32x2 tx = __builtin_elementwise_fma(qx, ax2, bx2);
        const f32x2 ty = __builtin_elementwise_fma(qy, ay2, by2);
        const f32x2 tz = __builtin_elementwise_fma(qz, az2, bz2);
        const float tmin = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, 0.0f));
        const float tmax = fminf(fminf(tx.y, ty.y), fminf(tz.y, limit));
        const float t = (tmin <= tmax) ? tmin : RT_F32_MAX; // absent children are inverted boxes: never entered
        k[i] = (__float_as_uint(t) & RT_KEY_MASK) | (uint32_t)i;
    }
    // keep the node's three 16-byte loads together (the reference words are only used after the sort; without
    // this the compiler sinks their load into the branch below, a fourth and dependent fetch)
    uint32_t w1x = w1.x, w1y = w1.y;
    asm volatile("" : "+v"(w1x), "+v"(w1y));
    // sort the four keys: children the ray does not enter end up last.  (The order among children whose entry
    // distances agree to 2 ulp is arbitrary: result-neutral, ties between HITS are resolved by triangle index.)
    uint32_t k0 = k[0], k1 = k[1], k2 = k[2], k3 = k[3];
    RT_KEY_SORT2(k0, k1)
    RT_KEY_SORT2(k2, k3)
    RT_KEY_SORT2(k0, k2)
    RT_KEY_SORT2(k1, k3)
    RT_KEY_SORT2(k1, k2)
    if (k0 >= RT_KEY_MISS) { // nothing entered: continue with the stack
        if (sp == 0) return false;
        sp--;
        cur = stack_pop<LDS_N>(stack, ovf, sp);
        return true;
    }
    // child references from the sorted slots: base + 4-bit offset
    const uint32_t child_off = (w0.w >> 24) | ((w1x >> 16) & 0xFF00u);
    const uint32_t base_inner = w1x & RT_DEV_NODE_BASE_MASK;
    const uint32_t base_leaf = (w1y & RT_DEV_TRI_BASE_MASK) | RT_DEV_LEAF_FLAG;
    const uint32_t n_inner4 = (w1y >> 25) & 0x1Cu; // 4 * n_inner
#define RT_CHILD_REF(key, out)                                                            \
    {                                                                                     \
        const uint32_t sh_ = ((key) << 2) & 12u;                                          \
        out = (sh_ < n_inner4 ? base_inner : base_leaf) + ((child_off >> sh_) & 15u);     \
    }
    // push the other entered children far-to-near without branching: always store three words (the launch
    // provides three spare entries), advance the pointer by the number of real ones
    const int extra = (k1 < RT_KEY_MISS ? 1 : 0) + (k2 < RT_KEY_MISS ? 1 : 0) + (k3 < RT_KEY_MISS ? 1 : 0);
    const uint32_t ek0 = extra == 3 ? k3 : (extra == 2 ? k2 : k1);
    const uint32_t ek1 = extra == 3 ? k2 : k1;
    uint32_t r0, e0, e1, r1;
    RT_CHILD_REF(k0, r0)
    RT_CHILD_REF(ek0, e0)
    RT_CHILD_REF(ek1, e1)
    RT_CHILD_REF(k1, r1)
#undef RT_CHILD_REF
    cur = r0;
    if (LDS_N == 0 || __ballot(sp + 3 > LDS_N) == 0ull) {
        stack[sp * WAVE] = e0;
        stack[(sp + 1) * WAVE] = e1;
        stack[(sp + 2) * WAVE] = r1;
    } else {
        stack_store<LDS_N>(stack, ovf, sp, e0);
        stack_store<LDS_N>(stack, ovf, sp + 1, e1);
        stack_store<LDS_N>(stack, ovf, sp + 2, r1);
    }
    sp += extra;
    return true;
}

// One visit of the 8-wide node `idx` (DevNode8, device_layout.h): the same conservative quantised-box filter as
// visit_node4 for eight slots, but no sort: the result is the 8-bit mask of the slots the ray enters (bit s = slot s);
// the caller visits them in increasing (slot XOR ray octant).  Returns the mask, already restricted to occupied slots.
template <bool COUNT>
__device__ __forceinline__ uint32_t visit_node8(const uint4* __restrict__ nodes, uint32_t idx, const FilterRay& fr, float closest_t, Counts& cnt,
                                                uint32_t& child_base, uint32_t& tri_base, uint32_t& imask, uint32_t& lmask) {
    const uint4* n = nodes + (size_t)idx * 5;
    const uint4 w0 = n[0], w1 = n[1], w2 = n[2], w3 = n[3], w4 = n[4];
    // keep the node's loads together: the words that are only used after the box tests would otherwise be fetched there
    uint32_t w1x = w1.x, w1y = w1.y, w1z = w1.z;
    asm volatile("" : "+v"(w1x), "+v"(w1y), "+v"(w1z));
    if (COUNT) cnt.nodes++;
    const float ax = ldexpf(fr.inv.x, (int)(int8_t)(w0.w & 0xFFu)), ay = ldexpf(fr.inv.y, (int)(int8_t)((w0.w >> 8) & 0xFFu)),
                az = ldexpf(fr.inv.z, (int)(int8_t)((w0.w >> 16) & 0xFFu));
    const float dox = __uint_as_float(w0.x) - fr.o.x, doy = __uint_as_float(w0.y) - fr.o.y, doz = __uint_as_float(w0.z) - fr.o.z;
    const float bx = dox * fr.inv.x, by = doy * fr.inv.y, bz = doz * fr.inv.z;
    const float ex = fmaf(255.0f, fabsf(ax), fabsf(bx)) * RT_FILTER_SLACK;
    const float ey = fmaf(255.0f, fabsf(ay), fabsf(by)) * RT_FILTER_SLACK;
    const float ez = fmaf(255.0f, fabsf(az), fabsf(bz)) * RT_FILTER_SLACK;
    const f32x2 pm = {-1.0f, 1.0f};
    const f32x2 bx2 = __builtin_elementwise_fma((f32x2){ex, ex}, pm, (f32x2){bx, bx});
    const f32x2 by2 = __builtin_elementwise_fma((f32x2){ey, ey}, pm, (f32x2){by, by});
    const f32x2 bz2 = __builtin_elementwise_fma((f32x2){ez, ez}, pm, (f32x2){bz, bz});
    const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az};
    const bool px = fr.inv.x >= 0.0f, py = fr.inv.y >= 0.0f, pz = fr.inv.z >= 0.0f;
    // [half]: w2 = qlo_x[0], qlo_x[1], qlo_y[0], qlo_y[1]; w3 = qlo_z[0], qlo_z[1], qhi_x[0], qhi_x[1]; w4 = qhi_y[0], qhi_y[1], qhi_z[0], qhi_z[1]
    const uint32_t nxw[2] = {px ? w2.x : w3.z, px ? w2.y : w3.w}, fxw[2] = {px ? w3.z : w2.x, px ? w3.w : w2.y};
    const uint32_t nyw[2] = {py ? w2.z : w4.x, py ? w2.w : w4.y}, fyw[2] = {py ? w4.x : w2.z, py ? w4.y : w2.w};
    const uint32_t nzw[2] = {pz ? w3.x : w4.z, pz ? w3.y : w4.w}, fzw[2] = {pz ? w4.z : w3.x, pz ? w4.w : w3.y};
    const float limit = closest_t * 1.0000153f;
    uint32_t miss = 0; // bit s: the ray does NOT enter slot s (sign bits of tmax - tmin, shifted in one per slot)
#pragma unroll
    for (int h = 1; h >= 0; h--) {
#pragma unroll
        for (int i = 3; i >= 0; i--) {
            const f32x2 qx = {(float)((nxw[h] >> (8 * i)) & 0xFFu), (float)((fxw[h] >> (8 * i)) & 0xFFu)};
            const f32x2 qy = {(float)((nyw[h] >> (8 * i)) & 0xFFu), (float)((fyw[h] >> (8 * i)) & 0xFFu)};
            const f32x2 qz = {(float)((nzw[h] >> (8 * i)) & 0xFFu), (float)((fzw[h] >> (8 * i)) & 0xFFu)};
            const f32x2 tx = __builtin_elementwise_fma(qx, ax2, bx2);
            const f32x2 ty = __builtin_elementwise_fma(qy, ay2, by2);
            const f32x2 tz = __builtin_elementwise_fma(qz, az2, bz2);
            const float tmin = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, 0.0f));
            const float tmax = fminf(fminf(tx.y, ty.y), fminf(tz.y, limit));
            // entered <=> tmin <= tmax <=> tmax - tmin >= +0 (all finite): v_alignbit shifts the sign bit in; slot 4h+i ends up in bit 4h+i
            miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tmax - tmin), 31);
        }
    }
    child_base = w1x;
    tri_base = w1y;
    lmask = w1z & 0xFFu;
    imask = w0.w >> 24;
    return ~miss & (imask | lmask);
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
            if (!visit_node4<COUNT>(nodes, fr, hit.t, stack, sp, cur, cnt)) break;
            continue;
        }
        if (test_leaf<COUNT, ANY_HIT>(sc.tris, cur, o, d, hit, cnt)) return;
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
    V3 dir_light_dir = ld3(L.neg_ndir); // -normalize(direction), lighting.rs:103 (per-light constant, DevLight)
    float dir_intensity = fmaxf(dot(normal, dir_light_dir), 0.0f) * L.intensity;
    V3 to_light = ld3(L.position) - point; // lighting.rs:120-122
    float distance = length(to_light);
    V3 pld = normalize(to_light);
    float att = f16_round_trip(1.0f / (1.0f + distance * distance * 0.01f)); // :125-127
    float point_intensity = fmaxf(dot(normal, pld), 0.0f) * L.intensity * att;
    float spot_factor = fmaxf(dot(dir_light_dir, pld), 0.0f); // :132 (the same -normalize(direction))
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
// geometric normal and material at a hit point: `sphere` selects DevScene::spheres[slot], else DevScene::tris[slot]
__device__ __forceinline__ void surface_at(const DevScene& sc, bool sphere, uint32_t slot, V3 point, V3& normal, uint32_t& material_id) {
    if (sphere) {
        const DevSphere& s = sc.spheres[slot];
        normal = normalize(point - ld3(s.center)); // intersection.rs:80
        material_id = s.material_id;
    } else {
        const float4* p = reinterpret_cast<const float4*>(sc.tris + slot);
        float4 q0 = p[0], q1 = p[1], q2 = p[2];
        asm volatile("" : "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w), "+v"(q2.x), "+v"(q2.y)); // one round trip for the record
        V3 e1 = v3(q0.w, q1.x, q1.y);
        V3 e2 = v3(q1.z, q1.w, q2.x);
        normal = normalize(cross(e1, e2)); // geometric, winding dependent, never flipped (intersection.rs:132)
        material_id = __float_as_uint(q2.y);
    }
}
__device__ __forceinline__ void hit_geometry(const DevScene& sc, const Hit& hit, V3 o, V3 d, V3& point, V3& normal, uint32_t& material_id) {
    point = o + d * hit.t; // Ray::at, ray.rs:56-58
    surface_at(sc, (hit.prim & RT_PRIM_SPHERE_FLAG) != 0, hit.slot, point, normal, material_id);
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
__device__ __forceinline__ PixelCoord block_pixel_at(const DevFrame& fr, uint32_t block, uint32_t lane) {
    uint32_t ox, oy, tw, th, blk;
    const uint32_t bpt_x = (fr.tile_size + 7u) >> 3;
    if (fr.single_tile) {
        ox = fr.tile_off_x; oy = fr.tile_off_y; tw = fr.tile_w; th = fr.tile_h;
        blk = block;
    } else {
        const uint32_t bpt = bpt_x * bpt_x;
        uint32_t k = block / bpt;
        blk = block - k * bpt;
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
__device__ __forceinline__ PixelCoord block_pixel(const DevFrame& fr) { return block_pixel_at(fr, blockIdx.x, threadIdx.x); }

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

__device__ __forceinline__ unsigned long long wave_sum(uint32_t v) {
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, WAVE);
    return s;
}


// Direction and length of the shadow segment toward light L from `point`: the same expressions
// light_contribution evaluates (lighting.rs:103, 120-122), so both produce the same bits.
__device__ __forceinline__ void shadow_segment(const DevLight& L, V3 point, V3& dir, float& dist) {
    if (L.light_type == 0) {
        dir = ld3(L.neg_ndir);
        dist = RT_F32_MAX;
    } else {
        V3 to_light = ld3(L.position) - point;
        dist = length(to_light);
        dir = normalize(to_light);
    }
}

// Wave-aggregated queue append: one atomicAdd per wave, slots handed out by ballot + prefix count.
__device__ __forceinline__ void wave_append(uint32_t* __restrict__ queue, uint32_t* __restrict__ counter, bool pred, uint32_t value) {
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, (int)leader, WAVE);
    if (pred) queue[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

} // namespace rtdev

#endif
