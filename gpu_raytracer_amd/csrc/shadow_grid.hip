// shadow_grid.hip — builds the per-light triangle lists of shadow_grid.h on the device (at scene upload).
//
// One thread per triangle record.  Point / spot light (cube map about L): the triangle is clipped against each face's
// frustum (Sutherland-Hodgman on the four side planes, widened by the margin), projected (u = x / z, v = y / z), and every
// cell its projection may touch gets an entry: the cells of the projected polygon's bounding box that no polygon edge
// separates from the polygon by more than the margin.  Directional light: the same on one orthographic grid, no clipping.
//
// Why the lists are supersets of what the triangle test can accept (the property the shadow stage relies on):
//  * a shadow segment starts at o = P + n * EXT_EPS and runs along d = fl(normalize(L - P)) to t = fl(|L - P|), so it does
//    not pass through L itself but within eps_eff = EXT_EPS + rounding of it; a point X of it at distance r from L is seen
//    from L within an angle eps_eff / r of -d.  A triangle whose nearest point to L is r_min away can therefore only be
//    hit by segments whose direction -d lies within eps_eff / r_min of the triangle's own directions: its projection is
//    dilated by that angle (x 3.5 for the stretch of the cube projection, 1 + u^2 + v^2 <= 3, and second order).  Triangles
//    so close that this exceeds 8 cells go to the near list, which every segment of the light tests;
//  * on top comes a margin of half a cell for everything that rounds: the segment's own cell (a quotient of its direction),
//    the clip / projection arithmetic here (relative 1e-6), and the triangle test's tolerance (it accepts rays that pass
//    a triangle's edge by ~1e-7 of the coordinates' magnitude): all of these are below a hundredth of a cell for any
//    scene whose grid is enabled (shadow_grid_build refuses scenes whose coordinates are too large for that);
//  * keys: a triangle is skipped by segments that end before its nearest point (point lights: key = r_min shrunk by 1e-4 and 1e-5 of its extent,
//    the segment's limit is its length + 2 eps_eff + slack; directional: the coordinate along the light's direction).
#include "shadow_grid.h"

#include <hip/hip_runtime.h>

#include <cstring> // rocPRIM's headers call memset on the host without including it

#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>

namespace {

#define SG_EXT_EPS RT_SG_EXT_EPS
#define SG_MARGIN_CELLS 0.5f
#define SG_MAX_CONE_CELLS 8.0f

struct SgParams {
    const DevTri* tris;
    uint32_t n_records;
    uint32_t kind, res, n_cells;
    float L[3];
    float eps_eff;
    float scale;         // cube: res / 2; ortho: cells per unit
    float au[3], av[3], aw[3];
    float u0, v0, key_top;
    float margin_cells;  // rounding margin (>= SG_MARGIN_CELLS)
    uint32_t* cell_count; // count pass: entries per cell; fill pass: cursor per cell
    const uint32_t* ovf_start; // fill pass: where a list's entries beyond its block start in `overflow`
    uint4* blocks;
    uint4* overflow;
};

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float comp(F3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// squared distance from the origin to the triangle (a, b, c): closest point by regions (Ericson, Real-Time Collision Detection 5.1.5)
__device__ float sg_origin_tri_dist2(F3 a, F3 b, F3 c) {
    const F3 ab = b - a, ac = c - a;
    const F3 ap = f3(-a.x, -a.y, -a.z);
    const float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0.0f && d2 <= 0.0f) return dot3(a, a);
    const F3 bp = f3(-b.x, -b.y, -b.z);
    const float d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3 >= 0.0f && d4 <= d3) return dot3(b, b);
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
        const float v = d1 / (d1 - d3);
        const F3 q = a + ab * v;
        return dot3(q, q);
    }
    const F3 cp = f3(-c.x, -c.y, -c.z);
    const float d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0.0f && d5 <= d6) return dot3(c, c);
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
        const float w = d2 / (d2 - d6);
        const F3 q = a + ac * w;
        return dot3(q, q);
    }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        const F3 q = b + (c - b) * w;
        return dot3(q, q);
    }
    const float denom = 1.0f / (va + vb + vc);
    const float v = vb * denom, w = vc * denom;
    const F3 q = a + ab * v + ac * w;
    return dot3(q, q);
}

struct SgTri { // what an entry carries besides its key
    uint4 q0, q1, q2; // q0.x is overwritten with the key
};

template <bool FILL>
__device__ __forceinline__ void sg_emit(const SgParams& p, uint32_t cell, const SgTri& tri, float key) {
    if (!FILL) {
        atomicAdd(&p.cell_count[cell], 1u);
    } else {
        const uint32_t k = atomicAdd(&p.cell_count[cell], 1u);
        const uint32_t in_block = cell < p.n_cells ? RT_SG_BLOCK_ENTRIES : 0u; // (the near list lives in the overflow array only)
        uint4* dst = k < in_block ? p.blocks + (size_t)cell * RT_SG_BLOCK_QUADS + 1u + RT_SG_ENTRY_QUADS * k
                                  : p.overflow + (size_t)(p.ovf_start[cell] + (k - in_block)) * RT_SG_ENTRY_QUADS;
        dst[0] = make_uint4(__float_as_uint(key), tri.q0.y, tri.q0.z, tri.q0.w);
        dst[1] = tri.q1;
        dst[2] = tri.q2;
    }
}

// Every cell of one face that the convex polygon (px, py)[0..n) - in cell coordinates - may touch when dilated by mc cells.
template <bool FILL>
__device__ void sg_raster(const SgParams& p, uint32_t face_base, const float* px, const float* py, int n, float mc, const SgTri& tri, float key, uint32_t lane) {
    const float R = (float)p.res;
    float minx = px[0], maxx = px[0], miny = py[0], maxy = py[0];
    for (int k = 1; k < n; k++) {
        minx = fminf(minx, px[k]);
        maxx = fmaxf(maxx, px[k]);
        miny = fminf(miny, py[k]);
        maxy = fmaxf(maxy, py[k]);
    }
    if (!(minx - mc < R) || !(maxx + mc >= 0.0f) || !(miny - mc < R) || !(maxy + mc >= 0.0f)) return; // outside the face (or NaN)
    const int x0 = (int)fmaxf(floorf(minx - mc), 0.0f), x1 = (int)fminf(floorf(maxx + mc), R - 1.0f);
    const int y0 = (int)fmaxf(floorf(miny - mc), 0.0f), y1 = (int)fminf(floorf(maxy + mc), R - 1.0f);
    if (x1 < x0 || y1 < y0) return;
    // edge functions: outward normal (nx, ny) of edge k, a cell is outside when its corner nearest to the polygon is
    // farther out than the margin (in the L1 norm of the normal: never less than the Euclidean distance asks for)
    float area2 = 0.0f; // twice the signed area, from coordinates relative to the box (so that it does not cancel)
    for (int k = 0; k < n; k++) {
        const int k1 = k + 1 == n ? 0 : k + 1;
        area2 += (px[k] - minx) * (py[k1] - miny) - (px[k1] - minx) * (py[k] - miny);
    }
    // slivers (whose orientation is not trustworthy in f32) and small boxes: the box is the answer
    const bool edges = n >= 3 && fabsf(area2) > 1e-3f * ((maxx - minx) * (maxy - miny)) + 1e-3f && (x1 - x0 > 1 || y1 - y0 > 1);
    const float sgn = area2 > 0.0f ? 1.0f : -1.0f;
    float nx[8], ny[8], lim[8];
    if (edges) {
        for (int k = 0; k < n; k++) {
            const int k1 = k + 1 == n ? 0 : k + 1;
            nx[k] = sgn * (py[k1] - py[k]);
            ny[k] = -sgn * (px[k1] - px[k]);
            lim[k] = (mc + 0.25f) * (fabsf(nx[k]) + fabsf(ny[k])); // + a quarter cell for the rounding of the products below
        }
    }
    // the 64 lanes of the wave that owns this triangle share the box's cells
    const uint32_t w = (uint32_t)(x1 - x0 + 1), cells = w * (uint32_t)(y1 - y0 + 1);
    for (uint32_t c = lane; c < cells; c += 64u) {
        const uint32_t row = c / w;
        const int x = x0 + (int)(c - row * w), y = y0 + (int)row;
        bool outside = false;
        if (edges) {
            for (int k = 0; k < n; k++) {
                const float cx = nx[k] > 0.0f ? (float)x : (float)(x + 1), cy = ny[k] > 0.0f ? (float)y : (float)(y + 1);
                if (nx[k] * (cx - px[k]) + ny[k] * (cy - py[k]) > lim[k]) {
                    outside = true;
                    break;
                }
            }
        }
        if (!outside) sg_emit<FILL>(p, face_base + (uint32_t)y * p.res + (uint32_t)x, tri, key);
    }
}

// Sutherland-Hodgman against the half space cx * x + cy * y - k * z <= 0 (face-local coordinates, z toward the face)
__device__ int sg_clip(const F3* in, int n, F3* out, float cx, float cy, float k) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        const F3 a = in[i], b = in[i + 1 == n ? 0 : i + 1];
        const float da = cx * a.x + cy * a.y - k * a.z, db = cx * b.x + cy * b.y - k * b.z;
        const bool ia = da <= 0.0f, ib = db <= 0.0f;
        if (ia && m < 8) out[m++] = a;
        if (ia != ib && m < 8) {
            const float t = da / (da - db);
            out[m++] = a + (b - a) * t;
        }
    }
    return m;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_sg_raster(SgParams p) {
    // one WAVE per triangle record: every lane works out the same polygons, the cells of a polygon's box are shared out over the
    // lanes (one thread per triangle spent 0.15 s per light on the few triangles that cover a hundred thousand cells)
    const uint32_t r = (uint32_t)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63u;
    if (r >= p.n_records) return;
    const uint32_t leaf = r & ~(RT_DEV_LEAF_STRIDE - 1u);
    if (r - leaf >= p.tris[leaf].leaf_count) return; // padding record
    const DevTri t = p.tris[r];
    SgTri tri;
    {
        const uint4* q = reinterpret_cast<const uint4*>(p.tris + r); // {v0.xyz, e1.x} {e1.yz, e2.xy} {e2.z, material, prim, count}
        const uint4 a = q[0], b = q[1], c = q[2];
        tri.q0 = make_uint4(0u, a.x, a.y, a.z);
        tri.q1 = make_uint4(a.w, b.x, b.y, b.z);
        tri.q2 = make_uint4(b.w, c.x, r, 0u);
    }
    // the triangle the test sees: (v0, v0 + e1, v0 + e2)
    const F3 v0 = f3(t.v0[0], t.v0[1], t.v0[2]);
    const F3 v1 = v0 + f3(t.e1[0], t.e1[1], t.e1[2]), v2 = v0 + f3(t.e2[0], t.e2[1], t.e2[2]);
    const float sum = (v0.x + v0.y + v0.z) + (v1.x + v1.y + v1.z) + (v2.x + v2.y + v2.z);
    if (!(fabsf(sum) < 3.0e38f)) return; // a vertex that is not finite: the triangle test accepts nothing (NaN / inf in every product)
    if (p.kind == RT_SG_KIND_ORTHO) {
        const F3 au = f3(p.au[0], p.au[1], p.au[2]), av = f3(p.av[0], p.av[1], p.av[2]), aw = f3(p.aw[0], p.aw[1], p.aw[2]);
        float px[3], py[3];
        const F3 v[3] = {v0, v1, v2};
        float smax = -3.0e38f;
        for (int k = 0; k < 3; k++) {
            px[k] = (dot3(v[k], au) - p.u0) * p.scale;
            py[k] = (dot3(v[k], av) - p.v0) * p.scale;
            smax = fmaxf(smax, dot3(v[k], aw));
        }
        sg_raster<FILL>(p, 0u, px, py, 3, p.margin_cells, tri, p.key_top - smax, lane);
        return;
    }
    const F3 L = f3(p.L[0], p.L[1], p.L[2]);
    const F3 q0 = v0 - L, q1 = v1 - L, q2 = v2 - L;
    // the nearest point's distance, shrunk: by 1e-4 of itself and by 1e-5 of the triangle's extent - on a sliver the closest-point
    // arithmetic places the point inexactly ALONG the triangle, which moves its distance in the second order only; a key may be
    // too small (the triangle is then tested by a few segments more) but never larger than the true distance
    const F3 ea = v1 - v0, eb = v2 - v0;
    const float extent = sqrtf(dot3(ea, ea)) + sqrtf(dot3(eb, eb));
    const float r_min = fmaxf(sqrtf(fmaxf(sg_origin_tri_dist2(q0, q1, q2), 0.0f)) * (1.0f - 1.0e-4f) - 1.0e-5f * extent, 0.0f);
    const float cone_cells = r_min > 0.0f ? 3.5f * (p.eps_eff / r_min) * p.scale : 3.0e38f;
    if (!(cone_cells <= SG_MAX_CONE_CELLS)) { // too close to the light for a bounded dilation: every segment of the light tests it
        if (lane == 0) sg_emit<FILL>(p, p.n_cells, tri, 0.0f);
        return;
    }
    const float mc = p.margin_cells + cone_cells;
    const float kk = 1.0f + (mc + 1.0f) / p.scale; // the face's frustum widened by the margin (and a cell for the clip's own rounding)
    for (uint32_t face = 0; face < 6u; face++) {
        const int a = (int)(face >> 1);
        const float s = (face & 1u) ? -1.0f : 1.0f;
        const int b = a == 2 ? 0 : a + 1, c = b == 2 ? 0 : b + 1;
        F3 poly[8], tmp[8];
        poly[0] = f3(comp(q0, b), comp(q0, c), s * comp(q0, a));
        poly[1] = f3(comp(q1, b), comp(q1, c), s * comp(q1, a));
        poly[2] = f3(comp(q2, b), comp(q2, c), s * comp(q2, a));
        if (poly[0].z <= 0.0f && poly[1].z <= 0.0f && poly[2].z <= 0.0f) continue; // behind this face
        int n = sg_clip(poly, 3, tmp, 1.0f, 0.0f, kk);
        if (n < 3) continue;
        n = sg_clip(tmp, n, poly, -1.0f, 0.0f, kk);
        if (n < 3) continue;
        n = sg_clip(poly, n, tmp, 0.0f, 1.0f, kk);
        if (n < 3) continue;
        n = sg_clip(tmp, n, poly, 0.0f, -1.0f, kk);
        if (n < 3) continue;
        float px[8], py[8];
        bool ok = true;
        for (int k = 0; k < n; k++) {
            const float z = poly[k].z;
            if (!(z > 0.0f)) ok = false; // cannot happen for r_min > 0 (inside the four planes z >= 0, zero only at the apex)
            px[k] = (poly[k].x / z + 1.0f) * p.scale;
            py[k] = (poly[k].y / z + 1.0f) * p.scale;
        }
        if (!ok) { // keep the superset property whatever the reason: the whole face
            const float qx[4] = {0.0f, (float)p.res, (float)p.res, 0.0f}, qy[4] = {0.0f, 0.0f, (float)p.res, (float)p.res};
            sg_raster<FILL>(p, face * p.res * p.res, qx, qy, 4, mc, tri, r_min, lane);
            continue;
        }
        sg_raster<FILL>(p, face * p.res * p.res, px, py, n, mc, tri, r_min, lane);
    }
}

// Over the cell counts (the near list, the last one, left out of the cell statistics): [0] all entries, summed in 64 bits (the 32-bit
// scan of the counts could wrap unnoticed), [1] cells that are not empty, [2] cells with more than `heavy` entries
__global__ __launch_bounds__(256) void k_sg_sum(const uint32_t* __restrict__ count, uint32_t n, uint32_t heavy, unsigned long long* out) {
    unsigned long long s = 0;
    uint32_t filled = 0, over = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t c = count[i];
        s += c;
        if (i + 1 < n) filled += c != 0u, over += c > heavy;
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        filled += __shfl_down(filled, off, 64);
        over += __shfl_down(over, off, 64);
    }
    if ((threadIdx.x & 63u) == 0 && s) {
        atomicAdd(&out[0], s);
        atomicAdd(&out[1], (unsigned long long)filled);
        atomicAdd(&out[2], (unsigned long long)over);
    }
}

// entries of each list that do not fit its block (the near list, index n_cells, has no block)
__global__ __launch_bounds__(256) void k_sg_overflow(const uint32_t* __restrict__ count, uint32_t* __restrict__ ovf, uint32_t n_cells) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_cells) return;
    const uint32_t n = count[c];
    ovf[c] = c < n_cells ? (n > RT_SG_BLOCK_ENTRIES ? n - RT_SG_BLOCK_ENTRIES : 0u) : n;
}
// ovf_start[n + 1] = ovf_start[n] + ovf[n] (the scan is exclusive); one thread
__global__ void k_sg_total(uint32_t* ovf_start, const uint32_t* ovf, uint32_t n) { ovf_start[n + 1] = ovf_start[n] + ovf[n]; }

// Orders every list of at most `heavy` entries (its nearest RT_SG_SORTED_PREFIX entries, by (key, record)) and writes the cells' header quads; statistics: [0] longest list
__global__ __launch_bounds__(256) void k_sg_sort(const uint32_t* __restrict__ count, const uint32_t* __restrict__ ovf_start, uint4* __restrict__ blocks,
                                                 uint4* __restrict__ overflow, uint32_t n_cells, uint32_t heavy, uint32_t* __restrict__ stats) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_cells) return;
    const uint32_t n = count[c], os = ovf_start[c];
    const uint32_t in_block = c < n_cells ? RT_SG_BLOCK_ENTRIES : 0u;
    uint4* __restrict__ blk = blocks + (size_t)c * RT_SG_BLOCK_QUADS; // (not touched for the near list)
    auto at = [&](uint32_t i) -> uint4* { return i < in_block ? blk + 1u + RT_SG_ENTRY_QUADS * i : overflow + (size_t)(os + (i - in_block)) * RT_SG_ENTRY_QUADS; };
    if (n) atomicMax(&stats[0], n);
    if (n > 1 && n <= heavy && c < n_cells) { // (the near list is tested whole, in any order)
        // selection of the RT_SG_SORTED_PREFIX smallest (key, record) in order: a segment's walk stops at the first key beyond its end
        // and gives up after RT_WF_GRID_WALK < RT_SG_SORTED_PREFIX entries, so the order of the later ones is never looked at (a full
        // insertion sort of the 48-byte entries took 8 ms per light, two thirds of the grids' build)
        const uint32_t m = n < RT_SG_SORTED_PREFIX ? n : RT_SG_SORTED_PREFIX;
        for (uint32_t i = 0; i + 1 < n && i < m; i++) {
            uint32_t best = i, bk = at(i)[0].x, br = at(i)[2].z;
            for (uint32_t j = i + 1; j < n; j++) {
                const uint32_t k = at(j)[0].x; // keys are non-negative floats: their bits order like the values
                if (k < bk || (k == bk && at(j)[2].z < br)) best = j, bk = k, br = at(j)[2].z;
            }
            if (best != i) {
                uint4 *a = at(i), *b = at(best);
                const uint4 a0 = a[0], a1 = a[1], a2 = a[2];
                a[0] = b[0], a[1] = b[1], a[2] = b[2];
                b[0] = a0, b[1] = a1, b[2] = a2;
            }
        }
    }
    if (c < n_cells) blk[0] = make_uint4(n, os, n > RT_SG_BLOCK_ENTRIES ? at(RT_SG_BLOCK_ENTRIES)[0].x : 0x7F800000u, 0u);
}

// What a walk can ever look at: a list's nearest RT_SG_SORTED_PREFIX entries when it has at most `heavy` (a walk gives up before the
// unordered rest), nothing of a longer one (its segments are handed to the BVH unseen); the near list whole.  Beyond the cell's block.
__global__ __launch_bounds__(256) void k_sg_keep(const uint32_t* __restrict__ count, uint32_t* __restrict__ keep, uint32_t n_cells, uint32_t heavy) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_cells) return;
    const uint32_t n = count[c];
    const uint32_t looked_at = n <= heavy ? (n < RT_SG_SORTED_PREFIX ? n : RT_SG_SORTED_PREFIX) : 0u;
    keep[c] = c < n_cells ? (looked_at > RT_SG_BLOCK_ENTRIES ? looked_at - RT_SG_BLOCK_ENTRIES : 0u) : n;
}
// ... moved from the array the lists were built in (every entry rasterised) to the one the frames read; the headers follow
__global__ __launch_bounds__(256) void k_sg_compact(const uint32_t* __restrict__ keep, const uint32_t* __restrict__ keep_start, const uint32_t* __restrict__ ovf_start,
                                                    const uint4* __restrict__ built, uint4* __restrict__ overflow, uint4* __restrict__ blocks, uint32_t n_cells) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_cells) return;
    const uint32_t n = keep[c], from = ovf_start[c], to = keep_start[c];
    for (uint32_t q = 0; q < n * RT_SG_ENTRY_QUADS; q++) overflow[(size_t)to * RT_SG_ENTRY_QUADS + q] = built[(size_t)from * RT_SG_ENTRY_QUADS + q];
    if (c < n_cells) reinterpret_cast<uint32_t*>(blocks + (size_t)c * RT_SG_BLOCK_QUADS)[1] = to;
}

#define SG_CHK(call)                  \
    do {                              \
        hipError_t e_ = (call);       \
        if (e_ != hipSuccess) {       \
            cleanup();                \
            return e_;                \
        }                             \
    } while (0)

uint32_t pow2_at_least(double v) {
    uint32_t r = 1;
    while ((double)r < v && r < (1u << 30)) r <<= 1;
    return r;
}

} // namespace

namespace rt {

hipError_t shadow_grid_build(const DevTri* d_tris, uint32_t n_records, const DevLight& light, const float lo[3], const float hi[3],
                             const ShadowGridOptions& opt, hipStream_t stream, ShadowGridBuild* out) {
    *out = ShadowGridBuild{};
    if (n_records == 0 || (n_records % RT_DEV_LEAF_STRIDE) != 0) return hipSuccess;
    for (int a = 0; a < 3; a++)
        if (!std::isfinite(lo[a]) || !std::isfinite(hi[a]) || !(lo[a] <= hi[a])) return hipSuccess;
    SgParams p{};
    p.tris = d_tris;
    p.n_records = n_records;
    double c_max = 0.0;
    for (int a = 0; a < 3; a++) c_max = std::max(c_max, std::max(std::fabs((double)lo[a]), std::fabs((double)hi[a])));
    const double n_real = (double)n_records / RT_DEV_LEAF_STRIDE; // at least this many triangles
    float limit_margin = 0.0f;
    if (light.light_type == 0) {
        double w[3] = {light.neg_ndir[0], light.neg_ndir[1], light.neg_ndir[2]};
        const double wl = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        if (!std::isfinite(wl) || !(std::fabs(wl - 1.0) < 1e-3)) return hipSuccess; // a zero or non-finite direction: no grid
        int least = 0;
        for (int a = 1; a < 3; a++)
            if (std::fabs(w[a]) < std::fabs(w[least])) least = a;
        double ax[3] = {0, 0, 0};
        ax[least] = 1.0;
        double u[3] = {w[1] * ax[2] - w[2] * ax[1], w[2] * ax[0] - w[0] * ax[2], w[0] * ax[1] - w[1] * ax[0]};
        const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        for (int a = 0; a < 3; a++) u[a] /= ul;
        double v[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
        double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300, smax = -1e300;
        for (int k = 0; k < 8; k++) {
            const double x[3] = {(k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2]};
            const double pu = x[0] * u[0] + x[1] * u[1] + x[2] * u[2], pv = x[0] * v[0] + x[1] * v[1] + x[2] * v[2];
            const double ps = x[0] * w[0] + x[1] * w[1] + x[2] * w[2];
            umin = std::min(umin, pu), umax = std::max(umax, pu), vmin = std::min(vmin, pv), vmax = std::max(vmax, pv), smax = std::max(smax, ps);
        }
        const uint32_t res = std::min(std::max(pow2_at_least(4.0 * std::sqrt(n_real)), 32u), std::max(32u, opt.res_dir));
        const double extent = std::max(std::max(umax - umin, vmax - vmin), 1e-6) * (1.0 + 4.0 / res) + 1e-5 * c_max + 1e-30;
        const double scale = res / extent;
        const double margin = SG_MARGIN_CELLS + 1.6e-6 * c_max * scale * 4.0; // the two sides' dot products round at ~4e-7 |coordinates| each
        if (!std::isfinite(scale) || !(margin < 3.0)) return hipSuccess;
        p.kind = RT_SG_KIND_ORTHO;
        p.res = res;
        p.n_cells = res * res;
        p.scale = (float)scale;
        p.u0 = (float)(0.5 * (umin + umax) - 0.5 * extent);
        p.v0 = (float)(0.5 * (vmin + vmax) - 0.5 * extent);
        p.key_top = (float)(smax + 1.0 + 1e-3 * c_max);
        p.margin_cells = (float)margin;
        for (int a = 0; a < 3; a++) p.au[a] = (float)u[a], p.av[a] = (float)v[a], p.aw[a] = light.neg_ndir[a];
        limit_margin = (float)(1.0e-4 + 4.0e-6 * c_max);
    } else {
        double d_max = 0.0;
        for (int a = 0; a < 3; a++) {
            if (!std::isfinite(light.position[a])) return hipSuccess;
            c_max = std::max(c_max, std::fabs((double)light.position[a]));
            const double far_side = std::max(std::fabs((double)light.position[a] - lo[a]), std::fabs((double)light.position[a] - hi[a]));
            d_max += far_side * far_side;
        }
        d_max = std::sqrt(d_max);
        const double eps_eff = SG_EXT_EPS * 1.01 + 2.0e-6 * (d_max + c_max);
        if (!std::isfinite(eps_eff) || !(eps_eff < 0.01 * (d_max + 1e-30) + SG_EXT_EPS * 2.0)) return hipSuccess; // coordinates too large for the offset to mean anything
        const uint32_t res = std::min(std::max(pow2_at_least(2.0 * std::sqrt(n_real)), 16u), std::max(16u, opt.res_point));
        p.kind = RT_SG_KIND_CUBE;
        p.res = res;
        p.n_cells = 6u * res * res;
        p.scale = 0.5f * (float)res;
        p.eps_eff = (float)eps_eff;
        p.margin_cells = SG_MARGIN_CELLS;
        for (int a = 0; a < 3; a++) p.L[a] = light.position[a];
        limit_margin = (float)(2.0 * eps_eff + 1.0e-5 * d_max);
    }

    uint32_t *count = nullptr, *ovf = nullptr, *ovf_start = nullptr, *keep = nullptr, *keep_start = nullptr, *stats = nullptr;
    unsigned long long* total64 = nullptr;
    uint4 *blocks = nullptr, *overflow = nullptr, *built = nullptr;
    void* temp = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(count);
        (void)hipFree(ovf);
        (void)hipFree(ovf_start);
        (void)hipFree(keep);
        (void)hipFree(keep_start);
        (void)hipFree(stats);
        (void)hipFree(total64);
        (void)hipFree(blocks);
        (void)hipFree(overflow);
        (void)hipFree(built);
        (void)hipFree(temp);
    };
    const size_t lists = (size_t)p.n_cells + 1; // + the near list
    SG_CHK(hipMalloc((void**)&count, lists * 4));
    SG_CHK(hipMalloc((void**)&ovf, lists * 4));
    SG_CHK(hipMalloc((void**)&ovf_start, (lists + 1) * 4));
    SG_CHK(hipMalloc((void**)&keep, lists * 4));
    SG_CHK(hipMalloc((void**)&keep_start, (lists + 1) * 4));
    SG_CHK(hipMalloc((void**)&stats, 2 * 4));
    SG_CHK(hipMalloc((void**)&total64, 3 * 8));
    SG_CHK(hipMemsetAsync(total64, 0, 3 * 8, stream));
    SG_CHK(hipMemsetAsync(count, 0, lists * 4, stream));
    SG_CHK(hipMemsetAsync(stats, 0, 2 * 4, stream));
    p.cell_count = count;
    const dim3 grid((n_records + 3u) / 4u), block(256); // a wave per record
    hipLaunchKernelGGL(k_sg_raster<false>, grid, block, 0, stream, p);
    SG_CHK(hipGetLastError());
    hipLaunchKernelGGL(k_sg_sum, dim3(1024), dim3(256), 0, stream, count, (uint32_t)lists, opt.heavy, total64);
    hipLaunchKernelGGL(k_sg_overflow, dim3((uint32_t)((lists + 255) / 256)), dim3(256), 0, stream, count, ovf, p.n_cells);
    size_t temp_bytes = 0;
    SG_CHK(rocprim::exclusive_scan(nullptr, temp_bytes, ovf, ovf_start, 0u, lists, rocprim::plus<uint32_t>(), stream));
    SG_CHK(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    SG_CHK(rocprim::exclusive_scan(temp, temp_bytes, ovf, ovf_start, 0u, lists, rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(k_sg_total, dim3(1), dim3(1), 0, stream, ovf_start, ovf, (uint32_t)(lists - 1));
    hipLaunchKernelGGL(k_sg_keep, dim3((uint32_t)((lists + 255) / 256)), dim3(256), 0, stream, count, keep, p.n_cells, opt.heavy);
    SG_CHK(rocprim::exclusive_scan(temp, temp_bytes, keep, keep_start, 0u, lists, rocprim::plus<uint32_t>(), stream)); // (same size and types as the scan above)
    hipLaunchKernelGGL(k_sg_total, dim3(1), dim3(1), 0, stream, keep_start, keep, (uint32_t)(lists - 1));
    uint32_t tail[2] = {0, 0}, built_tail[2] = {0, 0}; // [start of the near list, all overflow entries]: kept for the frames / as built
    unsigned long long sums[3] = {0, 0, 0};
    SG_CHK(hipMemcpyAsync(tail, keep_start + lists - 1, 8, hipMemcpyDeviceToHost, stream));
    SG_CHK(hipMemcpyAsync(built_tail, ovf_start + lists - 1, 8, hipMemcpyDeviceToHost, stream));
    SG_CHK(hipMemcpyAsync(sums, total64, 3 * 8, hipMemcpyDeviceToHost, stream));
    SG_CHK(hipStreamSynchronize(stream));
    const unsigned long long total = sums[0];
    out->n_entries = total;
    out->filled_cells = (uint32_t)sums[1];
    out->heavy_cells = (uint32_t)sums[2];
    const uint32_t near_count = tail[1] - tail[0]; // (kept whole)
    // A grid pays when its lists are short: 2.3 entries read per segment on the sponza-like scene (4 - 7 triangles per filled cell, one
    // cell in a thousand over 64 entries) against 11 node visits of the BVH.  Foliage-like clutter (bistro-like: 14 per filled cell, 4 % of
    // the cells over 64) lost against the BVH while a long walk held its whole wave up (rounds 2-3); with the walks parked
    // (k_wf_shadow_grid) and the lists cut to what a walk looks at, that scene's lights answer 81 % of their segments from the lists,
    // 6 entries each, and the frame is 12 % faster.  Refused now: lists beyond that kind of clutter.
    const bool long_lists = (double)total > opt.max_mean_list * (double)std::max<uint64_t>(1, sums[1]) || (double)sums[2] > opt.max_heavy_share * (double)sums[1];
    // (total bounds the overflow entries, so the 32-bit scan above did not wrap when it is accepted here)
    if (total == 0 || total > std::min<uint64_t>(opt.max_entries, 0xFFFFFFFFull / RT_SG_ENTRY_QUADS) || near_count > std::min(opt.heavy, 64u) || long_lists) {
        cleanup();
        return hipSuccess; // no grid for this light
    }
    const size_t block_bytes = (size_t)p.n_cells * RT_SG_BLOCK_QUADS * sizeof(uint4), ovf_bytes = ((size_t)tail[1] + 1) * RT_SG_ENTRY_QUADS * sizeof(uint4);
    if ((uint64_t)block_bytes + ovf_bytes > opt.max_bytes) { // the caller's memory budget: no grid, the BVH answers for this light
        cleanup();
        return hipSuccess;
    }
    // The lists are built whole (every entry rasterised: the nearest ones of a list are only known once it is complete) in an array that
    // lives for the build; what the frames read is what a walk can look at.  On cluttered scenes that is a fraction - the bistro-like
    // scene's lights: 256 M entries rasterised, most of them deep in long lists.
    SG_CHK(hipMalloc((void**)&blocks, block_bytes));
    SG_CHK(hipMalloc((void**)&built, ((size_t)built_tail[1] + 1) * RT_SG_ENTRY_QUADS * sizeof(uint4)));
    SG_CHK(hipMemsetAsync(blocks, 0, block_bytes, stream)); // (an untouched cell is an empty list)
    SG_CHK(hipMemsetAsync(count, 0, lists * 4, stream));
    p.ovf_start = ovf_start;
    p.blocks = blocks;
    p.overflow = built;
    hipLaunchKernelGGL(k_sg_raster<true>, grid, block, 0, stream, p);
    SG_CHK(hipGetLastError());
    hipLaunchKernelGGL(k_sg_sort, dim3((uint32_t)((lists + 255) / 256)), dim3(256), 0, stream, count, ovf_start, blocks, built, p.n_cells, opt.heavy, stats);
    SG_CHK(hipGetLastError());
    if (tail[1] == built_tail[1]) { // nothing to leave out
        overflow = built;
        built = nullptr;
    } else {
        SG_CHK(hipMalloc((void**)&overflow, ovf_bytes));
        hipLaunchKernelGGL(k_sg_compact, dim3((uint32_t)((lists + 255) / 256)), dim3(256), 0, stream, keep, keep_start, ovf_start, built, overflow, blocks, p.n_cells);
        SG_CHK(hipGetLastError());
    }
    uint32_t st[2] = {0, 0};
    SG_CHK(hipMemcpyAsync(st, stats, 8, hipMemcpyDeviceToHost, stream));
    SG_CHK(hipStreamSynchronize(stream));
    (void)hipFree(count);
    (void)hipFree(ovf);
    (void)hipFree(ovf_start);
    (void)hipFree(keep);
    (void)hipFree(keep_start);
    (void)hipFree(stats);
    (void)hipFree(total64);
    (void)hipFree(temp);
    (void)hipFree(built);

    DevShadowGrid& g = out->grid;
    g.kind = p.kind;
    g.res = p.res;
    g.n_cells = p.n_cells;
    g.heavy = opt.heavy;
    for (int a = 0; a < 3; a++) g.origin[a] = p.L[a], g.axis_u[a] = p.au[a], g.axis_v[a] = p.av[a], g.axis_w[a] = p.aw[a];
    g.scale = p.scale;
    g.u0 = p.u0;
    g.v0 = p.v0;
    g.key_top = p.key_top;
    g.limit_margin = limit_margin;
    g.near_begin = tail[0];
    g.near_end = tail[1];
    g.blocks = blocks;
    g.overflow = overflow;
    out->blocks = blocks;
    out->overflow = overflow;
    out->bytes = block_bytes + ovf_bytes;
    out->near_count = near_count;
    out->longest = st[0];
    return hipSuccess;
}

} // namespace rt
