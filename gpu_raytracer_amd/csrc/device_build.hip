// device_build.hip — the acceleration structure built ON the GPU (round 2).
//
// The reference rebuilds its BVH on the host whenever a scene is loaded (src/scene.rs:87-119 -> src/bvh.rs:104-122) and renders
// the next frame at once; round 1's host build (binned SAH + insertion-based optimisation) made rt_upload_scene cost more than
// a whole headline frame.  Here the build runs on the device in a few milliseconds:
//   1. per triangle: box, centroid, 63-bit Morton code of the centroid; rocPRIM radix sort (stable);
//   2. PLOC (Meister, Bittner 2018): clusters along the Morton curve; every round each cluster finds its nearest neighbour (area of
//      the union) within `radius` positions, mutual pairs merge; the new node's collapse costs (the dynamic program of
//      bvh_builder.cpp: cheapest cost of the subtree in 1..8 child slots) are computed when it is created, since its children are
//      final by then;
//   3. top-down, level by level: every queued wide node expands its children through the program's splits, assigns them to
//      octant slots and writes its DevNode8 and its leaves.  WHERE everything goes is known beforehand: next to the costs, the
//      program carries the number of wide nodes and leaves every subtree will produce, so each node computes the positions of
//      its children's blocks in the depth-first layout the host build produces by appending (a first version claimed ranges
//      with atomics: same tree, scattered siblings, 5 % slower frames).
// bvh_builder.cpp's method 1 (build_ploc + collapse8) is the host statement of the same algorithm: same Morton codes, same
// neighbour search and tie rule, same program, same layout - the arrays are byte-identical (tests/test_gpu_device_build.py).
// Images cannot depend on any of it: closest hits resolve ties by triangle index, any-hit is a boolean.  Measured (profiles/ab_r02.json): PLOC trees cost the headline frame +1 % against the
// host's binned SAH + reinsertion trees, with the same 11.6 node visits per segment.
#include "device_build.h"

#include <hip/hip_runtime.h>

#include <cstring> // rocPRIM's headers call memset on the host without including it

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

namespace rt {
namespace {

#define DB_INVALID 0xFFFFFFFFu
#define DB_W 8

struct DbDp { // collapse costs of one binary node (see collapse8 in bvh_builder.cpp)
    float c[DB_W];
    uint8_t split[DB_W];
    uint32_t count; // triangles below (saturating)
    uint32_t leaf;  // k = 1: the subtree is cheapest as a single leaf
    uint32_t wn[DB_W]; // [k-1]: wide nodes the subtree produces when it gets at most k child slots of its parent
    uint32_t ln[DB_W]; // [k-1]: leaves it produces
};
// effective number of slots the program uses when `k` are offered (expand of collapse8 walks down while the split is 0)
__device__ __forceinline__ int db_effective(const DbDp& d, int k) {
    while (k > 1 && d.split[k - 1] == 0) k--;
    return k;
}

struct DbArrays {
    const BuildTri* tris_in;
    float4* bmin; // [2n] node boxes (xyz)
    float4* bmax;
    uint2* child; // [2n] inner: (left, right); leaf: (DB_INVALID, input triangle)
    DbDp* dp;     // [2n]
    uint32_t n;   // input triangles
};

__device__ __forceinline__ float db_half_area(float3 mn, float3 mx) {
    const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
    if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ uint32_t db_ordered(float f) { // order-preserving float -> uint
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float db_unordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// ---- 1a. boxes, centroids, scene bounds of the centroids; bounds[0..2] min, [3..5] max (ordered uints), [6] finite triangles
__global__ __launch_bounds__(256) void k_db_bounds(const BuildTri* __restrict__ tris, uint32_t n, float4* __restrict__ pmin, float4* __restrict__ pmax,
                                                    uint32_t* __restrict__ bounds) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool finite = false;
    float3 c = make_float3(0.0f, 0.0f, 0.0f);
    if (i < n) {
        const BuildTri t = tris[i];
        finite = true;
        float mn[3], mx[3];
        for (int a = 0; a < 3; a++) {
            finite = finite && isfinite(t.v0[a]) && isfinite(t.v1[a]) && isfinite(t.v2[a]);
            mn[a] = fminf(fminf(t.v0[a], t.v1[a]), t.v2[a]);
            mx[a] = fmaxf(fmaxf(t.v0[a], t.v1[a]), t.v2[a]);
        }
        pmin[i] = make_float4(mn[0], mn[1], mn[2], finite ? 1.0f : 0.0f);
        pmax[i] = make_float4(mx[0], mx[1], mx[2], 0.0f);
        c = make_float3(0.5f * mn[0] + 0.5f * mx[0], 0.5f * mn[1] + 0.5f * mx[1], 0.5f * mn[2] + 0.5f * mx[2]); // Builder::cent
    }
    if (finite) {
        atomicMin(&bounds[0], db_ordered(c.x));
        atomicMin(&bounds[1], db_ordered(c.y));
        atomicMin(&bounds[2], db_ordered(c.z));
        atomicMax(&bounds[3], db_ordered(c.x));
        atomicMax(&bounds[4], db_ordered(c.y));
        atomicMax(&bounds[5], db_ordered(c.z));
    }
    const unsigned long long m = __ballot(finite);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&bounds[6], (uint32_t)__popcll(m));
}

__device__ __forceinline__ unsigned long long db_morton21(float x) { // bvh_builder.cpp morton21
    unsigned long long v = (unsigned long long)(uint32_t)fminf(fmaxf(x * 2097152.0f, 0.0f), 2097151.0f);
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// ---- 1b. Morton keys; triangles with a non-finite coordinate sort to the end (they are dropped: Möller–Trumbore can never accept them)
__global__ __launch_bounds__(256) void k_db_morton(uint32_t n, const float4* __restrict__ pmin, const float4* __restrict__ pmax, const uint32_t* __restrict__ bounds,
                                                    unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 mn = pmin[i], mx = pmax[i];
    unsigned long long key = ~0ull;
    if (mn.w != 0.0f) {
        const float lo[3] = {db_unordered(bounds[0]), db_unordered(bounds[1]), db_unordered(bounds[2])};
        const float hi[3] = {db_unordered(bounds[3]), db_unordered(bounds[4]), db_unordered(bounds[5])};
        const float c[3] = {0.5f * mn.x + 0.5f * mx.x, 0.5f * mn.y + 0.5f * mx.y, 0.5f * mn.z + 0.5f * mx.z};
        unsigned long long code = 0;
        for (int a = 0; a < 3; a++) {
            const float inv = hi[a] > lo[a] ? 1.0f / (hi[a] - lo[a]) : 0.0f;
            code |= db_morton21((c[a] - lo[a]) * inv) << (2 - a);
        }
        key = code;
    }
    keys[i] = key;
    vals[i] = i;
}

// ---- 2a. leaves: node i = the i-th finite triangle in Morton order
__global__ __launch_bounds__(256) void k_db_leaves(DbArrays A, uint32_t n_valid, const uint32_t* __restrict__ sorted, const float4* __restrict__ pmin,
                                                    const float4* __restrict__ pmax, uint32_t* __restrict__ cl, float cost_intersect) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_valid) return;
    const uint32_t prim = sorted[i];
    const float4 mn = pmin[prim], mx = pmax[prim];
    A.bmin[i] = mn;
    A.bmax[i] = mx;
    A.child[i] = make_uint2(DB_INVALID, prim);
    DbDp d;
    const float lc = cost_intersect * 1.0f * db_half_area(make_float3(mn.x, mn.y, mn.z), make_float3(mx.x, mx.y, mx.z));
    for (int k = 0; k < DB_W; k++) d.c[k] = lc, d.split[k] = 0, d.wn[k] = 0u, d.ln[k] = 1u;
    d.count = 1;
    d.leaf = 1;
    A.dp[i] = d;
    cl[i] = i;
}

// ---- 2b. nearest neighbour of every cluster within `radius` positions (first best: the lower position wins ties, as build_ploc)
#define DB_NN_BLOCK 256
#define DB_MAX_RADIUS 32
// Degenerate input (coincident triangles, chains) stalls the pairing at one mutual pair per round; a round that merges less than a
// sixteenth of the clusters escalates for the rest of the build (build_ploc has the same rule): level 1 prefers position i ^ 1
// among equal areas, level 2 pairs i and i ^ 1 outright.  Ordinary scenes never leave level 0.
__global__ __launch_bounds__(DB_NN_BLOCK) void k_db_nn(DbArrays A, const uint32_t* __restrict__ cl, uint32_t m, uint32_t radius, uint32_t level, uint32_t* __restrict__ nn) {
    __shared__ float s_mn[DB_NN_BLOCK + 2 * DB_MAX_RADIUS][3];
    __shared__ float s_mx[DB_NN_BLOCK + 2 * DB_MAX_RADIUS][3];
    const int base = (int)(blockIdx.x * DB_NN_BLOCK) - (int)radius;
    const int span = DB_NN_BLOCK + 2 * (int)radius;
    for (int k = threadIdx.x; k < span; k += DB_NN_BLOCK) {
        const int pos = base + k;
        if (pos >= 0 && pos < (int)m) {
            const uint32_t node = cl[pos];
            const float4 mn = A.bmin[node], mx = A.bmax[node];
            s_mn[k][0] = mn.x; s_mn[k][1] = mn.y; s_mn[k][2] = mn.z;
            s_mx[k][0] = mx.x; s_mx[k][1] = mx.y; s_mx[k][2] = mx.z;
        }
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * DB_NN_BLOCK + threadIdx.x;
    if (i >= m) return;
    const uint32_t buddy = i ^ 1u;
    if (level == 2u) {
        nn[i] = buddy < m ? buddy : i;
        return;
    }
    const int me = (int)threadIdx.x + (int)radius;
    const float3 imn = make_float3(s_mn[me][0], s_mn[me][1], s_mn[me][2]), imx = make_float3(s_mx[me][0], s_mx[me][1], s_mx[me][2]);
    auto union_area = [&](int j) {
        const int k = j - base;
        const float3 mn = make_float3(fminf(imn.x, s_mn[k][0]), fminf(imn.y, s_mn[k][1]), fminf(imn.z, s_mn[k][2]));
        const float3 mx = make_float3(fmaxf(imx.x, s_mx[k][0]), fmaxf(imx.y, s_mx[k][1]), fmaxf(imx.z, s_mx[k][2]));
        return db_half_area(mn, mx);
    };
    const int lo = i > radius ? (int)(i - radius) : 0, hi = (int)min(m - 1u, i + radius);
    float best = INFINITY;
    uint32_t bj = i;
    if (level == 1u && buddy < m) best = union_area((int)buddy), bj = buddy;
    for (int j = lo; j <= hi; j++) { // first best: the lower position wins ties
        if (j == (int)i) continue;
        const float a = union_area(j);
        if (a < best) best = a, bj = (uint32_t)j;
    }
    nn[i] = bj;
}

// role of cluster i this round: 0 stays, 1 merges with its partner (it is the lower of the pair and carries the new node), 2 disappears
__device__ __forceinline__ uint32_t db_role(const uint32_t* __restrict__ nn, uint32_t i) {
    const uint32_t j = nn[i];
    if (j == i || nn[j] != i) return 0u;
    return i < j ? 1u : 2u;
}

// ---- 2c. per block: clusters kept and pairs merged
__global__ __launch_bounds__(256) void k_db_count(const uint32_t* __restrict__ nn, uint32_t m, uint32_t* __restrict__ block_keep, uint32_t* __restrict__ block_merge) {
    __shared__ uint32_t s_keep, s_merge;
    if (threadIdx.x == 0) s_keep = s_merge = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t role = i < m ? db_role(nn, i) : 2u;
    const unsigned long long mk = __ballot(role != 2u), mm = __ballot(role == 1u);
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&s_keep, (uint32_t)__popcll(mk));
        atomicAdd(&s_merge, (uint32_t)__popcll(mm));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_keep[blockIdx.x] = s_keep;
        block_merge[blockIdx.x] = s_merge;
    }
}

// ---- 2d. exclusive scan of the block counts (one block); totals to out[0] (clusters after the round), out[1] (nodes created)
__global__ __launch_bounds__(1024) void k_db_scan(uint32_t* __restrict__ block_keep, uint32_t* __restrict__ block_merge, uint32_t n_blocks, uint32_t* __restrict__ out) {
    __shared__ uint32_t s_part[2][1024];
    const uint32_t per = (n_blocks + 1023u) / 1024u;
    const uint32_t b0 = threadIdx.x * per, b1 = min(n_blocks, b0 + per);
    uint32_t sk = 0, sm = 0;
    for (uint32_t b = b0; b < b1; b++) sk += block_keep[b], sm += block_merge[b];
    s_part[0][threadIdx.x] = sk;
    s_part[1][threadIdx.x] = sm;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ak = 0, am = 0;
        for (int t = 0; t < 1024; t++) {
            const uint32_t k = s_part[0][t], mm = s_part[1][t];
            s_part[0][t] = ak;
            s_part[1][t] = am;
            ak += k;
            am += mm;
        }
        out[0] = ak;
        out[1] = am;
    }
    __syncthreads();
    uint32_t ak = s_part[0][threadIdx.x], am = s_part[1][threadIdx.x];
    for (uint32_t b = b0; b < b1; b++) {
        const uint32_t k = block_keep[b], mm = block_merge[b];
        block_keep[b] = ak;
        block_merge[b] = am;
        ak += k;
        am += mm;
    }
}

// ---- 2e. compact the clusters (order along the curve is kept) and create the merged nodes with their collapse costs
__global__ __launch_bounds__(256) void k_db_merge(DbArrays A, const uint32_t* __restrict__ cl, const uint32_t* __restrict__ nn, uint32_t m, const uint32_t* __restrict__ block_keep,
                                                   const uint32_t* __restrict__ block_merge, uint32_t first_new_node, uint32_t* __restrict__ cl_out, float cost_traverse8,
                                                   float cost_intersect, uint32_t max_leaf) {
    __shared__ uint32_t s_wave[2][4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t role = i < m ? db_role(nn, i) : 2u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long mk = __ballot(role != 2u), mm = __ballot(role == 1u);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) {
        s_wave[0][wave] = (uint32_t)__popcll(mk);
        s_wave[1][wave] = (uint32_t)__popcll(mm);
    }
    __syncthreads();
    uint32_t off_k = block_keep[blockIdx.x], off_m = block_merge[blockIdx.x];
    for (uint32_t w = 0; w < wave; w++) off_k += s_wave[0][w], off_m += s_wave[1][w];
    off_k += (uint32_t)__popcll(mk & below);
    off_m += (uint32_t)__popcll(mm & below);
    if (i >= m || role == 2u) return;
    if (role == 0u) {
        cl_out[off_k] = cl[i];
        return;
    }
    const uint32_t p = first_new_node + off_m, L = cl[i], R = cl[nn[i]];
    cl_out[off_k] = p;
    const float4 lmn = A.bmin[L], lmx = A.bmax[L], rmn = A.bmin[R], rmx = A.bmax[R];
    const float3 mn = make_float3(fminf(lmn.x, rmn.x), fminf(lmn.y, rmn.y), fminf(lmn.z, rmn.z));
    const float3 mx = make_float3(fmaxf(lmx.x, rmx.x), fmaxf(lmx.y, rmx.y), fmaxf(lmx.z, rmx.z));
    A.bmin[p] = make_float4(mn.x, mn.y, mn.z, 1.0f);
    A.bmax[p] = make_float4(mx.x, mx.y, mx.z, 0.0f);
    A.child[p] = make_uint2(L, R);
    // the collapse program (collapse8): children are final
    const DbDp dl = A.dp[L], dr = A.dp[R];
    float dist[DB_W + 1];
    uint8_t arg[DB_W + 1];
    for (int k = 0; k <= DB_W; k++) dist[k] = INFINITY, arg[k] = 0;
    for (int k = 2; k <= DB_W; k++)
        for (int s = 1; s < k; s++) {
            const float v = dl.c[s - 1] + dr.c[k - s - 1];
            if (v < dist[k]) dist[k] = v, arg[k] = (uint8_t)s;
        }
    const float area = db_half_area(mn, mx);
    const float wide = cost_traverse8 * area + dist[DB_W];
    DbDp d;
    d.count = min(dl.count + dr.count, 1u << 20);
    const float leafc = d.count <= max_leaf ? cost_intersect * (float)d.count * area : INFINITY;
    // costs that are all inf / NaN (box areas beyond f32) leave arg at 0, which is not a split: same explicit rule as collapse8 (bvh_builder.cpp)
    if (arg[DB_W] == 0) arg[DB_W] = 1;
    d.leaf = (d.count <= max_leaf && !(wide < leafc)) ? 1u : 0u;
    d.c[0] = fminf(leafc, wide);
    d.split[0] = arg[DB_W];
    for (int k = 2; k <= DB_W; k++) {
        if (dist[k] < d.c[k - 2]) d.c[k - 1] = dist[k], d.split[k - 1] = arg[k];
        else d.c[k - 1] = d.c[k - 2], d.split[k - 1] = 0;
    }
    // what the subtree will produce: as one child (k = 1) a leaf, or a wide node whose 8 slots are split s0 : 8 - s0 ...
    {
        const int s0 = d.split[0];
        d.wn[0] = d.leaf ? 0u : 1u + dl.wn[s0 - 1] + dr.wn[DB_W - s0 - 1];
        d.ln[0] = d.leaf ? 1u : dl.ln[s0 - 1] + dr.ln[DB_W - s0 - 1];
    }
    for (int k = 2; k <= DB_W; k++) { // ... with k slots: through the split the program chose for the effective k
        const int kk = db_effective(d, k);
        if (kk > 1) {
            const int sl = d.split[kk - 1];
            d.wn[k - 1] = dl.wn[sl - 1] + dr.wn[kk - sl - 1];
            d.ln[k - 1] = dl.ln[sl - 1] + dr.ln[kk - sl - 1];
        } else {
            d.wn[k - 1] = d.wn[0];
            d.ln[k - 1] = d.ln[0];
        }
    }
    A.dp[p] = d;
}

// ---- 3. top-down emission, one launch per level
struct DbItem {
    uint32_t bin, dev, depth;
    uint32_t block; // where the node's inner children go (consecutive)
    uint32_t tri;   // where its leaves' triangle records go
};
struct DbOut {
    DevNode8* nodes;
    DevTri* tris;
    uint32_t* counters; // [2] items in the next queue, [3] depth, [4] leaves written, [5] error
    uint32_t cap_nodes, cap_tris, cap_queue;
};

__device__ __forceinline__ void db_put_tri(const BuildTri& t, DevTri* o) { // put_tri of bvh_builder.cpp
    for (int a = 0; a < 3; a++) {
        o->v0[a] = t.v0[a];
        o->e1[a] = t.v1[a] - t.v0[a];
        o->e2[a] = t.v2[a] - t.v0[a];
    }
    o->material_id = t.material_id;
    o->prim_id = t.prim_id;
    o->leaf_count = 0;
}

__global__ __launch_bounds__(64) void k_db_emit(DbArrays A, DbOut O, const DbItem* __restrict__ queue, uint32_t n_items, DbItem* __restrict__ next_queue) {
    const uint32_t q = blockIdx.x * 64u + threadIdx.x;
    if (q >= n_items) return;
    const DbItem it = queue[q];
    atomicMax(&O.counters[3], it.depth);
    // children of the wide node: the subtree of `bin` in at most 8 slots, through the program's splits (expand of collapse8)
    uint32_t ch[DB_W];
    int nch = 0;
    {
        uint32_t st_node[2 * DB_W];
        int st_k[2 * DB_W], sp = 0;
        const uint2 top = A.child[it.bin];
        const int s0 = A.dp[it.bin].split[0];
        st_node[sp] = top.y, st_k[sp++] = DB_W - s0;
        st_node[sp] = top.x, st_k[sp++] = s0;
        while (sp > 0) {
            const uint32_t mnode = st_node[--sp];
            int k = st_k[sp];
            const uint2 c = A.child[mnode];
            bool is_child = true;
            if (c.x != DB_INVALID) {
                const DbDp d = A.dp[mnode];
                while (k > 1 && d.split[k - 1] == 0) k--;
                if (k > 1) {
                    const int s = d.split[k - 1];
                    if (sp + 2 <= 2 * DB_W) {
                        st_node[sp] = c.y, st_k[sp++] = k - s;
                        st_node[sp] = c.x, st_k[sp++] = s;
                        is_child = false;
                    }
                }
            }
            if (is_child && nch < DB_W) ch[nch++] = mnode;
        }
    }
    // kinds, boxes, slots
    bool leafc[DB_W];
    float3 cmn[DB_W], cmx[DB_W];
    const float4 pmn4 = A.bmin[it.bin], pmx4 = A.bmax[it.bin];
    const float pmn[3] = {pmn4.x, pmn4.y, pmn4.z}, pmx[3] = {pmx4.x, pmx4.y, pmx4.z};
    float pc[3];
    for (int a = 0; a < 3; a++) pc[a] = 0.5f * pmn[a] + 0.5f * pmx[a];
    float score[DB_W][DB_W];
    for (int c = 0; c < nch; c++) {
        const uint2 cc = A.child[ch[c]];
        leafc[c] = cc.x == DB_INVALID || A.dp[ch[c]].leaf != 0u;
        const float4 mn = A.bmin[ch[c]], mx = A.bmax[ch[c]];
        cmn[c] = make_float3(mn.x, mn.y, mn.z);
        cmx[c] = make_float3(mx.x, mx.y, mx.z);
        const float off[3] = {(0.5f * mn.x + 0.5f * mx.x) - pc[0], (0.5f * mn.y + 0.5f * mx.y) - pc[1], (0.5f * mn.z + 0.5f * mx.z) - pc[2]};
        for (int sl = 0; sl < DB_W; sl++) score[c][sl] = (sl & 1 ? off[0] : -off[0]) + (sl & 2 ? off[1] : -off[1]) + (sl & 4 ? off[2] : -off[2]);
    }
    int slot_child[DB_W];
    bool done[DB_W];
    for (int sl = 0; sl < DB_W; sl++) slot_child[sl] = -1, done[sl] = false;
    for (int round = 0; round < nch; round++) { // greedy: the (child, slot) pair with the largest projection, as collapse8
        int bc = -1, bs = -1;
        float best = -INFINITY;
        for (int c = 0; c < nch; c++) {
            if (done[c]) continue;
            for (int sl = 0; sl < DB_W; sl++)
                if (slot_child[sl] < 0 && score[c][sl] > best) best = score[c][sl], bc = c, bs = sl;
        }
        if (bc < 0) {
            for (int c = 0; c < nch && bc < 0; c++)
                if (!done[c]) bc = c;
            for (int sl = 0; sl < DB_W && bs < 0; sl++)
                if (slot_child[sl] < 0) bs = sl;
        }
        slot_child[bs] = bc;
        done[bc] = true;
    }
    uint32_t imask = 0, lmask = 0;
    for (int sl = 0; sl < DB_W; sl++) {
        if (slot_child[sl] < 0) continue;
        if (leafc[slot_child[sl]]) lmask |= 1u << sl;
        else imask |= 1u << sl;
    }
    const uint32_t n_inner = (uint32_t)__popc(imask), n_leaf = (uint32_t)__popc(lmask);
    // the host build's layout (it appends while it walks depth first): this node's children block, then the blocks of its first
    // child's whole subtree, then the second child's ...; triangle records likewise (this node's leaves first)
    const uint32_t child_base = it.block, tri_base = it.tri;
    const uint32_t q_base = n_inner ? atomicAdd(&O.counters[2], n_inner) : 0u;
    if (n_leaf) atomicAdd(&O.counters[4], n_leaf);
    if (child_base + n_inner > O.cap_nodes || tri_base + RT_DEV_LEAF_STRIDE * n_leaf > O.cap_tris || q_base + n_inner > O.cap_queue || it.dev >= O.cap_nodes) {
        atomicOr(&O.counters[5], 1u); // the positions are derived from the program's own counts: cannot happen; never write out of bounds
        return;
    }
    // the node: per-axis power-of-two grid, planes rounded outward (quantise_axis / quantise_box of bvh_builder.cpp)
    DevNode8 d;
    uint32_t ex[3];
    for (int a = 0; a < 3; a++) {
        d.org[a] = pmn[a];
        const double extent = (double)pmx[a] - (double)pmn[a];
        int e = 1;
        if (extent > 0.0) {
            int fe;
            (void)frexp(extent / 255.0, &fe);
            e = fe + 127;
            if (e < 1) e = 1;
            if (e > 254) e = 254;
        }
        ex[a] = (uint32_t)(e - 127) & 0xFFu;
        const double scale = ldexp(1.0, e - 127);
        for (int h = 0; h < 2; h++) {
            uint32_t lo_word = 0, hi_word = 0;
            for (int i = 0; i < 4; i++) {
                const int sl = 4 * h + i;
                uint32_t qlo = 255, qhi = 0;
                if (slot_child[sl] >= 0) {
                    const float bmn = a == 0 ? cmn[slot_child[sl]].x : a == 1 ? cmn[slot_child[sl]].y : cmn[slot_child[sl]].z;
                    const float bmx = a == 0 ? cmx[slot_child[sl]].x : a == 1 ? cmx[slot_child[sl]].y : cmx[slot_child[sl]].z;
                    double lo = floor(((double)bmn - (double)d.org[a]) / scale);
                    double hi = ceil(((double)bmx - (double)d.org[a]) / scale);
                    lo = fmin(fmax(lo, 0.0), 255.0);
                    hi = fmin(fmax(hi, 0.0), 255.0);
                    while (lo > 0.0 && (double)d.org[a] + lo * scale > (double)bmn) lo -= 1.0;
                    while (hi < 255.0 && (double)d.org[a] + hi * scale < (double)bmx) hi += 1.0;
                    qlo = (uint32_t)lo;
                    qhi = (uint32_t)hi;
                }
                lo_word |= qlo << (8 * i);
                hi_word |= qhi << (8 * i);
            }
            d.qlo[a][h] = lo_word;
            d.qhi[a][h] = hi_word;
        }
    }
    d.ex_imask = ex[0] | (ex[1] << 8) | (ex[2] << 16) | (imask << 24);
    d.child_base = child_base;
    d.tri_base = tri_base;
    d.lmask = lmask;
    d._pad = 0;
    O.nodes[it.dev] = d;
    // children: inner ones queue up, leaves gather their triangles in index order
    uint32_t ri = 0, rl = 0;
    uint32_t next_block = child_base + n_inner, next_tri = tri_base + RT_DEV_LEAF_STRIDE * n_leaf;
    for (int sl = 0; sl < DB_W; sl++) {
        if (slot_child[sl] < 0) continue;
        const uint32_t cn = ch[slot_child[sl]];
        if (imask & (1u << sl)) {
            DbItem ni;
            ni.bin = cn;
            ni.dev = child_base + ri;
            ni.depth = it.depth + 1u;
            ni.block = next_block;
            ni.tri = next_tri;
            const DbDp cd = A.dp[cn];
            next_block += cd.wn[0] - 1u; // its descendants (the child itself sits in this node's block)
            next_tri += RT_DEV_LEAF_STRIDE * cd.ln[0];
            next_queue[q_base + ri] = ni;
            ri++;
        } else {
            uint32_t gathered[RT_DEV_LEAF_STRIDE], ng = 0, todo[2 * RT_DEV_LEAF_STRIDE + 2];
            int nt = 0;
            todo[nt++] = cn;
            while (nt > 0) {
                const uint2 u = A.child[todo[--nt]];
                if (u.x == DB_INVALID) {
                    if (ng < RT_DEV_LEAF_STRIDE) gathered[ng++] = u.y;
                } else if (nt + 2 <= (int)(2 * RT_DEV_LEAF_STRIDE + 2)) {
                    todo[nt++] = u.x;
                    todo[nt++] = u.y;
                }
            }
            for (uint32_t x = 1; x < ng; x++) // insertion sort by the caller's triangle index (the tie rule does not need it; the host build does the same)
                for (uint32_t y = x; y > 0 && A.tris_in[gathered[y]].prim_id < A.tris_in[gathered[y - 1]].prim_id; y--) {
                    const uint32_t tmp = gathered[y];
                    gathered[y] = gathered[y - 1];
                    gathered[y - 1] = tmp;
                }
            const uint32_t first = tri_base + RT_DEV_LEAF_STRIDE * rl;
            for (uint32_t x = 0; x < RT_DEV_LEAF_STRIDE; x++) {
                DevTri t;
                if (x < ng) db_put_tri(A.tris_in[gathered[x]], &t);
                else {
                    for (int a = 0; a < 3; a++) t.v0[a] = t.e1[a] = t.e2[a] = 0.0f;
                    t.material_id = t.prim_id = t.leaf_count = 0;
                }
                if (x == 0) t.leaf_count = ng;
                O.tris[first + x] = t;
            }
            rl++;
        }
    }
}

#define DB_CHK(call)                                      \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) {                           \
            err = e_;                                     \
            goto done;                                    \
        }                                                 \
    } while (0)

} // namespace

hipError_t device_build(const BuildTri* h_tris, size_t n_in, const BvhBuildOptions& opt, hipStream_t stream, DeviceBuild* out) {
    *out = DeviceBuild();
    if (n_in == 0) return hipSuccess;
    const uint32_t n = (uint32_t)n_in;
    const uint32_t radius = std::min<uint32_t>(std::max<uint32_t>(opt.ploc_radius, 1u), DB_MAX_RADIUS);
    const uint32_t max_leaf = std::min<uint32_t>(std::max<uint32_t>(opt.max_leaf, 1u), RT_DEV_LEAF_STRIDE);
    hipError_t err = hipSuccess;
    std::vector<void*> temps;
    auto alloc = [&](void** p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes ? bytes : 16);
        if (e == hipSuccess) temps.push_back(*p);
        return e;
    };
    BuildTri* d_tris = nullptr;
    float4 *pmin = nullptr, *pmax = nullptr;
    uint32_t *bounds = nullptr, *vals = nullptr, *vals2 = nullptr, *cl_a = nullptr, *cl_b = nullptr, *nn = nullptr, *block_keep = nullptr, *block_merge = nullptr, *totals = nullptr;
    unsigned long long *keys = nullptr, *keys2 = nullptr;
    void* sort_temp = nullptr;
    size_t sort_bytes = 0;
    DbArrays A{};
    DbOut O{};
    DbItem *q_a = nullptr, *q_b = nullptr;
    DevNode8* nodes_big = nullptr;
    DevTri* tris_big = nullptr;
    uint32_t h_bounds[8] = {0}, n_valid = 0, h_tot[2] = {0, 0}, h_cnt[8] = {0};
    const uint32_t blocks_n = (n + 255u) / 256u;

    DB_CHK(alloc((void**)&d_tris, (size_t)n * sizeof(BuildTri)));
    DB_CHK(hipMemcpyAsync(d_tris, h_tris, (size_t)n * sizeof(BuildTri), hipMemcpyHostToDevice, stream));
    DB_CHK(alloc((void**)&pmin, (size_t)n * 16));
    DB_CHK(alloc((void**)&pmax, (size_t)n * 16));
    DB_CHK(alloc((void**)&bounds, 8 * 4));
    DB_CHK(alloc((void**)&keys, (size_t)n * 8));
    DB_CHK(alloc((void**)&keys2, (size_t)n * 8));
    DB_CHK(alloc((void**)&vals, (size_t)n * 4));
    DB_CHK(alloc((void**)&vals2, (size_t)n * 4));
    {
        const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
        DB_CHK(hipMemcpyAsync(bounds, init, sizeof init, hipMemcpyHostToDevice, stream));
    }
    hipLaunchKernelGGL(k_db_bounds, dim3(blocks_n), dim3(256), 0, stream, d_tris, n, pmin, pmax, bounds);
    hipLaunchKernelGGL(k_db_morton, dim3(blocks_n), dim3(256), 0, stream, n, pmin, pmax, bounds, keys, vals);
    DB_CHK(hipGetLastError());
    DB_CHK(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, keys2, vals, vals2, (size_t)n, 0, 64, stream));
    DB_CHK(alloc(&sort_temp, sort_bytes));
    DB_CHK(rocprim::radix_sort_pairs(sort_temp, sort_bytes, keys, keys2, vals, vals2, (size_t)n, 0, 64, stream));
    DB_CHK(hipMemcpyAsync(h_bounds, bounds, sizeof h_bounds, hipMemcpyDeviceToHost, stream));
    DB_CHK(hipStreamSynchronize(stream));
    n_valid = h_bounds[6];
    if (n_valid == 0) goto done;

    A.tris_in = d_tris;
    A.n = n;
    DB_CHK(alloc((void**)&A.bmin, (size_t)2 * n_valid * 16));
    DB_CHK(alloc((void**)&A.bmax, (size_t)2 * n_valid * 16));
    DB_CHK(alloc((void**)&A.child, (size_t)2 * n_valid * 8));
    DB_CHK(alloc((void**)&A.dp, (size_t)2 * n_valid * sizeof(DbDp)));
    DB_CHK(alloc((void**)&cl_a, (size_t)n_valid * 4));
    DB_CHK(alloc((void**)&cl_b, (size_t)n_valid * 4));
    DB_CHK(alloc((void**)&nn, (size_t)n_valid * 4));
    {
        const uint32_t nb = (n_valid + 255u) / 256u;
        DB_CHK(alloc((void**)&block_keep, (size_t)nb * 4));
        DB_CHK(alloc((void**)&block_merge, (size_t)nb * 4));
        DB_CHK(alloc((void**)&totals, 2 * 4));
        hipLaunchKernelGGL(k_db_leaves, dim3(nb), dim3(256), 0, stream, A, n_valid, vals2, pmin, pmax, cl_a, opt.cost_intersect);
        DB_CHK(hipGetLastError());
    }
    {
        uint32_t m = n_valid, next_node = n_valid, level = 0;
        uint32_t* cl = cl_a;
        uint32_t* cl_next = cl_b;
        for (int round = 0; m > 1; round++) {
            if (round > 4096) { // PLOC merges at least one pair per round; this is a bug guard, not a limit
                err = hipErrorUnknown;
                goto done;
            }
            const uint32_t nb = (m + 255u) / 256u;
            hipLaunchKernelGGL(k_db_nn, dim3(nb), dim3(DB_NN_BLOCK), 0, stream, A, (const uint32_t*)cl, m, radius, level, nn);
            hipLaunchKernelGGL(k_db_count, dim3(nb), dim3(256), 0, stream, (const uint32_t*)nn, m, block_keep, block_merge);
            hipLaunchKernelGGL(k_db_scan, dim3(1), dim3(1024), 0, stream, block_keep, block_merge, nb, totals);
            hipLaunchKernelGGL(k_db_merge, dim3(nb), dim3(256), 0, stream, A, (const uint32_t*)cl, (const uint32_t*)nn, m, (const uint32_t*)block_keep, (const uint32_t*)block_merge,
                               next_node, cl_next, opt.cost_traverse8, opt.cost_intersect, max_leaf);
            DB_CHK(hipGetLastError());
            DB_CHK(hipMemcpyAsync(h_tot, totals, sizeof h_tot, hipMemcpyDeviceToHost, stream));
            DB_CHK(hipStreamSynchronize(stream));
            if (h_tot[0] >= m || h_tot[1] == 0 || next_node + h_tot[1] > 2u * n_valid) { // no progress / overflow: cannot happen
                err = hipErrorUnknown;
                goto done;
            }
            if (h_tot[1] < m / 16u && level < 2u) level++; // the round merged little: escalate for the rest of the build
            m = h_tot[0];
            next_node += h_tot[1];
            std::swap(cl, cl_next);
        }
        // root = the last cluster; a scene whose root is itself cheapest as ONE leaf is left to the host builder by the caller
        uint32_t root = 0;
        DB_CHK(hipMemcpyAsync(&root, cl, 4, hipMemcpyDeviceToHost, stream));
        DB_CHK(hipStreamSynchronize(stream));
        // ---- emission: the program knows how many wide nodes and leaves the root's subtree produces
        DbDp h_root;
        DB_CHK(hipMemcpyAsync(&h_root, A.dp + root, sizeof h_root, hipMemcpyDeviceToHost, stream));
        DB_CHK(hipStreamSynchronize(stream));
        if (h_root.leaf || h_root.wn[0] == 0 || h_root.wn[0] > n_valid || h_root.ln[0] > n_valid) { // (a one-leaf scene is the host builder's business)
            err = hipErrorUnknown;
            goto done;
        }
        O.cap_nodes = h_root.wn[0];
        O.cap_tris = RT_DEV_LEAF_STRIDE * h_root.ln[0];
        O.cap_queue = h_root.wn[0];
        DB_CHK(hipMalloc((void**)&nodes_big, (size_t)O.cap_nodes * sizeof(DevNode8)));
        DB_CHK(hipMalloc((void**)&tris_big, (size_t)O.cap_tris * sizeof(DevTri)));
        O.nodes = nodes_big;
        O.tris = tris_big;
        DB_CHK(alloc((void**)&O.counters, 8 * 4));
        DB_CHK(alloc((void**)&q_a, (size_t)O.cap_queue * sizeof(DbItem)));
        DB_CHK(alloc((void**)&q_b, (size_t)O.cap_queue * sizeof(DbItem)));
        {
            const uint32_t init[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            DB_CHK(hipMemcpyAsync(O.counters, init, sizeof init, hipMemcpyHostToDevice, stream));
            const DbItem first = {root, 0u, 1u, 1u, 0u}; // node 0 = the root, its children from position 1, its leaves from record 0
            DB_CHK(hipMemcpyAsync(q_a, &first, sizeof first, hipMemcpyHostToDevice, stream));
        }
        uint32_t n_items = 1;
        DbItem *q = q_a, *qn = q_b;
        for (int level = 0; n_items > 0; level++) {
            if (level > 4 * RT_DEV_MAX_BVH_DEPTH + 64) {
                err = hipErrorUnknown;
                goto done;
            }
            hipLaunchKernelGGL(k_db_emit, dim3((n_items + 63u) / 64u), dim3(64), 0, stream, A, O, (const DbItem*)q, n_items, qn);
            DB_CHK(hipGetLastError());
            DB_CHK(hipMemcpyAsync(h_cnt, O.counters, sizeof h_cnt, hipMemcpyDeviceToHost, stream));
            DB_CHK(hipStreamSynchronize(stream));
            if (h_cnt[5] != 0 || h_cnt[2] > O.cap_queue) {
                err = hipErrorUnknown;
                goto done;
            }
            n_items = h_cnt[2];
            const uint32_t zero = 0;
            DB_CHK(hipMemcpyAsync(O.counters + 2, &zero, 4, hipMemcpyHostToDevice, stream));
            std::swap(q, qn);
        }
        if (h_cnt[4] != h_root.ln[0]) { // every leaf the program announced was written
            err = hipErrorUnknown;
            goto done;
        }
        out->n_nodes = O.cap_nodes;
        out->n_tris = O.cap_tris;
        out->depth = h_cnt[3];
        out->n_leaves = h_cnt[4];
        out->nodes = nodes_big; // exactly sized: they become the scene's arrays
        out->tris = tris_big;
        nodes_big = nullptr;
        tris_big = nullptr;
    }
done:
    (void)hipStreamSynchronize(stream);
    for (void* p : temps) (void)hipFree(p);
    (void)hipFree(nodes_big);
    (void)hipFree(tris_big);
    if (err != hipSuccess) {
        (void)hipFree(out->nodes);
        (void)hipFree(out->tris);
        *out = DeviceBuild();
    }
    return err;
}

} // namespace rt
