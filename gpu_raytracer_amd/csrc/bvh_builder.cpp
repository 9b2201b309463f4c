// bvh_builder.cpp — binned-SAH binary build (host, multi-threaded) collapsed into the 8-wide quantised layout, see bvh_builder.h.
#include "bvh_builder.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <limits>
#include <numeric>
#include <thread>
#include <utility>

namespace rt {
namespace {

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int a = 0; a < 3; a++) {
            mn[a] = std::numeric_limits<float>::infinity();
            mx[a] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min(mn[a], b.mn[a]);
            mx[a] = std::max(mx[a], b.mx[a]);
        }
    }
    void grow(const float p[3]) {
        for (int a = 0; a < 3; a++) {
            mn[a] = std::min(mn[a], p[a]);
            mx[a] = std::max(mx[a], p[a]);
        }
    }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct TmpNode {
    Box box;
    uint32_t left, right; // inner: TmpNode indices; leaf: left = 0xFFFFFFFF
    uint32_t start, count;
};

#ifndef RT_BVH_REINSERT
#define RT_BVH_REINSERT 1
#endif
#ifndef RT_BVH_REINSERT_PASSES
#define RT_BVH_REINSERT_PASSES 2
#endif
#ifndef RT_BVH_REINSERT_MAX
#define RT_BVH_REINSERT_MAX 150000
#endif
#ifndef RT_BVH_REINSERT_FRACTION
#define RT_BVH_REINSERT_FRACTION 0.25
#endif
#ifndef RT_BVH_COLLAPSE_DP
#define RT_BVH_COLLAPSE_DP 1 /* measured: 0.8 % (sponza-like) to 1.5 % (bistro-like) over the greedy rule */
#endif
#ifndef RT_BVH_BINS
#define RT_BVH_BINS 32 /* SAH bins per axis: 32 instead of 16 gives 3 % fewer node visits on the sponza-like scene (+2 % throughput), 48 / 64 no more */
#endif
constexpr int kBins = RT_BVH_BINS;

struct Builder {
    const BuildTri* tris;
    std::vector<Box> boxes;
    std::vector<float> cent; // 3 per triangle
    std::vector<uint32_t> ids;
    std::vector<TmpNode> nodes;
    std::atomic<uint32_t> next_node{0};
    BvhBuildOptions opt;
    std::atomic<int> tasks_in_flight{0};
    int max_tasks = 1;

    uint32_t alloc() { return next_node.fetch_add(1); }

    static uint32_t ceil_log2(uint64_t v) {
        uint32_t r = 0;
        while ((1ull << r) < v) r++;
        return r;
    }

    void make_leaf(uint32_t me, uint32_t lo, uint32_t hi, const Box& box) {
        std::sort(ids.begin() + lo, ids.begin() + hi, [&](uint32_t a, uint32_t b) { return tris[a].prim_id < tris[b].prim_id; });
        nodes[me].box = box;
        nodes[me].left = nodes[me].right = 0xFFFFFFFFu;
        nodes[me].start = lo;
        nodes[me].count = hi - lo;
    }

    // depth = number of inner nodes above this node.  A subtree of n triangles split at the
    // object median needs ceil(log2(ceil(n / max_leaf))) more inner levels.
    void build(uint32_t me, uint32_t lo, uint32_t hi, uint32_t depth) {
        uint32_t n = hi - lo;
        Box box, cbox;
        box.reset();
        cbox.reset();
        for (uint32_t i = lo; i < hi; i++) {
            box.grow(boxes[ids[i]]);
            cbox.grow(&cent[3 * (size_t)ids[i]]);
        }
        if (n == 1) {
            make_leaf(me, lo, hi, box);
            return;
        }
        uint32_t levels_needed = ceil_log2((n + opt.max_leaf - 1) / opt.max_leaf);
        bool force_median = depth + levels_needed + 1 >= opt.max_depth;

        uint32_t mid = 0;
        bool have_split = false;
        if (!force_median) {
            float best_cost = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; axis++) {
                float cmin = cbox.mn[axis], cext = cbox.mx[axis] - cbox.mn[axis];
                if (!(cext > 0.0f)) continue;
                float scale = (float)kBins / cext;
                Box bin_box[kBins];
                uint32_t bin_cnt[kBins];
                for (int b = 0; b < kBins; b++) {
                    bin_box[b].reset();
                    bin_cnt[b] = 0;
                }
                for (uint32_t i = lo; i < hi; i++) {
                    int b = (int)((cent[3 * (size_t)ids[i] + axis] - cmin) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    bin_cnt[b]++;
                    bin_box[b].grow(boxes[ids[i]]);
                }
                float right_area[kBins];
                uint32_t right_cnt[kBins];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    acc.grow(bin_box[b]);
                    cnt += bin_cnt[b];
                    right_area[b] = acc.half_area();
                    right_cnt[b] = cnt;
                }
                acc.reset();
                cnt = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.grow(bin_box[b]);
                    cnt += bin_cnt[b];
                    if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = acc.half_area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = axis;
                        best_bin = b;
                    }
                }
            }
            if (best_axis >= 0) {
                float parent_area = box.half_area();
                float split_cost = opt.cost_traverse + opt.cost_intersect * (parent_area > 0.0f ? best_cost / parent_area : (float)n);
                float leaf_cost = opt.cost_intersect * (float)n;
                if (n <= opt.max_leaf && leaf_cost <= split_cost) {
                    make_leaf(me, lo, hi, box);
                    return;
                }
                float cmin = cbox.mn[best_axis], cext = cbox.mx[best_axis] - cbox.mn[best_axis];
                float scale = (float)kBins / cext;
                auto it = std::partition(ids.begin() + lo, ids.begin() + hi, [&](uint32_t id) {
                    int b = (int)((cent[3 * (size_t)id + best_axis] - cmin) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    return b <= best_bin;
                });
                mid = (uint32_t)(it - ids.begin());
                have_split = mid > lo && mid < hi;
            } else if (n <= opt.max_leaf) { // all centroids coincide
                make_leaf(me, lo, hi, box);
                return;
            }
        } else if (n <= opt.max_leaf) {
            make_leaf(me, lo, hi, box);
            return;
        }
        if (!have_split) { // object median along the widest centroid axis (ties by id: deterministic)
            int axis = 0;
            float ext = cbox.mx[0] - cbox.mn[0];
            for (int a = 1; a < 3; a++)
                if (cbox.mx[a] - cbox.mn[a] > ext) {
                    ext = cbox.mx[a] - cbox.mn[a];
                    axis = a;
                }
            mid = lo + n / 2;
            std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi, [&](uint32_t a, uint32_t b) {
                float ca = cent[3 * (size_t)a + axis], cb = cent[3 * (size_t)b + axis];
                return ca < cb || (ca == cb && a < b);
            });
        }
        uint32_t l = alloc(), r = alloc();
        nodes[me].box = box;
        nodes[me].left = l;
        nodes[me].right = r;
        nodes[me].start = nodes[me].count = 0;
        bool spawn = n > 32768 && tasks_in_flight.load(std::memory_order_relaxed) < max_tasks;
        if (spawn) {
            tasks_in_flight.fetch_add(1);
            auto fut = std::async(std::launch::async, [this, l, lo, mid, depth] { build(l, lo, mid, depth + 1); });
            build(r, mid, hi, depth + 1);
            fut.get();
            tasks_in_flight.fetch_sub(1);
        } else {
            build(l, lo, mid, depth + 1);
            build(r, mid, hi, depth + 1);
        }
    }
};

#if RT_BVH_REINSERT
// Insertion-based optimisation of the binary tree (after Bittner, Hapala, Havran, "Fast Insertion-Based Optimization of
// Bounding Volume Hierarchies", 2013, in its simplest form): take a subtree out (its parent goes with it, the sibling moves
// up), find by branch and bound the node next to which it enlarges the ancestors' boxes least, and put it back there.
// Candidates are the nodes with the largest area first.  The sum of the inner nodes' areas - what a top-down SAH build
// only approximates greedily - can only go down.  Returns the depth of the resulting tree.
static float union_area(const Box& a, const Box& b) {
    Box u = a;
    u.grow(b);
    return u.half_area();
}
static uint32_t reinsertion_optimize(std::vector<TmpNode>& nodes, uint32_t n_nodes, uint32_t& root, int passes, double fraction) {
    std::vector<uint32_t> parent(n_nodes, 0xFFFFFFFFu);
    auto is_leaf = [&](uint32_t x) { return nodes[x].left == 0xFFFFFFFFu; };
    {
        std::vector<uint32_t> st{root};
        while (!st.empty()) {
            const uint32_t x = st.back();
            st.pop_back();
            if (is_leaf(x)) continue;
            parent[nodes[x].left] = parent[nodes[x].right] = x;
            st.push_back(nodes[x].left);
            st.push_back(nodes[x].right);
        }
    }
    auto refit_up = [&](uint32_t x) { // recompute boxes from x to the root
        while (x != 0xFFFFFFFFu) {
            Box bx = nodes[nodes[x].left].box;
            bx.grow(nodes[nodes[x].right].box);
            nodes[x].box = bx;
            x = parent[x];
        }
    };
    struct Entry {
        float induced;
        uint32_t node;
        bool operator<(const Entry& o) const { return induced > o.induced; } // min-heap on the induced cost
    };
    for (int pass = 0; pass < passes; pass++) {
        std::vector<uint32_t> cand;
        for (uint32_t x = 0; x < n_nodes; x++)
            if (x != root && parent[x] != 0xFFFFFFFFu && parent[x] != root) cand.push_back(x);
        // a bounded number of candidates per pass (the step is serial: ~1.4 us per candidate): the largest nodes matter most
        const size_t take = std::min({cand.size(), (size_t)std::max(1.0, fraction * (double)cand.size()), (size_t)RT_BVH_REINSERT_MAX});
        std::partial_sort(cand.begin(), cand.begin() + (long)take, cand.end(), [&](uint32_t a, uint32_t b) {
            const float aa = nodes[a].box.half_area(), ab = nodes[b].box.half_area();
            return aa > ab || (aa == ab && a < b);
        });
        cand.resize(take);
        std::vector<Entry> heap;
        for (uint32_t nd : cand) {
            const uint32_t p = parent[nd];
            if (p == 0xFFFFFFFFu || p == root) continue; // moved under the root by an earlier step
            const uint32_t g = parent[p];
            const uint32_t sib = nodes[p].left == nd ? nodes[p].right : nodes[p].left;
            // take nd (and p) out: the sibling takes p's place
            if (nodes[g].left == p) nodes[g].left = sib;
            else nodes[g].right = sib;
            parent[sib] = g;
            refit_up(g);
            // branch and bound for the best neighbour
            const Box& nb = nodes[nd].box;
            const float na = nb.half_area();
            float best_cost = std::numeric_limits<float>::infinity();
            uint32_t best = sib;
            heap.clear();
            heap.push_back({0.0f, root});
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end());
                const Entry e = heap.back();
                heap.pop_back();
                if (e.induced + na >= best_cost) break; // nothing cheaper can follow (heap order)
                const float direct = union_area(nodes[e.node].box, nb);
                const float total = e.induced + direct;
                if (total < best_cost) {
                    best_cost = total;
                    best = e.node;
                }
                if (!is_leaf(e.node)) {
                    const float child_induced = total - nodes[e.node].box.half_area();
                    if (child_induced + na < best_cost) {
                        heap.push_back({child_induced, nodes[e.node].left});
                        std::push_heap(heap.begin(), heap.end());
                        heap.push_back({child_induced, nodes[e.node].right});
                        std::push_heap(heap.begin(), heap.end());
                    }
                }
            }
            // put it back: p becomes the parent of (best, nd) where best was
            const uint32_t bp = parent[best];
            nodes[p].left = best;
            nodes[p].right = nd;
            parent[best] = p;
            parent[nd] = p;
            parent[p] = bp;
            if (bp == 0xFFFFFFFFu) root = p;
            else if (nodes[bp].left == best) nodes[bp].left = p;
            else nodes[bp].right = p;
            refit_up(p);
        }
    }
    uint32_t depth = 0; // inner levels on the longest path
    std::vector<std::pair<uint32_t, uint32_t>> st{{root, 1u}};
    while (!st.empty()) {
        auto [x, d] = st.back();
        st.pop_back();
        if (is_leaf(x)) continue;
        depth = std::max(depth, d);
        st.push_back({nodes[x].left, d + 1});
        st.push_back({nodes[x].right, d + 1});
    }
    return depth;
}
#endif

void put_tri(const BuildTri& t, DevTri* o) {
    for (int a = 0; a < 3; a++) {
        o->v0[a] = t.v0[a];
        o->e1[a] = t.v1[a] - t.v0[a]; // shader/src/intersection.rs:104
        o->e2[a] = t.v2[a] - t.v0[a]; // :105
    }
    o->material_id = t.material_id;
    o->prim_id = t.prim_id;
    o->leaf_count = 0;
}

// ---- PLOC (Meister, Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume Hierarchy Construction", 2018): sort the
// triangles along a Morton curve, then repeatedly merge every pair of clusters that are each other's nearest neighbour (by the
// area of the union) within `radius` positions of the curve.  This host version is the executable statement of what the device
// build does (csrc/device_build.hip); it fills `nodes` (leaves first: node i = the i-th triangle in Morton order) and returns the root.
uint64_t morton21(float x) { // x in [0, 1]: 21 bits spread to every third bit (same expression in device_build.hip)
    uint64_t v = (uint64_t)(uint32_t)std::min(std::max(x * 2097152.0f, 0.0f), 2097151.0f);
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

uint32_t build_ploc(const std::vector<Box>& boxes, const std::vector<float>& cent, std::vector<uint32_t>& ids, std::vector<TmpNode>& nodes, uint32_t radius) {
    const size_t n = ids.size();
    Box cb;
    cb.reset();
    for (size_t i = 0; i < n; i++) cb.grow(&cent[3 * (size_t)ids[i]]);
    float inv[3];
    for (int a = 0; a < 3; a++) inv[a] = cb.mx[a] > cb.mn[a] ? 1.0f / (cb.mx[a] - cb.mn[a]) : 0.0f;
    std::vector<std::pair<uint64_t, uint32_t>> keyed(n);
    for (size_t i = 0; i < n; i++) {
        const float* c = &cent[3 * (size_t)ids[i]];
        const uint64_t code = (morton21((c[0] - cb.mn[0]) * inv[0]) << 2) | (morton21((c[1] - cb.mn[1]) * inv[1]) << 1) | morton21((c[2] - cb.mn[2]) * inv[2]);
        keyed[i] = {code, ids[i]};
    }
    std::sort(keyed.begin(), keyed.end());
    for (size_t i = 0; i < n; i++) ids[i] = keyed[i].second;
    nodes.resize(2 * n);
    std::vector<uint32_t> cl(n), nn(n), next;
    for (size_t i = 0; i < n; i++) {
        nodes[i].box = boxes[ids[i]];
        nodes[i].left = nodes[i].right = 0xFFFFFFFFu;
        nodes[i].start = (uint32_t)i;
        nodes[i].count = 1;
        cl[i] = (uint32_t)i;
    }
    uint32_t n_nodes = (uint32_t)n;
    size_t m = n;
    next.reserve(n);
    // Degenerate input stalls the pairing: with coincident triangles every cluster's nearest neighbour is the first of its
    // window, in a chain of growing triangles it is the predecessor - one mutual pair per round, and a tree as deep as the
    // input is long.  A round that merges less than a sixteenth of the clusters therefore escalates, for the rest of the build:
    // level 1 prefers position i ^ 1 among EQUAL areas (coincident triangles then pair up perfectly), level 2 pairs i and i ^ 1
    // outright.  Ordinary scenes never leave level 0.  (Level 1's rule at level 0 was measured: +5 % node visits on the
    // sponza-like scene, whose tessellated walls are full of exact ties.)
    int level = 0;
    while (m > 1) {
        for (size_t i = 0; i < m; i++) {
            const size_t buddy = i ^ 1;
            if (level == 2) {
                nn[i] = (uint32_t)(buddy < m ? buddy : i);
                continue;
            }
            const Box& bi = nodes[cl[i]].box;
            auto union_area = [&](size_t j) {
                Box u = bi;
                u.grow(nodes[cl[j]].box);
                return u.half_area();
            };
            float best = std::numeric_limits<float>::infinity();
            uint32_t bj = (uint32_t)i;
            if (level == 1 && buddy < m) best = union_area(buddy), bj = (uint32_t)buddy;
            const size_t lo = i > radius ? i - radius : 0, hi = std::min(m - 1, i + radius);
            for (size_t j = lo; j <= hi; j++) { // first best: the lower position wins ties
                if (j == i) continue;
                const float a = union_area(j);
                if (a < best) best = a, bj = (uint32_t)j;
            }
            nn[i] = bj;
        }
        next.clear();
        const uint32_t nodes_before = n_nodes;
        for (size_t i = 0; i < m; i++) {
            const uint32_t j = nn[i];
            if (j != i && nn[j] == i) {
                if (i < j) {
                    TmpNode& p = nodes[n_nodes];
                    p.box = nodes[cl[i]].box;
                    p.box.grow(nodes[cl[j]].box);
                    p.left = cl[i];
                    p.right = cl[j];
                    p.start = p.count = 0;
                    next.push_back(n_nodes++);
                }
            } else {
                next.push_back(cl[i]);
            }
        }
        if ((n_nodes - nodes_before) < m / 16 && level < 2) level++;
        m = next.size();
        std::copy(next.begin(), next.end(), cl.begin());
    }
    nodes.resize(n_nodes);
    return cl[0];
}

// ---- 8-wide collapse (DevNode8).  Which binary nodes become wide nodes, which are absorbed, which subtrees become one leaf is
// chosen by dynamic programming over the binary tree (minimum expected cost): c[k-1] = cheapest cost of a subtree when it may
// occupy at most k child slots of its wide parent: k = 1: either one leaf (<= max_leaf triangles) or a wide node of its own,
// area * cost_traverse8 + the best split of 8 slots between its two children; k > 1: the best split of k slots between its
// children, or k - 1 slots.  (Round 1's 4-wide collapse used the same program with four slots; the greedy "open the child with
// the largest area" rule it replaced was 0.8-1.5 % worse.)
constexpr int W8 = 8;
struct Dp8 {
    float c[W8];
    uint8_t split[W8]; // [k-1]: slots for the left child when k are distributed (0: use k - 1); [0]: the split of 8 when the node is a wide node
    uint8_t leaf;
};

void quantise_axis(const Box& parent, int a, float org, uint32_t& ex_out, double& scale_out) {
    double extent = (double)parent.mx[a] - (double)parent.mn[a];
    int e = 1;
    if (extent > 0.0) {
        int fe;
        std::frexp(extent / 255.0, &fe);
        e = fe + 127;
        if (e < 1) e = 1;
        if (e > 254) e = 254;
    }
    (void)org;
    ex_out = (uint32_t)(e - 127) & 0xFFu;
    scale_out = std::ldexp(1.0, e - 127);
}

void quantise_box(const Box& cb, int a, float org, double scale, uint32_t& qlo, uint32_t& qhi) {
    double lo = std::floor(((double)cb.mn[a] - (double)org) / scale);
    double hi = std::ceil(((double)cb.mx[a] - (double)org) / scale);
    lo = std::min(std::max(lo, 0.0), 255.0);
    hi = std::min(std::max(hi, 0.0), 255.0);
    while (lo > 0.0 && (double)org + lo * scale > (double)cb.mn[a]) lo -= 1.0;
    while (hi < 255.0 && (double)org + hi * scale < (double)cb.mx[a]) hi += 1.0;
    qlo = (uint32_t)lo;
    qhi = (uint32_t)hi;
}

void collapse8(std::vector<TmpNode>& nodes, std::vector<uint32_t>& ids, uint32_t n_nodes, uint32_t root, const BuildTri* tris_in, const BvhBuildOptions& opt,
               uint32_t max_leaf, BvhBuild& out) {
    auto is_leaf = [&](uint32_t t) { return nodes[t].left == 0xFFFFFFFFu; };
    DevTri blank;
    std::memset(&blank, 0, sizeof blank);
    if (is_leaf(root)) { // the whole scene is one leaf: a root node with that one child in slot 0, so that every walk starts at node 0
        const TmpNode& t = nodes[root];
        out.nodes.emplace_back();
        DevNode8& d = out.nodes[0];
        uint32_t ex[3];
        for (int a = 0; a < 3; a++) {
            d.org[a] = t.box.mn[a];
            double scale;
            quantise_axis(t.box, a, d.org[a], ex[a], scale);
            uint32_t qlo, qhi;
            quantise_box(t.box, a, d.org[a], scale, qlo, qhi);
            d.qlo[a][0] = qlo | 0xFFFFFF00u; // slots 1..7 empty: inverted
            d.qhi[a][0] = qhi;
            d.qlo[a][1] = 0xFFFFFFFFu;
            d.qhi[a][1] = 0u;
        }
        d.ex_imask = ex[0] | (ex[1] << 8) | (ex[2] << 16);
        d.child_base = 0;
        d.tri_base = 0;
        d.lmask = 1u;
        d._pad = 0;
        for (uint32_t i = 0; i < RT_DEV_LEAF_STRIDE; i++) {
            out.tris.push_back(blank);
            if (i < t.count) put_tri(tris_in[ids[t.start + i]], &out.tris[i]);
        }
        out.tris[0].leaf_count = t.count;
        out.n_leaves = 1;
        out.depth = 1;
        return;
    }
    std::vector<Dp8> dp(n_nodes);
    std::vector<uint32_t> sub_count(n_nodes, 0);
    const float inf = std::numeric_limits<float>::infinity();
    {
        struct Frame {
            uint32_t node;
            int phase;
        };
        std::vector<Frame> st;
        st.push_back({root, 0});
        while (!st.empty()) {
            Frame f = st.back();
            st.pop_back();
            const TmpNode& t = nodes[f.node];
            Dp8& d = dp[f.node];
            if (t.left == 0xFFFFFFFFu) {
                const float lc = opt.cost_intersect * (float)t.count * t.box.half_area();
                for (int k = 0; k < W8; k++) d.c[k] = lc, d.split[k] = 0;
                d.leaf = 1;
                sub_count[f.node] = t.count;
                continue;
            }
            if (f.phase == 0) {
                st.push_back({f.node, 1});
                st.push_back({t.left, 0});
                st.push_back({t.right, 0});
                continue;
            }
            // triangles of the subtree (saturating well above max_leaf): a subtree of <= max_leaf triangles may become one leaf,
            // whatever the order of its ids (the leaf gathers them)
            sub_count[f.node] = std::min<uint32_t>(sub_count[t.left] + sub_count[t.right], 1u << 20);
            const Dp8 &dl = dp[t.left], &dr = dp[t.right];
            float dist[W8 + 1];
            uint8_t arg[W8 + 1];
            for (int k = 0; k <= W8; k++) dist[k] = inf, arg[k] = 0;
            for (int k = 2; k <= W8; k++)
                for (int i = 1; i < k; i++) {
                    const float v = dl.c[i - 1] + dr.c[k - i - 1];
                    if (v < dist[k]) dist[k] = v, arg[k] = (uint8_t)i;
                }
            const float area = t.box.half_area();
            const float wide = opt.cost_traverse8 * area + dist[W8];
            const uint32_t cnt = sub_count[f.node];
            const float leafc = cnt && cnt <= max_leaf ? opt.cost_intersect * (float)cnt * area : inf;
            // Finite coordinates whose box area overflows f32 (~1e19 and beyond) make every cost inf or NaN: no `v < dist` above
            // succeeds and arg stays 0, which is not a split.  The rule then is explicit, the same in device_build.hip:
            // a leaf only when the count allows it, else a wide node split 1 : 7 - a poor tree, but a valid one.
            if (arg[W8] == 0) arg[W8] = 1;
            d.leaf = (cnt && cnt <= max_leaf && !(wide < leafc)) ? 1 : 0;
            d.c[0] = std::min(leafc, wide);
            d.split[0] = arg[W8];
            for (int k = 2; k <= W8; k++) {
                if (dist[k] < d.c[k - 2]) d.c[k - 1] = dist[k], d.split[k - 1] = arg[k];
                else d.c[k - 1] = d.c[k - 2], d.split[k - 1] = 0;
            }
        }
    }
    std::function<void(uint32_t, int, uint32_t*, int&)> expand = [&](uint32_t m, int k, uint32_t* ch, int& nch) {
        TmpNode& t = nodes[m];
        if (t.left != 0xFFFFFFFFu) {
            while (k > 1 && dp[m].split[k - 1] == 0) k--;
            if (k > 1) {
                const int i = dp[m].split[k - 1];
                expand(t.left, i, ch, nch);
                expand(t.right, k - i, ch, nch);
                return;
            }
            if (dp[m].leaf) { // the whole subtree as one leaf: gather its triangles (appended to ids), in index order
                uint32_t gathered[RT_DEV_LEAF_STRIDE], ng = 0, todo[2 * RT_DEV_LEAF_STRIDE], nt = 0;
                todo[nt++] = m;
                while (nt) {
                    const TmpNode& u = nodes[todo[--nt]];
                    if (u.left == 0xFFFFFFFFu) {
                        for (uint32_t i = 0; i < u.count && ng < RT_DEV_LEAF_STRIDE; i++) gathered[ng++] = ids[u.start + i];
                    } else {
                        todo[nt++] = u.left;
                        todo[nt++] = u.right;
                    }
                }
                std::sort(gathered, gathered + ng, [&](uint32_t x, uint32_t y) { return tris_in[x].prim_id < tris_in[y].prim_id; });
                t.start = (uint32_t)ids.size();
                t.count = ng;
                ids.insert(ids.end(), gathered, gathered + ng);
                t.left = t.right = 0xFFFFFFFFu;
            }
        }
        ch[nch++] = m;
    };
    struct Item {
        uint32_t tmp, dev, depth;
    };
    std::vector<Item> stack;
    out.nodes.reserve(nodes.size() / 4 + 1);
    out.nodes.emplace_back();
    stack.push_back({root, 0, 1});
    const float root_area = nodes[root].box.half_area();
    double cost = 0.0;
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        out.depth = std::max(out.depth, it.depth);
        const TmpNode t = nodes[it.tmp];
        uint32_t ch[W8];
        int nch = 0;
        {
            const int i = dp[it.tmp].split[0];
            expand(t.left, i, ch, nch);
            expand(t.right, W8 - i, ch, nch);
        }
        // Slot assignment: slot s lies toward the corner (s & 1 ? +x : -x, s & 2 ? +y : -y, s & 4 ? +z : -z) of the node.
        // Greedy: repeatedly give the (child, slot) pair with the largest projection of the child's centre offset on
        // the slot's diagonal.
        float pc[3];
        for (int a = 0; a < 3; a++) pc[a] = 0.5f * t.box.mn[a] + 0.5f * t.box.mx[a];
        float score[W8][W8];
        for (int c = 0; c < nch; c++) {
            const Box& cb = nodes[ch[c]].box;
            float off[3];
            for (int a = 0; a < 3; a++) off[a] = (0.5f * cb.mn[a] + 0.5f * cb.mx[a]) - pc[a];
            for (int sl = 0; sl < W8; sl++) score[c][sl] = (sl & 1 ? off[0] : -off[0]) + (sl & 2 ? off[1] : -off[1]) + (sl & 4 ? off[2] : -off[2]);
        }
        int slot_child[W8];
        bool child_done[W8] = {false, false, false, false, false, false, false, false};
        for (int sl = 0; sl < W8; sl++) slot_child[sl] = -1;
        for (int round = 0; round < nch; round++) {
            int bc = -1, bs = -1;
            float best = -inf;
            for (int c = 0; c < nch; c++) {
                if (child_done[c]) continue;
                for (int sl = 0; sl < W8; sl++)
                    if (slot_child[sl] < 0 && score[c][sl] > best) best = score[c][sl], bc = c, bs = sl;
            }
            if (bc < 0) { // only NaN scores are left (degenerate boxes): any free slot
                for (int c = 0; c < nch && bc < 0; c++)
                    if (!child_done[c]) bc = c;
                for (int sl = 0; sl < W8 && bs < 0; sl++)
                    if (slot_child[sl] < 0) bs = sl;
            }
            slot_child[bs] = bc;
            child_done[bc] = true;
        }
        uint32_t imask = 0, lmask = 0;
        for (int sl = 0; sl < W8; sl++) {
            if (slot_child[sl] < 0) continue;
            if (is_leaf(ch[slot_child[sl]])) lmask |= 1u << sl;
            else imask |= 1u << sl;
        }
        const uint32_t child_base = (uint32_t)out.nodes.size();
        const int n_inner = __builtin_popcount(imask);
        for (int c = 0; c < n_inner; c++) out.nodes.emplace_back();
        const uint32_t tri_base = (uint32_t)out.tris.size();
        int inner_rank = 0;
        std::vector<Item> pushes;
        for (int sl = 0; sl < W8; sl++) {
            if (slot_child[sl] < 0) continue;
            const uint32_t cn = ch[slot_child[sl]];
            if (imask & (1u << sl)) {
                pushes.push_back({cn, child_base + (uint32_t)inner_rank, it.depth + 1});
                inner_rank++;
            } else {
                const TmpNode& lf = nodes[cn];
                const size_t first = out.tris.size();
                for (uint32_t i = 0; i < RT_DEV_LEAF_STRIDE; i++) {
                    out.tris.push_back(blank);
                    if (i < lf.count) put_tri(tris_in[ids[lf.start + i]], &out.tris[first + i]);
                }
                out.tris[first].leaf_count = lf.count;
                out.n_leaves++;
                if (root_area > 0) cost += opt.cost_intersect * lf.count * lf.box.half_area() / root_area;
            }
        }
        for (int c = (int)pushes.size() - 1; c >= 0; c--) stack.push_back(pushes[c]);
        if (root_area > 0) cost += opt.cost_traverse8 * t.box.half_area() / root_area;
        DevNode8& d = out.nodes[it.dev];
        uint32_t ex[3];
        for (int a = 0; a < 3; a++) {
            d.org[a] = t.box.mn[a];
            double scale;
            quantise_axis(t.box, a, d.org[a], ex[a], scale);
            for (int h = 0; h < 2; h++) {
                uint32_t lo_word = 0, hi_word = 0;
                for (int i = 0; i < 4; i++) {
                    const int sl = 4 * h + i;
                    uint32_t qlo = 255, qhi = 0; // empty slot: inverted, never entered
                    if (slot_child[sl] >= 0) quantise_box(nodes[ch[slot_child[sl]]].box, a, d.org[a], scale, qlo, qhi);
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
    }
    // An empty slot can only be entered when the float evaluation cannot tell 255 grid steps apart (degenerate node, ray
    // origin ~1e8 grid steps away).  The traversal masks the hit bits with imask | lmask, so it is never followed.
    out.sah_cost = cost;
}

} // namespace

void build_bvh(const BuildTri* tris_in, size_t n_in, const BvhBuildOptions& opt_in, BvhBuild& out) {
    out = BvhBuild();
    Builder b;
    b.opt = opt_in;
    if (b.opt.max_leaf < 1) b.opt.max_leaf = 1;
    if (b.opt.max_leaf > RT_DEV_LEAF_STRIDE) b.opt.max_leaf = RT_DEV_LEAF_STRIDE; // a leaf owns RT_DEV_LEAF_STRIDE triangle records
    if (b.opt.max_depth > RT_DEV_MAX_BVH_DEPTH) b.opt.max_depth = RT_DEV_MAX_BVH_DEPTH;
    int hw = (int)std::thread::hardware_concurrency();
    b.max_tasks = std::max(1, (b.opt.threads > 0 ? b.opt.threads : (hw > 0 ? hw : 1)) - 1);
    b.tris = tris_in;
    b.ids.reserve(n_in);
    b.boxes.resize(n_in);
    b.cent.resize(3 * n_in);
    for (size_t i = 0; i < n_in; i++) {
        const BuildTri& t = tris_in[i];
        bool finite = true;
        for (int a = 0; a < 3; a++) finite = finite && std::isfinite(t.v0[a]) && std::isfinite(t.v1[a]) && std::isfinite(t.v2[a]);
        if (!finite) continue;
        Box bx;
        bx.reset();
        bx.grow(t.v0);
        bx.grow(t.v1);
        bx.grow(t.v2);
        b.boxes[i] = bx;
        for (int a = 0; a < 3; a++) b.cent[3 * i + a] = 0.5f * bx.mn[a] + 0.5f * bx.mx[a];
        b.ids.push_back((uint32_t)i);
    }
    size_t n = b.ids.size();
    if (n == 0) return;
    if (opt_in.method == 1) { // PLOC on the host: the statement of the device build (quality / structure reference)
        std::vector<TmpNode> pn;
        const uint32_t proot = build_ploc(b.boxes, b.cent, b.ids, pn, opt_in.ploc_radius);
        collapse8(pn, b.ids, (uint32_t)pn.size(), proot, tris_in, b.opt, b.opt.max_leaf, out);
        return;
    }
    b.nodes.resize(2 * n);
    uint32_t root = b.alloc();
    b.build(root, 0, (uint32_t)n, 0);
#if RT_BVH_REINSERT
    if (n > 8 && opt_in.reinsert) { // keep the top-down tree when the optimised one would exceed the depth the kernels' stacks are sized for
        std::vector<TmpNode> keep(b.nodes.begin(), b.nodes.begin() + b.next_node.load());
        const uint32_t keep_root = root;
        const uint32_t d = reinsertion_optimize(b.nodes, b.next_node.load(), root, RT_BVH_REINSERT_PASSES, RT_BVH_REINSERT_FRACTION);
        if (d > b.opt.max_depth) {
            std::copy(keep.begin(), keep.end(), b.nodes.begin());
            root = keep_root;
        }
    }
#endif

    collapse8(b.nodes, b.ids, b.next_node.load(), root, tris_in, b.opt, b.opt.max_leaf, out);
}

} // namespace rt
