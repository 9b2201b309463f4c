// bvh_builder.cpp — binned-SAH BVH2 build (host, multi-threaded), see bvh_builder.h.
#include "bvh_builder.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <future>
#include <limits>
#include <numeric>
#include <thread>

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

constexpr int kBins = 16;

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

void put_tri(const BuildTri& t, DevTri* o) {
    for (int a = 0; a < 3; a++) {
        o->v0[a] = t.v0[a];
        o->e1[a] = t.v1[a] - t.v0[a]; // shader/src/intersection.rs:104
        o->e2[a] = t.v2[a] - t.v0[a]; // :105
    }
    o->material_id = t.material_id;
    o->prim_id = t.prim_id;
    o->_pad = 0;
}

} // namespace

void build_bvh(const BuildTri* tris_in, size_t n_in, const BvhBuildOptions& opt_in, BvhBuild& out) {
    out = BvhBuild();
    Builder b;
    b.opt = opt_in;
    if (b.opt.max_leaf < 1) b.opt.max_leaf = 1;
    if (b.opt.max_leaf > RT_DEV_MAX_LEAF_TRIS) b.opt.max_leaf = RT_DEV_MAX_LEAF_TRIS;
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
    b.nodes.resize(2 * n);
    uint32_t root = b.alloc();
    b.build(root, 0, (uint32_t)n, 0);

    // Emit: triangles in leaf (= ids) order; inner nodes in depth-first pre-order.
    out.tris.resize(n);
    for (size_t i = 0; i < n; i++) put_tri(tris_in[b.ids[i]], &out.tris[i]);
    auto leaf_ref = [&](const TmpNode& t) { return RT_DEV_LEAF_FLAG | (t.count << RT_DEV_LEAF_COUNT_SHIFT) | t.start; };
    const TmpNode& rt_node = b.nodes[root];
    if (rt_node.left == 0xFFFFFFFFu) { // whole scene fits one leaf
        out.root_ref = leaf_ref(rt_node);
        out.n_leaves = 1;
        out.depth = 0;
        return;
    }
    // count inner nodes, then assign indices iteratively (explicit stack: no deep recursion)
    struct Item {
        uint32_t tmp, dev, depth;
    };
    std::vector<Item> stack;
    out.nodes.reserve(n);
    out.nodes.emplace_back();
    stack.push_back({root, 0, 1});
    double cost = 0.0;
    float root_area = rt_node.box.half_area();
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        out.depth = std::max(out.depth, it.depth);
        const TmpNode& t = b.nodes[it.tmp];
        const TmpNode* ch[2] = {&b.nodes[t.left], &b.nodes[t.right]};
        uint32_t refs[2];
        for (int c = 0; c < 2; c++) {
            if (ch[c]->left == 0xFFFFFFFFu) {
                refs[c] = leaf_ref(*ch[c]);
                out.n_leaves++;
                if (root_area > 0) cost += opt_in.cost_intersect * ch[c]->count * ch[c]->box.half_area() / root_area;
            } else {
                refs[c] = (uint32_t)out.nodes.size();
                out.nodes.emplace_back();
                if (root_area > 0) cost += opt_in.cost_traverse * ch[c]->box.half_area() / root_area;
            }
        }
        // right pushed first so the left subtree is laid out right after its parent
        if (!(refs[1] & RT_DEV_LEAF_FLAG)) stack.push_back({t.right, refs[1], it.depth + 1});
        if (!(refs[0] & RT_DEV_LEAF_FLAG)) stack.push_back({t.left, refs[0], it.depth + 1});
        DevNode& d = out.nodes[it.dev];
        std::memcpy(d.c0_min, ch[0]->box.mn, 12);
        std::memcpy(d.c0_max, ch[0]->box.mx, 12);
        std::memcpy(d.c1_min, ch[1]->box.mn, 12);
        std::memcpy(d.c1_max, ch[1]->box.mx, 12);
        d.child0 = refs[0];
        d.child1 = refs[1];
        d._pad0 = d._pad1 = 0;
    }
    out.root_ref = 0;
    out.sah_cost = cost + opt_in.cost_traverse;
}

} // namespace rt
