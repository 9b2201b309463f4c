// rt_oracle_bvh.cpp — restatement of the reference's host BVH builder
// (src/bvh.rs:104-374).  TEST INFRASTRUCTURE ONLY, see rt_oracle.h.
//
//  * empty scene       -> one empty leaf with inverted infinite bounds (:105-114)
//  * > 100,000 tris    -> build_chunked (:154-189) + build_simple_top_level_bvh (:192-247): EXACT
//  * otherwise         -> one triangle per leaf, pre-order flatten with left = parent + 1
//                         (:278-374).  The reference takes the topology from the third-party
//                         `bvh` crate 0.11 (Cargo.toml:19, call site src/bvh.rs:142), which is
//                         not under /root/reference: topology UNPINNED.  We split at the object
//                         median of the widest centroid axis; closest-hit results do not depend
//                         on topology (except equal-t ties).

#include "rt_oracle.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>

namespace {

rt_aabb aabb_new(const float mn[3], const float mx[3]) { // shared/src/lib.rs:753-760
    rt_aabb a;
    a.min[0] = mn[0]; a.min[1] = mn[1]; a.min[2] = mn[2]; a._padding0 = 0.0f;
    a.max[0] = mx[0]; a.max[1] = mx[1]; a.max[2] = mx[2]; a._padding1 = 0.0f;
    return a;
}
rt_aabb aabb_empty() { // :763-768
    float inf = std::numeric_limits<float>::infinity();
    float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    return aabb_new(mn, mx);
}
rt_aabb aabb_union(const rt_aabb& a, const rt_aabb& b) { // :771-784 (f32::min/max)
    float mn[3], mx[3];
    for (int i = 0; i < 3; i++) {
        mn[i] = fminf(a.min[i], b.min[i]);
        mx[i] = fmaxf(a.max[i], b.max[i]);
    }
    return aabb_new(mn, mx);
}
rt_aabb tri_aabb(const rt_triangle& t, const rt_vertex* v) { // Triangle::bounding_box :671-685
    const float* a = v[t.v0_index].position;
    const float* b = v[t.v1_index].position;
    const float* c = v[t.v2_index].position;
    float mn[3], mx[3];
    for (int i = 0; i < 3; i++) {
        mn[i] = fminf(fminf(a[i], b[i]), c[i]);
        mx[i] = fmaxf(fmaxf(a[i], b[i]), c[i]);
    }
    return aabb_new(mn, mx);
}
rt_bvh_node node_leaf(const rt_aabb& b, uint32_t start, uint32_t count) { // BvhNode::leaf :806-814
    return rt_bvh_node{b, 0xFFFFFFFFu, 0xFFFFFFFFu, start, count};
}
rt_bvh_node node_internal(const rt_aabb& b, uint32_t l, uint32_t r) { // BvhNode::internal :817-825
    return rt_bvh_node{b, l, r, 0, 0};
}
bool is_leaf(const rt_bvh_node& n) { return n.left_child == 0xFFFFFFFFu && n.right_child == 0xFFFFFFFFu; } // :828-830

// build_simple_top_level_bvh — src/bvh.rs:192-247
std::vector<rt_bvh_node> top_level(std::vector<rt_bvh_node> leaves) {
    if (leaves.size() <= 1) return leaves;
    std::vector<rt_bvh_node> nodes;
    std::vector<rt_bvh_node> current = std::move(leaves);
    while (current.size() > 1) {
        std::vector<rt_bvh_node> next;
        for (size_t i = 0; i < current.size(); i += 2) {
            if (i + 1 < current.size()) {
                rt_aabb u = aabb_union(current[i].bounds, current[i + 1].bounds);
                uint32_t l = (uint32_t)nodes.size();
                nodes.push_back(current[i]);
                nodes.push_back(current[i + 1]);
                next.push_back(node_internal(u, l, l + 1));
            } else { // odd node out, promoted under a one-child internal node
                uint32_t idx = (uint32_t)nodes.size();
                nodes.push_back(current[i]);
                next.push_back(node_internal(current[i].bounds, idx, 0xFFFFFFFFu));
            }
        }
        current = std::move(next);
    }
    if (!current.empty()) nodes.push_back(current[0]);
    std::reverse(nodes.begin(), nodes.end());
    uint32_t offset = (uint32_t)nodes.size() - 1;
    for (auto& n : nodes) {
        if (!is_leaf(n)) {
            if (n.left_child != 0xFFFFFFFFu) n.left_child = offset - n.left_child;
            if (n.right_child != 0xFFFFFFFFu) n.right_child = offset - n.right_child;
        }
    }
    return nodes;
}

struct StdBuild {
    const rt_triangle* tris;
    const rt_vertex* verts;
    std::vector<rt_aabb> boxes;
    std::vector<float> cx, cy, cz;
    std::vector<rt_bvh_node> nodes;
    std::vector<uint32_t> indices;

    // convert_node_recursive's output order (:278-374): parent, whole left subtree, whole right subtree
    uint32_t build(uint32_t* ids, uint32_t n) {
        uint32_t me = (uint32_t)nodes.size();
        if (n == 1) {
            uint32_t start = (uint32_t)indices.size();
            indices.push_back(ids[0]);
            nodes.push_back(node_leaf(boxes[ids[0]], start, 1));
            return me;
        }
        rt_aabb bounds = aabb_empty();
        float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < n; i++) {
            bounds = aabb_union(bounds, boxes[ids[i]]);
            float c[3] = {cx[ids[i]], cy[ids[i]], cz[ids[i]]};
            for (int a = 0; a < 3; a++) {
                cmin[a] = fminf(cmin[a], c[a]);
                cmax[a] = fmaxf(cmax[a], c[a]);
            }
        }
        int axis = 0;
        float ext = cmax[0] - cmin[0];
        for (int a = 1; a < 3; a++)
            if (cmax[a] - cmin[a] > ext) { ext = cmax[a] - cmin[a]; axis = a; }
        const std::vector<float>& key = axis == 0 ? cx : (axis == 1 ? cy : cz);
        uint32_t mid = n / 2;
        std::nth_element(ids, ids + mid, ids + n, [&](uint32_t a, uint32_t b) {
            return key[a] < key[b] || (key[a] == key[b] && a < b);
        });
        nodes.push_back(node_internal(bounds, 0, 0));
        uint32_t l = build(ids, mid);
        uint32_t r = build(ids + mid, n - mid);
        nodes[me] = node_internal(bounds, l, r);
        return me;
    }
};

} // namespace

static int build_impl(const rt_triangle* tris, uint32_t n_tris, const rt_vertex* verts, uint32_t n_verts, rt_bvh_node* out_nodes, uint32_t* n_nodes,
                      uint32_t* out_indices, uint32_t* n_indices, bool force_standard) {
    if (!n_nodes || !n_indices) return -1;
    for (uint32_t i = 0; i < n_tris; i++)
        if (tris[i].v0_index >= n_verts || tris[i].v1_index >= n_verts || tris[i].v2_index >= n_verts) return -1;
    std::vector<rt_bvh_node> nodes;
    std::vector<uint32_t> indices;
    if (n_tris == 0) { // :105-114
        nodes.push_back(node_leaf(aabb_empty(), 0, 0));
    } else if (n_tris > 100000u && !force_standard) { // build_chunked :154-189
        size_t per_leaf = std::max<size_t>(n_tris / 10000u, 32);
        std::vector<rt_bvh_node> leaves;
        for (size_t base = 0, chunk_idx = 0; base < n_tris; base += per_leaf, chunk_idx++) {
            size_t len = std::min(per_leaf, (size_t)n_tris - base);
            rt_aabb box = aabb_empty();
            for (size_t i = 0; i < len; i++) box = aabb_union(box, tri_aabb(tris[base + i], verts));
            uint32_t start = (uint32_t)indices.size();
            for (size_t i = 0; i < len; i++) indices.push_back((uint32_t)(chunk_idx * per_leaf + i));
            leaves.push_back(node_leaf(box, start, (uint32_t)len));
        }
        nodes = leaves.size() > 1 ? top_level(std::move(leaves)) : leaves;
    } else { // build_standard :125-151
        StdBuild sb;
        sb.tris = tris;
        sb.verts = verts;
        sb.boxes.resize(n_tris);
        sb.cx.resize(n_tris); sb.cy.resize(n_tris); sb.cz.resize(n_tris);
        for (uint32_t i = 0; i < n_tris; i++) {
            sb.boxes[i] = tri_aabb(tris[i], verts);
            sb.cx[i] = (sb.boxes[i].min[0] + sb.boxes[i].max[0]) * 0.5f;
            sb.cy[i] = (sb.boxes[i].min[1] + sb.boxes[i].max[1]) * 0.5f;
            sb.cz[i] = (sb.boxes[i].min[2] + sb.boxes[i].max[2]) * 0.5f;
        }
        std::vector<uint32_t> ids(n_tris);
        std::iota(ids.begin(), ids.end(), 0u);
        sb.nodes.reserve(2 * (size_t)n_tris);
        sb.build(ids.data(), n_tris);
        nodes = std::move(sb.nodes);
        indices = std::move(sb.indices);
    }
    if (out_nodes) {
        if (*n_nodes < nodes.size() || *n_indices < indices.size()) return -2;
        std::copy(nodes.begin(), nodes.end(), out_nodes);
        if (out_indices) std::copy(indices.begin(), indices.end(), out_indices);
    }
    *n_nodes = (uint32_t)nodes.size();
    *n_indices = (uint32_t)indices.size();
    return 0;
}

extern "C" int oracle_build_bvh(const rt_triangle* tris, uint32_t n_tris, const rt_vertex* verts, uint32_t n_verts, rt_bvh_node* out_nodes,
                                uint32_t* n_nodes, uint32_t* out_indices, uint32_t* n_indices) {
    return build_impl(tris, n_tris, verts, n_verts, out_nodes, n_nodes, out_indices, n_indices, false);
}

// Baseline flavour (ii) only: one triangle per leaf at every size (the reference switches to 32+-triangle chunks above
// 100,000 triangles, which is what makes its traversal slow there).
extern "C" int oracle_build_bvh_per_triangle(const rt_triangle* tris, uint32_t n_tris, const rt_vertex* verts, uint32_t n_verts, rt_bvh_node* out_nodes,
                                             uint32_t* n_nodes, uint32_t* out_indices, uint32_t* n_indices) {
    return build_impl(tris, n_tris, verts, n_verts, out_nodes, n_nodes, out_indices, n_indices, true);
}
