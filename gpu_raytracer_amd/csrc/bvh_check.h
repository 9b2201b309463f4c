// bvh_check.h — structural validation of an 8-wide tree (DevNode8 + DevTri arrays) the way the kernels decode it.  Shared by
// tests/check_bvh.cpp (host builder under ASan/UBSan) and rt_debug_check_bvh (the tree a context holds on the device, i.e. what
// the device build produced): see tests/check_bvh.cpp for what is checked.
#ifndef RT_BVH_CHECK_H
#define RT_BVH_CHECK_H

#include <cmath>
#include <cstdio>
#include <cstdint>
#include <limits>
#include <vector>

#include "bvh_builder.h"

namespace rtcheck {
using namespace rt;

static int g_fail = 0;
static bool g_print = true;
#define CHECK(cond, ...)                       \
    do {                                       \
        if (!(cond)) {                         \
            if (g_print && g_fail < 20) {      \
                std::printf("FAIL: " __VA_ARGS__); \
                std::printf("\n");             \
            }                                  \
            g_fail++;                          \
        }                                      \
    } while (0)

struct Ctx {
    const BvhBuild* b;
    std::vector<uint32_t> seen; // per prim id
    uint32_t max_depth = 0;
    size_t leaves = 0;
};

static void check_leaf(Ctx& c, uint32_t start, const double lo[3], const double hi[3], bool have_box) {
    CHECK(start < c.b->tris.size(), "leaf start %u out of %zu", start, c.b->tris.size());
    if (start >= c.b->tris.size()) return;
    const uint32_t count = c.b->tris[start].leaf_count;
    CHECK(count >= 1 && count <= RT_DEV_LEAF_STRIDE, "leaf at %u has count %u", start, count);
    CHECK((size_t)start + RT_DEV_LEAF_STRIDE <= c.b->tris.size(), "leaf at %u runs past the array", start);
    if (count < 1 || count > RT_DEV_LEAF_STRIDE || (size_t)start + RT_DEV_LEAF_STRIDE > c.b->tris.size()) return;
    for (uint32_t i = count; i < RT_DEV_LEAF_STRIDE; i++) CHECK(c.b->tris[start + i].leaf_count == 0, "padding record %u carries a count", start + i);
    c.leaves++;
    for (uint32_t i = 0; i < count; i++) {
        const DevTri& t = c.b->tris[start + i];
        if (i > 0) CHECK(t.leaf_count == 0, "triangle %u inside a leaf carries a count", start + i);
        CHECK(t.prim_id < c.seen.size(), "prim id %u out of range", t.prim_id);
        if (t.prim_id < c.seen.size()) c.seen[t.prim_id]++;
        if (!have_box) continue;
        for (int v = 0; v < 3; v++)
            for (int a = 0; a < 3; a++) {
                const double p = v == 0 ? (double)t.v0[a] : v == 1 ? (double)t.v0[a] + (double)t.e1[a] : (double)t.v0[a] + (double)t.e2[a];
                // e1 / e2 are rounded differences: allow the vertex to stick out by one float ulp of its magnitude
                const double tol = std::ldexp(std::fabs(p) + std::fabs((double)t.v0[a]), -22);
                CHECK(p >= lo[a] - tol && p <= hi[a] + tol, "triangle %u vertex %d axis %d: %.9g outside [%.9g, %.9g]", start + i, v, a, p, lo[a], hi[a]);
            }
    }
}

static void check_node(Ctx& c, uint32_t node, uint32_t depth, const double plo[3], const double phi[3], bool have_box) {
    CHECK(node < c.b->nodes.size(), "node %u out of %zu", node, c.b->nodes.size());
    if (node >= c.b->nodes.size()) return;
    c.max_depth = depth > c.max_depth ? depth : c.max_depth;
    CHECK(depth <= RT_DEV_MAX_BVH_DEPTH, "depth %u beyond the bound", depth);
    if (depth > RT_DEV_MAX_BVH_DEPTH) return;
    const DevNode8& n = c.b->nodes[node];
    const uint32_t imask = n.ex_imask >> 24, lmask = n.lmask & 0xFFu;
    CHECK((imask & lmask) == 0, "node %u: slot both inner and leaf (imask %02x lmask %02x)", node, imask, lmask);
    CHECK((n.lmask >> 8) == 0, "node %u: lmask has high bits", node);
    double scale[3];
    for (int a = 0; a < 3; a++) {
        const int k = (int)(int8_t)((n.ex_imask >> (8 * a)) & 0xFFu);
        CHECK(k >= -126 && k <= 127, "node %u axis %d exponent %d", node, a, k);
        scale[a] = std::ldexp(1.0, k);
    }
    int present = 0;
    for (int s = 0; s < 8; s++) {
        double lo[3], hi[3];
        bool inverted = false;
        for (int a = 0; a < 3; a++) {
            const uint32_t qlo = (n.qlo[a][s >> 2] >> (8 * (s & 3))) & 0xFFu, qhi = (n.qhi[a][s >> 2] >> (8 * (s & 3))) & 0xFFu;
            if (qlo > qhi) inverted = true;
            lo[a] = (double)n.org[a] + qlo * scale[a];
            hi[a] = (double)n.org[a] + qhi * scale[a];
        }
        const bool inner = (imask >> s) & 1u, leaf = (lmask >> s) & 1u;
        if (!inner && !leaf) {
            CHECK(inverted, "node %u: empty slot %d has a box that can be entered", node, s);
            continue;
        }
        CHECK(!inverted, "node %u: occupied slot %d has an inverted box", node, s);
        present++;
        if (have_box)
            for (int a = 0; a < 3; a++) { // a child's box may stick out of its parent's by the quantisation step, not more
                CHECK(lo[a] >= plo[a] - 2 * scale[a] - 1e-30 && hi[a] <= phi[a] + 2 * scale[a] + 1e-30, "node %u slot %d axis %d box [%.9g,%.9g] far outside parent [%.9g,%.9g]",
                      node, s, a, lo[a], hi[a], plo[a], phi[a]);
            }
        const uint32_t below = (1u << s) - 1u;
        if (leaf) check_leaf(c, n.tri_base + RT_DEV_LEAF_STRIDE * (uint32_t)__builtin_popcount(lmask & below), lo, hi, true);
        else check_node(c, n.child_base + (uint32_t)__builtin_popcount(imask & below), depth + 1, lo, hi, true);
    }
    CHECK(present >= (node == 0 ? 1 : 2), "node %u has %d children", node, present);
}


// Checks the whole tree; `seen` must hold one zero per caller triangle (prim id).  Returns the number of failures.
static int check_tree(const BvhBuild& b, std::vector<uint32_t>& seen, uint32_t* real_depth, size_t* leaves) {
    g_fail = 0;
    Ctx c;
    c.b = &b;
    c.seen.swap(seen);
    const double inf = std::numeric_limits<double>::infinity();
    const double lo[3] = {-inf, -inf, -inf}, hi[3] = {inf, inf, inf};
    if (!b.nodes.empty()) {
        check_node(c, 0, 1, lo, hi, false);
        CHECK(b.tris.size() == c.leaves * RT_DEV_LEAF_STRIDE, "%zu triangle records for %zu leaves", b.tris.size(), c.leaves);
        CHECK(b.n_leaves == c.leaves, "n_leaves %u, found %zu", b.n_leaves, c.leaves);
    } else {
        CHECK(b.tris.empty(), "a build without nodes has triangle records");
    }
    CHECK(c.max_depth <= b.depth, "real depth %u exceeds the reported %u", c.max_depth, b.depth);
    CHECK(2 * b.depth + 2 <= RT_DEV_MAX_STACK_ENTRIES, "depth %u needs more stack than the kernels provide", b.depth);
    if (real_depth) *real_depth = c.max_depth;
    if (leaves) *leaves = c.leaves;
    seen.swap(c.seen);
    return g_fail;
}

} // namespace rtcheck
#endif
