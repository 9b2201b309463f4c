// bvh_builder.h — host-side acceleration-structure build for the HIP layout.
//
// The reference builds its BVH with the third-party `bvh` crate (<= 100k triangles,
// src/bvh.rs:125-151) or in mesh-order chunks of 32+ triangles (> 100k, :154-247) and
// its kernel neither orders children nor culls by the closest hit
// (shader/src/bvh.rs:40-85).  Closest-hit results do not depend on topology, so the
// library builds its own: binned-SAH binary tree, <= 4 triangles per leaf, bounded depth,
// collapsed into a 4-wide tree (the child with the largest surface area is opened until a
// node has four children) and emitted in the 48-byte quantised layout of device_layout.h.
#ifndef RT_BVH_BUILDER_H
#define RT_BVH_BUILDER_H

#include <cstddef>
#include <cstdint>
#include <vector>

#include "device_layout.h"

namespace rt {

struct BuildTri {
    float v0[3], v1[3], v2[3];
    uint32_t material_id;
    uint32_t prim_id;
};

struct BvhBuild {
    std::vector<DevNode4> nodes;
    std::vector<DevTri> tris; // leaf order
    uint32_t root_ref = RT_DEV_REF_NONE; // nothing to visit
    uint32_t depth = 0;                   // inner-node levels of the 4-wide tree on the longest root-to-leaf path
    uint32_t n_leaves = 0;
    double sah_cost = 0.0;
    // 8-wide variant of the same binary tree (BvhBuildOptions::wide8); an inner root always exists when nodes8 is not empty
    std::vector<DevNode8> nodes8;
    std::vector<DevTri> tris8; // RT_DEV_LEAF_STRIDE records per leaf
    uint32_t depth8 = 0;
    double sah_cost8 = 0.0;
};

struct BvhBuildOptions {
    int threads = 0;        // 0 = hardware_concurrency
    uint32_t max_leaf = 4;  // <= RT_DEV_MAX_LEAF_TRIS
    uint32_t max_depth = RT_DEV_MAX_BVH_DEPTH;
    float cost_traverse = 0.7f; // relative to cost_intersect: measured optimum on the headline frame (0.5-0.75: +1 % over 1.0)
    float cost_intersect = 1.0f;
    bool wide8 = false;          // also emit the 8-wide variant
    float cost_traverse8 = 1.0f; // its traversal cost per node (a visit tests eight boxes)
};

// Triangles with a non-finite coordinate are dropped: Möller–Trumbore can never accept
// them (every comparison with NaN fails, shader/src/intersection.rs:109-130).
void build_bvh(const BuildTri* tris, size_t n, const BvhBuildOptions& opt, BvhBuild& out);

} // namespace rt
#endif
