// bvh_builder.h — host-side acceleration-structure build for the HIP layout.
//
// The reference builds its BVH with the third-party `bvh` crate (<= 100k triangles,
// src/bvh.rs:125-151) or in mesh-order chunks of 32+ triangles (> 100k, :154-247) and
// its kernel neither orders children nor culls by the closest hit
// (shader/src/bvh.rs:40-85).  Closest-hit results do not depend on topology, so the
// library builds its own: binned-SAH binary tree, insertion-based optimisation, <= 4 triangles
// per leaf, bounded depth, collapsed into an 8-wide tree by a dynamic program over the binary
// tree and emitted in the 80-byte quantised layout of device_layout.h.
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
    std::vector<DevNode8> nodes; // empty: no (finite) triangles; otherwise nodes[0] is the root
    std::vector<DevTri> tris;    // RT_DEV_LEAF_STRIDE records per leaf, in slot order of the nodes
    uint32_t depth = 0;          // inner-node levels on the longest root-to-leaf path
    uint32_t n_leaves = 0;
    double sah_cost = 0.0;
};

struct BvhBuildOptions {
    int threads = 0;        // 0 = hardware_concurrency
    uint32_t max_leaf = 4;  // <= RT_DEV_LEAF_STRIDE
    uint32_t max_depth = RT_DEV_MAX_BVH_DEPTH;
    float cost_traverse = 0.7f; // of a BINARY node while the binary tree is built, relative to cost_intersect (measured optimum 0.5-0.75)
    float cost_intersect = 1.0f;
    float cost_traverse8 = 1.0f; // of an 8-wide node in the collapse (0.7 ... 1.5 measure the same)
    int method = 0;              // 0: binned SAH top-down + insertion-based optimisation (quality); 1: PLOC (what the device build does)
    bool reinsert = true;        // method 0: run the insertion-based optimisation
    uint32_t ploc_radius = 8;    // PLOC (host statement and device build): search radius along the Morton curve (8 / 16 / 32 measure within 1.5 %)
};

// Triangles with a non-finite coordinate are dropped: Möller–Trumbore can never accept
// them (every comparison with NaN fails, shader/src/intersection.rs:109-130).
void build_bvh(const BuildTri* tris, size_t n, const BvhBuildOptions& opt, BvhBuild& out);

} // namespace rt
#endif
