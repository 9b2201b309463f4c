// device_layout.h — HBM layout of the scene as the HIP kernels read it.
//
// The reference kernel reads a 48-byte AoS BvhNode as 8-10 scalar u32 loads per
// visit (shader/src/scene_access.rs:110-159), two visits per tree level, and reaches a
// triangle through three dependent gathers: index -> Triangle -> 3 x Vertex = 56 bytes
// (shader/src/bvh.rs:113-122, triangle_access.rs:26-47).  Here:
//
//   DevNode8 80 B, one per INNER node of an 8-wide BVH: the boxes of up to eight children,
//            quantised to 8 bits per plane on a per-node grid (origin + power-of-two scale per
//            axis, boxes rounded outward).  One visit = 5 x dwordx4 and replaces three to four
//            visits of a binary tree.  Quantised boxes are supersets of the exact ones: they
//            only filter, every hit is decided by the triangle test, results are unchanged.
//            Leaves have no node record: a leaf is a run of the triangle array (length in its first record).
//   DevTri   48 B, pre-gathered (v0, e1 = v1-v0, e2 = v2-v0, material, original
//            index) stored in leaf order: one contiguous 48-byte read (3 x dwordx4)
//            per triangle test.  e1/e2 are the same f32 subtractions the reference
//            does per test (shader/src/intersection.rs:104-105), done once.
#ifndef RT_DEVICE_LAYOUT_H
#define RT_DEVICE_LAYOUT_H

#include <stdint.h>

#define RT_DEV_LEAF_FLAG 0x80000000u
#define RT_DEV_LEAF_START_MASK 0x7FFFFFFFu
#define RT_DEV_REF_NONE 0xFFFFFFFFu /* "nothing to visit" (a leaf reference no tree contains) */
#define RT_DEV_MAX_LEAF_TRIS 4u
#define RT_DEV_MAX_TRIS 0x07FFFFFFu
#define RT_DEV_MAX_BVH_DEPTH 32 /* binary build depth bound; the 8-wide tree is at most this deep */
/* A visit parks at most two (base, mask) groups - the siblings still to visit and a postponed leaf group - so a tree of
   depth D needs 2*D + 2 64-bit stack entries per lane (DevScene::stack_entries). */
#define RT_DEV_MAX_STACK_ENTRIES (2 * RT_DEV_MAX_BVH_DEPTH + 2)

#define RT_PRIM_MISS 0xFFFFFFFFu
#define RT_PRIM_SPHERE_FLAG 0x80000000u

// After Ylitie, Karras, Laine, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", 2017 (round 1 used a
// 4-wide node with a 4-key distance sort; measured against it in round 2, profiles/ab_r02.json):
// 80 bytes = 5 x dwordx4, up to eight children in SLOTS chosen by the builder so that a
// child's slot number says toward which corner of the node it lies (bit a set: the high side of axis a).  A ray visits
// the children it enters in increasing (slot XOR ray octant): no distance sort.  Inner children are consecutive nodes in
// slot order (child = child_base + number of inner slots below it), every leaf child owns RT_DEV_LEAF_STRIDE consecutive
// triangle records in slot order (first = tri_base + stride * number of leaf slots below it; the run length is in the
// first record as before), so what is left of a node after a visit is a (base, 8-bit mask) pair: one stack entry
// per visit instead of up to three.
struct DevNode8 { // 80 bytes, 16-byte aligned
    float org[3];         // quantisation origin (the node's box minimum)
    uint32_t ex_imask;    // byte 0..2: signed exponent of axis x,y,z; byte 3: slots holding an inner child
    uint32_t child_base;  // node index of the first inner child
    uint32_t tri_base;    // triangle record of the first leaf child
    uint32_t lmask;       // byte 0: slots holding a leaf
    uint32_t _pad;
    uint32_t qlo[3][2];   // [axis][half]: byte i = quantised lower plane of slot 4 * half + i
    uint32_t qhi[3][2];   // upper planes; empty slots are inverted (lo 255, hi 0)
};
#define RT_DEV_LEAF_STRIDE 4u
#define RT_DEV_MAX_NODES 0x3FFFFFFFu

struct DevTri { // 48 bytes, 16-byte aligned
    float v0[3];
    float e1[3];
    float e2[3];
    uint32_t material_id;
    uint32_t prim_id;    // index in the caller's triangle array
    uint32_t leaf_count; // on the first triangle of a leaf: triangles in the leaf (1..4); 0 on the others
};

struct DevMaterial { // 32 bytes: the 8 words of Material the kernel reads (shader/src/material.rs:16-63), f16 fields decoded
    float albedo[3];
    float metallic;
    float emission[3];
    float ior;
    float transmission;
    float roughness; // only used by the extended mode
    float _pad[2];
};

struct DevSphere { // 32 bytes
    float center[3];
    float radius;
    uint32_t material_id;
    uint32_t _pad[3];
};

struct DevLight { // 64 bytes: the 11 words of Light the kernel reads (shader/src/scene_access.rs:60-107) + one per-light constant
    float position[3];
    uint32_t light_type;
    float color[3];
    float intensity;
    float direction[3];
    uint32_t _pad;
    float neg_ndir[3]; // -normalize(direction) (lighting.rs:103, 132): the same for every pixel, so evaluated once at upload, by
    uint32_t _pad2;    // the host, in the same f32 operation order (x*x + y*y + z*z, IEEE sqrt and divide): same bits
};

struct DevScene {
    const DevNode8* nodes;
    const DevTri* tris; // RT_DEV_LEAF_STRIDE records per leaf, in the order of the nodes' leaf slots
    const DevSphere* spheres;
    const DevLight* lights;
    const DevMaterial* materials;
    uint32_t n_nodes; // 0: no triangles.  Otherwise node 0 is the root (a scene of one leaf gets a root with that one child)
    uint32_t n_tris;  // triangle records incl. padding
    uint32_t n_spheres;
    uint32_t n_lights;
    uint32_t n_materials; // `materials.len()` of shader/src/lib.rs:307 := material_count
    uint32_t stack_entries; // 64-bit stack entries per lane the kernels must provide: 2 * depth + 2
};

// Camera terms that do not depend on the pixel, computed once on the host in the
// reference's operation order (shader/src/ray.rs:33-44).
struct DevCamera {
    float origin[3];
    float forward[3];
    float right[3];   // forward x up      (not normalised, ray.rs:43)
    float true_up[3]; // right x forward   (not normalised, ray.rs:44)
    float width_f, height_f;
    float aspect;    // width / height
    float fov_scale; // tan(fov * 0.5 * pi / 180)
};

struct DevFrame {
    DevCamera cam;
    uint32_t width, height;
    uint32_t tile_size;
    uint32_t tiles_x, tiles_y;
    uint32_t tile_first, tile_stride, n_owned_tiles; // owned tile k has row-major index tile_first + k * tile_stride
    uint32_t mode;                                   // RT_MODE_*
    uint32_t channel_mask;                           // bit c set: write channel texture c (rt_dispatch_tile writes one)
    uint32_t cur_bounce, max_bounce;                 // mode 1 pass selection (shader/src/lib.rs:117-121)
    uint32_t spp, frame_seed;
    uint32_t flags;                                  // RT_FLAG_* of rt_render_params
    // when single_tile != 0 the launch covers exactly one tile given explicitly (rt_dispatch_tile)
    uint32_t single_tile, tile_off_x, tile_off_y, tile_w, tile_h;
};

struct DevTargets {
    float* rgba32f;      // width*height*4 floats (r,g,b,1)
    uint8_t* chan[3];    // three Rgba8Unorm channel textures, width*height*4 bytes each
    uint32_t* prim_id;   // per-pixel closest primitive (modes 0/1)
    float* hit_t;        // per-pixel hit distance       (modes 0/1)
    unsigned long long* counters; // [0] rays [1] node visits [2] tri tests [3] camera [4] continuation [5] shadow segments
};

#endif
