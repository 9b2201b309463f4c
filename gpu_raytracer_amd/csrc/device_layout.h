// device_layout.h — HBM layout of the scene as the HIP kernels read it.
//
// The reference kernel reads a 48-byte AoS BvhNode as 8-10 scalar u32 loads per
// visit (shader/src/scene_access.rs:110-159) and reaches a triangle through three
// dependent gathers: index -> Triangle -> 3 x Vertex = 56 bytes
// (shader/src/bvh.rs:113-122, triangle_access.rs:26-47).  Here:
//
//   DevNode  64 B, one per INNER node, holding BOTH child boxes and both child
//            references: one aligned 64-byte fetch (4 x dwordx4) per traversal
//            step instead of two 48-byte node reads.  Leaves have no node record:
//            a leaf is a (start,count) run in the triangle array, encoded in the
//            parent's child reference.
//   DevTri   48 B, pre-gathered (v0, e1 = v1-v0, e2 = v2-v0, material, original
//            index) stored in leaf order: one contiguous 48-byte read (3 x dwordx4)
//            per triangle test.  e1/e2 are the same f32 subtractions the reference
//            does per test (shader/src/intersection.rs:104-105), done once.
#ifndef RT_DEVICE_LAYOUT_H
#define RT_DEVICE_LAYOUT_H

#include <stdint.h>

#define RT_DEV_LEAF_FLAG 0x80000000u
#define RT_DEV_LEAF_COUNT_SHIFT 27
#define RT_DEV_LEAF_START_MASK 0x07FFFFFFu
#define RT_DEV_MAX_LEAF_TRIS 15u
#define RT_DEV_MAX_TRIS 0x07FFFFFFu
#define RT_DEV_STACK_DEPTH 32 /* per-lane traversal stack entries held in LDS */
#define RT_DEV_MAX_BVH_DEPTH 32 /* builder guarantee: inner-node depth <= this, so the stack cannot overflow */

#define RT_PRIM_MISS 0xFFFFFFFFu
#define RT_PRIM_SPHERE_FLAG 0x80000000u

struct DevNode { // 64 bytes, 64-byte aligned
    float c0_min[3];
    uint32_t child0; // inner: node index; leaf: RT_DEV_LEAF_FLAG | count << 27 | start
    float c0_max[3];
    uint32_t child1;
    float c1_min[3];
    uint32_t _pad0;
    float c1_max[3];
    uint32_t _pad1;
};

struct DevTri { // 48 bytes, 16-byte aligned
    float v0[3];
    float e1[3];
    float e2[3];
    uint32_t material_id;
    uint32_t prim_id; // index in the caller's triangle array
    uint32_t _pad;
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

struct DevLight { // 48 bytes: the 11 words of Light the kernel reads (shader/src/scene_access.rs:60-107)
    float position[3];
    uint32_t light_type;
    float color[3];
    float intensity;
    float direction[3];
    uint32_t _pad;
};

struct DevScene {
    const DevNode* nodes;
    const DevTri* tris;
    const DevSphere* spheres;
    const DevLight* lights;
    const DevMaterial* materials;
    uint32_t n_nodes;
    uint32_t n_tris;
    uint32_t n_spheres;
    uint32_t n_lights;
    uint32_t n_materials; // `materials.len()` of shader/src/lib.rs:307 := material_count
    uint32_t root_ref;    // child reference of the root (a leaf reference for tiny scenes)
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
