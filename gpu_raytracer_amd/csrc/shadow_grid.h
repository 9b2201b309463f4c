// shadow_grid.h — per-light triangle lists for the shadow segments of the wavefront pipeline.
//
// Every shadow segment of the extended mode ends at a light (lighting.rs:96-139: directional, point and spot lights; there
// are no area lights), so for ONE light all segments lie on lines through one point (or along one direction).  Seen from
// the light a segment is a point: it can only be stopped by triangles whose projection covers that point.  Per light the
// scene is therefore rasterised ONCE, at upload, into a grid over the light's directions - a cube map about a point or
// spot light, an orthographic grid across a directional one - each cell holding the triangles whose (conservatively
// dilated) projection touches it, sorted by distance from the light.  A shadow segment then needs no tree: it looks up
// its cell and runs the reference's Möller-Trumbore test (device_common.h test_triangle, the same arithmetic the leaves
// of the BVH run) over the cell's list, nearest to the light first, until a triangle is accepted or the list reaches
// triangles beyond the segment's own end.  The answer is the BVH's answer - "is any triangle accepted" does not depend on
// which superset of the accepted triangles is tested - provided the lists ARE supersets: the rasteriser dilates every
// triangle by the cone of directions its possible hit points subtend plus a fixed margin that covers all rounding
// (shadow_grid.hip), triangles too close to the light for that go to a "near" list every segment of the light tests, and
// a light whose near list would be long (or whose geometry is not finite) simply has no grid.  Cells with more than
// `heavy` triangles (foliage-like clutter seen end-on) and lights without a grid hand their segments on to the BVH
// traversal kernel, and so do segments still undecided after RT_SG_SORTED_PREFIX - 1 entries; what a grid stores of a list is what such
// a walk can look at.  A grid only pays when its lists are not too long: the build measures them and refuses the grid otherwise
// (shadow_grid.hip), and a scene keeps its grids only when every light that casts shadows has one (rt_api.cpp).  tests/test_gpu_shadow_grid.py renders every kind of scene and
// light placement with the lists and with RT_FLAG_NO_SHADOW_GRID and asks for equal bits; the oracle parity tests run on the lists.
#ifndef RT_SHADOW_GRID_H
#define RT_SHADOW_GRID_H

#include <hip/hip_runtime.h>

#include "device_layout.h"

struct DevShadowGrid { // one per light, 128 bytes
    uint32_t kind;     // 0: no grid (the BVH decides), 1: cube map about `origin`, 2: orthographic grid across axis_w
    uint32_t res;      // cells per side of a face
    uint32_t n_cells;  // faces * res * res; the pseudo-cell n_cells is the near list
    uint32_t heavy;    // cells with more entries than this are left to the BVH
    float origin[3];   // kind 1: the light's position
    float scale;       // kind 1: res / 2 (cells per unit of u = x / z);  kind 2: cells per world unit
    float axis_u[3];   // kind 2: the grid's axes (unit vectors across the light's direction) ...
    float u0;          // ... and the coordinates of its corner
    float axis_v[3];
    float v0;
    float axis_w[3];   // kind 2: the direction TOWARD the light (DevLight::neg_ndir)
    float key_top;     // kind 2: a triangle's key is key_top - (largest coordinate along axis_w)
    float limit_margin; // a segment tests entries whose key is below (its own key + limit_margin)
    uint32_t near_begin, near_end; // the near list: entries [near_begin, near_end) of `overflow`
    uint32_t _pad;
    // Cell c owns the 128-byte block blocks[8 c .. 8 c + 8) - ONE line of the L2, which is what a scattered read moves from HBM:
    //   quad 0      {entries in the cell's list, where its 3rd and later entries start in `overflow`, key of the 3rd entry (+inf if none), 0}
    //   quads 1-3   the list's first entry, quads 4-6 its second, quad 7 unused
    // An entry is 48 bytes, {key, v0.xyz} {e1.xyz, e2.x} {e2.yz, triangle record, 0}: the triangle travels with its key (f32; kind 1:
    // distance light - triangle, kind 2: see key_top).  Lists are sorted by key, a segment reads 2.2 entries on average (headline
    // scene), so most segments are decided by their cell's one line; the stage is bound by HBM bandwidth (5.6 TB/s measured with
    // separate offset and entry arrays, 335 bytes per segment), so lines per segment are what counts (scripts/micro/random_read.hip).
    const uint4* blocks;
    const uint4* overflow;
    uint32_t _pad2[4];
};

static_assert(sizeof(DevShadowGrid) == 128, "DevShadowGrid is staged in LDS and uploaded as an array");

#define RT_SG_EXT_EPS 0.001f /* the origin offset of a shadow segment the lists' dilation is derived from: device_common.h EXT_EPS (checked in wavefront.hip) */
#define RT_SG_KIND_NONE 0u
#define RT_SG_KIND_CUBE 1u
#define RT_SG_KIND_ORTHO 2u
#define RT_SG_ENTRY_QUADS 3u
#define RT_SG_BLOCK_QUADS 8u   /* 128 bytes per cell */
#define RT_SG_BLOCK_ENTRIES 2u /* entries held in the cell's own block */
#ifndef RT_SG_SORTED_PREFIX
#define RT_SG_SORTED_PREFIX 32u /* a list's first 32 entries are its 32 nearest in order; later ones follow unordered (a walk gives up before them) */
#endif

namespace rt {

struct ShadowGridOptions {
    uint32_t res_point = 1024;   // cells per side of a cube face
    uint32_t res_dir = 2048;     // cells per side of an orthographic grid
    uint32_t heavy = 128;        // longest list a segment walks itself (its nearest RT_SG_SORTED_PREFIX - 1 entries at most); 64 until the end of round 3
    uint64_t max_entries = 400ull << 20; // per light; beyond it the light gets no grid
    uint64_t max_bytes = ~0ull;  // per light, cell blocks + list entries; a grid that would take more is refused (before anything of it is allocated)
    double max_mean_list = 32.0; // entries per filled cell ... (9 and 1 % until the end of round 3: no grids for cluttered scenes then)
    double max_heavy_share = 0.05; // ... and share of cells over `heavy` beyond which a grid does not pay (shadow_grid.hip)
};

struct ShadowGridBuild {
    DevShadowGrid grid{};      // kind 0 when the light gets no grid (nothing allocated then)
    void* blocks = nullptr;    // device allocations behind grid.blocks / grid.overflow, owned by the caller
    void* overflow = nullptr;
    uint64_t bytes = 0;        // of the two
    uint64_t n_entries = 0;
    uint32_t near_count = 0;
    uint32_t longest = 0;      // longest cell list
    uint32_t heavy_cells = 0;  // cells left to the BVH
    uint32_t filled_cells = 0; // cells with a list (n_entries, heavy_cells, filled_cells are also set when the grid is refused)
};

// Rasterises the triangle records `d_tris` (device memory, leaf order, RT_DEV_LEAF_STRIDE records per leaf) for one light on the
// current device.  lo / hi: the bounding box of the finite triangles (host).  Synchronises the stream.
hipError_t shadow_grid_build(const DevTri* d_tris, uint32_t n_records, const DevLight& light, const float lo[3], const float hi[3],
                             const ShadowGridOptions& opt, hipStream_t stream, ShadowGridBuild* out);

} // namespace rt
#endif
