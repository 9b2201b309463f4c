/*
 * rt_oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * The oracle is a scalar C++ restatement of the reference's device kernel
 * (the .rs files under shader/src) written from the source text, consuming the reference's
 * own buffer layout (binding 1 combined u32 metadata buffer, bindings 2-4
 * Triangle buffers, binding 5 materials, 128-byte push constants).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing under gpu_raytracer_amd/ links, imports or calls it.
 *
 * PARITY PINNING: the reference ships no golden vectors, fixtures or tests for
 * this path and cannot be built or run offline, so the
 * oracle is pinned by (a) the reference's host-side unit tests that touch the
 * contract (restated in tests/test_contract.py) and (b) hand-derived
 * known-answer tests K1-K7 of SURVEY.md §8c (tests/test_oracle_kat.py).
 * For ray-gen / traversal / intersection / shading there are no reference
 * vectors: that part is "parity unpinned by reference vectors, pinned by
 * source reading".
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_shared.h"

#ifdef __cplusplus
extern "C" {
#endif

/* What the kernel's 8 bindings see (shader/src/lib.rs:28-36). `*_len` are the
 * `.len()` of the bound slices, i.e. buffer capacities in elements. */
typedef struct oracle_bindings {
    const uint32_t* scene_metadata; /* binding 1 */
    uint64_t scene_metadata_len;    /* in u32 words */
    const rt_triangle* triangles[3]; /* bindings 2-4 */
    uint64_t triangles_len[3];
    const rt_material* materials;   /* binding 5 */
    uint64_t materials_len;
} oracle_bindings;

typedef struct oracle_counters {
    uint64_t rays;        /* ray segments traced */
    uint64_t node_visits; /* nodes popped and bounds-tested (shader/src/bvh.rs:40-55) */
    uint64_t tri_tests;   /* triangles fetched and tested (shader/src/bvh.rs:108-130) */
    uint64_t sphere_tests;
    uint64_t stack_drops; /* pushes dropped at stack_ptr >= 63 (shader/src/bvh.rs:76-83) */
    uint64_t oob_reads;   /* out-of-range buffer reads (returned 0, as robust buffer access does) */
} oracle_counters;

/* One dispatch of main_cs over one tile and one colour channel, exactly as
 * src/compute.rs:212-251 issues it: every invocation id in
 * [0, ceil(tw/16)*16) x [0, ceil(th/16)*16) runs main_cs (shader/src/lib.rs:25-89).
 * `image` is the bound Rgba8Unorm storage texture (img_w x img_h x 4 bytes). */
int oracle_dispatch(const oracle_bindings* b, const rt_push_constants* pc,
                    uint8_t* image, uint32_t img_w, uint32_t img_h,
                    oracle_counters* counters /* nullable, accumulated */);

/* Whole frame = the tile x channel loop of src/compute.rs:137-191, with `threads`
 * host threads dealing tiles from an atomic counter.
 *   base_pc      : push constants of any tile (tile fields and channel are overwritten)
 *   rgba8[3]     : red / green / blue channel textures (each w*h*4), nullable
 *   rgb32f       : w*h*3 floats, component c taken from the channel-c pass, nullable
 *   prim_ids, ts : per-pixel closest hit (original triangle index, 0x80000000|i for
 *                  sphere i, 0xFFFFFFFF miss) and distance, nullable
 *   faithful3    : 1 = trace three times, once per channel, as the reference does;
 *                  0 = trace once and shade three times (bit-identical, 3x cheaper) */
int oracle_render_frame(const oracle_bindings* b, const rt_push_constants* base_pc,
                        uint32_t tile_size, int threads, int faithful3,
                        uint8_t* rgba8_red, uint8_t* rgba8_green, uint8_t* rgba8_blue,
                        float* rgb32f, uint32_t* prim_ids, float* ts,
                        oracle_counters* counters);

/* Extended mode (RT_MODE_EXTENDED of rt_hip.h): the CPU statement of the build's own path tracer
 * (jittered samples, shadow rays, real bounces) built on the reference's declared-but-stub
 * wavefront API.  NO reference implementation exists (generate_continuation_rays is a stub,
 * shader/src/wavefront.rs:340-355), so this is the SPECIFICATION the HIP kernel is checked
 * against, not a restatement: parity for this mode is "unpinned".  It is anchored to the pinned
 * part by construction: with spp = 1, max_bounces = 0 and ORACLE_EXT_NO_SHADOWS it reproduces
 * mode 1 bit for bit (tests/test_oracle_extended.py).  Full rules in DESIGN.md "Extended mode".
 *   segments[4]: camera, continuation, shadow segments traced, and paths ended by russian roulette */
#define ORACLE_EXT_NO_SHADOWS 2u /* same bit as RT_FLAG_NO_SHADOWS */
int oracle_render_extended(const oracle_bindings* b, const rt_push_constants* base_pc,
                           uint32_t spp, uint32_t max_bounces, uint32_t flags, int threads,
                           float* rgb32f, uint64_t* segments, oracle_counters* counters);
/* The same for the pixel rectangle [x0, x0+rw) x [y0, y0+rh) of the frame whose size base_pc names: rgb32f holds rw*rh*3
 * floats.  A pixel's samples depend only on its coordinates in the full frame, so a crop of a 3840x2160 frame can be
 * checked without rendering the whole of it on the CPU. */
int oracle_render_extended_region(const oracle_bindings* b, const rt_push_constants* base_pc,
                                  uint32_t spp, uint32_t max_bounces, uint32_t flags, int threads,
                                  uint32_t x0, uint32_t y0, uint32_t rw, uint32_t rh,
                                  float* rgb32f, uint64_t* segments, oracle_counters* counters);

/* Restatement of BvhBuilder::build (src/bvh.rs:104-122): empty scene -> one empty
 * leaf (:105-114); > 100,000 triangles -> chunked mesh-order leaves + bottom-up
 * pairing (:154-247, exact); otherwise one triangle per leaf, pre-order flattened
 * (:278-374) over a topology the absent `bvh` crate would have chosen — here a
 * median/SAH split of our own (topology unpinned, image independent of it).
 * Call with nodes == NULL to get the counts. */
int oracle_build_bvh(const rt_triangle* tris, uint32_t n_tris, const rt_vertex* verts, uint32_t n_verts,
                     rt_bvh_node* nodes, uint32_t* n_nodes, uint32_t* tri_indices, uint32_t* n_indices);

/* Baseline flavour (ii) of SURVEY 8(d) — NOT the reference's algorithm, used for a second CPU timing only: one
 * triangle per leaf at every scene size, and an ordered, distance-culled traversal (same image: checked by the tests). */
int oracle_build_bvh_per_triangle(const rt_triangle* tris, uint32_t n_tris, const rt_vertex* verts, uint32_t n_verts,
                                  rt_bvh_node* nodes, uint32_t* n_nodes, uint32_t* tri_indices, uint32_t* n_indices);
void oracle_set_fast_traversal(int on);

/* f16 helpers used by the restatement (round-to-nearest-even), exposed for tests. */
uint16_t oracle_f32_to_f16(float v);
float oracle_f16_to_f32(uint16_t h);

#ifdef __cplusplus
}
#endif
#endif
