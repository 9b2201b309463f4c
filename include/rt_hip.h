/*
 * rt_hip.h — C ABI of librt_hip.so, the MI355X (gfx950) replacement for the
 * reference's device kernel `main_cs` and the wgpu plumbing around it.
 *
 * The reference has no FFI layer; the seam this ABI replaces is
 *   - src/buffers.rs:157-470  (BufferManager::update_*: host Vec<T> -> GPU buffers,
 *                              bindings 1-5 of src/renderer.rs:250-341),
 *   - src/compute.rs:137-251  (execute_compute_pass / process_tile /
 *                              process_color_channel: push constants + dispatch),
 *   - shader/src/lib.rs:25-89 (main_cs, what each dispatch computes),
 *   - shader/src/lib.rs:367-391 (main_fs, the 3-texture combine).
 * Each entry point below names the reference interface it stands in for.
 *
 * Conventions
 *   - Plain C: pointers + sizes, the Pod structs of rt_shared.h by pointer.
 *   - Every input pointer is borrowed for the duration of the call and copied
 *     (as queue.write_buffer does, src/buffers.rs:236-240); outputs go to
 *     caller-allocated memory; the context owns all device memory.
 *   - Return value: 0 = RT_OK, negative = error class; text via rt_last_error.
 *     Nothing throws or unwinds across this boundary.
 *   - A context is single-caller (not re-entrant), as the reference's
 *     RenderState is only touched from the winit thread (src/main.rs:239-292).
 *   - rt_render is synchronous: it returns after the device finished, so timing
 *     is well defined.  rt_dispatch_tile returns after the launch, as
 *     queue.submit does (src/compute.rs:165: the reference never waits); the
 *     next call that needs the result or an idle device (rt_read_*, rt_get_stats,
 *     rt_upload_*, rt_render, rt_destroy) waits for it, and rt_get_stats then
 *     reports the kernel time of the last dispatch.
 *   - There is no CPU fallback: without a HIP device every compute entry
 *     point fails with RT_ERR_HIP.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include "rt_shared.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_BAD_ARG (-1)
#define RT_ERR_OOM (-2)
#define RT_ERR_HIP (-3)
#define RT_ERR_NOT_UPLOADED (-4)
#define RT_ERR_INTERNAL (-5)

/* Render modes of rt_render. */
#define RT_MODE_LEGACY 0u    /* shader/src/lib.rs:58-79: one pixel-centre ray, miss = black           */
#define RT_MODE_WAVEFRONT 1u /* shader/src/lib.rs:92-149 + wavefront.rs:116-165: same, miss = sky      */
#define RT_MODE_EXTENDED 2u  /* jittered spp + real bounces built on the reference's declared-but-stub
                                wavefront API (SimpleRng, WavefrontRay types, russian roulette);
                                no reference implementation exists: see DESIGN.md "extended mode"      */

/* rt_render_params.flags */
#define RT_FLAG_COUNTERS 1u   /* run the counting variant of the kernel: fills node_visits / tri_tests  */
#define RT_FLAG_NO_SHADOWS 2u /* extended mode only: lights are not gated by shadow rays (as in the     */
                              /* reference, which never traces them)                                    */
#define RT_FLAG_KERNEL_V1 4u  /* extended mode only: run the nested-loop megakernel (v1) instead of the   */
                              /* wavefront pipeline (A/B measurements; same results)                    */
#define RT_FLAG_KERNEL_SM 8u  /* extended mode only: run the state-machine megakernel (v2) instead      */
#define RT_FLAG_NO_SHADOW_GRID 16u /* extended mode only: every shadow segment walks the BVH instead of its  */
                              /* light's triangle lists (A/B measurements and tests; same results)      */

#define RT_FLAG_KERNEL_PIPELINE 32u /* extended mode only: always the queue pipeline, also for the frames that take the one-pass kernel */
                              /* by rule (max_bounces 0 over a tiny tree; tests and A/B measurements; same results)        */

#define RT_FLAG_NO_BEAMS 64u  /* extended mode only: camera segments walk the tree like every other segment instead of testing their   */
                              /* pixel block's leaf list (A/B measurements and tests; same results)                                 */

#define RT_FLAG_STAGE_TIMES 128u /* extended mode only: time every launch of the frame's dominant stage kernel with HIP events on its stream  */
                              /* (bench.py's roofline of that kernel; read with rt_debug_stage_times)                                   */

typedef struct rt_ctx rt_ctx;

typedef struct rt_render_params {
    rt_camera camera;
    uint32_t width, height;
    uint32_t spp;         /* samples per pixel; modes 0/1 have no spp in the reference: they trace   */
                          /* the single pixel-centre ray and ignore this field                       */
    uint32_t max_bounces; /* mode 1: max_bounce_depth of pack_flags; mode 2: continuation segments,  */
                          /* at most RT_MAX_BOUNCES (RT_ERR_BAD_ARG beyond)                          */
    uint32_t mode;        /* RT_MODE_*                                                               */
    uint32_t frame_seed;  /* PushConstants::frame_seed (shared/src/lib.rs:226)                       */
    uint32_t tile_size;   /* 0 = RT_TILE_SIZE (128). Tile grid = TileHelper::calculate_tile_count    */
    uint32_t tile_rank;   /* this context renders tiles with (row-major index % tile_world) ==       */
    uint32_t tile_world;  /* tile_rank; 0/1 = all tiles. Used by one-process-per-GPU launches.       */
                          /* A context over several devices splits those tiles among its devices.   */
    uint32_t flags;       /* RT_FLAG_*                                                               */
} rt_render_params;

typedef struct rt_stats {
    uint64_t rays;         /* ray segments traced by the last render/dispatch (primary + continuation + shadow) */
    uint64_t primary_rays; /* camera segments among them                                             */
    uint64_t continuation_rays; /* bounce segments (extended mode)                                   */
    uint64_t shadow_rays;  /* shadow segments (extended mode)                                        */
    uint64_t pixels;       /* pixels written                                                          */
    uint64_t node_visits;  /* BVH nodes fetched   (only with RT_FLAG_COUNTERS, else 0)                */
    uint64_t tri_tests;    /* triangle records fetched and tested (only with RT_FLAG_COUNTERS)        */
    double kernel_ms;      /* HIP-event time of the trace kernel(s) on the launch stream              */
    double wall_ms;        /* host wall time of the call                                              */
    uint64_t node_bytes;   /* bytes of one BVH node record in the device layout                       */
    uint64_t tri_bytes;    /* bytes of one triangle record in the device layout                       */
    uint64_t scene_bytes;  /* device bytes held by the scene (nodes + triangles + materials + lights) */
    uint32_t bvh_nodes;    /* nodes in the device BVH                                                 */
    uint32_t bvh_depth;    /* its depth                                                               */
    uint32_t n_devices;
    uint32_t flags;        /* RT_STAT_* of the last render                                            */
    uint64_t texture_bytes; /* bytes of texture data last handed to rt_upload_textures (never sampled) */
    uint32_t n_textures;   /* TextureInfo records last handed to rt_upload_textures                   */
    uint32_t tree_build;   /* how the tree in use was built: 2 on the device (rt_upload_scene*, milliseconds), 0 by the host builder  */
                           /* (tiny scenes, the fallback for degenerate input, rt_prepare RT_PREPARE_QUALITY_TREE), 1 host PLOC (development) */
    uint64_t grid_bytes;   /* device bytes of the per-light triangle lists ("light grids") of the extended mode's shadow stage;   */
                           /* 0 until an extended-mode frame (or rt_prepare) has built them: rt_upload_scene* builds none          */
    double grid_build_ms;  /* host wall time that build took (once per uploaded scene)                                            */
} rt_stats;

/* rt_stats.flags */
#define RT_STAT_MEGAKERNEL_FALLBACK 1u /* extended mode: the frame did not fit the queue pipeline (more than 32 lights: one
                                          visibility bit per light; or a device share beyond the addressable path slots) and
                                          was rendered by the state-machine megakernel: same image, about 3x slower */

#define RT_STAT_SINGLE_PASS 2u /* extended mode: primary rays only (max_bounces 0) over a tiny tree: rendered by the one-pass kernel that
                                  keeps a pixel's samples in registers and stores it once (same image as the pipeline, several times faster) */

#define RT_MAX_BOUNCES 255u /* extended mode: bounce depths travel in 8 bits, as in pack_flags (shared/src/lib.rs:1154-1179) */

/* Create a context on `n_devices` HIP devices (ids in device_ids; NULL = device 0..n-1).
 * Replaces RenderState::new's adapter/device acquisition (src/renderer.rs:93-125).
 * With n_devices > 1 the scene is replicated and tiles are interleaved over the devices. */
int rt_create(rt_ctx** out, const int* device_ids, int n_devices);

/* Upload a scene from the host Vecs of SceneState (src/scene.rs:8-18).
 * Replaces BufferManager::update_scene_metadata / update_triangles / update_materials
 * (src/buffers.rs:157-377).  ref_nodes / ref_tri_indices (the output of
 * BvhBuilder::build, src/bvh.rs:104-122) are optional and only validated: the
 * library always builds its own acceleration structure, closest-hit results do
 * not depend on BVH topology. */
int rt_upload_scene(rt_ctx* ctx,
                    const rt_sphere* spheres, uint32_t n_spheres,
                    const rt_light* lights, uint32_t n_lights,
                    const rt_vertex* vertices, uint32_t n_vertices,
                    const rt_triangle* triangles, uint32_t n_triangles,
                    const rt_material* materials, uint32_t n_materials,
                    const rt_bvh_node* ref_nodes, uint32_t n_ref_nodes,
                    const uint32_t* ref_tri_indices, uint32_t n_ref_tri_indices);

/* Upload byte-for-byte what bindings 1-5 carry (src/renderer.rs:250-341):
 * binding 1 = scene_metadata [spheres|lights|bvh_nodes|triangle_indices|vertices] with
 * offsets in u32 words (src/buffers.rs:213-268), bindings 2-4 = Triangle buffers split
 * every triangles_per_buffer (src/buffers.rs:274-336), binding 5 = materials. */
int rt_upload_scene_packed(rt_ctx* ctx,
                           const uint32_t* scene_metadata, size_t n_u32,
                           const rt_scene_metadata_offsets* offsets,
                           const rt_triangle* const tri_buffers[3], const uint32_t tri_counts[3],
                           uint32_t triangles_per_buffer,
                           const rt_material* materials, uint32_t n_materials);

/* Hand over what bindings 6-7 carry (src/renderer.rs:250-341): the TextureInfo array of
 * BufferManager::update_textures (src/buffers.rs:381-419) and the texture bytes of update_texture_data
 * (src/buffers.rs:422-470, which packs them four to a u32).  main_cs binds both and reads neither
 * (shader/src/lib.rs:34-35), so they are validated (every texture inside the data) and recorded in rt_stats, nothing
 * else: a host that keeps its update_* call sequence maps one to one.  Independent of rt_upload_scene*. */
int rt_upload_textures(rt_ctx* ctx, const rt_texture_info* textures, uint32_t n_textures,
                       const uint8_t* texture_data, size_t n_bytes);

/* Optional: pay now what rt_render would otherwise pay on the first frame that needs it.  RT_PREPARE_SHADOW_GRIDS: the per-light
 * triangle lists of the extended mode's shadow stage (rt_stats.grid_bytes / grid_build_ms; 45-50 ms and 7.5 GB for a 262k-triangle
 * scene with five lights).  A host that only renders the reference's modes 0/1 (src/compute.rs:12-50) never calls this and never
 * pays: rt_upload_scene* builds only the tree.  No reference counterpart (the reference traces no shadow segments). */
#define RT_PREPARE_SHADOW_GRIDS 1u
/* RT_PREPARE_QUALITY_TREE: for scenes that stay - rebuild the acceleration structure with the host builder (binned SAH + insertion-based
 * optimisation, 0.35 s per 262 k triangles, 4.8 s for 3.8 M) in place of the tree rt_upload_scene* built on the device in milliseconds:
 * frames 2 % (sponza-like) to 9 % (bistro-like) faster, same images.  rt_stats.tree_build tells which tree is in use. */
#define RT_PREPARE_QUALITY_TREE 2u
int rt_prepare(rt_ctx* ctx, uint32_t what);

/* Render a whole frame (all tiles of this context's share, all three colour channels in
 * one pass).  Replaces ComputeRenderer::run_compute's tile x channel loop
 * (src/compute.rs:12-50, 137-251). */
int rt_render(rt_ctx* ctx, const rt_render_params* params);

/* Exact analogue of ONE process_color_channel dispatch (src/compute.rs:212-251): renders
 * the tile named by the push constants into the rgba8 texture of channel
 * pc->packed_flags & 0xFF (0..2; anything else is RT_ERR_BAD_ARG, as
 * get_compute_bind_group fails, src/renderer.rs:769-776).  Scene counts inside
 * pc->metadata_offsets are ignored in favour of the uploaded scene.  Asynchronous (see above). */
int rt_dispatch_tile(rt_ctx* ctx, const rt_push_constants* pc);

/* Read back the float RGB framebuffer of the last rt_render: width*height*3 floats, row-major, y down. */
int rt_read_rgb32f(rt_ctx* ctx, float* out, size_t n_floats);

/* Read back the three Rgba8Unorm channel textures (src/renderer.rs:452-475): width*height*4 bytes each. */
int rt_read_rgba8_channels(rt_ctx* ctx, uint8_t* red, uint8_t* green, uint8_t* blue, size_t n_bytes_each);

/* main_fs combine (shader/src/lib.rs:383-388): (red.x, green.y, blue.z, 255); width*height*4 bytes. */
int rt_read_rgba8_combined(rt_ctx* ctx, uint8_t* out, size_t n_bytes);

/* Per-pixel closest-hit record of the last mode-0/1 rt_render: original triangle index
 * (0xFFFFFFFF = miss, 0x80000000|i = sphere i) and hit distance t.  Index parity hook. */
int rt_read_hits(rt_ctx* ctx, uint32_t* prim_ids, float* t, size_t n_pixels);

int rt_get_stats(rt_ctx* ctx, rt_stats* out);

/* Last error text of this context (or of the failed rt_create when ctx is NULL). */
const char* rt_last_error(rt_ctx* ctx);

void rt_destroy(rt_ctx* ctx);

/* Library build info: "gfx950 strict|fast ..." */
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
