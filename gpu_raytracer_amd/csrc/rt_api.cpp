// rt_api.cpp — the C ABI of include/rt_hip.h: context, scene upload, frame / tile
// dispatch, read-back.  Replaces the reference's wgpu plumbing
// (src/buffers.rs:157-470 uploads, src/compute.rs:137-251 dispatch loop,
// src/renderer.rs:452-475 channel textures).  No CPU fallback: every compute entry point
// needs a HIP device and fails with RT_ERR_HIP otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "bvh_builder.h"
#include "bvh_check.h"
#include "device_build.h"
#include "device_layout.h"
#include "half.h"
#include "kernels.h"
#include "shadow_grid.h"
#include "wavefront.h"

#ifndef RT_SINGLE_PASS_MAX_NODES
#define RT_SINGLE_PASS_MAX_NODES 64u /* "BVH depth <= 4" scenes of BASELINE configs[1]; larger scenes: measured in profiles/ab_r03.json */
#endif
#define RT_DEVICE_BUILD_MIN_TRIS 1024u /* below this the host build takes well under a millisecond */

namespace {

thread_local std::string g_create_error;

struct DeviceState {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr; // the second lane of the wavefront pipeline (batches alternate between the two)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_start = nullptr, ev_join = nullptr, ev_res[2] = {nullptr, nullptr};
    DevNode8* nodes = nullptr;
    DevTri* tris = nullptr;
    DevSphere* spheres = nullptr;
    DevLight* lights = nullptr;
    DevMaterial* materials = nullptr;
    float* rgba32f = nullptr;
    uint8_t* chan[3] = {nullptr, nullptr, nullptr};
    uint32_t* prim_id = nullptr;
    float* hit_t = nullptr;
    unsigned long long* counters = nullptr;
    uint32_t fb_w = 0, fb_h = 0;
    void* readback_dev = nullptr;  // epilogue output (combined rgba8 / packed rgb32f), whole-frame single-device reads
    void* readback_host = nullptr; // pinned staging for it
    size_t readback_bytes = 0;
    uint32_t tile_first = 0, tile_stride = 1, n_owned = 0; // of the last rt_render
    DevShadowGrid* grids = nullptr;                        // one per light (shadow_grid.h), null when no light has a grid
    std::vector<void*> grid_allocs;
    std::vector<rt::ShadowGridBuild> grid_info;
    bool grids_tried = false;                              // the light grids of the current scene were built (or refused) on this device
    bool grids_partial = false;                            // ... and a light that has triangles to cast shadows was left without one: then no light keeps its grid
    rt::WfBuffers wf{};                                    // wavefront pipeline state (extended mode)
    rt::WfBuffers wf2{};                                   // ... of the second lane (its own path state, queues and counters; the pixel sums are shared)
    std::vector<void*> wf2_allocs;
    bool used_wavefront = false;
    bool used_two_lanes = false;
    uint32_t wf_lights = 0;
    uint32_t wf_spp = 0; // spp the current wavefront allocation was sized for
    std::vector<hipEvent_t> stage_events; // RT_FLAG_STAGE_TIMES: event pairs around the launches of the dominant stage kernel
    uint32_t stage_events_used = 0;
    bool wf_beams_off = false; // this frame walks the tree for its camera segments too (RT_FLAG_NO_BEAMS)
    std::vector<void*> wf_allocs;
};

} // namespace

struct rt_ctx {
    std::vector<DeviceState> devs;
    std::string err;
    bool uploaded = false;
    bool pending_dispatch = false; // rt_dispatch_tile launches are not waited for (the reference's queue.submit is not either); see sync_pending
    DevScene scene_counts{}; // counts; pointers are per device
    rt_stats stats{};
    uint32_t frame_w = 0, frame_h = 0, frame_tile = RT_TILE_SIZE, frame_tiles_x = 0, frame_tiles_y = 0;
    bool frame_valid = false;
    unsigned long long diag[8] = {0}; // diagnostics of the counting kernel variant (rt_debug_counters)
    unsigned long long grid_diag[2] = {0}; // ... of the light grids: shadow segments they answered, list entries read
    double stage_ms[2] = {0.0, 0.0};       // RT_FLAG_STAGE_TIMES: [0] sum of the k_wf_shadow_grid launch durations of the last frame (device 0), [1] launches
    int fail_upload_at = -1;          // test hook: the next scene upload fails before its k-th device array (rt_debug_fail_upload)
    uint32_t n_input_tris = 0;        // triangles handed to the last scene upload (prim ids are < this)
    int build_method = 0;             // how its tree was built: 0 host SAH, 1 host PLOC, 2 device PLOC
    uint32_t tree_tris_uploaded = 0;  // triangle records of the tree the devices hold (scene_bytes accounting across rt_prepare)
    std::vector<rt::BuildTri> build_tris; // the triangles of the last upload as the builders take them (rt_prepare RT_PREPARE_QUALITY_TREE rebuilds from them)
    std::vector<DevLight> host_lights; // what the lazy light-grid build needs of the last upload: the lights, ...
    float box_lo[3] = {0, 0, 0}, box_hi[3] = {0, 0, 0}; // ... the box of the triangles with finite vertices
    uint32_t n_textures = 0;          // bindings 6-7 as last handed over (rt_upload_textures); never sampled, like the reference
    uint64_t texture_bytes = 0;

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

namespace {

#ifndef RT_WF_LANES_DEFAULT
#define RT_WF_LANES_DEFAULT 2 /* measured on the headline frame: 179.7 - 188.1 ms against 197.6 - 199.0 with one lane; 16 spp: 48.8 against 51.9 */
#endif

#define HIPCHK(ctx, call)                                                                                       \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess)                                                                                   \
            return (ctx)->fail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                               hipGetErrorString(e_), __FILE__, __LINE__);                                     \
    } while (0)

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

void free_scene(DeviceState& d) {
    (void)hipSetDevice(d.device);
    (void)hipFree(d.nodes); (void)hipFree(d.tris); (void)hipFree(d.spheres); (void)hipFree(d.lights); (void)hipFree(d.materials);
    d.nodes = nullptr; d.tris = nullptr; d.spheres = nullptr; d.lights = nullptr; d.materials = nullptr;
    for (void* p : d.grid_allocs) (void)hipFree(p);
    d.grid_allocs.clear();
    d.grid_info.clear();
    d.grids = nullptr;
    d.grids_tried = false;
    d.grids_partial = false;
}
void free_targets(DeviceState& d) {
    (void)hipSetDevice(d.device);
    (void)hipFree(d.rgba32f); (void)hipFree(d.chan[0]); (void)hipFree(d.chan[1]); (void)hipFree(d.chan[2]); (void)hipFree(d.prim_id); (void)hipFree(d.hit_t);
    d.rgba32f = nullptr; d.chan[0] = d.chan[1] = d.chan[2] = nullptr; d.prim_id = nullptr; d.hit_t = nullptr;
    d.fb_w = d.fb_h = 0;
    (void)hipFree(d.readback_dev);
    if (d.readback_host) (void)hipHostFree(d.readback_host);
    d.readback_dev = d.readback_host = nullptr;
    d.readback_bytes = 0;
}

void free_wavefront(DeviceState& d) {
    (void)hipSetDevice(d.device);
    for (void* p : d.wf_allocs) (void)hipFree(p);
    d.wf_allocs.clear();
    for (void* p : d.wf2_allocs) (void)hipFree(p);
    d.wf2_allocs.clear();
    d.wf2 = rt::WfBuffers{};
    d.wf = rt::WfBuffers{};
    d.wf_lights = 0;
}

template <class T>
int upload_array(rt_ctx* ctx, T** dst, const std::vector<T>& src) {
    *dst = nullptr;
    if (src.empty()) return RT_OK;
    HIPCHK(ctx, hipMalloc((void**)dst, src.size() * sizeof(T)));
    HIPCHK(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

int ensure_targets(rt_ctx* ctx, DeviceState& d, uint32_t w, uint32_t h) {
    if (d.fb_w == w && d.fb_h == h && d.rgba32f) return RT_OK;
    if (d.stream) HIPCHK(ctx, hipStreamSynchronize(d.stream)); // dispatches in flight may still write the old targets
    free_targets(d);
    HIPCHK(ctx, hipSetDevice(d.device));
    size_t n = (size_t)w * h;
    HIPCHK(ctx, hipMalloc((void**)&d.rgba32f, n * 16));
    for (int c = 0; c < 3; c++) HIPCHK(ctx, hipMalloc((void**)&d.chan[c], n * 4));
    HIPCHK(ctx, hipMalloc((void**)&d.prim_id, n * 4));
    HIPCHK(ctx, hipMalloc((void**)&d.hit_t, n * 4));
    // fresh textures read as zero, like newly created wgpu textures.  On d.stream: it is a non-blocking stream, so a
    // memset on the null stream would not be ordered against the kernels that write these targets next
    HIPCHK(ctx, hipMemsetAsync(d.rgba32f, 0, n * 16, d.stream));
    for (int c = 0; c < 3; c++) HIPCHK(ctx, hipMemsetAsync(d.chan[c], 0, n * 4, d.stream));
    HIPCHK(ctx, hipMemsetAsync(d.prim_id, 0xFF, n * 4, d.stream));
    HIPCHK(ctx, hipMemsetAsync(d.hit_t, 0, n * 4, d.stream));
    d.fb_w = w;
    d.fb_h = h;
    return RT_OK;
}

// Per-frame camera constants in the reference's operation order (shader/src/ray.rs:33-44;
// wavefront.rs:87-98 for mode 1, which uses the f32 resolution directly).
DevCamera make_camera(const rt_camera& cam, float res_x, float res_y, bool wavefront) {
    DevCamera c;
    float wf, hf;
    if (wavefront) {
        wf = res_x;
        hf = res_y;
    } else { // `resolution[i] as u32` then `as f32` (ray.rs:23-24, 28-29, 33)
        auto as_u32 = [](float f) -> uint32_t { return !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)f); };
        wf = (float)as_u32(res_x);
        hf = (float)as_u32(res_y);
    }
    c.width_f = wf;
    c.height_f = hf;
    c.aspect = wf / hf;
    c.fov_scale = tanf(cam.fov * 0.5f * 3.14159265358979323846f / 180.0f);
    const float* f = cam.direction;
    const float* u = cam.up;
    // glam cross: (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
    float r[3] = {f[1] * u[2] - u[1] * f[2], f[2] * u[0] - u[2] * f[0], f[0] * u[1] - u[0] * f[1]};
    float t[3] = {r[1] * f[2] - f[1] * r[2], r[2] * f[0] - f[2] * r[0], r[0] * f[1] - f[0] * r[1]};
    for (int a = 0; a < 3; a++) {
        c.origin[a] = cam.position[a];
        c.forward[a] = f[a];
        c.right[a] = r[a];
        c.true_up[a] = t[a];
    }
    return c;
}

DevScene scene_for(const rt_ctx* ctx, const DeviceState& d) {
    DevScene s = ctx->scene_counts;
    s.nodes = d.nodes;
    s.tris = d.tris;
    s.spheres = d.spheres;
    s.lights = d.lights;
    s.materials = d.materials;
    return s;
}

DevTargets targets_for(const DeviceState& d) {
    DevTargets t;
    t.rgba32f = d.rgba32f;
    for (int c = 0; c < 3; c++) t.chan[c] = d.chan[c];
    t.prim_id = d.prim_id;
    t.hit_t = d.hit_t;
    t.counters = d.counters;
    return t;
}

int sync_pending(rt_ctx* ctx);

int upload_common(rt_ctx* ctx, const rt_sphere* spheres, uint32_t n_spheres, const rt_light* lights, uint32_t n_lights,
                  const rt_vertex* vertices, uint32_t n_vertices, const std::vector<rt_triangle>& tris,
                  const std::vector<uint32_t>& prim_ids, const rt_material* materials, uint32_t n_materials) {
    if (int rcp = sync_pending(ctx)) return rcp; // dispatches in flight still read the old scene
    if (tris.size() > RT_DEV_MAX_TRIS) return ctx->fail(RT_ERR_BAD_ARG, "too many triangles: %zu > %u", tris.size(), RT_DEV_MAX_TRIS);
    std::vector<rt::BuildTri> bt(tris.size());
    for (size_t i = 0; i < tris.size(); i++) {
        const rt_triangle& t = tris[i];
        if (t.v0_index >= n_vertices || t.v1_index >= n_vertices || t.v2_index >= n_vertices)
            return ctx->fail(RT_ERR_BAD_ARG, "triangle %zu references vertex out of range (%u,%u,%u >= %u)", i, t.v0_index,
                             t.v1_index, t.v2_index, n_vertices);
        std::memcpy(bt[i].v0, vertices[t.v0_index].position, 12);
        std::memcpy(bt[i].v1, vertices[t.v1_index].position, 12);
        std::memcpy(bt[i].v2, vertices[t.v2_index].position, 12);
        bt[i].material_id = t.material_id;
        bt[i].prim_id = prim_ids.empty() ? (uint32_t)i : prim_ids[i];
    }
    rt::BvhBuild bvh;
    rt::BvhBuildOptions opt;
    if (const char* e = std::getenv("RT_BVH_COST_TRAVERSE")) opt.cost_traverse = (float)std::atof(e); // tuning knobs (development)
    if (const char* e = std::getenv("RT_BVH_MAX_LEAF")) opt.max_leaf = (uint32_t)std::atoi(e);
    if (const char* e = std::getenv("RT_BVH8_COST_TRAVERSE")) opt.cost_traverse8 = (float)std::atof(e);
    if (const char* e = std::getenv("RT_BUILD_METHOD")) opt.method = std::atoi(e);
    if (const char* e = std::getenv("RT_BUILD_REINSERT")) opt.reinsert = std::atoi(e) != 0;
    if (const char* e = std::getenv("RT_PLOC_RADIUS")) opt.ploc_radius = (uint32_t)std::atoi(e);
    // Where the tree is built.  Default (method 2): ON the device (device_build.hip: Morton sort + PLOC + collapse + emission,
    // a few milliseconds), as soon as the scene is large enough for that to matter; the host builders remain for tiny scenes, as
    // the fallback when a device-built tree would be deeper than the kernels' stacks allow (degenerate inputs; the host build
    // bounds its depth), and on request: RT_BUILD_METHOD=0 binned SAH + insertion-based optimisation (1 % faster frames, 0.35 s
    // per 262 k triangles), 1 PLOC on the host (the statement the device build is checked against).
    int method = 2;
    if (const char* e = std::getenv("RT_BUILD_METHOD")) method = std::atoi(e);
    if (method == 2 && bt.size() < RT_DEVICE_BUILD_MIN_TRIS) method = 0;
    opt.method = method == 1 ? 1 : 0;
    std::vector<rt::DeviceBuild> dbuilds;
    uint32_t tree_nodes = 0, tree_tris = 0, tree_depth = 0;
    if (method == 2) {
        if (ctx->fail_upload_at == 0) {
            ctx->fail_upload_at = -1;
            ctx->uploaded = false;
            ctx->frame_valid = false;
            ctx->scene_counts = DevScene{};
            for (auto& d : ctx->devs) free_scene(d);
            return ctx->fail(RT_ERR_OOM, "rt_upload: allocation failure injected by rt_debug_fail_upload");
        }
        dbuilds.resize(ctx->devs.size());
        bool ok = true;
        for (size_t j = 0; j < ctx->devs.size() && ok; j++) {
            DeviceState& d = ctx->devs[j];
            HIPCHK(ctx, hipSetDevice(d.device));
            const hipError_t e = rt::device_build(bt.data(), bt.size(), opt, d.stream, &dbuilds[j]);
            ok = e == hipSuccess && dbuilds[j].depth <= RT_DEV_MAX_BVH_DEPTH && dbuilds[j].n_nodes <= RT_DEV_MAX_NODES &&
                 dbuilds[j].n_nodes == dbuilds[0].n_nodes && dbuilds[j].n_tris == dbuilds[0].n_tris;
        }
        if (!ok) { // fall back to the host build (depth bound, or the device ran out of memory for the temporaries)
            for (size_t j = 0; j < dbuilds.size(); j++) {
                (void)hipSetDevice(ctx->devs[j].device);
                (void)hipFree(dbuilds[j].nodes);
                (void)hipFree(dbuilds[j].tris);
            }
            dbuilds.clear();
            method = 0;
        } else {
            tree_nodes = dbuilds[0].n_nodes;
            tree_tris = dbuilds[0].n_tris;
            tree_depth = dbuilds[0].depth;
        }
    }
    if (method != 2) {
        rt::build_bvh(bt.data(), bt.size(), opt, bvh);
        if (bvh.depth > RT_DEV_MAX_BVH_DEPTH) return ctx->fail(RT_ERR_INTERNAL, "BVH depth %u exceeds the bound %d", bvh.depth, RT_DEV_MAX_BVH_DEPTH);
        if (bvh.nodes.size() > RT_DEV_MAX_NODES) return ctx->fail(RT_ERR_BAD_ARG, "scene needs %zu BVH nodes > %u", bvh.nodes.size(), RT_DEV_MAX_NODES);
        tree_nodes = (uint32_t)bvh.nodes.size();
        tree_tris = (uint32_t)bvh.tris.size();
        tree_depth = bvh.depth;
    }
    ctx->n_input_tris = (uint32_t)bt.size();
    ctx->build_method = method;

    std::vector<DevSphere> ds(n_spheres);
    for (uint32_t i = 0; i < n_spheres; i++) {
        std::memcpy(ds[i].center, spheres[i].center, 12);
        ds[i].radius = spheres[i].radius;
        ds[i].material_id = spheres[i].material_id;
        ds[i]._pad[0] = ds[i]._pad[1] = ds[i]._pad[2] = 0;
    }
    std::vector<DevLight> dl(n_lights);
    for (uint32_t i = 0; i < n_lights; i++) {
        std::memcpy(dl[i].position, lights[i].position, 12);
        dl[i].light_type = lights[i].light_type;
        std::memcpy(dl[i].color, lights[i].color, 12);
        dl[i].intensity = lights[i].intensity;
        std::memcpy(dl[i].direction, lights[i].direction, 12);
        dl[i]._pad = 0;
        { // -normalize(direction): glam scalar-math normalize = v * (1 / sqrt(dot(v, v))), dot = (x*x + y*y) + z*z; this file is built -ffp-contract=off
            const float* v = lights[i].direction;
            const volatile float len2 = (v[0] * v[0]) + (v[1] * v[1]) + (v[2] * v[2]);
            const volatile float inv = 1.0f / std::sqrt(len2);
            for (int a = 0; a < 3; a++) dl[i].neg_ndir[a] = -(v[a] * inv);
            dl[i]._pad2 = 0;
        }
    }
    std::vector<DevMaterial> dm(n_materials);
    for (uint32_t i = 0; i < n_materials; i++) { // MaterialEvaluator accessors, shader/src/material.rs:16-63
        const rt_material& m = materials[i];
        std::memcpy(dm[i].albedo, m.albedo, 12);
        std::memcpy(dm[i].emission, m.emission, 12);
        dm[i].metallic = rt::f16_bits_to_f32((uint16_t)(m.metallic_roughness_f16 & 0xFFFF));
        dm[i].roughness = rt::f16_bits_to_f32((uint16_t)(m.metallic_roughness_f16 >> 16));
        dm[i].ior = rt::f16_bits_to_f32((uint16_t)(m.ior_transmission_f16 & 0xFFFF));
        dm[i].transmission = rt::f16_bits_to_f32((uint16_t)(m.ior_transmission_f16 >> 16));
        dm[i]._pad[0] = dm[i]._pad[1] = 0.0f;
    }

    // From here on the old scene is gone: the context counts as "nothing uploaded" until EVERY device holds the whole new
    // scene, so a failure half way (out of memory on the triangles, or on the second device) leaves no stale counts
    // pointing at freed or partly filled arrays - the next rt_render answers RT_ERR_NOT_UPLOADED.
    ctx->uploaded = false;
    ctx->frame_valid = false;
    ctx->scene_counts = DevScene{};
    for (auto& d : ctx->devs) free_scene(d);
    float box_lo[3] = {INFINITY, INFINITY, INFINITY}, box_hi[3] = {-INFINITY, -INFINITY, -INFINITY}; // of the triangles with finite vertices
    for (const rt::BuildTri& t : bt) {
        const float* vs[3] = {t.v0, t.v1, t.v2};
        bool finite = true;
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) finite = finite && std::isfinite(vs[k][a]);
        if (!finite) continue;
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) box_lo[a] = std::min(box_lo[a], vs[k][a]), box_hi[a] = std::max(box_hi[a], vs[k][a]);
    }
    auto upload_all = [&](DeviceState& d) -> int {
        HIPCHK(ctx, hipSetDevice(d.device));
        int rc;
        int k = 0;
        auto hook = [&]() { return ctx->fail_upload_at >= 0 && ctx->fail_upload_at == k++; }; // test hook (rt_debug_fail_upload)
        const size_t j = (size_t)(&d - ctx->devs.data());
        if (hook()) return ctx->fail(RT_ERR_OOM, "rt_upload: allocation failure injected by rt_debug_fail_upload");
        if (!dbuilds.empty()) { // the tree was built on this device: take the arrays over
            d.nodes = dbuilds[j].nodes;
            dbuilds[j].nodes = nullptr;
        } else if ((rc = upload_array(ctx, &d.nodes, bvh.nodes)) != RT_OK) return rc;
        if (hook()) return ctx->fail(RT_ERR_OOM, "rt_upload: allocation failure injected by rt_debug_fail_upload");
        if (!dbuilds.empty()) {
            d.tris = dbuilds[j].tris;
            dbuilds[j].tris = nullptr;
        } else if ((rc = upload_array(ctx, &d.tris, bvh.tris)) != RT_OK) return rc;
        if (hook()) return ctx->fail(RT_ERR_OOM, "rt_upload: allocation failure injected by rt_debug_fail_upload");
        if ((rc = upload_array(ctx, &d.spheres, ds)) != RT_OK) return rc;
        if ((rc = upload_array(ctx, &d.lights, dl)) != RT_OK) return rc;
        if ((rc = upload_array(ctx, &d.materials, dm)) != RT_OK) return rc;
        // the copies went through the null stream and d.stream is non-blocking: make the order explicit
        HIPCHK(ctx, hipDeviceSynchronize());
        return RT_OK;
    };
    for (auto& d : ctx->devs) {
        const int rc = upload_all(d);
        if (rc != RT_OK) {
            ctx->fail_upload_at = -1;
            for (auto& e : ctx->devs) free_scene(e);
            for (size_t j = 0; j < dbuilds.size(); j++) { // device-built arrays not yet handed to a device state
                (void)hipSetDevice(ctx->devs[j].device);
                (void)hipFree(dbuilds[j].nodes);
                (void)hipFree(dbuilds[j].tris);
            }
            return rc;
        }
    }
    ctx->fail_upload_at = -1;
    DevScene& sc = ctx->scene_counts;
    sc = DevScene{};
    sc.n_nodes = tree_nodes;
    sc.n_tris = tree_tris;
    sc.n_spheres = n_spheres;
    sc.n_lights = n_lights;
    sc.n_materials = n_materials;
    sc.stack_entries = 2u * tree_depth + 2u; // a visit parks at most two groups
    ctx->stats = rt_stats{};
    ctx->stats.node_bytes = sizeof(DevNode8);
    ctx->stats.tri_bytes = sizeof(DevTri);
    ctx->stats.scene_bytes = (size_t)tree_nodes * sizeof(DevNode8) + (size_t)tree_tris * sizeof(DevTri) + ds.size() * sizeof(DevSphere) +
                             dl.size() * sizeof(DevLight) + dm.size() * sizeof(DevMaterial);
    ctx->stats.bvh_nodes = sc.n_nodes;
    ctx->stats.bvh_depth = tree_depth;
    ctx->stats.tree_build = (uint32_t)method;
    ctx->tree_tris_uploaded = tree_tris;
    ctx->stats.n_devices = (uint32_t)ctx->devs.size();
    ctx->stats.n_textures = ctx->n_textures;
    ctx->stats.texture_bytes = ctx->texture_bytes;
    ctx->build_tris.swap(bt);
    ctx->host_lights = dl; // the light grids of the extended mode's shadow stage are built when a frame first needs them (ensure_grids)
    for (int a = 0; a < 3; a++) ctx->box_lo[a] = box_lo[a], ctx->box_hi[a] = box_hi[a];
    ctx->uploaded = true;
    ctx->frame_valid = false;
    return RT_OK;
}

// Per-light triangle lists for the shadow segments of the wavefront pipeline (shadow_grid.h), rasterised on the device from the
// triangle records of the uploaded scene.  LAZY: the reference's flow (src/scene.rs:87-119 replace_with_gltf, then
// src/compute.rs:12-50 every frame) only ever renders modes 0/1, which trace no shadow segments, so rt_upload_scene* builds
// nothing; the first extended-mode frame that traces shadow segments through the pipeline pays the build once per scene (or the
// caller asks for it ahead of time: rt_prepare).  rt_stats reports what it cost (grid_build_ms) and holds (grid_bytes).
// RT_SHADOW_GRID=0 and the RT_SHADOW_GRID_* variables are development knobs only.
int ensure_grids(rt_ctx* ctx, DeviceState& d) {
    if (d.grids_tried) return RT_OK;
    d.grids_tried = true;
    const uint32_t n_lights = (uint32_t)ctx->host_lights.size(), tree_tris = ctx->scene_counts.n_tris;
    bool grids_on = true;
    rt::ShadowGridOptions gopt;
    if (const char* e = std::getenv("RT_SHADOW_GRID")) grids_on = std::atoi(e) != 0;
    if (const char* e = std::getenv("RT_SHADOW_GRID_RES")) gopt.res_point = (uint32_t)std::atoi(e), gopt.res_dir = 2u * (uint32_t)std::atoi(e);
    if (const char* e = std::getenv("RT_SHADOW_GRID_HEAVY")) gopt.heavy = (uint32_t)std::atoi(e);
    if (const char* e = std::getenv("RT_SHADOW_GRID_MEAN")) gopt.max_mean_list = std::atof(e), gopt.max_heavy_share = 1.0; // (forces grids onto cluttered scenes)
    if (!grids_on || n_lights == 0 || n_lights > RT_WF_MAX_LIGHTS || tree_tris == 0) return RT_OK;
    const double t0 = now_ms();
    HIPCHK(ctx, hipSetDevice(d.device));
    HIPCHK(ctx, hipStreamSynchronize(d.stream));
    std::vector<DevShadowGrid> hg(n_lights);
    bool any = false;
    size_t free_b = 0, total_b = 0;
    HIPCHK(ctx, hipMemGetInfo(&free_b, &total_b));
    // all grids together may take a quarter of what is free now - cell blocks (128 bytes per cell: 0.8 GB per cube map at 1024 cells
    // per face side) AND list entries count against it (only the entries did); a light whose grid would not fit gets none
    rt::ShadowGridOptions lopt = gopt;
    lopt.max_bytes = free_b / 4 / n_lights;
    lopt.max_entries = std::min<uint64_t>(gopt.max_entries, lopt.max_bytes / (RT_SG_ENTRY_QUADS * sizeof(uint4)));
    uint64_t bytes = 0;
    for (uint32_t i = 0; i < n_lights; i++) {
        rt::ShadowGridBuild gb;
        const hipError_t e = rt::shadow_grid_build(d.tris, tree_tris, ctx->host_lights[i], ctx->box_lo, ctx->box_hi, lopt, d.stream, &gb);
        if (e != hipSuccess) { // out of memory for a grid is not an error: the BVH answers for this light
            (void)hipGetLastError();
            gb = rt::ShadowGridBuild{};
        }
        hg[i] = gb.grid;
        if (gb.blocks) d.grid_allocs.push_back(gb.blocks);
        if (gb.overflow) d.grid_allocs.push_back(gb.overflow);
        if (gb.grid.kind != RT_SG_KIND_NONE) bytes += gb.bytes;
        any = any || gb.grid.kind != RT_SG_KIND_NONE;
        if (gb.grid.kind == RT_SG_KIND_NONE && gb.n_entries != 0) d.grids_partial = true; // lists were counted and found too long (or too large): this light's segments walk the tree
        d.grid_info.push_back(gb);
    }
    if (any && d.grids_partial) {
        // All lights or none.  The segments of a light without a grid are handed on one wave-load at a time, an atomic on the next queue's
        // counter each: a trickle when the lists leave 0.1 - 20 % to the tree, but with a whole light's segments it is millions of atomics
        // on one address per frame (bistro-like with one light of four on lists: 67 - 81 ms against 54 - 60 with no grids at all).
        HIPCHK(ctx, hipDeviceSynchronize());
        for (void* p : d.grid_allocs) (void)hipFree(p);
        d.grid_allocs.clear();
        for (uint32_t i = 0; i < n_lights; i++) {
            hg[i] = DevShadowGrid{};
            d.grid_info[i].grid = DevShadowGrid{};
            d.grid_info[i].blocks = d.grid_info[i].overflow = nullptr;
            d.grid_info[i].bytes = 0;
        }
        any = false;
        bytes = 0;
    }
    if (any) {
        int rc;
        if ((rc = upload_array(ctx, &d.grids, hg)) != RT_OK) return rc;
        d.grid_allocs.push_back(d.grids);
        HIPCHK(ctx, hipDeviceSynchronize());
    }
    if (&d == &ctx->devs[0]) { // (every device holds the same grids)
        ctx->stats.grid_bytes = bytes;
        ctx->stats.grid_build_ms = now_ms() - t0;
    }
    return RT_OK;
}

// Path-state arrays and queues of the wavefront pipeline, sized for `batch` samples per owned pixel block.
int ensure_wavefront(rt_ctx* ctx, DeviceState& d, uint32_t n_blocks, uint32_t batch, uint32_t n_lights, bool two_lanes) {
    const uint32_t capacity = n_blocks * batch * 64u;
    // overflow entries (64-bit) per lane beyond the LDS part of the stack
    const uint32_t ovf_entries = ctx->scene_counts.stack_entries + 2u > RT_WF8_LDS_STACK ? ctx->scene_counts.stack_entries + 2u - RT_WF8_LDS_STACK : 1u;
    if (d.wf.capacity >= capacity && d.wf.n_blocks == n_blocks && d.wf.batch == batch && d.wf_lights >= n_lights && d.wf.counters &&
        d.wf.ovf_entries >= ovf_entries && d.wf.stack_ovf && (!two_lanes || (d.wf2.counters && d.wf2.capacity >= capacity && d.wf2.stack_ovf)))
        return RT_OK;
    free_wavefront(d);
    HIPCHK(ctx, hipSetDevice(d.device));
    // an allocation that fails half way must not leave a lane that looks complete to the reuse test above
    struct Guard {
        DeviceState& d;
        bool ok = false;
        ~Guard() { if (!ok) free_wavefront(d); }
    } guard{d};
    for (int lane = 0; lane < (two_lanes ? 2 : 1); lane++) {
    std::vector<void*>& owned = lane ? d.wf2_allocs : d.wf_allocs;
    auto alloc = [&](void** p, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(p, bytes ? bytes : 16);
        if (e == hipSuccess) owned.push_back(*p);
        return e;
    };
    rt::WfBuffers& w = lane ? d.wf2 : d.wf;
    const size_t P = capacity ? capacity : 64;
    HIPCHK(ctx, alloc((void**)&w.ray_o, P * 16));
    HIPCHK(ctx, alloc((void**)&w.ray_d, P * 16));
    HIPCHK(ctx, alloc((void**)&w.hit, P * 16));
    HIPCHK(ctx, alloc((void**)&w.thr, P * 16));
    HIPCHK(ctx, alloc((void**)&w.rad, P * 16));
    HIPCHK(ctx, alloc((void**)&w.vtx, P * 32));
    HIPCHK(ctx, alloc((void**)&w.sample_rad, P * 16));
    HIPCHK(ctx, alloc((void**)&w.pxy, P * 4));
    // producers reserve queue space in windows (wavefront.hip): wf_queue_slots is the bound on real entries + padding
    const uint32_t lights = std::max(1u, n_lights);
    const size_t ext_slots = rt::wf_queue_slots(P, 1), shadow_slots = rt::wf_queue_slots(P * (size_t)lights, lights);
    if (ext_slots > 0xFFFFFFFFull || shadow_slots > 0xFFFFFFFFull)
        return ctx->fail(RT_ERR_INTERNAL, "wavefront queues of %zu / %zu slots exceed 32-bit positions", ext_slots, shadow_slots);
    HIPCHK(ctx, alloc((void**)&w.q_ext[0], ext_slots * 4));
    HIPCHK(ctx, alloc((void**)&w.q_ext[1], ext_slots * 4));
    HIPCHK(ctx, alloc((void**)&w.q_shadow, shadow_slots * 4));
    HIPCHK(ctx, alloc((void**)&w.q_shadow2, (P * (size_t)lights + 64) * 4)); // written densely: one slot per entry
    w.q_ext_cap = (uint32_t)ext_slots;
    w.q_shadow_cap = (uint32_t)shadow_slots;
    HIPCHK(ctx, alloc((void**)&w.counters, rt::WF_N_COUNTERS * sizeof(uint32_t)));
    HIPCHK(ctx, alloc((void**)&w.totals, 16 * sizeof(unsigned long long)));
    if (lane == 0) HIPCHK(ctx, alloc((void**)&w.accum, (size_t)std::max(1u, n_blocks) * 64 * 16));
    else w.accum = d.wf.accum; // one running sum per pixel: the lanes' resolves are ordered by events
    if (lane == 0) { // the camera beams of the frame: per block a leaf list, built once per frame and read by both lanes
        HIPCHK(ctx, alloc((void**)&w.beam_count, ((size_t)std::max(1u, n_blocks) + 1) * 4));
        HIPCHK(ctx, alloc((void**)&w.beam_ref, (size_t)std::max(1u, n_blocks) * RT_BEAM_CAP * 4));
        HIPCHK(ctx, alloc((void**)&w.beam_dist, (size_t)std::max(1u, n_blocks) * RT_BEAM_CAP * 4));
    } else {
        w.beam_count = d.wf.beam_count;
        w.beam_ref = d.wf.beam_ref;
        w.beam_dist = d.wf.beam_dist;
    }
    HIPCHK(ctx, alloc((void**)&w.stack_ovf, (size_t)rt::wf_persistent_waves() * ovf_entries * 64 * 8));
    w.ovf_entries = ovf_entries;
    w.n_blocks = n_blocks;
    w.batch = batch;
    w.capacity = capacity;
    }
    d.wf_lights = n_lights;
    guard.ok = true;
    return RT_OK;
}

// Samples per pixel kept in flight by the wavefront pipeline.  Every stage of every bounce ends in a tail where
// the persistent waves drain, and late bounces carry few paths, so batches should be as large as memory allows:
// measured on the headline frame, 3 samples per batch (8 M paths) 4,640 Mrays/s, 32 per batch (71 M paths, 12 GB
// of the 288 GB) 5,840.  Target 64 M paths, bounded by half the free device memory, the 27-bit path id of a shadow
// queue entry and the 32-bit queue positions; the samples are then spread evenly over the batches.
// RT_WF_BATCH (samples per batch) / RT_WF_TARGET_PATHS override.
// Largest number of path slots the wavefront pipeline can address: 27-bit path ids in shadow queue entries, 32-bit queue
// positions with up to 4 slots per (path, light) plus per-wave slack (wf_queue_slots).
uint64_t wavefront_max_paths(uint32_t n_lights) {
    const uint64_t lights = std::max(1u, n_lights);
    return std::min<uint64_t>((uint64_t)RT_WF_ID_MASK + 1, (1ull << 29) / lights);
}

uint32_t wavefront_batch(uint32_t n_blocks, uint32_t spp, uint32_t n_lights, size_t free_bytes, bool two_lanes) {
    const uint64_t per_sample = std::max<uint64_t>(1, (uint64_t)n_blocks * 64u);
    const uint64_t hard_limit = std::max<uint64_t>(1, wavefront_max_paths(n_lights) / per_sample); // samples per batch the ids allow
    if (const char* e = std::getenv("RT_WF_BATCH")) {
        const uint32_t want = (uint32_t)std::strtoul(e, nullptr, 10);
        if (want) return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min(want, spp), hard_limit));
    }
    uint64_t target_paths = 64ull << 20;
    if (const char* e = std::getenv("RT_WF_TARGET_PATHS")) target_paths = std::max<uint64_t>(1, std::strtoull(e, nullptr, 10));
    const uint64_t lights = std::max(1u, n_lights);
    const uint64_t bytes_per_path = 8 * 16 + 4 + 2 * 16 + 16 * lights + 4 * lights; // path state, vis, pxy, two extension queues, shadow queue (4 slots per entry), the handed-on shadow queue (dense)
    target_paths = std::min<uint64_t>(target_paths, free_bytes / 2 / bytes_per_path);
    target_paths = std::min<uint64_t>(target_paths, wavefront_max_paths(n_lights));
    uint32_t max_batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(spp, target_paths / per_sample));
    if (two_lanes && spp >= 2) max_batch = std::min(max_batch, (spp + 1) / 2); // two lanes want two batches, also of a frame that would fit one
    uint32_t n_batches = (spp + max_batch - 1) / max_batch;
    if (two_lanes && n_batches > 1 && (n_batches & 1u)) {
        // an even number of batches, so that both lanes get the same share (measured on the headline frame, round 3: 3 batches of 22/21/21
        // samples 176.8 ms, 2 of 32 170.9, 4 of 16 174.3): one batch fewer if that stays within a tenth of the target and the hard limits
        const uint32_t fewer = n_batches - 1, b = (spp + fewer - 1) / fewer;
        if ((uint64_t)b * per_sample * 10 <= target_paths * 11 && b <= hard_limit && (uint64_t)b * per_sample <= free_bytes / 2 / bytes_per_path) n_batches = fewer;
        else n_batches++;
    }
    return std::max(1u, (spp + n_batches - 1) / std::max(1u, n_batches));
}

// Largest tree (8-wide nodes) whose primary-rays-only frames take the one-pass kernel instead of the queue pipeline (run_frame).
uint32_t single_pass_max_nodes() {
    if (const char* e = std::getenv("RT_SINGLE_PASS_MAX_NODES")) return (uint32_t)std::strtoul(e, nullptr, 10); // development knob
    return RT_SINGLE_PASS_MAX_NODES;
}

// rt_dispatch_tile returns after the launch, like `queue.submit` in src/compute.rs:165.  Whatever needs the result or
// the device idle (read-back, statistics, a new scene, teardown) waits here first; the kernel time reported afterwards
// is that of the LAST dispatch.
int sync_pending(rt_ctx* ctx) {
    if (!ctx->pending_dispatch) return RT_OK;
    ctx->pending_dispatch = false;
    DeviceState& d = ctx->devs[0];
    HIPCHK(ctx, hipSetDevice(d.device));
    HIPCHK(ctx, hipStreamSynchronize(d.stream));
    float ms = 0.0f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, d.ev0, d.ev1));
    ctx->stats.kernel_ms = ms;
    return RT_OK;
}

// Launch one frame (or one explicit tile) on every device and wait.
int run_frame(rt_ctx* ctx, DevFrame fr, bool counters, uint32_t world, uint32_t rank, bool single_tile) {
    double w0 = now_ms();
    if (!single_tile || counters) {
        int rcp = sync_pending(ctx);
        if (rcp != RT_OK) return rcp;
    }
    size_t nd = single_tile ? 1 : ctx->devs.size();
    uint32_t total_tiles = fr.tiles_x * fr.tiles_y;
    bool fallback = false; // the extended mode was asked for its default (queue) pipeline and got the megakernel
    bool single = false;   // ... took the one-pass kernel by rule (primary rays only, tiny tree)
    struct WfRun { // one device's walk through the batches and bounces of its share of a wavefront-pipeline frame
        enum Phase { GENERATE, BOUNCE, RESOLVE, DONE };
        size_t dev = 0;
        DevFrame f{};
        DevScene dsc{};
        uint32_t batch = 1, first = 0, j = 0, it = 0, alive = 0;
        bool two = false;
        Phase phase = GENERATE;
        hipStream_t poll_stream = nullptr;
    };
    std::vector<WfRun> runs;
    runs.reserve(nd); // (polls hold pointers into it)
    for (size_t j = 0; j < nd; j++) ctx->devs[j].stage_events_used = 0;
    ctx->stage_ms[0] = ctx->stage_ms[1] = 0.0;
    for (size_t j = 0; j < nd; j++) {
        DeviceState& d = ctx->devs[j];
        HIPCHK(ctx, hipSetDevice(d.device));
        int rc = ensure_targets(ctx, d, fr.width, fr.height);
        if (rc != RT_OK) return rc;
        DevFrame f = fr;
        if (!single_tile) {
            // the context owns the tiles i with i % world == rank (rt_hip.h); its device j takes every nd-th of those
            uint32_t stride = world * (uint32_t)nd, first = rank + world * (uint32_t)j;
            f.tile_first = first;
            f.tile_stride = stride;
            f.n_owned_tiles = first < total_tiles ? (total_tiles - first + stride - 1) / stride : 0;
            d.tile_first = first;
            d.tile_stride = stride;
            d.n_owned = f.n_owned_tiles;
        }
        HIPCHK(ctx, hipMemsetAsync(d.counters, 0, 16 * sizeof(unsigned long long), d.stream));
        HIPCHK(ctx, hipEventRecord(d.ev0, d.stream));
        // the queue-based pipeline needs one visibility bit per light and every path slot of ONE sample per owned pixel block
        // addressable; anything else (more than 32 lights, a single device's share beyond ~134 M pixels) takes the megakernel
        // Primary rays only over a tiny tree (BASELINE configs[1]: Cornell, 12 triangles, max_bounces 0): the queue pipeline would move
        // 132 bytes of path state per sample through six kernels for a closest hit that costs a few dozen instructions (measured 15.9 ms
        // and 46.7 GB of HBM traffic per 1080p 64-spp frame); the nested-loop kernel loops the pixel's samples in registers, traces the
        // closest hit and the per-light shadow segments inline and stores the pixel once: 2.6 ms, same bits (the per-sample order is
        // fixed by the CPU statement).  SURVEY 7 step 6, shader/src/lib.rs:86-88 (one write per pixel).  RT_FLAG_KERNEL_PIPELINE keeps the pipeline.
        const bool single_pass = f.mode == RT_MODE_EXTENDED && !(f.flags & (RT_FLAG_KERNEL_V1 | RT_FLAG_KERNEL_SM | RT_FLAG_KERNEL_PIPELINE)) && f.max_bounce == 0 &&
                                 ctx->scene_counts.n_nodes <= single_pass_max_nodes();
        if (single_pass) {
            single = true;
            f.flags |= RT_FLAG_KERNEL_V1; // (launch_render_extended picks its kernel by this bit: the nested loops, not the state machine)
        }
        const bool wavefront = f.mode == RT_MODE_EXTENDED && !(f.flags & (RT_FLAG_KERNEL_V1 | RT_FLAG_KERNEL_SM)) && !single_pass &&
                               ctx->scene_counts.n_lights <= RT_WF_MAX_LIGHTS &&
                               (uint64_t)f.n_owned_tiles * rt::blocks_per_tile(f.tile_size) * 64u <= wavefront_max_paths(ctx->scene_counts.n_lights);
        if (f.mode == RT_MODE_EXTENDED && !wavefront && !single_pass && !(f.flags & (RT_FLAG_KERNEL_V1 | RT_FLAG_KERNEL_SM))) fallback = true;
        if (wavefront) {
            if (!(f.flags & (RT_FLAG_NO_SHADOW_GRID | RT_FLAG_NO_SHADOWS))) { // the first frame that traces shadow segments through the pipeline builds the light grids
                rc = ensure_grids(ctx, d);
                if (rc != RT_OK) return rc;
            }
            const uint32_t n_blocks = f.n_owned_tiles * rt::blocks_per_tile(f.tile_size);
            size_t free_b = 0, total_b = 0;
            HIPCHK(ctx, hipMemGetInfo(&free_b, &total_b));
            const char* lanes_env = std::getenv("RT_WF_LANES"); // (read per frame: tests and A/B runs switch it)
            // by default only for frames whose shadow stage runs on light grids: with the BVH walk both big stages are VALU-bound and sharing
            // the chip gains nothing (bistro-like 4K share: 136.7 ms on two lanes against 127.3 on one)
            bool two_lanes_on = lanes_env ? std::atoi(lanes_env) >= 2 : (RT_WF_LANES_DEFAULT >= 2 && d.grids != nullptr && !(f.flags & RT_FLAG_NO_SHADOW_GRID));
            // ... and not for a frame of long paths that fits ONE batch: two lanes would cut it in two, every launch would come twice, and
            // beyond four bounces most launches serve few paths and cost their latency whatever they serve (bistro-like 4K share, 64 spp,
            // 8 bounces, 66 M paths: one batch on one lane 105.3 ms, two of 32 samples on two lanes 116.7; 4 bounces: the two lanes win)
            if (!lanes_env && two_lanes_on && f.max_bounce > 4 && wavefront_batch(n_blocks, f.spp, ctx->scene_counts.n_lights, free_b, false) >= f.spp) two_lanes_on = false;
            const uint32_t batch = d.wf.n_blocks == n_blocks && d.wf_spp == f.spp && d.wf.batch && d.wf_lights >= ctx->scene_counts.n_lights && d.used_two_lanes == (two_lanes_on && f.spp >= 2) ? d.wf.batch // same frame shape as last time: keep the allocation
                                                                                               : wavefront_batch(n_blocks, f.spp, ctx->scene_counts.n_lights, two_lanes_on ? free_b / 2 : free_b, two_lanes_on);
            d.wf_spp = f.spp;
            // Two lanes (RT_WF_LANES=2, the default when a frame has two or more batches): batches alternate between two streams with their own
            // path state, so that one batch's shadow stage (bound by HBM bandwidth since the light grids) runs beside the other's closest-hit
            // traversal (bound by VALU issue).  The pixel sums are added in batch order: a resolve waits for the previous batch's.
            const uint32_t n_batches = (f.spp + batch - 1) / batch;
            const bool two = two_lanes_on && n_batches >= 2 && d.stream2;
            rc = ensure_wavefront(ctx, d, n_blocks, batch, ctx->scene_counts.n_lights, two);
            if (rc != RT_OK) return rc;
            HIPCHK(ctx, hipMemsetAsync(d.wf.totals, 0, 16 * sizeof(unsigned long long), d.stream));
            HIPCHK(ctx, hipEventRecord(d.ev0, d.stream)); // re-record: allocation above is not part of the kernel time
            { // camera beams: on by default, RT_WF_BEAMS=0 (development knob) or RT_FLAG_NO_BEAMS walks the tree for every camera segment
                const char* be = std::getenv("RT_WF_BEAMS");
                const bool beams = !(f.flags & RT_FLAG_NO_BEAMS) && (be ? std::atoi(be) != 0 : true);
                d.wf_beams_off = !beams;
                if (beams) HIPCHK(ctx, rt::wf_beams(scene_for(ctx, d), f, d.wf, d.stream));
            }
            if (two) {
                HIPCHK(ctx, hipEventRecord(d.ev_start, d.stream));
                HIPCHK(ctx, hipStreamWaitEvent(d.stream2, d.ev_start, 0));
                HIPCHK(ctx, hipMemsetAsync(d.wf2.totals, 0, 16 * sizeof(unsigned long long), d.stream2));
            }
            d.wf.grids = d.wf2.grids = (f.flags & RT_FLAG_NO_SHADOW_GRID) ? nullptr : d.grids;
            { const char* pe = std::getenv("RT_WF_PROBE"); d.wf.probe = d.wf2.probe = pe ? (uint32_t)std::atoi(pe) : 0u; }
            // the launches themselves follow below, one step per device in turn (WfRun): a device's every-8-bounces read-back of its live
            // paths must not hold back the launches of the next device (it was inside this loop)
            WfRun r;
            r.dev = j;
            r.f = f;
            r.dsc = scene_for(ctx, d);
            r.batch = batch;
            r.two = two;
            runs.push_back(r);
            d.used_two_lanes = two;
            d.used_wavefront = true;
        } else if (f.mode == RT_MODE_EXTENDED) {
            d.used_wavefront = false;
            HIPCHK(ctx, rt::launch_render_extended(scene_for(ctx, d), f, targets_for(d), counters, d.stream));
        } else
            HIPCHK(ctx, rt::launch_render_reference(scene_for(ctx, d), f, targets_for(d), counters, d.stream));
        if (!wavefront) HIPCHK(ctx, hipEventRecord(d.ev1, d.stream));
    }
    // The wavefront pipeline's launches: every device advances by one step (a batch's generation, one bounce, a batch's resolve) in
    // turn, so that several devices of one context fill up side by side; the devices that have just passed a multiple of 8 bounces are
    // then polled together ("long paths: stop as soon as every path has ended").
    for (bool busy = !runs.empty(); busy;) {
        busy = false;
        std::vector<WfRun*> polls;
        for (WfRun& r : runs) {
            if (r.phase == WfRun::DONE) continue;
            busy = true;
            DeviceState& d = ctx->devs[r.dev];
            HIPCHK(ctx, hipSetDevice(d.device));
            const uint32_t n = std::min(r.batch, r.f.spp - r.first);
            const uint32_t lane = r.two ? (r.j & 1u) : 0u;
            rt::WfBuffers w = lane ? d.wf2 : d.wf; // (a copy: the kernels take it by value)
            if (d.wf_beams_off) w.beam_count = nullptr;
            hipStream_t st = lane ? d.stream2 : d.stream;
            if (r.phase == WfRun::GENERATE) {
                HIPCHK(ctx, rt::wf_generate(r.dsc, r.f, w, r.first, n, st));
                r.it = 0;
                r.phase = WfRun::BOUNCE;
            } else if (r.phase == WfRun::BOUNCE) {
                hipEvent_t* gev = nullptr;
                if ((r.f.flags & RT_FLAG_STAGE_TIMES) && r.dev == 0 && w.grids) { // (bench.py: the dominant kernel's launch durations, measured live)
                    if (d.stage_events.size() < (size_t)d.stage_events_used + 2) {
                        hipEvent_t a = nullptr, b = nullptr;
                        HIPCHK(ctx, hipEventCreate(&a));
                        HIPCHK(ctx, hipEventCreate(&b));
                        d.stage_events.push_back(a);
                        d.stage_events.push_back(b);
                    }
                    gev = d.stage_events.data() + d.stage_events_used;
                    d.stage_events_used += 2;
                }
                HIPCHK(ctx, rt::wf_bounce(r.dsc, r.f, w, r.it, n, counters, st, gev));
                if (r.it >= r.f.max_bounce) r.phase = WfRun::RESOLVE;
                else if ((r.it & 7u) == 7u) {
                    HIPCHK(ctx, hipMemcpyAsync(&r.alive, w.counters + rt::WF_EXT_COUNT, 4, hipMemcpyDeviceToHost, st));
                    r.poll_stream = st;
                    polls.push_back(&r);
                } else r.it++;
            } else { // RESOLVE
                if (r.two && r.j > 0) HIPCHK(ctx, hipStreamWaitEvent(st, d.ev_res[(r.j - 1u) & 1u], 0));
                HIPCHK(ctx, rt::wf_resolve(r.f, w, targets_for(d), n, r.first == 0, r.first + n >= r.f.spp, st));
                if (r.two) HIPCHK(ctx, hipEventRecord(d.ev_res[r.j & 1u], st));
                r.first += r.batch;
                r.j++;
                if (r.first < r.f.spp) r.phase = WfRun::GENERATE;
                else {
                    if (r.two) { // the frame ends when both lanes have
                        HIPCHK(ctx, hipEventRecord(d.ev_join, d.stream2));
                        HIPCHK(ctx, hipStreamWaitEvent(d.stream, d.ev_join, 0));
                    }
                    HIPCHK(ctx, hipEventRecord(d.ev1, d.stream));
                    r.phase = WfRun::DONE;
                }
            }
        }
        for (WfRun* r : polls) { // (after every device has its next launches queued)
            HIPCHK(ctx, hipSetDevice(ctx->devs[r->dev].device));
            HIPCHK(ctx, hipStreamSynchronize(r->poll_stream));
            if (r->alive == 0) r->phase = WfRun::RESOLVE;
            else r->it++;
        }
    }
    double kernel_ms = 0.0;
    unsigned long long cnt[16] = {0}, grid_cnt[2] = {0, 0};
    const bool extended = fr.mode == RT_MODE_EXTENDED;
    uint64_t pixels = 0;
    if (single_tile && !counters && !extended) { // one explicit tile of the reference's dispatch sequence: do not wait
        const uint64_t px = (uint64_t)std::min(fr.tile_w, fr.width - std::min(fr.width, fr.tile_off_x)) *
                            std::min(fr.tile_h, fr.height - std::min(fr.height, fr.tile_off_y));
        rt_stats& st = ctx->stats;
        st.pixels = px;
        st.rays = st.primary_rays = (fr.mode == RT_MODE_LEGACY || fr.cur_bounce <= fr.max_bounce) ? px : 0;
        st.continuation_rays = st.shadow_rays = st.node_visits = st.tri_tests = 0;
        st.kernel_ms = 0.0; // filled in by sync_pending
        st.wall_ms = now_ms() - w0;
        ctx->pending_dispatch = true;
        return RT_OK;
    }
    for (size_t j = 0; j < nd; j++) {
        DeviceState& d = ctx->devs[j];
        HIPCHK(ctx, hipSetDevice(d.device));
        HIPCHK(ctx, hipStreamSynchronize(d.stream));
        float ms = 0.0f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, d.ev0, d.ev1));
        kernel_ms = std::max(kernel_ms, (double)ms);
        if (j == 0 && d.stage_events_used) {
            if (d.stream2) HIPCHK(ctx, hipStreamSynchronize(d.stream2));
            for (uint32_t k = 0; k + 1 < d.stage_events_used; k += 2) {
                float sm = 0.0f;
                HIPCHK(ctx, hipEventElapsedTime(&sm, d.stage_events[k], d.stage_events[k + 1]));
                ctx->stage_ms[0] += sm;
                ctx->stage_ms[1] += 1.0;
            }
        }
        if (extended && d.used_wavefront) {
            unsigned long long t[16];
            HIPCHK(ctx, hipMemcpy(t, d.wf.totals, sizeof t, hipMemcpyDeviceToHost));
            if (d.used_two_lanes) {
                unsigned long long t2[16];
                HIPCHK(ctx, hipMemcpy(t2, d.wf2.totals, sizeof t2, hipMemcpyDeviceToHost));
                for (int k = 0; k < 16; k++) t[k] = k == 5 ? std::max(t[k], t2[k]) : t[k] + t2[k]; // ([5] is a high-water mark)
            }
            cnt[0] += t[0] + t[1] + t[2];
            cnt[3] += t[0];
            cnt[4] += t[1];
            cnt[5] += t[2];
            cnt[1] += t[3];
            cnt[2] += t[4];
            if (t[WF_TOTAL_ERROR] != 0)
                return ctx->fail(RT_ERR_INTERNAL, "wavefront pipeline: %s (frame discarded)",
                                 (t[WF_TOTAL_ERROR] & 2ull) ? "a traversal stack grew beyond the depth the tree reports" : "a queue reservation exceeded its allocation");
            cnt[8] = std::max(cnt[8], t[5]); // diagnostics: stack high-water mark, visits with > 16 / > 24 entries
            cnt[9] += t[6];
            cnt[10] += t[7];
            for (int k = 0; k < 5; k++) cnt[11 + k] += t[8 + k]; // wave-level step counts of the traversal stages
            grid_cnt[0] += t[13];
            grid_cnt[1] += t[14];
        } else if (counters || extended) {
            unsigned long long c[16];
            HIPCHK(ctx, hipMemcpy(c, d.counters, sizeof c, hipMemcpyDeviceToHost));
            for (int k = 0; k < 16; k++) cnt[k] += c[k];
        }
        if (single_tile) {
            pixels += (uint64_t)std::min(fr.tile_w, fr.width - std::min(fr.width, fr.tile_off_x)) *
                      std::min(fr.tile_h, fr.height - std::min(fr.height, fr.tile_off_y));
        } else {
            for (uint32_t k = 0; k < d.n_owned; k++) {
                uint32_t tile = d.tile_first + k * d.tile_stride;
                uint32_t ty = tile / fr.tiles_x, tx = tile % fr.tiles_x;
                pixels += (uint64_t)std::min(fr.tile_size, fr.width - tx * fr.tile_size) * std::min(fr.tile_size, fr.height - ty * fr.tile_size);
            }
        }
    }
    rt_stats& st = ctx->stats;
    st.pixels = pixels;
    bool traced = fr.mode == RT_MODE_LEGACY || fr.cur_bounce <= fr.max_bounce;
    if (extended) { // segments counted on the device
        st.rays = cnt[0];
        st.primary_rays = cnt[3];
        st.continuation_rays = cnt[4];
        st.shadow_rays = cnt[5];
    } else {
        st.rays = traced ? pixels : 0; // modes 0/1: one segment per pixel
        st.primary_rays = st.rays;
        st.continuation_rays = st.shadow_rays = 0;
    }
    st.node_visits = counters ? cnt[1] : 0;
    st.tri_tests = counters ? cnt[2] : 0;
    for (int k = 0; k < 8; k++) ctx->diag[k] = counters ? cnt[8 + k] : 0;
    for (int k = 0; k < 2; k++) ctx->grid_diag[k] = counters ? grid_cnt[k] : 0;
    st.kernel_ms = kernel_ms;
    st.wall_ms = now_ms() - w0;
    st.flags = (fallback ? RT_STAT_MEGAKERNEL_FALLBACK : 0u) | (single ? RT_STAT_SINGLE_PASS : 0u);
    return RT_OK;
}

int gather(rt_ctx* ctx, uint8_t* out, size_t elem, int which);

// A dispatch sequence runs on the context's first device and starts from the textures as they are (tiles that are
// not dispatched keep their texels, like the reference's storage textures).  After an rt_render that was split over
// several devices those texels are spread over the devices: bring them together on the first one (a slow path through
// host memory, taken once at the transition).
int consolidate_on_first_device(rt_ctx* ctx, uint32_t w, uint32_t h) {
    if (ctx->devs.size() < 2 || ctx->frame_w != w || ctx->frame_h != h) return RT_OK;
    DeviceState& d0 = ctx->devs[0];
    if (d0.fb_w != w || d0.fb_h != h) return RT_OK;
    bool spread = false;
    for (size_t j = 1; j < ctx->devs.size(); j++) spread = spread || (ctx->devs[j].n_owned > 0 && ctx->devs[j].fb_w == w && ctx->devs[j].fb_h == h);
    if (!spread) return RT_OK;
    for (auto& d : ctx->devs) {
        HIPCHK(ctx, hipSetDevice(d.device));
        HIPCHK(ctx, hipStreamSynchronize(d.stream));
    }
    const size_t n = (size_t)w * h;
    std::vector<uint8_t> buf;
    HIPCHK(ctx, hipSetDevice(d0.device));
    for (int which = 0; which < 6; which++) {
        const size_t elem = which == 0 ? 16 : 4;
        void* dst = which == 0 ? (void*)d0.rgba32f : which <= 3 ? (void*)d0.chan[which - 1] : which == 4 ? (void*)d0.prim_id : (void*)d0.hit_t;
        buf.resize(n * elem);
        HIPCHK(ctx, hipSetDevice(d0.device));
        HIPCHK(ctx, hipMemcpy(buf.data(), dst, n * elem, hipMemcpyDeviceToHost)); // what the first device holds outside everyone's tiles
        int rc = gather(ctx, buf.data(), elem, which);
        if (rc != RT_OK) return rc;
        HIPCHK(ctx, hipSetDevice(d0.device));
        HIPCHK(ctx, hipMemcpy(dst, buf.data(), n * elem, hipMemcpyHostToDevice));
    }
    HIPCHK(ctx, hipDeviceSynchronize());
    d0.tile_first = 0;
    d0.tile_stride = 1;
    d0.n_owned = ctx->frame_tiles_x * ctx->frame_tiles_y;
    for (size_t j = 1; j < ctx->devs.size(); j++) ctx->devs[j].n_owned = 0;
    return RT_OK;
}

} // namespace

extern "C" {

const char* rt_version(void) { return "librt_hip 0.1 gfx950 (fp-contract=off, IEEE div/sqrt)"; }

const char* rt_last_error(rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt_create(rt_ctx** out, const int* device_ids, int n_devices) {
    if (!out || n_devices < 1) {
        g_create_error = "rt_create: bad arguments";
        return RT_ERR_BAD_ARG;
    }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1) {
        g_create_error = std::string("rt_create: no HIP device (") + hipGetErrorString(e) + "); this library has no CPU fallback";
        return RT_ERR_HIP;
    }
    rt_ctx* ctx = new rt_ctx();
    for (int i = 0; i < n_devices; i++) {
        DeviceState d;
        d.device = device_ids ? device_ids[i] : i;
        if (d.device < 0 || d.device >= count) {
            g_create_error = "rt_create: device id out of range";
            rt_destroy(ctx);
            return RT_ERR_BAD_ARG;
        }
        if ((e = hipSetDevice(d.device)) != hipSuccess || (e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking)) != hipSuccess || (e = hipStreamCreateWithFlags(&d.stream2, hipStreamNonBlocking)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&d.ev_start, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&d.ev_join, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&d.ev_res[0], hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&d.ev_res[1], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreate(&d.ev0)) != hipSuccess || (e = hipEventCreate(&d.ev1)) != hipSuccess ||
            (e = hipMalloc((void**)&d.counters, 16 * sizeof(unsigned long long))) != hipSuccess) {
            g_create_error = std::string("rt_create: ") + hipGetErrorString(e);
            ctx->devs.push_back(d);
            rt_destroy(ctx);
            return RT_ERR_HIP;
        }
        ctx->devs.push_back(d);
    }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_ctx* ctx) {
    if (!ctx) return;
    for (auto& d : ctx->devs) {
        if (d.device < 0) continue;
        (void)hipSetDevice(d.device);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        free_scene(d);
        free_targets(d);
        free_wavefront(d);
        (void)hipFree(d.counters);
        if (d.ev0) (void)hipEventDestroy(d.ev0);
        if (d.ev1) (void)hipEventDestroy(d.ev1);
        for (hipEvent_t e : {d.ev_start, d.ev_join, d.ev_res[0], d.ev_res[1]})
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : d.stage_events) (void)hipEventDestroy(e);
        if (d.stream2) (void)hipStreamDestroy(d.stream2);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    delete ctx;
}

int rt_upload_scene(rt_ctx* ctx, const rt_sphere* spheres, uint32_t n_spheres, const rt_light* lights, uint32_t n_lights,
                    const rt_vertex* vertices, uint32_t n_vertices, const rt_triangle* triangles, uint32_t n_triangles,
                    const rt_material* materials, uint32_t n_materials, const rt_bvh_node* ref_nodes, uint32_t n_ref_nodes,
                    const uint32_t* ref_tri_indices, uint32_t n_ref_tri_indices) {
    if (!ctx) return RT_ERR_BAD_ARG;
    if ((n_spheres && !spheres) || (n_lights && !lights) || (n_vertices && !vertices) || (n_triangles && !triangles) ||
        (n_materials && !materials))
        return ctx->fail(RT_ERR_BAD_ARG, "rt_upload_scene: null array with non-zero count");
    // The reference BVH is accepted for interface compatibility and sanity-checked only.
    if (ref_nodes) {
        for (uint32_t i = 0; i < n_ref_nodes; i++) {
            const rt_bvh_node& n = ref_nodes[i];
            bool leaf = n.left_child == RT_INVALID_INDEX;
            if (!leaf && (n.left_child >= n_ref_nodes || (n.right_child != RT_INVALID_INDEX && n.right_child >= n_ref_nodes)))
                return ctx->fail(RT_ERR_BAD_ARG, "reference BVH node %u has a child out of range", i);
            if (leaf && ref_tri_indices && (uint64_t)n.triangle_start + n.triangle_count > n_ref_tri_indices)
                return ctx->fail(RT_ERR_BAD_ARG, "reference BVH leaf %u exceeds triangle_indices", i);
        }
    }
    std::vector<rt_triangle> tris(triangles, triangles + n_triangles);
    return upload_common(ctx, spheres, n_spheres, lights, n_lights, vertices, n_vertices, tris, {}, materials, n_materials);
}

int rt_upload_scene_packed(rt_ctx* ctx, const uint32_t* md, size_t n_u32, const rt_scene_metadata_offsets* off,
                           const rt_triangle* const tri_buffers[3], const uint32_t tri_counts[3], uint32_t triangles_per_buffer,
                           const rt_material* materials, uint32_t n_materials) {
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!off || (n_u32 && !md) || !tri_buffers || !tri_counts || (n_materials && !materials))
        return ctx->fail(RT_ERR_BAD_ARG, "rt_upload_scene_packed: null argument");
    auto section_ok = [&](uint32_t o, uint32_t count, uint32_t words) { return (uint64_t)o + (uint64_t)count * words <= n_u32; };
    if (!section_ok(off->spheres_offset, off->spheres_count, RT_SPHERE_WORDS) ||
        !section_ok(off->lights_offset, off->lights_count, RT_LIGHT_WORDS) ||
        !section_ok(off->bvh_nodes_offset, off->bvh_nodes_count, RT_BVH_NODE_WORDS) ||
        !section_ok(off->triangle_indices_offset, off->triangle_indices_count, 1) ||
        !section_ok(off->vertices_offset, off->vertices_count, RT_VERTEX_WORDS))
        return ctx->fail(RT_ERR_BAD_ARG, "rt_upload_scene_packed: a metadata section exceeds the buffer (%zu words)", n_u32);
    // Sections are plain reinterpretations of the host Vec<T> bytes (src/buffers.rs:213-234).
    const rt_sphere* spheres = reinterpret_cast<const rt_sphere*>(md + off->spheres_offset);
    const rt_light* lights = reinterpret_cast<const rt_light*>(md + off->lights_offset);
    const rt_vertex* vertices = reinterpret_cast<const rt_vertex*>(md + off->vertices_offset);
    std::vector<rt_triangle> tris;
    std::vector<uint32_t> prim_ids;
    for (int b = 0; b < 3; b++) {
        if (tri_counts[b] == 0) continue;
        if (!tri_buffers[b]) return ctx->fail(RT_ERR_BAD_ARG, "rt_upload_scene_packed: triangle buffer %d is null", b);
        if (triangles_per_buffer == 0 || tri_counts[b] > triangles_per_buffer)
            return ctx->fail(RT_ERR_BAD_ARG, "rt_upload_scene_packed: buffer %d holds %u > triangles_per_buffer %u", b, tri_counts[b],
                             triangles_per_buffer);
        for (uint32_t i = 0; i < tri_counts[b]; i++) {
            tris.push_back(tri_buffers[b][i]);
            prim_ids.push_back((uint32_t)b * triangles_per_buffer + i); // logical index, shader/src/triangle_access.rs:26-27
        }
    }
    return upload_common(ctx, spheres, off->spheres_count, lights, off->lights_count, vertices, off->vertices_count, tris, prim_ids,
                         materials, n_materials);
}

int rt_upload_textures(rt_ctx* ctx, const rt_texture_info* textures, uint32_t n_textures, const uint8_t* texture_data, size_t n_bytes) {
    if (!ctx) return RT_ERR_BAD_ARG;
    if ((n_textures && !textures) || (n_bytes && !texture_data)) return ctx->fail(RT_ERR_BAD_ARG, "rt_upload_textures: null array with non-zero count");
    for (uint32_t i = 0; i < n_textures; i++)
        if ((uint64_t)textures[i].offset + textures[i].size > n_bytes)
            return ctx->fail(RT_ERR_BAD_ARG, "rt_upload_textures: texture %u (offset %u, %u bytes) exceeds the %zu bytes of texture data", i,
                             textures[i].offset, textures[i].size, n_bytes);
    // main_cs binds both buffers and reads neither (`_textures`, `_texture_data`, shader/src/lib.rs:34-35): nothing goes to the device
    ctx->n_textures = n_textures;
    ctx->texture_bytes = n_bytes;
    ctx->stats.n_textures = n_textures;
    ctx->stats.texture_bytes = n_bytes;
    return RT_OK;
}

// Development aid (not part of rt_hip.h): make the next scene upload fail before its k-th device array, as an allocation failure would.
int rt_debug_fail_upload(rt_ctx* ctx, int k) {
    if (!ctx) return RT_ERR_BAD_ARG;
    ctx->fail_upload_at = k;
    return RT_OK;
}

// Development aid (not part of rt_hip.h): download the tree the context holds on its first device and validate it the way the
// kernels decode it (bvh_check.h: slots and masks, every finite triangle in exactly one leaf, conservative boxes, depth).
// out[0] nodes, [1] leaves, [2] reported depth, [3] real depth, [4] triangles placed exactly once, [5] build method, [6] / [7] FNV-1a
// hashes of the node and triangle arrays (the device build lays its tree out exactly as its host statement does); returns the number of failures.
int rt_debug_check_bvh(rt_ctx* ctx, uint32_t out[8]) {
    if (!ctx || !ctx->uploaded) return -1;
    DeviceState& d = ctx->devs[0];
    if (hipSetDevice(d.device) != hipSuccess) return -1;
    (void)hipStreamSynchronize(d.stream);
    rt::BvhBuild b;
    b.nodes.resize(ctx->scene_counts.n_nodes);
    b.tris.resize(ctx->scene_counts.n_tris);
    if (!b.nodes.empty() && hipMemcpy(b.nodes.data(), d.nodes, b.nodes.size() * sizeof(DevNode8), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (!b.tris.empty() && hipMemcpy(b.tris.data(), d.tris, b.tris.size() * sizeof(DevTri), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    b.depth = ctx->stats.bvh_depth;
    b.n_leaves = (uint32_t)(b.tris.size() / RT_DEV_LEAF_STRIDE);
    std::vector<uint32_t> seen(ctx->n_input_tris, 0);
    uint32_t real_depth = 0;
    size_t leaves = 0;
    rtcheck::g_print = true;
    int fails = rtcheck::check_tree(b, seen, &real_depth, &leaves);
    uint32_t once = 0;
    for (uint32_t v : seen) {
        once += v == 1 ? 1u : 0u;
        if (v > 1) fails++;
    }
    if (out) {
        out[0] = (uint32_t)b.nodes.size();
        out[1] = (uint32_t)leaves;
        out[2] = b.depth;
        out[3] = real_depth;
        out[4] = once;
        out[5] = (uint32_t)ctx->build_method;
        auto fnv = [](const void* p, size_t n) {
            uint32_t h = 2166136261u;
            const unsigned char* c = static_cast<const unsigned char*>(p);
            for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 16777619u;
            return h;
        };
        out[6] = fnv(b.nodes.data(), b.nodes.size() * sizeof(DevNode8));
        out[7] = fnv(b.tris.data(), b.tris.size() * sizeof(DevTri));
    }
    return fails;
}

// Development aids (not part of rt_hip.h): the queue allocation bound and the window rule, host-only arithmetic (tests/test_queue_bound.py).
unsigned long long rt_debug_queue_slots(unsigned long long max_entries, uint32_t per_lane, unsigned long long waves) {
    return rt::wf_queue_slots_for((size_t)max_entries, per_lane, (size_t)waves);
}
uint32_t rt_debug_pick_window(uint32_t iterations, uint32_t per_lane) { return rt::wf_pick_window(iterations, per_lane); }

// RT_PREPARE_QUALITY_TREE: the tree of the uploaded scene rebuilt by the host builder (bvh_builder.cpp method 0: multithreaded binned SAH +
// insertion-based optimisation + the same 8-slot collapse) in place of the one rt_upload_scene* built on the device in milliseconds.  For
// scenes that stay: 0.35 s per 262 k triangles, 4.8 s for 3.8 M; frames 2 % (sponza-like) to 9 % (bistro-like) faster, same images (closest
// hits do not depend on the tree).  The light grids hold triangle records in leaf order: they go with the old tree and are built again on demand.
static int prepare_quality_tree(rt_ctx* ctx) {
    if (ctx->build_method == 0 || ctx->build_tris.empty()) return RT_OK; // already the host builder's (tiny scenes, a fallback, an earlier call)
    for (auto& d : ctx->devs) {
        HIPCHK(ctx, hipSetDevice(d.device));
        HIPCHK(ctx, hipDeviceSynchronize());
    }
    rt::BvhBuild bvh;
    rt::BvhBuildOptions opt;
    opt.method = 0;
    rt::build_bvh(ctx->build_tris.data(), ctx->build_tris.size(), opt, bvh);
    if (bvh.depth > RT_DEV_MAX_BVH_DEPTH || bvh.nodes.size() > RT_DEV_MAX_NODES || bvh.nodes.empty())
        return ctx->fail(RT_ERR_INTERNAL, "rt_prepare: the host build gave %zu nodes at depth %u (the device-built tree stays)", bvh.nodes.size(), bvh.depth);
    // new arrays first, so that a failure leaves the old tree in place
    std::vector<DevNode8*> nn(ctx->devs.size(), nullptr);
    std::vector<DevTri*> nt(ctx->devs.size(), nullptr);
    int rc = RT_OK;
    for (size_t j = 0; j < ctx->devs.size() && rc == RT_OK; j++) {
        if (hipSetDevice(ctx->devs[j].device) != hipSuccess) rc = ctx->fail(RT_ERR_HIP, "rt_prepare: hipSetDevice failed");
        if (rc == RT_OK) rc = upload_array(ctx, &nn[j], bvh.nodes);
        if (rc == RT_OK) rc = upload_array(ctx, &nt[j], bvh.tris);
    }
    if (rc != RT_OK) {
        for (size_t j = 0; j < ctx->devs.size(); j++) {
            (void)hipSetDevice(ctx->devs[j].device);
            (void)hipFree(nn[j]);
            (void)hipFree(nt[j]);
        }
        return rc;
    }
    for (size_t j = 0; j < ctx->devs.size(); j++) {
        DeviceState& d = ctx->devs[j];
        (void)hipSetDevice(d.device);
        (void)hipDeviceSynchronize();
        (void)hipFree(d.nodes);
        (void)hipFree(d.tris);
        d.nodes = nn[j];
        d.tris = nt[j];
        for (void* p : d.grid_allocs) (void)hipFree(p);
        d.grid_allocs.clear();
        d.grid_info.clear();
        d.grids = nullptr;
        d.grids_tried = false;
        d.grids_partial = false;
    }
    DevScene& sc = ctx->scene_counts;
    sc.n_nodes = (uint32_t)bvh.nodes.size();
    sc.n_tris = (uint32_t)bvh.tris.size();
    sc.stack_entries = 2u * bvh.depth + 2u;
    ctx->stats.scene_bytes += (uint64_t)bvh.nodes.size() * sizeof(DevNode8) + (uint64_t)bvh.tris.size() * sizeof(DevTri) -
                              ((uint64_t)ctx->stats.bvh_nodes * sizeof(DevNode8) + (uint64_t)ctx->tree_tris_uploaded * sizeof(DevTri));
    ctx->tree_tris_uploaded = sc.n_tris;
    ctx->stats.bvh_nodes = sc.n_nodes;
    ctx->stats.bvh_depth = bvh.depth;
    ctx->stats.grid_bytes = 0;
    ctx->stats.grid_build_ms = 0.0;
    ctx->stats.tree_build = 0;
    ctx->build_method = 0;
    ctx->frame_valid = false;
    return RT_OK;
}

int rt_prepare(rt_ctx* ctx, uint32_t what) {
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!ctx->uploaded) return ctx->fail(RT_ERR_NOT_UPLOADED, "rt_prepare: no scene uploaded");
    const uint32_t known = RT_PREPARE_SHADOW_GRIDS | RT_PREPARE_QUALITY_TREE;
    if (what & ~known) return ctx->fail(RT_ERR_BAD_ARG, "rt_prepare: unknown bits 0x%x", what & ~known);
    if (int rcp = sync_pending(ctx)) return rcp;
    if (what & RT_PREPARE_QUALITY_TREE)
        if (int rc = prepare_quality_tree(ctx)) return rc;
    if (what & RT_PREPARE_SHADOW_GRIDS)
        for (auto& d : ctx->devs)
            if (int rc = ensure_grids(ctx, d)) return rc;
    return RT_OK;
}

int rt_render(rt_ctx* ctx, const rt_render_params* p) {
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!p) return ctx->fail(RT_ERR_BAD_ARG, "rt_render: null params");
    if (!ctx->uploaded) return ctx->fail(RT_ERR_NOT_UPLOADED, "rt_render: no scene uploaded");
    if (p->width == 0 || p->height == 0 || p->width > 65535u * 8u || p->height > 65535u * 8u)
        return ctx->fail(RT_ERR_BAD_ARG, "rt_render: bad resolution %ux%u", p->width, p->height);
    if (p->mode > RT_MODE_EXTENDED) return ctx->fail(RT_ERR_BAD_ARG, "rt_render: mode %u not supported", p->mode);
    if (p->mode == RT_MODE_EXTENDED && (p->spp == 0 || p->spp > 65536u))
        return ctx->fail(RT_ERR_BAD_ARG, "rt_render: spp %u out of range", p->spp);
    // bounce depths travel in 8 bits in the reference (pack_flags, shared/src/lib.rs:1154-1179); modes 0/1 mask as the
    // reference does, the extended mode refuses what it cannot represent rather than looping 2^32 times
    if (p->mode == RT_MODE_EXTENDED && p->max_bounces > RT_MAX_BOUNCES)
        return ctx->fail(RT_ERR_BAD_ARG, "rt_render: max_bounces %u > %u", p->max_bounces, RT_MAX_BOUNCES);
    uint32_t world = p->tile_world ? p->tile_world : 1, rank = p->tile_rank;
    if (rank >= world) return ctx->fail(RT_ERR_BAD_ARG, "rt_render: tile_rank %u >= tile_world %u", rank, world);
    DevFrame fr{};
    fr.width = p->width;
    fr.height = p->height;
    fr.tile_size = p->tile_size ? p->tile_size : RT_TILE_SIZE;
    if (fr.tile_size > 4096) return ctx->fail(RT_ERR_BAD_ARG, "rt_render: tile_size %u too large", fr.tile_size);
    fr.tiles_x = (fr.width + fr.tile_size - 1) / fr.tile_size; // TileHelper::calculate_tile_count, shared/src/lib.rs:1187-1191
    fr.tiles_y = (fr.height + fr.tile_size - 1) / fr.tile_size;
    fr.mode = p->mode;
    fr.channel_mask = 7u;
    fr.cur_bounce = 0; // the bounce-0 pass of src/compute.rs:413-479 (later passes redraw the same pixels)
    fr.max_bounce = p->mode == RT_MODE_EXTENDED ? p->max_bounces : (p->max_bounces & 0xFF);
    fr.spp = p->mode == RT_MODE_EXTENDED ? p->spp : 1;
    fr.flags = p->flags;
    fr.frame_seed = p->frame_seed;
    fr.cam = make_camera(p->camera, (float)p->width, (float)p->height, p->mode != RT_MODE_LEGACY);
    int rc = run_frame(ctx, fr, (p->flags & RT_FLAG_COUNTERS) != 0, world, rank, false);
    if (rc != RT_OK) return rc;
    ctx->frame_w = fr.width;
    ctx->frame_h = fr.height;
    ctx->frame_tile = fr.tile_size;
    ctx->frame_tiles_x = fr.tiles_x;
    ctx->frame_tiles_y = fr.tiles_y;
    ctx->frame_valid = true;
    return RT_OK;
}

int rt_dispatch_tile(rt_ctx* ctx, const rt_push_constants* pc) {
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!pc) return ctx->fail(RT_ERR_BAD_ARG, "rt_dispatch_tile: null push constants");
    if (!ctx->uploaded) return ctx->fail(RT_ERR_NOT_UPLOADED, "rt_dispatch_tile: no scene uploaded");
    uint32_t channel = pc->packed_flags & 0xFF;
    if (channel > 2) return ctx->fail(RT_ERR_BAD_ARG, "rt_dispatch_tile: invalid color channel %u", channel); // src/renderer.rs:769-776
    auto as_u32 = [](float f) -> uint32_t { return !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)f); };
    uint32_t w = as_u32(pc->resolution[0]), h = as_u32(pc->resolution[1]);
    if (w == 0 || h == 0 || w > 65535u * 8u || h > 65535u * 8u)
        return ctx->fail(RT_ERR_BAD_ARG, "rt_dispatch_tile: bad resolution %gx%g", pc->resolution[0], pc->resolution[1]);
    uint32_t mode = (pc->packed_flags >> 24) & 0xFF;
    DevFrame fr{};
    fr.width = w;
    fr.height = h;
    fr.single_tile = 1;
    fr.tile_off_x = pc->tile_offset[0];
    fr.tile_off_y = pc->tile_offset[1];
    fr.tile_w = pc->tile_size_packed & 0xFFFF; // unpack_tile_size, shared/src/lib.rs:1146-1150
    fr.tile_h = (pc->tile_size_packed >> 16) & 0xFFFF;
    fr.tile_size = std::max(1u, std::max(fr.tile_w, fr.tile_h));
    fr.tiles_x = fr.tiles_y = 1;
    fr.mode = mode ? RT_MODE_WAVEFRONT : RT_MODE_LEGACY;
    fr.channel_mask = 1u << channel;
    fr.cur_bounce = (pc->packed_flags >> 8) & 0xFF;
    fr.max_bounce = (pc->packed_flags >> 16) & 0xFF;
    fr.spp = 1;
    fr.frame_seed = pc->frame_seed;
    fr.cam = make_camera(pc->camera, pc->resolution[0], pc->resolution[1], mode != 0);
    // Only texture `channel` receives texels (the other two keep theirs, as separate bind groups do);
    // the float and hit targets of the tile are refreshed as a by-product.
    DeviceState& d = ctx->devs[0];
    int rc = consolidate_on_first_device(ctx, w, h);
    if (rc != RT_OK) return rc;
    rc = run_frame(ctx, fr, false, 1, 0, true);
    if (rc != RT_OK) return rc;
    ctx->frame_w = w;
    ctx->frame_h = h;
    ctx->frame_tile = RT_TILE_SIZE;
    ctx->frame_tiles_x = (w + RT_TILE_SIZE - 1) / RT_TILE_SIZE;
    ctx->frame_tiles_y = (h + RT_TILE_SIZE - 1) / RT_TILE_SIZE;
    ctx->frame_valid = true;
    // a dispatch sequence is owned entirely by device 0
    d.tile_first = 0;
    d.tile_stride = 1;
    d.n_owned = ctx->frame_tiles_x * ctx->frame_tiles_y;
    for (size_t j = 1; j < ctx->devs.size(); j++) ctx->devs[j].n_owned = 0; // whatever they own dates from an earlier rt_render
    return RT_OK;
}

// Copy `elem` bytes per pixel of every device's owned tiles into `out` (full frame, row-major).
namespace {
int gather(rt_ctx* ctx, uint8_t* out, size_t elem, int which /*0 rgba32f, 1..3 chan, 4 prim, 5 t*/) {
    uint32_t w = ctx->frame_w, h = ctx->frame_h;
    size_t n = (size_t)w * h;
    std::vector<uint8_t> tmp;
    for (size_t j = 0; j < ctx->devs.size(); j++) {
        DeviceState& d = ctx->devs[j];
        if (d.fb_w != w || d.fb_h != h || d.n_owned == 0) continue;
        const void* src = which == 0 ? (const void*)d.rgba32f : which <= 3 ? (const void*)d.chan[which - 1] : which == 4 ? (const void*)d.prim_id : (const void*)d.hit_t;
        HIPCHK(ctx, hipSetDevice(d.device));
        bool all = d.tile_stride == 1 && d.tile_first == 0;
        if (all) {
            HIPCHK(ctx, hipMemcpy(out, src, n * elem, hipMemcpyDeviceToHost));
            continue;
        }
        tmp.resize(n * elem);
        HIPCHK(ctx, hipMemcpy(tmp.data(), src, n * elem, hipMemcpyDeviceToHost));
        uint32_t ts = ctx->frame_tile;
        for (uint32_t k = 0; k < d.n_owned; k++) {
            uint32_t tile = d.tile_first + k * d.tile_stride;
            uint32_t ty = tile / ctx->frame_tiles_x, tx = tile % ctx->frame_tiles_x;
            uint32_t x0 = tx * ts, y0 = ty * ts, tw = std::min(ts, w - x0), th = std::min(ts, h - y0);
            for (uint32_t y = y0; y < y0 + th; y++)
                std::memcpy(out + ((size_t)y * w + x0) * elem, tmp.data() + ((size_t)y * w + x0) * elem, (size_t)tw * elem);
        }
    }
    return RT_OK;
}
} // namespace

// Whole frame on one device: run the epilogue there and bring the result back through pinned staging
// (`which` 0: packed rgb32f, 1: combined rgba8).  Returns RT_OK, an error, or 1 when the caller must use the gather path.
static int read_epilogue(rt_ctx* ctx, int which, void* out, size_t bytes) {
    if (ctx->devs.size() != 1) return 1;
    DeviceState& d = ctx->devs[0];
    if (d.fb_w != ctx->frame_w || d.fb_h != ctx->frame_h || !(d.tile_stride == 1 && d.tile_first == 0)) return 1;
    HIPCHK(ctx, hipSetDevice(d.device));
    if (d.readback_bytes < bytes) {
        (void)hipFree(d.readback_dev);
        if (d.readback_host) (void)hipHostFree(d.readback_host);
        d.readback_dev = d.readback_host = nullptr;
        d.readback_bytes = 0;
        HIPCHK(ctx, hipMalloc(&d.readback_dev, bytes));
        HIPCHK(ctx, hipHostMalloc(&d.readback_host, bytes, hipHostMallocDefault));
        d.readback_bytes = bytes;
    }
    const size_t n = (size_t)ctx->frame_w * ctx->frame_h;
    if (which == 0) HIPCHK(ctx, rt::launch_pack_rgb32f(d.rgba32f, (float*)d.readback_dev, n, d.stream));
    else HIPCHK(ctx, rt::launch_combine_rgba8(d.chan[0], d.chan[1], d.chan[2], (uint8_t*)d.readback_dev, n, d.stream));
    HIPCHK(ctx, hipMemcpyAsync(d.readback_host, d.readback_dev, bytes, hipMemcpyDeviceToHost, d.stream));
    HIPCHK(ctx, hipStreamSynchronize(d.stream));
    std::memcpy(out, d.readback_host, bytes);
    return RT_OK;
}

int rt_read_rgb32f(rt_ctx* ctx, float* out, size_t n_floats) {
    if (!ctx) return RT_ERR_BAD_ARG;
    if (int rcp = sync_pending(ctx)) return rcp;
    if (!ctx->frame_valid) return ctx->fail(RT_ERR_NOT_UPLOADED, "rt_read_rgb32f: nothing rendered yet");
    size_t n = (size_t)ctx->frame_w * ctx->frame_h;
    if (!out || n_floats != n * 3) return ctx->fail(RT_ERR_BAD_ARG, "rt_read_rgb32f: expected %zu floats, got %zu", n * 3, n_floats);
    int rc = read_epilogue(ctx, 0, out, n * 12);
    if (rc <= 0) return rc;
    std::vector<float> tmp(n * 4, 0.0f);
    rc = gather(ctx, reinterpret_cast<uint8_t*>(tmp.data()), 16, 0);
    if (rc != RT_OK) return rc;
    for (size_t i = 0; i < n; i++) {
        out[3 * i + 0] = tmp[4 * i + 0];
        out[3 * i + 1] = tmp[4 * i + 1];
        out[3 * i + 2] = tmp[4 * i + 2];
    }
    return RT_OK;
}

int rt_read_rgba8_channels(rt_ctx* ctx, uint8_t* red, uint8_t* green, uint8_t* blue, size_t n_bytes_each) {
    if (!ctx) return RT_ERR_BAD_ARG;
    if (int rcp = sync_pending(ctx)) return rcp;
    if (!ctx->frame_valid) return ctx->fail(RT_ERR_NOT_UPLOADED, "rt_read_rgba8_channels: nothing rendered yet");
    size_t n = (size_t)ctx->frame_w * ctx->frame_h * 4;
    if (n_bytes_each != n) return ctx->fail(RT_ERR_BAD_ARG, "rt_read_rgba8_channels: expected %zu bytes each, got %zu", n, n_bytes_each);
    uint8_t* outs[3] = {red, green, blue};
    for (int c = 0; c < 3; c++) {
        if (!outs[c]) continue;
        std::memset(outs[c], 0, n);
        int rc = gather(ctx, outs[c], 4, 1 + c);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

int rt_read_rgba8_combined(rt_ctx* ctx, uint8_t* out, size_t n_bytes) {
    if (!ctx) return RT_ERR_BAD_ARG;
    if (int rcp = sync_pending(ctx)) return rcp;
    if (!ctx->frame_valid) return ctx->fail(RT_ERR_NOT_UPLOADED, "rt_read_rgba8_combined: nothing rendered yet");
    size_t n = (size_t)ctx->frame_w * ctx->frame_h * 4;
    if (!out || n_bytes != n) return ctx->fail(RT_ERR_BAD_ARG, "rt_read_rgba8_combined: expected %zu bytes, got %zu", n, n_bytes);
    int rc = read_epilogue(ctx, 1, out, n);
    if (rc <= 0) return rc;
    std::vector<uint8_t> r(n), g(n), b(n);
    rc = rt_read_rgba8_channels(ctx, r.data(), g.data(), b.data(), n);
    if (rc != RT_OK) return rc;
    for (size_t i = 0; i < n; i += 4) { // main_fs, shader/src/lib.rs:383-388
        out[i + 0] = r[i + 0];
        out[i + 1] = g[i + 1];
        out[i + 2] = b[i + 2];
        out[i + 3] = 255;
    }
    return RT_OK;
}

int rt_read_hits(rt_ctx* ctx, uint32_t* prim_ids, float* t, size_t n_pixels) {
    if (!ctx) return RT_ERR_BAD_ARG;
    if (int rcp = sync_pending(ctx)) return rcp;
    if (!ctx->frame_valid) return ctx->fail(RT_ERR_NOT_UPLOADED, "rt_read_hits: nothing rendered yet");
    size_t n = (size_t)ctx->frame_w * ctx->frame_h;
    if (n_pixels != n) return ctx->fail(RT_ERR_BAD_ARG, "rt_read_hits: expected %zu pixels, got %zu", n, n_pixels);
    int rc;
    if (prim_ids) {
        std::memset(prim_ids, 0xFF, n * 4);
        if ((rc = gather(ctx, reinterpret_cast<uint8_t*>(prim_ids), 4, 4)) != RT_OK) return rc;
    }
    if (t) {
        std::memset(t, 0, n * 4);
        if ((rc = gather(ctx, reinterpret_cast<uint8_t*>(t), 4, 5)) != RT_OK) return rc;
    }
    return RT_OK;
}

// Development aid (not part of rt_hip.h): wave-level diagnostics of the last RT_FLAG_COUNTERS render of the
// state-machine kernel: transition passes, lanes served, node iterations, lanes active, leaf iterations, lanes
// active, cycles in transition phases, cycles in traversal phases (summed over waves).
// Development aid: the camera beams of the last extended-mode frame on the first device: per owned 8x8 pixel block the length of its
// triangle list (0xFFFFFFFF: no list).  Returns the number of blocks (<= n), negative on error.
int rt_debug_beams(rt_ctx* ctx, uint32_t* counts, uint32_t n) {
    if (!ctx || ctx->devs.empty()) return RT_ERR_BAD_ARG;
    DeviceState& d = ctx->devs[0];
    if (!d.wf.beam_count) return 0;
    if (hipSetDevice(d.device) != hipSuccess) return RT_ERR_HIP;
    (void)hipStreamSynchronize(d.stream);
    const uint32_t m = std::min(n, d.wf.n_blocks);
    if (m && hipMemcpy(counts, d.wf.beam_count, (size_t)m * 4, hipMemcpyDeviceToHost) != hipSuccess) return RT_ERR_HIP;
    for (uint32_t i = 0; i < m; i++)
        if (counts[i] & RT_BEAM_OVERFLOW) counts[i] = 0xFFFFFFFFu;
    return (int)m;
}

// Development aid / bench.py: with RT_FLAG_STAGE_TIMES, the launches of the frame's dominant stage kernel (k_wf_shadow_grid) on the first
// device, timed with HIP events on the stream they were launched on: out[0] sum of their durations in ms, out[1] their number.
int rt_debug_stage_times(rt_ctx* ctx, double out[2]) {
    if (!ctx || !out) return RT_ERR_BAD_ARG;
    out[0] = ctx->stage_ms[0];
    out[1] = ctx->stage_ms[1];
    return RT_OK;
}

int rt_debug_counters(rt_ctx* ctx, unsigned long long out[8]) {
    if (!ctx || !out) return RT_ERR_BAD_ARG;
    for (int k = 0; k < 8; k++) out[k] = ctx->diag[k];
    return RT_OK;
}

// Development aid: what the light grids (shadow_grid.h) of the first device look like.  light < n_lights: out = {kind (0: refused), cells per
// side, entries, near-list length, longest list, cells left to the BVH, cells with a list, 0}; light == 0xFFFFFFFF: out = {lights with a grid, all
// entries, bytes, shadow segments the grids answered in the last frame rendered with RT_FLAG_COUNTERS, list entries read, 0, 0, 0}.
int rt_debug_shadow_grid(rt_ctx* ctx, uint32_t light, unsigned long long out[8]) {
    if (!ctx || !out) return RT_ERR_BAD_ARG;
    for (int k = 0; k < 8; k++) out[k] = 0;
    if (ctx->devs.empty()) return RT_OK;
    const DeviceState& d = ctx->devs[0];
    if (light == 0xFFFFFFFFu) {
        for (const rt::ShadowGridBuild& g : d.grid_info) {
            if (g.grid.kind == RT_SG_KIND_NONE) continue;
            out[0]++;
            out[1] += g.n_entries;
            out[2] += g.bytes;
        }
        out[3] = ctx->grid_diag[0];
        out[4] = ctx->grid_diag[1];
        return RT_OK;
    }
    if (light >= d.grid_info.size()) return RT_OK;
    const rt::ShadowGridBuild& g = d.grid_info[light];
    out[0] = g.grid.kind;
    out[1] = g.grid.res;
    out[2] = g.n_entries;
    out[3] = g.near_count;
    out[4] = g.longest;
    out[5] = g.heavy_cells;
    out[6] = g.filled_cells;
    return RT_OK;
}

int rt_get_stats(rt_ctx* ctx, rt_stats* out) {
    if (!ctx) return RT_ERR_BAD_ARG;
    if (!out) return ctx->fail(RT_ERR_BAD_ARG, "rt_get_stats: null out");
    if (int rcp = sync_pending(ctx)) return rcp;
    *out = ctx->stats;
    return RT_OK;
}

} // extern "C"
