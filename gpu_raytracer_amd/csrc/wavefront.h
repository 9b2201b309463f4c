// wavefront.h — the queue-based ("wavefront") pipeline of the extended mode.
//
// The reference declares a wavefront path tracer (WavefrontRay / WavefrontCounters, shared/src/lib.rs:165-194,
// host loop src/compute.rs:413-479) but implements neither queues nor continuation rays.  This is that design
// built for MI355X: path state and ray queues live in HBM (288 GB makes millions of paths in flight cheap),
// each stage is its own kernel so every wave runs ONE code path, queue slots are handed out with
// wave-aggregated atomics (ballot + prefix count), and the two traversal stages are persistent kernels whose
// idle lanes refill themselves from the queue.
#ifndef RT_WAVEFRONT_H
#define RT_WAVEFRONT_H

#include <hip/hip_runtime.h>

#include "device_layout.h"
#include "shadow_grid.h"

namespace rt {

// counters[] slots (device memory, uint32)
enum WfCounter : uint32_t {
    WF_EXT_COUNT = 0,    // entries in the current extension queue
    WF_EXT_NEXT = 1,     // entries appended to the next extension queue
    WF_SHADOW2_COUNT = 2, // shadow segments the light grids handed on to the BVH traversal (dense: no windows)
    WF_SHADOW_COUNT = 3, // shadow segments of this iteration
    WF_EXT_CURSOR = 4,   // persistent-kernel fetch cursors
    WF_SHADOW_CURSOR = 5,
    WF_EXT_WINDOW = 6,      // reservation window (a power of two) the current extension queue was written with
    WF_SHADOW_WINDOW = 7,   // ... the shadow queue
    WF_EXT_WINDOW_NEXT = 8, // ... the next extension queue
    WF_SHADOW2_CURSOR = 9,
    WF_SHADOW2_WINDOW = 10, // stays 0: the handed-on queue is written densely
    WF_UNUSED_11 = 11,
    WF_FB_COUNT = 12,       // camera segments of pixel blocks without a beam list (they walk the tree): entries in q_ext[1] at depth 0, dense
    WF_FB_CURSOR = 13,
    WF_N_COUNTERS = 16
};

struct WfBuffers {
    // per path slot (P = samples_in_batch * n_blocks * 64)
    float4* ray_o;       // xyz origin
    float4* ray_d;       // xyz direction (also the incoming direction of the current vertex)
    uint4* hit;          // xyz = hit point bits, w = RT_PRIM_MISS | RT_PRIM_SPHERE_FLAG + sphere index | triangle slot
    float4* thr;         // xyz throughput, w = bits: hero channel | depth << 8
    float4* rad;         // xyz radiance of the sample so far, w = bits: rng state
    float4* vtx;         // two per path, always used together, so one 32-byte record = one cache line per gather:
                         //   [2 id]     xyz vertex position, w = bits: material id
                         //   [2 id + 1] xyz geometric normal, w = bits: the lights with a shadow segment from the current vertex (k_wf_shade), less those the shadow stage found occluded
    float4* sample_rad;  // xyz final radiance of the sample (written when the path ends)
    uint32_t* pxy;       // x | y << 16, 0xFFFFFFFF = no pixel (tile edge)
    // queues
    uint32_t* q_ext[2];  // path ids to extend (double buffered)
    uint32_t* q_shadow;  // path id | light << 27 (frames without light grids; with them the shadow stage goes by vertex, over q_ext)
    uint32_t* q_shadow2; // the same entries, those the light grids (shadow_grid.h) leave to the BVH; dense
    const DevShadowGrid* grids; // one per light, or null: every shadow segment walks the BVH
    uint32_t* counters;  // WfCounter
    unsigned long long* totals; // [0] camera [1] continuation [2] shadow segments, [3] node visits, [4] triangle tests, [13] shadow segments answered by a light grid, [14] list entries they read, [15] error word
    float4* accum;       // per owned pixel slot: running sum over samples (in sample order)
    uint32_t q_ext_cap;  // slots allocated for each extension queue / the shadow queue: a window reservation that would
    uint32_t q_shadow_cap; // end beyond it raises totals[WF_TOTAL_ERROR] instead of writing (window_reserve)
    uint32_t* stack_ovf; // global overflow part of the traversal stacks: [persistent wave][entry][lane], 64-bit entries
    uint32_t ovf_entries; // entries per lane in it
    uint32_t n_blocks;   // 8x8 pixel blocks owned by this device
    uint32_t batch;      // samples per pixel in flight
    uint32_t capacity;   // path slots
    // camera beams (round 3): per owned 8x8 pixel block the leaves its frustum touches, nearest first; the camera segments of the block
    // test them directly instead of walking the tree (k_wf_beams / k_wf_trace_camera).  Null: every camera segment walks the tree.
    uint32_t* beam_count; // [n_blocks] triangles in the block's list, or RT_BEAM_OVERFLOW | j: no list (too many: its segments walk the tree); [n_blocks]: blocks without a list
    uint32_t* beam_ref;   // [n_blocks][RT_BEAM_CAP] triangle records (index into DevScene::tris)
    float* beam_dist;     // [n_blocks][RT_BEAM_CAP] a lower bound of the distance from the camera to the triangle, ascending
    uint32_t probe;      // development probes of the counting kernel variants (RT_WF_PROBE), 0 otherwise
};

#ifndef RT_BEAM_CAP
#define RT_BEAM_CAP 128u /* triangles per block list: 90 % of the headline frame's blocks need fewer (64 / 192 / 256 measured in profiles/ab_r03.json) */
#endif
#define RT_BEAM_OVERFLOW 0x80000000u /* beam_count[b]: this bit = no list; the low bits then number the blocks without a list (their place in the walk queue) */
#define WF_TOTAL_ERROR 15 /* totals[] slot: non-zero = a queue reservation did not fit; every later stage kernel of the frame returns at once */
#ifndef RT_WF8_LDS_STACK
#define RT_WF8_LDS_STACK 8 /* 64-bit traversal stack entries per lane kept in LDS by the persistent kernels; deeper ones overflow to HBM */
#endif
#define RT_WF_MAX_LIGHTS 32u /* visibility is one bit per light in a 32-bit word */
#define RT_WF_ID_MASK 0x07FFFFFFu

uint32_t wf_shading_blocks(); // grid size (256-thread blocks) of the generate / shade / finish kernels
size_t wf_queue_slots(size_t max_entries, uint32_t per_lane); // allocation bound of a queue holding up to max_entries real entries, written with up to per_lane entries per lane and iteration
size_t wf_queue_slots_for(size_t max_entries, uint32_t per_lane, size_t producing_waves); // the same for a given number of producing waves (host-only arithmetic, unit-tested)
uint32_t wf_pick_window(uint32_t iterations, uint32_t per_lane); // the reservation window a producing wave uses (host copy of the device rule, unit-tested)
uint32_t wf_persistent_waves(); // grid size (in 64-lane blocks) of the persistent traversal kernels on the current device
hipError_t wf_beams(const DevScene& sc, const DevFrame& fr, const WfBuffers& wb, hipStream_t s); // once per frame and device, before the batches
hipError_t wf_generate(const DevScene& sc, const DevFrame& fr, const WfBuffers& wb, uint32_t first_sample, uint32_t n_samples, hipStream_t s);
hipError_t wf_bounce(const DevScene& sc, const DevFrame& fr, const WfBuffers& wb, uint32_t iteration, uint32_t n_samples, bool counters, hipStream_t s,
                     hipEvent_t* grid_events = nullptr); // n_samples: samples per pixel in this batch (iteration 0); grid_events: two events recorded around the k_wf_shadow_grid launch
hipError_t wf_resolve(const DevFrame& fr, const WfBuffers& wb, const DevTargets& tg, uint32_t n_samples, bool first_batch, bool last_batch, hipStream_t s);

} // namespace rt
#endif
