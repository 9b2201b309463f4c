// wavefront.hip — queue-based pipeline of the extended mode (see wavefront.h).
//
// Stages per bounce iteration (each a kernel, one code path per wave):
//   k_wf_trace<closest>  persistent; lanes pull path ids from the extension queue, walk the BVH, store the hit
//   k_wf_shade           consume the hit: miss / invalid material end the path; otherwise store the vertex and
//                        enqueue one shadow segment per light with a non-zero contribution
//   k_wf_shadow_grid     shadow segments of lights that have a triangle-list grid (shadow_grid.h): look up the segment's cell as seen
//                        from the light, test the cell's few triangles; sets a visibility bit per (path, light); what it cannot
//                        decide cheaply goes on to
//   k_wf_trace<any hit>  persistent; shadow segments walk the BVH and set the same visibility bits
//   k_wf_finish          sum the visible contributions IN LIGHT ORDER, add emission, then terminal shading or
//                        continuation sampling + russian roulette; survivors go to the next extension queue
//   k_wf_advance         one thread: queue sizes move on, cursors are reset
// Per-path arithmetic and its order are exactly those of the CPU statement (and of the megakernels in
// kernels.hip); queues only change WHEN a path's next step runs, so images stay bit-identical.
#include "wavefront.h"
#include <cstdlib>

#include <algorithm>

#include "device_common.h"

using namespace rtdev;

#ifndef RT_WF_REFILL
#define RT_WF_REFILL 16 /* idle lanes in a wave before it fetches new segments */
#endif
#ifndef RT_WF_CHUNK
#define RT_WF_CHUNK 256 /* queue entries a wave claims per atomic (64 / 128: -37 % / -6 %, cursor atomics; 1024 / 2048 / 8192: -1.5 / -5 / -12 %: small chunks keep the
                           waves on neighbouring parts of the queue, i.e. on neighbouring rays) */
#endif
#ifndef RT_WF_ADAPTIVE_CHUNK
#define RT_WF_ADAPTIVE_CHUNK 1
#endif
#ifndef RT_WF_WINDOW
#define RT_WF_WINDOW 512 /* queue slots a producing wave reserves per atomic */
#endif
#ifndef RT_WF_BLOCK_MAJOR
#define RT_WF_BLOCK_MAJOR 1 /* path slots: all samples of a pixel block adjacent (1) or all blocks of a sample adjacent (0) */
#endif
#ifndef RT_WF_SHADE_WAVES
#define RT_WF_SHADE_WAVES 4 /* waves per SIMD the shading stages are register-allocated for */
#endif
#ifndef RT_WF_WAVES_PER_CU
#define RT_WF_WAVES_PER_CU 24
#endif
#ifndef RT_WF_WAVES_SMALL
#define RT_WF_WAVES_SMALL 12u /* of RT_WF_WAVES_PER_CU: what a launch over a batch below 24 Mi path slots uses: fewer, longer-lived waves drain a short queue with less tail (ab_r03: waves_per_cu_by_batch) */
#endif
#ifndef RT_WF_SHADOW_LIGHT_MAJOR
#define RT_WF_SHADOW_LIGHT_MAJOR 1
#endif

namespace {

__device__ __forceinline__ V3 f4v(float4 a) { return v3(a.x, a.y, a.z); }
#define RT_KEEP4_EARLY(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w)) /* (RT_KEEP4, usable before its definition below) */

// Queue output through wave-private windows.  A single counter sustains ~90 M atomics/s (MI355X_MICROARCH.md,
// "dequeue"): one atomic per wave-iteration made the shading stages atomic-bound, and so did fixed 512-slot
// windows once batches grew (160 M shadow entries = 312 k atomics = 3.5 ms of a 5 ms launch).  A producing wave
// therefore reserves a window sized from the launch: a quarter of what it can emit in total, as a power of two
// between the minimum and RT_WF_WINDOW_MAX, i.e. about four atomics per wave and launch.  What a wave does not
// use is filled with WF_SENTINEL entries.  All windows of a launch have the same size W and start at multiples of
// W, and a window's sentinels are always its tail, so a consumer that meets a sentinel skips to the next multiple
// of W (k_wf_trace); W travels through the counters.  Real entries are tallied separately (totals) because
// queue lengths include the padding.
#ifndef RT_WF_WINDOW_MAX
#define RT_WF_WINDOW_MAX 16384u
#endif
__host__ __device__ inline uint32_t wf_min_window(uint32_t per_lane) { // >= 2 x the largest request of one wave iteration
    uint32_t w = RT_WF_WINDOW;
    while (w < 128u * per_lane) w <<= 1;
    return w;
}
// iters: iterations a producing wave will run at most; per_lane: entries a lane can emit per iteration
__host__ __device__ inline uint32_t pick_window(uint32_t iters, uint32_t per_lane) {
    const uint32_t most = iters * WAVE * per_lane;
    uint32_t w = wf_min_window(per_lane);
    while (w < most / 4u && w < RT_WF_WINDOW_MAX) w <<= 1;
    return w;
}
struct OutWindow {
    uint32_t next, end; // wave-uniform
};
#define WF_SENTINEL 0xFFFFFFFFu
__device__ __forceinline__ void window_close(uint32_t* __restrict__ queue, OutWindow& w) {
    for (uint32_t i = w.next + (threadIdx.x & 63u); i < w.end; i += WAVE) queue[i] = WF_SENTINEL;
    w.next = w.end;
}
#define WF_NO_SLOT 0xFFFFFFFFu
// returns the first slot for this lane; `mine` entries per lane, `incl` = inclusive prefix of `mine` over the wave.
// A reservation that would end beyond the queue's allocation (`capacity` slots) is not used: the error word is raised,
// the caller gets WF_NO_SLOT and must not write, every later stage kernel of the frame returns at once and the host
// reports RT_ERR_INTERNAL (the allocation bound wf_queue_slots is derived below; this is the belt to its braces).
__device__ __forceinline__ uint32_t window_reserve(uint32_t* __restrict__ queue, uint32_t* __restrict__ counter, OutWindow& w, uint32_t window,
                                                  uint32_t mine, uint32_t incl, uint32_t total, uint32_t capacity, unsigned long long* __restrict__ error) {
    if (w.next + total > w.end) {
        window_close(queue, w);
        uint32_t base = 0;
        if ((threadIdx.x & 63u) == 0) base = atomicAdd(counter, window); // total <= window / 2 by construction (wf_min_window)
        base = __shfl(base, 0, WAVE);
        if (base > capacity || window > capacity - base) { // wave-uniform
            if ((threadIdx.x & 63u) == 0) atomicOr(error, 1ull);
            w.next = w.end = 0u;
            return WF_NO_SLOT;
        }
        w.next = base;
        w.end = base + window;
    }
    const uint32_t at = w.next + incl - mine;
    w.next += total;
    return at;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t u = __shfl_up(v, off, WAVE);
        if ((int)lane >= off) v += u;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// generation: one lane per path slot (sample k of pixel (block b, lane l))
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wf_generate(DevFrame fr, rt::WfBuffers wb, uint32_t first_sample, uint32_t n_slots_blocks) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * 4u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    OutWindow win = {0u, 0u};
    uint32_t real = 0;
    const uint32_t window = pick_window((n_slots_blocks + n_waves - 1u) / n_waves, 1u);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        wb.counters[rt::WF_EXT_WINDOW] = window;
        if (wb.beam_count) wb.counters[rt::WF_FB_COUNT] = wb.beam_count[wb.n_blocks] * (n_slots_blocks / max(1u, wb.n_blocks)) * WAVE; // blocks without a list x samples x 64
    }
    // path slot p = (b * n_samples + k) * 64 + lane: all samples of an 8x8 pixel block are neighbours in the queue, so
    // the segments in flight at any moment (queue order survives compaction) come from a small part of the image
    const uint32_t n_samples = n_slots_blocks / max(1u, wb.n_blocks);
    for (uint32_t sb = wave; sb < n_slots_blocks; sb += n_waves) {
#if RT_WF_BLOCK_MAJOR
        const uint32_t b = sb / n_samples, k = sb - b * n_samples; // sb = b * n_samples + k
#else
        const uint32_t k = sb / wb.n_blocks, b = sb - k * wb.n_blocks; // sb = k * n_blocks + b
#endif
        const PixelCoord px = block_pixel_at(fr, b, lane);
        const uint32_t p = sb * WAVE + lane;
        if (px.valid) {
            SimpleRng rng = rng_for(fr.frame_seed + px.x + px.y * fr.width, first_sample + k);
            float jx = 0.5f, jy = 0.5f;
            if (fr.spp > 1) {
                jx = rng.next_f32();
                jy = rng.next_f32();
            }
            V3 o, d;
            camera_ray(fr.cam, (float)px.x + jx, (float)px.y + jy, true, o, d);
            wb.ray_o[p] = make_float4(o.x, o.y, o.z, 0.0f);
            wb.ray_d[p] = make_float4(d.x, d.y, d.z, 0.0f);
            wb.thr[p] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(3u)); // channel 3 (none), depth 0
            wb.rad[p] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(rng.seed));
            wb.pxy[p] = px.x | (px.y << 16);
        } else {
            wb.pxy[p] = 0xFFFFFFFFu;
            wb.sample_rad[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        const uint32_t mine = px.valid ? 1u : 0u;
        const uint32_t incl = wave_incl_scan(mine), total = __shfl(incl, WAVE - 1, WAVE);
        if (total) {
            const uint32_t at = window_reserve(wb.q_ext[0], &wb.counters[rt::WF_EXT_COUNT], win, window, mine, incl, total, wb.q_ext_cap, &wb.totals[WF_TOTAL_ERROR]);
            if (mine && at != WF_NO_SLOT) wb.q_ext[0][at] = p;
            real += mine;
            // camera segments of a block without a beam list walk the tree in the persistent kernel: a second, dense queue in the buffer
            // that becomes the next extension queue only after this depth's traversal (k_wf_finish writes it from position 0 then)
        }
        // camera segments of a block without a beam list walk the tree in the persistent kernel: a second queue, in the buffer that
        // becomes the next extension queue only after this depth's traversal.  No atomics: the j-th block without a list owns the
        // entries [(j n_samples + k) 64, + 64), lanes without a pixel leave a sentinel
        if (wb.beam_count) {
            const uint32_t bc = wb.beam_count[b];
            if (bc & RT_BEAM_OVERFLOW) wb.q_ext[1][((size_t)(bc & ~RT_BEAM_OVERFLOW) * n_samples + k) * WAVE + lane] = px.valid ? p : WF_SENTINEL;
        }
    }
    window_close(wb.q_ext[0], win);
    const unsigned long long r = wave_sum(real);
    if (lane == 0 && r) atomicAdd(&wb.totals[0], r);
}

// ---------------------------------------------------------------------------------------------------------
// persistent traversal kernel (closest hit for extension segments, any hit for shadow segments) on the 8-wide tree.
// A wave owns a private chunk of 256 queue entries; when >= RT_WF_REFILL lanes are idle they take the next entries
// (__ballot + prefix count of the idle mask, no atomic until the chunk is used up).  Traversal is speculative with one
// postponed leaf GROUP per lane: what is left of a visited node is a group (base, hit mask | kind mask << 8) - G for its
// inner children, T for its leaves; a lane keeps descending through G while T waits, triangle tests run when
// RT_WF8_LEAF_THRESHOLD lanes hold a T (result-neutral: ties resolve by triangle index, not by visiting order).  A visit
// parks at most two 64-bit entries; children are taken from a group in increasing (slot XOR ray octant) through a 2 KB
// table in LDS (device_common.h first_slot computes the same function arithmetically for the other kernels).
// ---------------------------------------------------------------------------------------------------------
#ifndef RT_WF8_LDS_STACK
#define RT_WF8_LDS_STACK 8 /* 64-bit entries per lane kept in LDS; deeper ones go to the HBM overflow area */
#endif
#ifndef RT_WF8_MIN_WAVES
#define RT_WF8_MIN_WAVES 6 /* 80 VGPRs: measured, 82 VGPRs (5 waves per SIMD) cost 4 % */
#endif
#ifndef RT_WF8_LEAF_THRESHOLD
#define RT_WF8_LEAF_THRESHOLD 24
#endif
#define WF8_KIND_T 0x80000000u
#define WF8_NONE 0xFFFFFFFFu

template <bool COUNT, bool ANY>
__global__ __launch_bounds__(WAVE, RT_WF8_MIN_WAVES) void k_wf_trace(DevScene sc, rt::WfBuffers wb, const uint32_t* __restrict__ queue, uint32_t count_slot,
                                                    uint32_t cursor_slot, uint32_t window_slot) {
    extern __shared__ uint32_t s_mem8[];
    if (wb.totals[WF_TOTAL_ERROR] != 0ull) return;
    const uint32_t lane = threadIdx.x;
    uint2* __restrict__ stack = reinterpret_cast<uint2*>(s_mem8) + lane;                  // entry k of this lane: stack[k * 64]
    uint8_t* __restrict__ lut = reinterpret_cast<uint8_t*>(s_mem8 + RT_WF8_LDS_STACK * 2 * WAVE); // [octant][mask] -> slot to take next
    for (uint32_t e = lane; e < 2048u; e += WAVE) {
        const uint32_t oct = e >> 8, m = e & 255u;
        uint32_t best = 0u, best_key = 99u;
        for (uint32_t i = 0; i < 8u; i++)
            if (((m >> i) & 1u) && (i ^ oct) < best_key) best_key = i ^ oct, best = i;
        lut[e] = (uint8_t)best;
    }
    __syncthreads();
    uint2* __restrict__ ovf = reinterpret_cast<uint2*>(wb.stack_ovf) + (size_t)blockIdx.x * wb.ovf_entries * WAVE + lane;
    const uint4* __restrict__ nodes = reinterpret_cast<const uint4*>(sc.nodes);
    const uint32_t count = wb.counters[count_slot];
    const uint32_t window_mask = max(wb.counters[window_slot], 1u) - 1u;
    uint32_t* cursor = &wb.counters[cursor_slot];
    // Queue entries a wave claims per atomic: RT_WF_CHUNK (256) when the queue holds that much for every wave; a short queue - the shadow
    // segments the light grids hand on (2 % of them: 0.5 M entries per launch), the camera segments without a beam list, the last
    // bounces - is shared out in smaller chunks, down to one wave-load: the launch then ends after one generation of segments per wave
    // instead of four (the drain was ~0.2 ms of every such launch; 60 of them in a headline frame)
#if RT_WF_ADAPTIVE_CHUNK
    uint32_t chunk = RT_WF_CHUNK;
    while (chunk > WAVE && (unsigned long long)chunk * gridDim.x > count) chunk >>= 1;
#else
    const uint32_t chunk = RT_WF_CHUNK;
#endif
    Counts cnt = {0u, 0u};
    int sp_max = 0;
    uint32_t d_node_steps = 0, d_leaf_steps = 0, d_leaf_lanes = 0, d_leaf_trips = 0, d_refills = 0;
    uint32_t d_empty = 0, d_entered = 0; // counting variant: visits that enter no child, children entered
    bool second = false;                 // counting variant, RT_WF_PROBE=1: the segment's second walk, started with its own hit distance (only that walk is counted)

    bool active = false, exhausted = false;
    uint32_t chunk_next = 0, chunk_end = 0;
    uint32_t id = 0, li = 0;
    V3 o = v3(0.0f, 0.0f, 0.0f), d = o;
    FilterRay fray;
    fray.o = o;
    fray.inv = o;
    Hit hit;
    hit.t = RT_F32_MAX;
    hit.prim = RT_PRIM_MISS;
    hit.slot = 0;
    uint32_t cur = WF8_NONE;          // node to visit next
    uint32_t g_base = 0, g_bits = 0;  // inner children still to visit: child_base, hit mask | imask << 8
    uint32_t t_base = 0, t_bits = 0;  // the postponed leaf group: tri_base, hit mask | lmask << 8
    uint32_t oct = 0;                 // table row of this segment: (dx < 0) | (dy < 0) << 1 | (dz < 0) << 2, times 256
    int sp = 0;

    for (;;) {
        const unsigned long long idle = __ballot(!active);
        if (!exhausted && (__popcll(idle) >= RT_WF_REFILL || idle == ~0ull)) {
            if (chunk_next >= chunk_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(cursor, chunk);
                base = __shfl(base, 0, WAVE);
                chunk_next = base;
                chunk_end = min(base + chunk, count);
                if (base >= count) exhausted = true;
            }
            if (!exhausted) {
                if (COUNT) d_refills++;
                const uint32_t n_fetch = (uint32_t)__popcll(idle);
                const uint32_t idx = chunk_next + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t last = min(chunk_next + n_fetch, chunk_end) - 1u;
                chunk_next += n_fetch;
                uint32_t e = WF_SENTINEL;
                if (!active && idx < chunk_end) e = queue[idx];
                if (__ballot(!active && idx == last && e == WF_SENTINEL) != 0ull) chunk_next = max(chunk_next, (last + window_mask + 1u) & ~window_mask);
                if (e != WF_SENTINEL) {
                    if (ANY) {
                        id = e & RT_WF_ID_MASK;
                        li = e >> 27;
                        const V3 point = f4v(wb.vtx[2 * (size_t)id]), normal = f4v(wb.vtx[2 * (size_t)id + 1]);
                        float dist;
                        shadow_segment(sc.lights[li], point, d, dist);
                        o = point + normal * EXT_EPS;
                        hit.t = dist;
                    } else {
                        id = e;
                        o = f4v(wb.ray_o[id]);
                        d = f4v(wb.ray_d[id]);
                        hit.t = RT_F32_MAX;
                    }
                    hit.prim = RT_PRIM_MISS;
                    hit.slot = 0;
                    fray = make_filter_ray(o, d);
                    oct = ((fray.inv.x < 0.0f ? 1u : 0u) | (fray.inv.y < 0.0f ? 2u : 0u) | (fray.inv.z < 0.0f ? 4u : 0u)) << 8;
                    // Shadow segments walk their children FAR to near (the complemented octant): any accepted triangle ends the walk, so
                    // the order is free, and measured the first occluder lies nearer the light than the vertex - occluded segments (61 %
                    // of them) needed 11.4 node visits near-first against 9.6 for visible ones; far-first: -3 % frame time, same bits.
                    if (ANY) oct ^= 7u << 8;
                    test_spheres(sc, o, d, hit);
                    sp = 0;
                    cur = WF8_NONE;
                    t_bits = 0;
                    // the root as the only child of a group: node 0 (the builder gives every non-empty scene an inner root)
                    g_base = 0;
                    g_bits = sc.n_nodes ? (1u | (1u << 8)) : 0u;
                    active = true;
                    if (ANY && hit.prim != RT_PRIM_MISS) g_bits = t_bits = 0; // occluded by a sphere already
                }
            }
        }
        if (__ballot(active) == 0ull) {
            if (exhausted) break;
            continue;
        }
        for (;;) {
            // ---- fetch: lanes without a node take the next child of their group, or the next group off the stack
            if (cur == WF8_NONE) { // (idle lanes have empty groups and an empty stack: nothing happens)
                if ((g_bits & 0xFFu) == 0u && sp > 0) {
                    const int k = sp - 1;
                    // an unconditional ds_read_b64 (clamped index), overridden on the rare lanes whose top entry is in HBM: left to
                    // itself the compiler merges the two loads into one flat load through a selected 64-bit generic address
                    uint2 e = stack[min(k, RT_WF8_LDS_STACK - 1) * WAVE];
                    asm volatile("" : "+v"(e.x), "+v"(e.y)); // (pins the LDS read in front of the branch)
                    if (k >= RT_WF8_LDS_STACK) e = ovf[(k - RT_WF8_LDS_STACK) * WAVE];
                    if (!(e.x & WF8_KIND_T)) {
                        sp = k;
                        g_base = e.x;
                        g_bits = e.y;
                    } else if ((t_bits & 0xFFu) == 0u) {
                        sp = k;
                        t_base = e.x & ~WF8_KIND_T;
                        t_bits = e.y;
                    }
                }
                if (g_bits & 0xFFu) {
                    const uint32_t i = lut[oct + (g_bits & 0xFFu)];
                    g_bits ^= 1u << i;                                                   // the bit is set
                    cur = g_base + (uint32_t)__popc(__builtin_amdgcn_ubfe(g_bits, 8u, i)); // inner slots below i
                }
            }
            // an idle lane has cur == WF8_NONE and empty groups, so these are plain compares: one v_cmp per mask (a ballot of a
            // bool that went through control flow costs a v_cndmask + v_cmp round trip)
            const bool can_node = cur != WF8_NONE;
            const bool has_leaf = (t_bits & 0xFFu) != 0u;
            const unsigned long long m_node = __builtin_amdgcn_uicmp(cur, WF8_NONE, 33 /* ne */), m_leaf = __builtin_amdgcn_uicmp(t_bits & 0xFFu, 0u, 33);
            if (m_node != 0ull && __popcll(m_leaf) < RT_WF8_LEAF_THRESHOLD) {
                if (COUNT) d_node_steps++;
                if (can_node) {
                    uint32_t cb, tb, im, lm;
                    const uint32_t hm = visit_node8<COUNT>(nodes, cur, fray, hit.t, cnt, cb, tb, im, lm);
                    if (COUNT) {
                        if (wb.probe && !ANY && !second) cnt.nodes--;
                        else d_empty += hm == 0u ? 1u : 0u, d_entered += (uint32_t)__popc(hm);
                    }
                    cur = WF8_NONE;
                    if (g_bits & 0xFFu) { // siblings still to visit: park them
                        if (sp < RT_WF8_LDS_STACK) stack[sp * WAVE] = make_uint2(g_base, g_bits);
                        else if ((uint32_t)(sp - RT_WF8_LDS_STACK) < wb.ovf_entries) ovf[(sp - RT_WF8_LDS_STACK) * WAVE] = make_uint2(g_base, g_bits);
                        else atomicOr(&wb.totals[WF_TOTAL_ERROR], 2ull); // a tree deeper than its reported depth: never written past the allocation, the frame is discarded
                        sp++;
                    }
                    g_base = cb;
                    g_bits = (hm & im) | (im << 8);
                    const uint32_t nt = hm & lm;
                    if (nt) {
                        if ((t_bits & 0xFFu) == 0u) {
                            t_base = tb;
                            t_bits = nt | (lm << 8);
                        } else {
                            if (sp < RT_WF8_LDS_STACK) stack[sp * WAVE] = make_uint2(tb | WF8_KIND_T, nt | (lm << 8));
                            else if ((uint32_t)(sp - RT_WF8_LDS_STACK) < wb.ovf_entries) ovf[(sp - RT_WF8_LDS_STACK) * WAVE] = make_uint2(tb | WF8_KIND_T, nt | (lm << 8));
                            else atomicOr(&wb.totals[WF_TOTAL_ERROR], 2ull);
                            sp++;
                        }
                    }
                    if (COUNT) sp_max = max(sp_max, sp);
                }
            } else if (m_leaf != 0ull) {
                uint32_t trips = 0;
                if (has_leaf) {
                    const uint32_t before = cnt.tris;
                    const uint32_t i = lut[oct + (t_bits & 0xFFu)];
                    t_bits ^= 1u << i;
                    const uint32_t first = t_base + RT_DEV_LEAF_STRIDE * (uint32_t)__popc(__builtin_amdgcn_ubfe(t_bits, 8u, i));
                    if (test_leaf<COUNT, ANY>(sc.tris, RT_DEV_LEAF_FLAG | first, o, d, hit, cnt)) { // occluded: nothing more to do
                        cur = WF8_NONE;
                        g_bits = t_bits = 0u;
                        sp = 0;
                    }
                    trips = cnt.tris - before;
                    if (COUNT && wb.probe && !ANY && !second) cnt.tris = before;
                }
                if (COUNT) {
                    d_leaf_steps++;
                    d_leaf_lanes += (uint32_t)__popcll(m_leaf);
                    for (int off = 32; off > 0; off >>= 1) trips = max(trips, (uint32_t)__shfl_xor((int)trips, off, WAVE));
                    d_leaf_trips += trips;
                }
            }
            uint32_t busy = ((g_bits | t_bits) & 0xFFu) | (uint32_t)sp | (cur + 1u); // 0: nothing left to visit (and 0 on idle lanes)
            if (COUNT && !ANY && wb.probe && active && busy == 0u && !second) { // (development probe) walk the segment again, knowing where it ends
                second = true;
                hit.t = fminf(hit.t * 1.00001f, RT_F32_MAX);
                hit.prim = RT_PRIM_MISS;
                g_base = 0;
                g_bits = sc.n_nodes ? (1u | (1u << 8)) : 0u;
                busy = g_bits & 0xFFu;
            }
            if (active && busy == 0u) { // segment finished
                second = false;
                if (ANY) {
                    if (hit.prim != RT_PRIM_MISS) atomicAnd(reinterpret_cast<uint32_t*>(&wb.vtx[2 * (size_t)id + 1]) + 3, ~(1u << li)); // occluded: the light leaves the visibility word (k_wf_shade set it)
                } else {
                    const V3 hp = o + d * hit.t;
                    const uint32_t code = hit.prim == RT_PRIM_MISS ? RT_PRIM_MISS : ((hit.prim & RT_PRIM_SPHERE_FLAG) ? hit.prim : hit.slot);
                    wb.hit[id] = make_uint4(__float_as_uint(hp.x), __float_as_uint(hp.y), __float_as_uint(hp.z), code);
                }
                active = false;
            }
            const unsigned long long still = __builtin_amdgcn_uicmp(busy, 0u, 33 /* ne */);
            if (still == 0ull) break;
            if (!exhausted && __popcll(~still) >= RT_WF_REFILL) break;
        }
    }
    if (COUNT) {
        unsigned long long n0 = wave_sum(cnt.nodes), n1 = wave_sum(cnt.tris);
        for (int off = 32; off > 0; off >>= 1) sp_max = max(sp_max, __shfl_down(sp_max, off, WAVE));
        if (lane == 0) {
            atomicAdd(&wb.totals[3], n0);
            atomicAdd(&wb.totals[4], n1);
            atomicMax(&wb.totals[5], (unsigned long long)sp_max);
            atomicAdd(&wb.totals[8], (unsigned long long)d_node_steps);
            atomicAdd(&wb.totals[9], (unsigned long long)d_leaf_steps);
            atomicAdd(&wb.totals[10], (unsigned long long)d_leaf_lanes);
            atomicAdd(&wb.totals[11], (unsigned long long)d_leaf_trips);
            atomicAdd(&wb.totals[12], (unsigned long long)d_refills);
        }
        const unsigned long long e0 = wave_sum(d_empty), e1 = wave_sum(d_entered);
        if (lane == 0) {
            atomicAdd(&wb.totals[6], e0);
            atomicAdd(&wb.totals[7], e1);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Camera beams (round 3).  All camera segments leave one point, and the 64 samples x 64 pixels of an 8x8 pixel block leave it through
// a pyramid a quarter of a degree wide: the leaves of the tree that pyramid touches are few (median 16 triangles, 98 % of the
// headline frame's blocks under 128 leaves) and the same for every segment of the block.  k_wf_beams collects them once per frame -
// a frustum walk of the 8-wide tree, one wave per block, eight nodes x eight child slots per step - and orders them by a lower
// bound of their distance; k_wf_trace_camera then tests a block's segments against its list, nearest leaf first, every lane on the
// same triangle (no stack, no divergence, full lanes), until the next leaf lies beyond every lane's hit.  This replaces the tree walk
// (14 node visits x ~200 instructions at 74 % of the lanes) for a third of the closest-hit segments.  Same results: which triangle is
// hit is decided by the same Moeller-Trumbore statement with the same tie rule (lowest triangle index among equal t), and the list is a
// superset of the triangles that statement can accept for any segment of the block - the pyramid is widened by an eighth of a pixel
// (600 x the rounding of a direction) and the leaf boxes by 2e-6 of their magnitude.  Blocks whose list would exceed RT_BEAM_CAP leaves
// (views along a colonnade) or whose pyramid is degenerate (NaN cameras) keep the per-lane tree walk.
// Replaces, for depth-0 segments, BvhTraverser::traverse_and_intersect (shader/src/bvh.rs:18-88).
// ---------------------------------------------------------------------------------------------------------
#ifndef RT_BEAM_STACK
#define RT_BEAM_STACK 512
#endif
#ifndef RT_BEAM_LEAVES
#define RT_BEAM_LEAVES 1024 /* leaves the frustum walk may collect before it gives up (their triangles are then tested one by one against the pyramid) */
#endif
#ifndef RT_BEAM_MARGIN_PX
#define RT_BEAM_MARGIN_PX 0.125f
#endif
#ifndef RT_BEAM_SAMPLES_PER_WAVE
#define RT_BEAM_SAMPLES_PER_WAVE 8u
#endif
#ifndef RT_BEAM_DIST_SCALE
#define RT_BEAM_DIST_SCALE 0.99999f
#endif
#ifndef RT_BEAM_EARLY_OUT
#define RT_BEAM_EARLY_OUT 1
#endif
#ifndef RT_BEAM_TRI_CULL
#define RT_BEAM_TRI_CULL 1
#endif
struct BeamPlanes {
    V3 n[4]; // inward normals of the pyramid's four sides (through the camera position)
    V3 o;
};
// is the box [lo, hi] (already padded) outside the pyramid?  The corner farthest along each plane's inward normal decides.
__device__ __forceinline__ bool beam_outside(const BeamPlanes& bp, const float lo[3], const float hi[3]) {
    bool outside = false;
    for (int k = 0; k < 4; k++) {
        const V3 pv = v3((bp.n[k].x > 0.0f ? hi[0] : lo[0]) - bp.o.x, (bp.n[k].y > 0.0f ? hi[1] : lo[1]) - bp.o.y, (bp.n[k].z > 0.0f ? hi[2] : lo[2]) - bp.o.z);
        outside = outside || dot(bp.n[k], pv) < 0.0f;
    }
    return outside;
}
// a lower bound of the distance from the camera to anything inside the box (segment directions are unit vectors: t is a distance)
__device__ __forceinline__ float beam_box_distance(const V3 o, const float lo[3], const float hi[3]) {
    const float dx = fmaxf(fmaxf(lo[0] - o.x, o.x - hi[0]), 0.0f), dy = fmaxf(fmaxf(lo[1] - o.y, o.y - hi[1]), 0.0f), dz = fmaxf(fmaxf(lo[2] - o.z, o.z - hi[2]), 0.0f);
    return sqrtf(dx * dx + dy * dy + dz * dz) * RT_BEAM_DIST_SCALE;
}
__global__ __launch_bounds__(WAVE) void k_wf_beams(DevScene sc, DevFrame fr, rt::WfBuffers wb) {
    __shared__ uint32_t s_nodes[RT_BEAM_STACK];
    __shared__ uint32_t s_leaf[RT_BEAM_LEAVES];
    __shared__ uint32_t s_ref[RT_BEAM_CAP];
    __shared__ float s_dist[RT_BEAM_CAP];
    const uint32_t lane = threadIdx.x, b = blockIdx.x;
    const PixelCoord p0 = block_pixel_at(fr, b, 0); // the block's first pixel (defined also when it lies outside the image)
    const float x0 = (float)p0.x - RT_BEAM_MARGIN_PX, x1 = (float)p0.x + 8.0f + RT_BEAM_MARGIN_PX;
    const float y0 = (float)p0.y - RT_BEAM_MARGIN_PX, y1 = (float)p0.y + 8.0f + RT_BEAM_MARGIN_PX;
    const DevCamera& cam = fr.cam;
    auto dir = [&](float sx, float sy) { // camera_ray's direction before it is normalised
        const float u = sx / cam.width_f, v = sy / cam.height_f;
        const float cx = (u * 2.0f - 1.0f) * cam.aspect * cam.fov_scale, cy = (1.0f - v * 2.0f) * cam.fov_scale;
        return ld3(cam.forward) + ld3(cam.right) * cx + ld3(cam.true_up) * cy;
    };
    const V3 cs[4] = {dir(x0, y0), dir(x1, y0), dir(x1, y1), dir(x0, y1)};
    const V3 cc = dir(0.5f * (x0 + x1), 0.5f * (y0 + y1));
    BeamPlanes bp;
    bp.o = ld3(cam.origin);
    bool bad = false;
    for (int k = 0; k < 4; k++) {
        bp.n[k] = cross(cs[k], cs[(k + 1) & 3]);
        const float s = dot(bp.n[k], cc);
        if (s < 0.0f) bp.n[k] = -bp.n[k];
        // the four corner directions must lie on the inner side of every plane (a pyramid narrower than 180 degrees, finite numbers)
        for (int j = 0; j < 4; j++)
            bad = bad || !(dot(bp.n[k], cs[j]) >= -1e-6f * (fabsf(bp.n[k].x) + fabsf(bp.n[k].y) + fabsf(bp.n[k].z)) * (fabsf(cs[j].x) + fabsf(cs[j].y) + fabsf(cs[j].z)));
        bad = bad || !(fabsf(s) > 0.0f) || !(fabsf(s) < RT_F32_MAX);
    }
    uint32_t n_stack = 0, n_leaf = 0;
    bool overflow = bad;
    if (sc.n_nodes) {
        if (lane == 0) s_nodes[0] = 0u;
        n_stack = 1;
    }
    __syncthreads();
    const uint32_t slot = lane & 7u;
    const unsigned long long lower = (1ull << lane) - 1ull;
    // ---- the frustum walk: eight nodes x eight child slots per step
    while (n_stack > 0 && !overflow) {
        const uint32_t take = min(n_stack, 8u), base = n_stack - take;
        const bool have = (lane >> 3) < take;
        const uint32_t node = have ? s_nodes[base + (lane >> 3)] : 0u;
        __syncthreads(); // (the reads above precede the pushes below)
        n_stack = base;
        bool inner = false, leaf = false;
        uint32_t ref = 0;
        if (have) {
            const DevNode8& nd = sc.nodes[node];
            const uint32_t ex = nd.ex_imask, imask = ex >> 24, lmask = nd.lmask & 0xFFu;
            if (((imask | lmask) >> slot) & 1u) {
                float lo[3], hi[3];
                for (int a = 0; a < 3; a++) {
                    const float sca = ldexpf(1.0f, (int)(int8_t)((ex >> (8 * a)) & 0xFFu));
                    const float ql = (float)((nd.qlo[a][slot >> 2] >> (8u * (slot & 3u))) & 0xFFu), qh = (float)((nd.qhi[a][slot >> 2] >> (8u * (slot & 3u))) & 0xFFu);
                    lo[a] = nd.org[a] + ql * sca;
                    hi[a] = nd.org[a] + qh * sca;
                    const float pad = (fabsf(lo[a]) + fabsf(hi[a])) * 2.0e-6f + 1.0e-30f; // the rounding of the two sums above, and vertices an ulp outside their box (bvh_check.h)
                    lo[a] -= pad;
                    hi[a] += pad;
                }
                if (!beam_outside(bp, lo, hi)) {
                    const uint32_t below = (1u << slot) - 1u;
                    if ((imask >> slot) & 1u) {
                        inner = true;
                        ref = nd.child_base + (uint32_t)__popc(imask & below);
                    } else {
                        leaf = true;
                        ref = nd.tri_base + RT_DEV_LEAF_STRIDE * (uint32_t)__popc(lmask & below);
                    }
                }
            }
        }
        const unsigned long long mi = __ballot(inner), ml = __ballot(leaf);
        const uint32_t pi = n_stack + (uint32_t)__popcll(mi & lower), pc = n_leaf + (uint32_t)__popcll(ml & lower);
        if (inner && pi < RT_BEAM_STACK) s_nodes[pi] = ref;
        if (leaf && pc < RT_BEAM_LEAVES) s_leaf[pc] = ref;
        n_stack += (uint32_t)__popcll(mi);
        n_leaf += (uint32_t)__popcll(ml);
        overflow = n_stack > RT_BEAM_STACK || n_leaf > RT_BEAM_LEAVES;
        __syncthreads();
    }
    // ---- the triangles of those leaves, each against the pyramid by its own box
    uint32_t n_tri = 0;
    for (uint32_t l0 = 0; l0 < n_leaf && !overflow; l0 += WAVE) {
        const uint32_t li = l0 + lane;
        uint32_t first = 0, count = 0;
        if (li < n_leaf) {
            first = s_leaf[li];
            count = sc.tris[first].leaf_count;
        }
        for (uint32_t k = 0; k < RT_DEV_LEAF_STRIDE; k++) { // (wave-uniform trip count: the appends are wave operations)
            bool keep = false;
            float dist = 0.0f;
            if (k < count) {
                const DevTri& t = sc.tris[first + k];
                float lo[3], hi[3];
                for (int a = 0; a < 3; a++) {
                    const float va = t.v0[a], vb = t.v0[a] + t.e1[a], vc = t.v0[a] + t.e2[a];
                    lo[a] = fminf(va, fminf(vb, vc));
                    hi[a] = fmaxf(va, fmaxf(vb, vc));
                    const float pad = (fabsf(lo[a]) + fabsf(hi[a])) * 2.0e-6f + 1.0e-30f;
                    lo[a] -= pad;
                    hi[a] += pad;
                }
                keep = !RT_BEAM_TRI_CULL || !(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]) || !beam_outside(bp, lo, hi); // (non-finite boxes are kept: the test decides)
                dist = beam_box_distance(bp.o, lo, hi);
                if (!(dist >= 0.0f)) dist = 0.0f;
            }
            const unsigned long long mk = __ballot(keep);
            const uint32_t pos = n_tri + (uint32_t)__popcll(mk & lower);
            if (keep && pos < RT_BEAM_CAP) {
                s_ref[pos] = first + k;
                s_dist[pos] = dist;
            }
            n_tri += (uint32_t)__popcll(mk);
        }
        overflow = n_tri > RT_BEAM_CAP;
    }
    __syncthreads();
    if (overflow) { // no list: the block's camera segments walk the tree; it takes the next place in the queue k_wf_generate fills for them
        if (lane == 0) wb.beam_count[b] = RT_BEAM_OVERFLOW | atomicAdd(&wb.beam_count[wb.n_blocks], 1u);
        return;
    }
    // nearest first: every entry finds its rank (ties by position)
    for (uint32_t i = lane; i < n_tri; i += WAVE) {
        const float di = s_dist[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n_tri; j++) {
            const float dj = s_dist[j];
            rank += (dj < di || (dj == di && j < i)) ? 1u : 0u;
        }
        wb.beam_ref[(size_t)b * RT_BEAM_CAP + rank] = s_ref[i];
        wb.beam_dist[(size_t)b * RT_BEAM_CAP + rank] = di;
    }
    if (lane == 0) wb.beam_count[b] = n_tri;
}

// Depth-0 closest hits from the block lists.  One wave per (pixel block, RT_BEAM_SAMPLES_PER_WAVE samples): the list's triangle records are
// staged in LDS and those samples of the block's 64 pixels (path slots (b * n_samples + k) * 64 + lane, as k_wf_generate lays them out) are
// tested against them, all lanes on the same triangle (broadcast LDS reads).  Blocks without a list are skipped: k_wf_generate put their
// segments on a queue of their own for k_wf_trace.
template <bool COUNT>
__global__ __launch_bounds__(WAVE) void k_wf_trace_camera(DevScene sc, rt::WfBuffers wb, uint32_t n_samples) {
    extern __shared__ uint4 s_cam[]; // RT_BEAM_CAP * 3 quads + RT_BEAM_CAP floats (+ a per-lane walk stack behind them for the development probe)
    if (wb.totals[WF_TOTAL_ERROR] != 0ull) return;
    const uint32_t lane = threadIdx.x;
    float* s_d = reinterpret_cast<float*>(s_cam + RT_BEAM_CAP * 3u);
    Counts cnt = {0u, 0u};
    // one wave per (block, RT_BEAM_SAMPLES_PER_WAVE samples): a block's samples in one wave would be a chain of 22 ... 64 dependent
    // round trips with 16 waves per CU on an eighth of the frame (2.4 ms per launch measured); staging the list again per chunk is cheap
    const uint32_t chunks = (n_samples + RT_BEAM_SAMPLES_PER_WAVE - 1u) / RT_BEAM_SAMPLES_PER_WAVE;
    for (uint32_t item = blockIdx.x; item < wb.n_blocks * chunks; item += gridDim.x) {
        const uint32_t b = item / chunks, k_first = (item - b * chunks) * RT_BEAM_SAMPLES_PER_WAVE, k_end = min(n_samples, k_first + RT_BEAM_SAMPLES_PER_WAVE);
        const uint32_t n_list = (uint32_t)__builtin_amdgcn_readfirstlane((int)wb.beam_count[b]);
        if (n_list & RT_BEAM_OVERFLOW) continue; // (k_wf_generate queues this block's segments for the tree walk)
        {
            __syncthreads(); // (the previous block's reads are done)
            for (uint32_t i = lane; i < n_list; i += WAVE) {
                const uint32_t slot = wb.beam_ref[(size_t)b * RT_BEAM_CAP + i];
                const uint4* __restrict__ rec = reinterpret_cast<const uint4*>(sc.tris + slot);
                const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
                s_cam[3u * i] = q0;
                s_cam[3u * i + 1u] = q1;
                s_cam[3u * i + 2u] = make_uint4(q2.x, q2.y, q2.z, slot); // (the leaf length of the record is not needed here: its place carries the record's index)
                s_d[i] = wb.beam_dist[(size_t)b * RT_BEAM_CAP + i];
            }
            __syncthreads();
        }
        for (uint32_t k = k_first; k < k_end; k++) {
            const uint32_t sb = RT_WF_BLOCK_MAJOR ? b * n_samples + k : k * wb.n_blocks + b;
            const uint32_t p = sb * WAVE + lane;
            const bool valid = wb.pxy[p] != 0xFFFFFFFFu;
            if (__ballot(valid) == 0ull) continue;
            V3 o = v3(0.0f, 0.0f, 0.0f), d = v3(0.0f, 0.0f, 1.0f);
            if (valid) {
                float4 ro = wb.ray_o[p], rd = wb.ray_d[p];
                RT_KEEP4_EARLY(ro);
                RT_KEEP4_EARLY(rd);
                o = f4v(ro);
                d = f4v(rd);
            }
            Hit hit;
            hit.t = valid ? RT_F32_MAX : -1.0f; // (lanes without a path never ask for another triangle)
            hit.prim = RT_PRIM_MISS;
            hit.slot = 0;
            if (valid) test_spheres(sc, o, d, hit);
            {
                for (uint32_t i = 0; i < n_list; i++) {
                    if (RT_BEAM_EARLY_OUT && __ballot(s_d[i] < hit.t) == 0ull) break; // sorted: every later triangle is farther still
                    const uint4 q0 = s_cam[3u * i], q1 = s_cam[3u * i + 1u], q2 = s_cam[3u * i + 2u];
                    if (COUNT && valid) cnt.tris++;
                    float t = 0.0f;
                    const bool inside = moller_trumbore(v3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z)), v3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y)),
                                                        v3(__uint_as_float(q1.z), __uint_as_float(q1.w), __uint_as_float(q2.x)), o, d, t);
                    // the acceptance rule of test_triangle (device_common.h): lowest index among equal t, strict against a sphere.  Written as
                    // selects under ONE mask: with the three assignments inside an `if`, hipcc (ROCm 7.2) kept the old record index on the lanes
                    // accepted through the tie clause (found as 1 wrong pixel per 2 M segments; tests/test_gpu_beams.py holds the tie scene)
                    const uint32_t prim = q2.z;
                    const bool accept = inside && t > RT_MIN_RAY_DISTANCE && (t < hit.t || (t == hit.t && prim < hit.prim && hit.prim < RT_PRIM_SPHERE_FLAG));
                    hit.t = accept ? t : hit.t;
                    hit.prim = accept ? prim : hit.prim;
                    hit.slot = accept ? q2.w : hit.slot;
                }
            }
            if (COUNT && wb.probe == 2u && valid) { // development probe: the same segment by the tree walk (the launch provides the extra LDS)
                Hit h2;
                h2.t = RT_F32_MAX;
                h2.prim = RT_PRIM_MISS;
                h2.slot = 0;
                Counts c2 = {0u, 0u};
                test_spheres(sc, o, d, h2);
                traverse<false, false>(sc, o, d, reinterpret_cast<uint2*>(s_cam + RT_BEAM_CAP * 4u) + lane, h2, c2);
                if (h2.prim != hit.prim || h2.t != hit.t) {
                    if (atomicAdd(&wb.totals[6], 1ull) == 0ull) {
                        wb.totals[8] = b;
                        wb.totals[9] = ((unsigned long long)hit.prim << 32) | h2.prim;
                        wb.totals[10] = ((unsigned long long)__float_as_uint(hit.t) << 32) | __float_as_uint(h2.t);
                        wb.totals[11] = ((unsigned long long)n_list << 32) | h2.slot;
                        wb.totals[12] = p;
                    }
                }
            }
            if (valid) {
                const V3 hp = o + d * hit.t;
                const uint32_t code = hit.prim == RT_PRIM_MISS ? RT_PRIM_MISS : ((hit.prim & RT_PRIM_SPHERE_FLAG) ? hit.prim : hit.slot);
                wb.hit[p] = make_uint4(__float_as_uint(hp.x), __float_as_uint(hp.y), __float_as_uint(hp.z), code);
            }
        }
    }
    if (COUNT) {
        const unsigned long long n0 = wave_sum(cnt.nodes), n1 = wave_sum(cnt.tris);
        if (lane == 0) {
            atomicAdd(&wb.totals[3], n0);
            atomicAdd(&wb.totals[4], n1);
        }
    }
}

// The shading stages are bound by memory latency, not arithmetic (22 % VALU busy, 83 % of the wave time in s_waitcnt):
// what counts is the number of DEPENDENT round trips per path.  Lights come from LDS (staged once per block), and
// each group of per-path loads is issued together: RT_KEEP4 pins the loaded values at one point so the
// compiler cannot split a record by first use and sink the later words behind a branch (each a further round trip).
#define RT_KEEP4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
__device__ __forceinline__ void stage_lights(DevLight* __restrict__ s_lights, const DevScene& sc) {
    const uint32_t words = sc.n_lights * (uint32_t)(sizeof(DevLight) / 4);
    const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(sc.lights);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s_lights);
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}
__device__ __forceinline__ DevMaterial load_material(const DevScene& sc, uint32_t material_id) {
    const float4* p = reinterpret_cast<const float4*>(sc.materials + material_id);
    float4 a = p[0], b = p[1];
    RT_KEEP4(a);
    RT_KEEP4(b);
    DevMaterial m;
    m.albedo[0] = a.x; m.albedo[1] = a.y; m.albedo[2] = a.z; m.metallic = a.w;
    m.emission[0] = b.x; m.emission[1] = b.y; m.emission[2] = b.z; m.ior = b.w;
    const float2 c = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(sc.materials + material_id) + 8);
    m.transmission = c.x; m.roughness = c.y;
    m._pad[0] = m._pad[1] = 0.0f;
    return m;
}

// ---------------------------------------------------------------------------------------------------------
// shadow segments through the per-light grids (shadow_grid.h).  One thread per vertex, no refill machinery: the work of a segment is a
// short dependent chain of fetches (cell block -> further list entries; measured 2.2 entries per segment on the headline scene), so what
// counts is the number of segments in flight - a persistent version with k_wf_trace's refill was latency-bound at a twelfth of the VALU
// rate and slower than the traversal it replaces.  A segment goes through its cell's list nearest to the light first until the
// reference's triangle test accepts an entry (occluded) or the keys pass the segment's own end (visible).  Segments that are still
// undecided after RT_WF_GRID_WALK entries, segments in cells longer than the grid's `heavy` and segments of lights without a grid are
// appended, densely, to q_shadow2 for k_wf_trace<any hit>.
// ---------------------------------------------------------------------------------------------------------
#ifndef RT_WF_GRID_WALK
#define RT_WF_GRID_WALK 31 /* entries a segment looks at before it is handed on (12 while a long walk held its whole wave up: rounds 2-3) */
#endif
static_assert(EXT_EPS == RT_SG_EXT_EPS, "the light grids' dilation is derived from the shadow segments' origin offset");
static_assert(RT_WF_GRID_WALK < RT_SG_SORTED_PREFIX, "a walk may only look at the ordered part of a list");
#ifndef RT_WF_GRID_BLOCKS_PER_CU
#define RT_WF_GRID_BLOCKS_PER_CU 32 /* 256-thread blocks per CU in the grid-stride launch: 8 measured 3 % slower than 16 (round 2); 32 / 48 / 64: -0.7 % / -0.6 % / -0.6 % on the
                                        headline frame, an eighth of it +0.4 % / +2 % / +2 % (round 3, profiles/ab_r03.json) */
#endif
#ifndef RT_WF_GRID_MIN_WAVES
#define RT_WF_GRID_MIN_WAVES 7 /* waves per SIMD asked of the register allocator.  By segment (rounds 2-3): 8 (64 VGPRs) spilled five registers inside the
                                loop - scratch traffic in a stage bound by memory requests; 7 (72 VGPRs): -2.7 % on the headline frame, 6 the same.  By vertex
                                the block's 30 KB of LDS (the parking lists) allow five blocks per CU, the compiler knows it and takes 86 VGPRs, no spills;
                                5, 6 and 7 waves had measured the same before (profiles/ab_r03.json) */
#endif
// One shadow segment (vertex `point` / `normal` toward light li) against the head of its cell's list: what the cell's own 128-byte block
// holds (header, two entries, the key of the third).  Outcome GRID_VISIBLE / GRID_OCCLUDED / GRID_FORWARD (left to the BVH), or
// GRID_PENDING: the list goes on beyond the block and the segment has not met its occluder or its end yet - `pend` is then what the
// second part of the walk (grid_walk_on) needs.
enum : uint32_t { GRID_VISIBLE = 0u, GRID_OCCLUDED = 1u, GRID_FORWARD = 2u, GRID_PENDING = 3u };
struct GridPending {
    V3 o, d;
    float dist, limit;
    uint32_t at;    // the next entry, as an index into the grid's overflow array
    uint32_t i, count; // ... which is entry i of `count`
};
static_assert(RT_SG_BLOCK_ENTRIES == 2u, "grid_segment_head tests the block's two entries by name");
template <bool COUNT>
__device__ __forceinline__ uint32_t grid_segment_head(const DevScene& sc, const DevLight& light, const DevShadowGrid& g, V3 point, V3 normal, GridPending& pend,
                                                      uint32_t& n_tests, uint32_t& n_entries) {
    V3 d;
    float dist;
    shadow_segment(light, point, d, dist);
    const V3 o = point + normal * EXT_EPS;
    Hit hit;
    hit.t = dist;
    hit.prim = RT_PRIM_MISS;
    hit.slot = 0;
    test_spheres(sc, o, d, hit);
    if (hit.prim != RT_PRIM_MISS) return GRID_OCCLUDED; // by a sphere: nothing left to do
    const uint32_t kind = g.kind;
    uint32_t cell = 0xFFFFFFFFu; // no cell: an empty list
    float limit = 0.0f;
    if (kind == RT_SG_KIND_CUBE) {
        // the direction from the light toward the vertex picks the face (largest component) and the cell (the other two over it)
        const float wx = -d.x, wy = -d.y, wz = -d.z;
        const float ax = fabsf(wx), ay = fabsf(wy), az = fabsf(wz);
        const uint32_t a = (ax >= ay && ax >= az) ? 0u : (ay >= az ? 1u : 2u);
        const float wa = a == 0u ? wx : (a == 1u ? wy : wz), wb_ = a == 0u ? wy : (a == 1u ? wz : wx), wc = a == 0u ? wz : (a == 1u ? wx : wy);
        const float inv = __builtin_amdgcn_rcpf(fabsf(wa)); // (an ulp either way is far inside the lists' margin)
        const float fu = (wb_ * inv + 1.0f) * g.scale, fv = (wc * inv + 1.0f) * g.scale;
        const uint32_t top = g.res - 1u;
        const uint32_t ix = min((uint32_t)max((int)floorf(fu), 0), top), iy = min((uint32_t)max((int)floorf(fv), 0), top);
        cell = ((2u * a + (wa < 0.0f ? 1u : 0u)) * g.res + iy) * g.res + ix;
        limit = dist + g.limit_margin;
    } else if (kind == RT_SG_KIND_ORTHO) {
        const float fu = (dot(o, ld3(g.axis_u)) - g.u0) * g.scale, fv = (dot(o, ld3(g.axis_v)) - g.v0) * g.scale;
        const float r = (float)g.res;
        if (fu >= 0.0f && fu < r && fv >= 0.0f && fv < r) cell = (uint32_t)fv * g.res + (uint32_t)fu; // outside: nothing projects there
        limit = (g.key_top - dot(o, ld3(g.axis_w))) + g.limit_margin;
    } else {
        return GRID_FORWARD; // a light without a grid
    }
    // the cell's block: header and the list's first entry in the first half of its 128-byte line, the second entry in the other half
    if (cell == 0xFFFFFFFFu && g.near_begin == g.near_end) return GRID_VISIBLE;
    uint4 hd = make_uint4(0u, 0u, 0x7F800000u, 0u), q0 = hd, q1 = hd, q2 = hd;
    const uint4* __restrict__ blk = g.blocks + (size_t)(cell == 0xFFFFFFFFu ? 0u : cell) * RT_SG_BLOCK_QUADS;
    if (cell != 0xFFFFFFFFu) {
        hd = blk[0];
        q0 = blk[1], q1 = blk[2], q2 = blk[3];
        RT_KEEP4(hd);
        RT_KEEP4(q0);
        RT_KEEP4(q1);
        RT_KEEP4(q2);
    }
    const uint32_t count = hd.x;
    if (count > g.heavy) return GRID_FORWARD;
    const uint4* __restrict__ ovf = g.overflow; // 48-byte entries: {key, v0} {e1, e2.x} {e2.yz, record, 0}
    // triangles too close to the light for a bounded dilation: tested by every segment of the light (normally none)
    for (uint32_t k = g.near_begin; k < g.near_end; k++) {
        const uint4 n0 = ovf[3 * (size_t)k], n1 = ovf[3 * (size_t)k + 1], n2 = ovf[3 * (size_t)k + 2];
        if (COUNT) n_tests++;
        float t;
        if (moller_trumbore(v3(__uint_as_float(n0.y), __uint_as_float(n0.z), __uint_as_float(n0.w)), v3(__uint_as_float(n1.x), __uint_as_float(n1.y), __uint_as_float(n1.z)),
                            v3(__uint_as_float(n1.w), __uint_as_float(n2.x), __uint_as_float(n2.y)), o, d, t) &&
            t > RT_MIN_RAY_DISTANCE && t < dist)
            return GRID_OCCLUDED;
    }
    // entries come nearest to the light first: a key beyond the segment's own end means every later triangle lies beyond it too
    if (count == 0u || !(__uint_as_float(q0.x) < limit)) return GRID_VISIBLE;
    float t;
    if (COUNT) n_entries++, n_tests++;
    // the acceptance of test_triangle for a segment that has hit nothing yet: 1e-5 < t < its length
    if (moller_trumbore(v3(__uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w)), v3(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z)),
                        v3(__uint_as_float(q1.w), __uint_as_float(q2.x), __uint_as_float(q2.y)), o, d, t) &&
        t > RT_MIN_RAY_DISTANCE && t < dist)
        return GRID_OCCLUDED;
    if (count == 1u) return GRID_VISIBLE;
    q0 = blk[4], q1 = blk[5], q2 = blk[6];
    if (!(__uint_as_float(q0.x) < limit)) return GRID_VISIBLE;
    if (COUNT) n_entries++, n_tests++;
    if (moller_trumbore(v3(__uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w)), v3(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z)),
                        v3(__uint_as_float(q1.w), __uint_as_float(q2.x), __uint_as_float(q2.y)), o, d, t) &&
        t > RT_MIN_RAY_DISTANCE && t < dist)
        return GRID_OCCLUDED;
    if (count == 2u || !(__uint_as_float(hd.z) < limit)) return GRID_VISIBLE; // (the third entry's key travels in the header)
    pend.o = o, pend.d = d, pend.dist = dist, pend.limit = limit;
    pend.at = hd.y, pend.i = 2u, pend.count = count;
    return GRID_PENDING;
}
// ... and the list beyond the block, at most `trips` entries further: entries 2, 3, ... follow each other in the overflow array, each is
// fetched when the one before it has decided nothing.  GRID_PENDING again: `pend` has moved on.
template <bool COUNT>
__device__ __forceinline__ uint32_t grid_walk_on(const DevShadowGrid& g, GridPending& pend, uint32_t trips, uint32_t& n_tests, uint32_t& n_entries) {
    const uint4* __restrict__ ovf = g.overflow;
    for (uint32_t k = 0; k < trips; k++) {
        const size_t at = 3 * (size_t)pend.at;
        uint4 q0 = ovf[at], q1 = ovf[at + 1], q2 = ovf[at + 2];
        RT_KEEP4(q0);
        RT_KEEP4(q1);
        RT_KEEP4(q2);
        if (!(__uint_as_float(q0.x) < pend.limit)) return GRID_VISIBLE;
        if (COUNT) n_entries++, n_tests++;
        float t;
        if (moller_trumbore(v3(__uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w)), v3(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z)),
                            v3(__uint_as_float(q1.w), __uint_as_float(q2.x), __uint_as_float(q2.y)), pend.o, pend.d, t) &&
            t > RT_MIN_RAY_DISTANCE && t < pend.dist)
            return GRID_OCCLUDED;
        pend.at++;
        pend.i++;
        if (pend.i >= pend.count) return GRID_VISIBLE;
        if (pend.i >= RT_WF_GRID_WALK) return GRID_FORWARD;
    }
    return GRID_PENDING;
}

// One thread per vertex of the bounce (the entries of the extension queue, as k_wf_shade and k_wf_finish read it); k_wf_shade has left the
// lights that need a segment as the vertex record's visibility word, and what this stage (and k_wf_trace<any hit> after it) does is CLEAR
// the bits of occluded segments - every update of the word is an atomic and-not, so their order does not matter.  The wave goes through the
// lights together - neighbouring vertices toward one light fall into neighbouring cells - and a lane skips the lights it has no segment for.
// Against one thread per (vertex, light) entry of a shadow queue (rounds 1-2): the 32-byte vertex record is read once instead of once per
// light, and the queue (4 bytes written and read per segment) is gone.
//
// List walks differ in length (headline scene: 2.2 entries on average, but 22 % of the segments need more than the two entries of their
// cell's block, 9 % more than four, 2 % more than nine - so that nearly every wave of 64 had a lane going to the walk's limit, the others
// idle behind it: a fifth of the lanes at work, and each trip a dependent fetch the whole wave waits for).  So a lane only looks at its
// cell's block; a segment undecided after that is parked in the wave's list in LDS (44 bytes), and when 64 are parked the wave walks them
// on together, RT_WF_GRID_PASS entries at a time, parking again what is still undecided.
#ifndef RT_WF_GRID_PASS
#define RT_WF_GRID_PASS 3u
#endif
#define RT_WF_GRID_PARK 128u /* slots of a wave's list: fewer than 64 parked before a light's segments add at most 64 */
#define RT_WF_GRID_PARK_WORDS 11u
template <bool COUNT>
__global__ __launch_bounds__(256, RT_WF_GRID_MIN_WAVES) void k_wf_shadow_grid(DevScene sc, rt::WfBuffers wb, const uint32_t* __restrict__ queue) {
    __shared__ DevLight s_lights[RT_WF_MAX_LIGHTS];
    __shared__ DevShadowGrid s_grids[RT_WF_MAX_LIGHTS];
    __shared__ uint32_t s_fwd[4][128]; // per wave: entries to hand on, not yet appended
    __shared__ uint32_t s_park[4][RT_WF_GRID_PARK_WORDS][RT_WF_GRID_PARK];
    if (wb.totals[WF_TOTAL_ERROR] != 0ull) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t n_fwd = 0, n_park = 0; // (wave-uniform)
    stage_lights(s_lights, sc);
    {
        const uint32_t words = sc.n_lights * (uint32_t)(sizeof(DevShadowGrid) / 4);
        const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(wb.grids);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_grids);
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    uint32_t n_tests = 0, n_entries = 0, n_answered = 0;
    uint32_t(*park)[RT_WF_GRID_PARK] = s_park[wave];
    // hand on: collected per wave in LDS and appended 64 or more at a time (one atomic on the queue's counter per append: an atomic per
    // wave and iteration, 1.5 M of them on one address, cost more than the whole list walk)
    auto hand_on = [&](bool forward, uint32_t entry, bool flush) {
        const unsigned long long fm = __ballot(forward);
        if (fm != 0ull) {
            if (forward) s_fwd[wave][n_fwd + (uint32_t)__popcll(fm & below)] = entry;
            n_fwd += (uint32_t)__popcll(fm);
        }
        if (n_fwd >= 64u || (flush && n_fwd)) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&wb.counters[rt::WF_SHADOW2_COUNT], n_fwd);
            at = __shfl(at, 0, WAVE);
            for (uint32_t k = lane; k < n_fwd; k += WAVE) wb.q_shadow2[at + k] = s_fwd[wave][k];
            n_fwd = 0;
        }
    };
    auto park_push = [&](bool pending, const GridPending& pd, uint32_t entry) {
        const unsigned long long pm = __ballot(pending);
        if (pm == 0ull) return;
        if (pending) {
            const uint32_t s = n_park + (uint32_t)__popcll(pm & below);
            park[0][s] = __float_as_uint(pd.o.x), park[1][s] = __float_as_uint(pd.o.y), park[2][s] = __float_as_uint(pd.o.z);
            park[3][s] = __float_as_uint(pd.d.x), park[4][s] = __float_as_uint(pd.d.y), park[5][s] = __float_as_uint(pd.d.z);
            park[6][s] = __float_as_uint(pd.dist), park[7][s] = __float_as_uint(pd.limit);
            park[8][s] = pd.at, park[9][s] = pd.i | (pd.count << 8), park[10][s] = entry;
        }
        n_park += (uint32_t)__popcll(pm);
    };
    // the parked segments on top of the list (64, or all of them), RT_WF_GRID_PASS entries further
    auto walk_parked = [&]() {
        const uint32_t take = min(n_park, 64u), first = n_park - take;
        const bool have = lane < take;
        GridPending pd = {};
        uint32_t entry = 0, outcome = GRID_VISIBLE;
        if (have) {
            const uint32_t s = first + lane;
            pd.o = v3(__uint_as_float(park[0][s]), __uint_as_float(park[1][s]), __uint_as_float(park[2][s]));
            pd.d = v3(__uint_as_float(park[3][s]), __uint_as_float(park[4][s]), __uint_as_float(park[5][s]));
            pd.dist = __uint_as_float(park[6][s]), pd.limit = __uint_as_float(park[7][s]);
            pd.at = park[8][s];
            const uint32_t ic = park[9][s];
            pd.i = ic & 0xFFu, pd.count = ic >> 8;
            entry = park[10][s];
            outcome = grid_walk_on<COUNT>(s_grids[entry >> 27], pd, RT_WF_GRID_PASS, n_tests, n_entries);
            if (outcome == GRID_OCCLUDED) atomicAnd(reinterpret_cast<uint32_t*>(&wb.vtx[2 * (size_t)(entry & RT_WF_ID_MASK) + 1]) + 3, ~(1u << (entry >> 27)));
            if (COUNT && outcome <= GRID_OCCLUDED) n_answered++;
        }
        n_park = first;
        park_push(have && outcome == GRID_PENDING, pd, entry);
        hand_on(have && outcome == GRID_FORWARD, entry, false);
    };
    const uint32_t count = wb.counters[rt::WF_EXT_COUNT]; // slots, window padding (sentinels) included
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t id_next = blockIdx.x * blockDim.x + threadIdx.x < count ? queue[blockIdx.x * blockDim.x + threadIdx.x] : WF_SENTINEL;
    for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride) { // (block-uniform bound: the appends are wave operations)
        const uint32_t id = id_next;
        const uint32_t i_next = base + stride + threadIdx.x; // the next entry is fetched a whole iteration ahead
        id_next = i_next < count ? queue[i_next] : WF_SENTINEL;
        uint32_t want = 0;
        V3 point = v3(0, 0, 0), normal = point;
        if (id != WF_SENTINEL) {
            float4 vp = wb.vtx[2 * (size_t)id], vn = wb.vtx[2 * (size_t)id + 1];
            RT_KEEP4(vp);
            RT_KEEP4(vn);
            point = f4v(vp), normal = f4v(vn);
            if (__float_as_uint(vp.w) != 0xFFFFFFFFu) want = __float_as_uint(vn.w); // ("no vertex": the path ended in k_wf_shade)
        }
        uint32_t occluded = 0;
        for (uint32_t li = 0; li < sc.n_lights; li++) {
            const bool mine = (want >> li) & 1u;
            if (__ballot(mine) == 0ull) continue;
            uint32_t outcome = GRID_VISIBLE;
            GridPending pd = {};
            if (mine) {
                outcome = grid_segment_head<COUNT>(sc, s_lights[li], s_grids[li], point, normal, pd, n_tests, n_entries);
                if (outcome == GRID_OCCLUDED) occluded |= 1u << li;
                if (COUNT && outcome <= GRID_OCCLUDED) n_answered++;
            }
            const uint32_t entry = id | (li << 27);
            park_push(mine && outcome == GRID_PENDING, pd, entry);
            hand_on(mine && outcome == GRID_FORWARD, entry, false);
            while (n_park >= 64u) walk_parked(); // (a pass parks again what it leaves undecided: below 64 before the next light adds its own)
        }
        if (occluded != 0u) atomicAnd(reinterpret_cast<uint32_t*>(&wb.vtx[2 * (size_t)id + 1]) + 3, ~occluded);
    }
    while (n_park != 0u) walk_parked();
    hand_on(false, 0u, true);
    if (COUNT) {
        const unsigned long long t = wave_sum(n_tests), a = wave_sum(n_answered), en = wave_sum(n_entries);
        if ((threadIdx.x & 63u) == 0) {
            atomicAdd(&wb.totals[4], t);
            atomicAdd(&wb.totals[13], a);
            atomicAdd(&wb.totals[14], en);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// shading stage 1: consume the closest hit, store the vertex, enqueue shadow segments
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wf_end_path(const rt::WfBuffers& wb, uint32_t id, V3 radiance) {
    wb.sample_rad[id] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
}

__global__ __launch_bounds__(256, RT_WF_SHADE_WAVES) void k_wf_shade(DevScene sc, DevFrame fr, rt::WfBuffers wb, const uint32_t* __restrict__ queue) {
    __shared__ DevLight s_lights[RT_WF_MAX_LIGHTS];
    if (wb.totals[WF_TOTAL_ERROR] != 0ull) return;
    stage_lights(s_lights, sc);
    const uint32_t count = wb.counters[rt::WF_EXT_COUNT];
    const bool shadows = (fr.flags & 2u) == 0;
    // with light grids the shadow stage works per vertex (k_wf_shadow_grid): the lights that need a segment go into the vertex record's
    // visibility word as a mask, not into the shadow queue as entries
    const bool to_grids = shadows && wb.grids != nullptr;
    const uint32_t stride = gridDim.x * blockDim.x;
    OutWindow win = {0u, 0u};
    uint32_t n_shadow = 0;
    const uint32_t window = pick_window((count + stride - 1u) / stride, max(1u, sc.n_lights));
    if (blockIdx.x == 0 && threadIdx.x == 0) wb.counters[rt::WF_SHADOW_WINDOW] = window;
    // every wave runs the same number of iterations so the wave-aggregated appends see whole waves
    uint32_t id_next = blockIdx.x * blockDim.x + threadIdx.x < count ? queue[blockIdx.x * blockDim.x + threadIdx.x] : WF_SENTINEL;
    for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride) {
        const uint32_t id = id_next;
        const uint32_t i_next = base + stride + threadIdx.x; // the next entry is fetched a whole iteration ahead
        id_next = i_next < count ? queue[i_next] : WF_SENTINEL;
        const bool have = id != WF_SENTINEL;
        bool vertex = false;
        V3 point = v3(0, 0, 0), normal = point;
        uint32_t material_id = 0;
        DevMaterial m = {};
        if (have) {
            uint4 h = wb.hit[id];
            asm volatile("" : "+v"(h.x), "+v"(h.y), "+v"(h.z), "+v"(h.w));
            // throughput / radiance are only read where a path ends here (a quarter of this stage's read traffic otherwise)
            if (h.w == RT_PRIM_MISS) { // process_wavefront_ray, wavefront.rs:146-151
                const V3 radiance = f4v(wb.rad[id]) + v3(0.1f, 0.2f, 0.3f) * f4v(wb.thr[id]);
                wf_end_path(wb, id, radiance);
            } else {
                point = v3(__uint_as_float(h.x), __uint_as_float(h.y), __uint_as_float(h.z));
                surface_at(sc, (h.w & RT_PRIM_SPHERE_FLAG) != 0, h.w & ~RT_PRIM_SPHERE_FLAG, point, normal, material_id);
                if (material_id >= sc.n_materials) {
                    const V3 radiance = f4v(wb.rad[id]) + v3(1.0f, 0.0f, 1.0f) * f4v(wb.thr[id]);
                    wf_end_path(wb, id, radiance);
                } else {
                    vertex = true;
                    m = load_material(sc, material_id);
                    wb.vtx[2 * (size_t)id] = make_float4(point.x, point.y, point.z, __uint_as_float(material_id));
                    if (!shadows) wb.vtx[2 * (size_t)id + 1] = make_float4(normal.x, normal.y, normal.z, 0.0f);
                }
            }
        }
        if (have && !vertex) wb.vtx[2 * (size_t)id] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu)); // "no vertex": k_wf_finish skips it
        if (shadows) {
            // which lights need a shadow segment (non-zero contribution)?  One bit per light, then ONE
            // aggregated append per wave: exclusive scan of the per-lane counts + a single atomicAdd.
            uint32_t mask = 0;
            if (vertex) {
                for (uint32_t li = 0; li < sc.n_lights; li++) {
                    V3 sdir;
                    float sdist;
                    const V3 contrib = light_contribution(s_lights[li], m, point, normal, sdir, sdist);
                    if (contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f) mask |= 1u << li;
                }
            }
            // the visibility word of the vertex record starts as this mask; the shadow stage clears the bits of occluded segments
            if (vertex) wb.vtx[2 * (size_t)id + 1] = make_float4(normal.x, normal.y, normal.z, __uint_as_float(mask));
            if (to_grids) { // (k_wf_shadow_grid goes by vertex, not by queue entry)
                n_shadow += (uint32_t)__popc(mask);
                continue;
            }
#if RT_WF_SHADOW_LIGHT_MAJOR
            // light-major order inside the wave's slice: consecutive entries (one traversal wave's refill) go
            // toward the same light from neighbouring vertices
            uint32_t total = 0;
            for (uint32_t li = 0; li < sc.n_lights; li++) total += (uint32_t)__popcll(__ballot((mask >> li) & 1u));
            if (total) {
                uint32_t at = window_reserve(wb.q_shadow, &wb.counters[rt::WF_SHADOW_COUNT], win, window, 0u, 0u, total, wb.q_shadow_cap, &wb.totals[WF_TOTAL_ERROR]);
                n_shadow += (uint32_t)__popc(mask);
                const unsigned long long below = (1ull << (threadIdx.x & 63u)) - 1ull;
                for (uint32_t li = 0; at != WF_NO_SLOT && li < sc.n_lights; li++) {
                    const bool want = (mask >> li) & 1u;
                    const unsigned long long b = __ballot(want);
                    if (want) wb.q_shadow[at + (uint32_t)__popcll(b & below)] = id | (li << 27);
                    at += (uint32_t)__popcll(b);
                }
            }
#else
            const uint32_t mine = (uint32_t)__popc(mask);
            const uint32_t incl = wave_incl_scan(mine), total = __shfl(incl, WAVE - 1, WAVE);
            if (total) {
                uint32_t at = window_reserve(wb.q_shadow, &wb.counters[rt::WF_SHADOW_COUNT], win, window, mine, incl, total, wb.q_shadow_cap, &wb.totals[WF_TOTAL_ERROR]);
                n_shadow += mine;
                while (mask && at != WF_NO_SLOT) {
                    const uint32_t li = (uint32_t)__ffs((int)mask) - 1u;
                    mask &= mask - 1u;
                    wb.q_shadow[at++] = id | (li << 27);
                }
            }
#endif
        }
    }
    window_close(wb.q_shadow, win);
    const unsigned long long ns = wave_sum(n_shadow);
    if ((threadIdx.x & 63u) == 0 && ns) atomicAdd(&wb.totals[2], ns);
}

// End of a bounce iteration: queue sizes move on, cursors are reset (one thread, a launch of its own).  Round 3 tried it inside
// k_wf_finish, run by the last block to end (every block: __syncthreads, __threadfence, atomicAdd on a done counter): the agent-scope
// release fence after a kernel that has just written 64 bytes per path costs more than the launch it saves - k_wf_finish 16.9 -> 20.7 ms
// per headline frame, +0.24 ms per launch on an eighth of it (profiles/ab_r03.json).
__global__ void k_wf_advance(rt::WfBuffers wb) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t* c = wb.counters;
    c[rt::WF_EXT_COUNT] = c[rt::WF_EXT_NEXT];
    c[rt::WF_EXT_WINDOW] = c[rt::WF_EXT_WINDOW_NEXT];
    c[rt::WF_EXT_NEXT] = 0;
    c[rt::WF_SHADOW_COUNT] = 0;
    c[rt::WF_SHADOW2_COUNT] = 0;
    c[rt::WF_SHADOW2_CURSOR] = 0;
    c[rt::WF_EXT_CURSOR] = 0;
    c[rt::WF_SHADOW_CURSOR] = 0;
}

// ---------------------------------------------------------------------------------------------------------
// shading stage 2: ordered light sum, terminal shading or continuation
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, RT_WF_SHADE_WAVES) void k_wf_finish(DevScene sc, DevFrame fr, rt::WfBuffers wb, const uint32_t* __restrict__ queue,
                                                  uint32_t* __restrict__ next_queue) {
    __shared__ DevLight s_lights[RT_WF_MAX_LIGHTS];
    if (wb.totals[WF_TOTAL_ERROR] != 0ull) return;
    stage_lights(s_lights, sc);
    const uint32_t count = wb.counters[rt::WF_EXT_COUNT];
    const bool shadows = (fr.flags & 2u) == 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    OutWindow win = {0u, 0u};
    uint32_t n_cont = 0;
    const uint32_t window = pick_window((count + stride - 1u) / stride, 1u);
    if (blockIdx.x == 0 && threadIdx.x == 0) wb.counters[rt::WF_EXT_WINDOW_NEXT] = window;
    uint32_t id_next = blockIdx.x * blockDim.x + threadIdx.x < count ? queue[blockIdx.x * blockDim.x + threadIdx.x] : WF_SENTINEL;
    for (uint32_t base = blockIdx.x * blockDim.x; base < count; base += stride) {
        bool cont = false;
        const uint32_t id = id_next;
        const uint32_t i_next = base + stride + threadIdx.x; // the next entry is fetched a whole iteration ahead
        id_next = i_next < count ? queue[i_next] : WF_SENTINEL;
        float4 vp = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0xFFFFFFFFu));
        float4 vn = vp, th = vp, ra = vp, rdin = vp;
        if (id != WF_SENTINEL) { // the whole path record in one round trip
            vp = wb.vtx[2 * (size_t)id];
            vn = wb.vtx[2 * (size_t)id + 1];
            th = wb.thr[id];
            ra = wb.rad[id];
            rdin = wb.ray_d[id];
            RT_KEEP4(vp);
            RT_KEEP4(vn);
            RT_KEEP4(th);
            RT_KEEP4(ra);
            RT_KEEP4(rdin);
        }
        const uint32_t vis = __float_as_uint(vn.w); // one bit per light: k_wf_shade's mask of lights with a segment, less those the shadow stage found occluded
        if (__float_as_uint(vp.w) != 0xFFFFFFFFu) { // paths that ended in k_wf_shade carry the "no vertex" marker
            const V3 point = f4v(vp), normal = f4v(vn);
            const uint32_t material_id = __float_as_uint(vp.w);
            const DevMaterial m = load_material(sc, material_id);
            V3 throughput = f4v(th), radiance = f4v(ra);
            const uint32_t tw = __float_as_uint(th.w);
            uint32_t channel = tw & 0xFFu;
            const uint32_t depth = tw >> 8;
            SimpleRng rng = {__float_as_uint(ra.w)};
            const bool terminal = depth >= fr.max_bounce;
            // direct light: the reference's loop order (lighting.rs:33-43), occluded lights skipped
            V3 lighting = v3(0.0f, 0.0f, 0.0f);
            if (terminal) lighting = lighting + ld3(m.albedo) * 0.1f;
            for (uint32_t li = 0; li < sc.n_lights; li++) {
                V3 sdir;
                float sdist;
                const V3 contrib = light_contribution(s_lights[li], m, point, normal, sdir, sdist);
                const bool nonzero = contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f;
                if (nonzero && shadows && !((vis >> li) & 1u)) continue;
                lighting = lighting + contrib;
            }
            lighting = lighting + ld3(m.emission);
            const float tf = fminf(fmaxf(m.transmission, 0.0f), 1.0f);
            if (terminal) {
                V3 out = lighting;
                if (tf > 0.0f) out = transmission_mix(m, lighting, tf);
                radiance = radiance + out * throughput;
                wf_end_path(wb, id, radiance);
            } else {
                radiance = radiance + (lighting * (1.0f - tf)) * throughput;
                const V3 din = f4v(rdin);
                const bool front = dot(normal, din) < 0.0f;
                const V3 nf = front ? normal : -normal;
                bool transmit = false;
                if (tf > 0.0f) transmit = rng.next_f32() < tf;
                V3 ndir, norigin;
                const V3 albedo = ld3(m.albedo);
                bool absorbed = false;
                if (transmit) {
                    if (channel == 3) {
                        uint32_t c = (uint32_t)(rng.next_f32() * 3.0f);
                        channel = c < 2 ? c : 2;
                        throughput = v3(channel == 0 ? throughput.x * 3.0f : 0.0f, channel == 1 ? throughput.y * 3.0f : 0.0f,
                                        channel == 2 ? throughput.z * 3.0f : 0.0f);
                    }
                    const float offs = channel == 0 ? -0.018f : (channel == 1 ? 0.0f : 0.035f);
                    const float ior_c = m.ior + offs;
                    const float eta = front ? (1.0f / ior_c) : ior_c;
                    const float cos_i = -dot(nf, din);
                    const float sin2_t = eta * eta * (1.0f - cos_i * cos_i);
                    if (sin2_t > 1.0f) {
                        ndir = din - nf * (2.0f * dot(din, nf));
                        norigin = point + nf * EXT_EPS;
                    } else {
                        const float cos_t = sqrtf(1.0f - sin2_t);
                        ndir = din * eta + nf * (eta * cos_i - cos_t);
                        norigin = point - nf * EXT_EPS;
                    }
                    ndir = normalize(ndir);
                    throughput = throughput * albedo;
                } else if (m.metallic > 0.5f) {
                    const float u1 = rng.next_f32(), u2 = rng.next_f32();
                    const V3 r = din - nf * (2.0f * dot(din, nf));
                    ndir = normalize(r + unit_vector(u1, u2) * m.roughness);
                    absorbed = !(dot(ndir, nf) > 0.0f);
                    norigin = point + nf * EXT_EPS;
                    if (!absorbed) throughput = throughput * albedo;
                } else {
                    const float u1 = rng.next_f32(), u2 = rng.next_f32();
                    V3 w = nf + unit_vector(u1, u2);
                    if (dot(w, w) < 1e-12f) w = nf;
                    ndir = normalize(w);
                    norigin = point + nf * EXT_EPS;
                    throughput = throughput * albedo;
                }
                if (!absorbed && depth >= 2) {
                    const float p = fminf(fmaxf(fmaxf(fmaxf(throughput.x, throughput.y), throughput.z), 0.05f), 1.0f);
                    if (rng.next_f32() > p) absorbed = true;
                    else throughput = v3(throughput.x / p, throughput.y / p, throughput.z / p);
                }
                if (absorbed) {
                    wf_end_path(wb, id, radiance);
                } else {
                    cont = true;
                    wb.ray_o[id] = make_float4(norigin.x, norigin.y, norigin.z, 0.0f);
                    wb.ray_d[id] = make_float4(ndir.x, ndir.y, ndir.z, 0.0f);
                    wb.thr[id] = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(channel | ((depth + 1u) << 8)));
                    wb.rad[id] = make_float4(radiance.x, radiance.y, radiance.z, __uint_as_float(rng.seed));
                }
            }
        }
        const uint32_t mine = cont ? 1u : 0u;
        const uint32_t incl = wave_incl_scan(mine), total = __shfl(incl, WAVE - 1, WAVE);
        if (total) {
            const uint32_t at = window_reserve(next_queue, &wb.counters[rt::WF_EXT_NEXT], win, window, mine, incl, total, wb.q_ext_cap, &wb.totals[WF_TOTAL_ERROR]);
            if (cont && at != WF_NO_SLOT) next_queue[at] = id;
            n_cont += mine;
        }
    }
    window_close(next_queue, win);
    const unsigned long long nc = wave_sum(n_cont);
    if ((threadIdx.x & 63u) == 0 && nc) atomicAdd(&wb.totals[1], nc);
}

// ---------------------------------------------------------------------------------------------------------
// resolve: add the batch's samples to each pixel's running sum IN SAMPLE ORDER; write the image on the last batch
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void k_wf_resolve(DevFrame fr, rt::WfBuffers wb, DevTargets tg, uint32_t n_samples, uint32_t first_batch,
                                                     uint32_t last_batch) {
    const uint32_t lane = threadIdx.x, b = blockIdx.x;
    const uint32_t q = b * WAVE + lane;
    float4 acc = first_batch ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : wb.accum[q];
    V3 sum = f4v(acc);
    for (uint32_t k = 0; k < n_samples; k++) sum = sum + f4v(wb.sample_rad[(size_t)(RT_WF_BLOCK_MAJOR ? b * n_samples + k : k * wb.n_blocks + b) * WAVE + lane]);
    wb.accum[q] = make_float4(sum.x, sum.y, sum.z, 0.0f);
    if (!last_batch) return;
    const PixelCoord px = block_pixel_at(fr, b, lane);
    if (!px.valid) return;
    const float n = (float)fr.spp;
    const V3 color = v3(sum.x / n, sum.y / n, sum.z / n);
    const size_t pix = (size_t)px.y * fr.width + px.x;
    if (tg.rgba32f) reinterpret_cast<float4*>(tg.rgba32f)[pix] = make_float4(color.x, color.y, color.z, 1.0f);
    if (tg.chan[0]) reinterpret_cast<uint32_t*>(tg.chan[0])[pix] = unorm8(color.x) | 0xFF000000u;
    if (tg.chan[1]) reinterpret_cast<uint32_t*>(tg.chan[1])[pix] = (unorm8(color.y) << 8) | 0xFF000000u;
    if (tg.chan[2]) reinterpret_cast<uint32_t*>(tg.chan[2])[pix] = (unorm8(color.z) << 16) | 0xFF000000u;
}

int g_cu_count = 0;
int cu_count() {
    if (g_cu_count == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cu_count = prop.multiProcessorCount;
        if (g_cu_count <= 0) g_cu_count = 256;
    }
    return g_cu_count;
}

} // namespace

namespace rt {

#ifndef RT_WF_SHADE_BLOCKS_PER_CU
#define RT_WF_SHADE_BLOCKS_PER_CU 16 /* 256-thread blocks per CU for generate / shade / finish: 4, 5, 10 measure 1-3 % slower than 8; 12…32 within 0.5 %, 16 best (-0.7 %) */
#endif
uint32_t wf_shading_blocks() {
    static const int per_cu = [] {
        const char* e = std::getenv("RT_WF_SHADE_BLOCKS_PER_CU"); // development knob
        const int v = e ? std::atoi(e) : 0;
        return v > 0 && v <= 64 ? v : RT_WF_SHADE_BLOCKS_PER_CU;
    }();
    return (uint32_t)(cu_count() * per_cu);
}
// Slots a queue can take up.  A producing wave asks for at most 64 x per_lane entries at a time and a window is at
// least twice that (wf_min_window), so a window that is closed holds more than half real entries: closed windows take
// <= 2 x real entries.  On top comes the one window each wave leaves partly unused at the end.  Its size W is
// pick_window(iterations, per_lane) <= max(minimum, half of what the wave could emit), and "what the wave could emit"
// is derived from the length of the queue it CONSUMES, padding included.  With C the allocation of the consumed queue,
// the last windows of all waves sum to <= C x per_lane / 2 + waves x (minimum + one iteration).  For an extension
// queue (per_lane 1, consuming an extension queue of the same allocation): C <= 2 P + C / 2 + slack, i.e. C <= 4 P +
// slack; for the shadow queue (per_lane L, consuming an extension queue): <= 2 P L + (4 P) L / 2 = 4 P L.  Hence 4 x.
// (Round 1 allocated 2.5 x on an argument that ignored the padding of the consumed queue; the worst case worked out
// to 2.48 x - a 1 % margin.  window_reserve now also checks every reservation against the allocation.)
size_t wf_queue_slots_for(size_t max_entries, uint32_t per_lane, size_t waves) {
    return max_entries * 4 + (waves + 1) * 2 * ((size_t)wf_min_window(per_lane) + 64u * per_lane) + RT_WF_WINDOW_MAX;
}
size_t wf_queue_slots(size_t max_entries, uint32_t per_lane) { return wf_queue_slots_for(max_entries, per_lane, (size_t)wf_shading_blocks() * 4); }
uint32_t wf_pick_window(uint32_t iters, uint32_t per_lane) { return pick_window(iters, per_lane); }
uint32_t wf_persistent_waves() { return (uint32_t)(cu_count() * RT_WF_WAVES_PER_CU); } // the most a launch uses (what the stacks' overflow area is sized for)
// Waves per CU of one traversal launch, from the path slots of the batch it serves (round 3, profiles/ab_r03.json: waves_per_cu_by_batch): the
// fewer segments a launch has, the more its end - waves finishing their last chunk one by one - weighs, and the shorter that end is with
// fewer, busier waves.  An eighth of the headline frame (8 M path slots per batch) is 3.5 % faster with 10-12 waves per CU than with 24
// (23.9 against 24.8 ms), a quarter is level (42.6 ms either way; 16 and 20 are slower), the whole frame (33 M per batch) wants all 24
// (166.2 against 167.6 with 20, 170.2 with 18).
static uint32_t trace_waves_per_cu(uint32_t path_slots) {
    static const int forced = [] { const char* e = std::getenv("RT_WF_WAVES_PER_CU"); const int v = e ? std::atoi(e) : 0; return v > 0 && v <= RT_WF_WAVES_PER_CU ? v : 0; }(); // development knob
    if (forced) return (uint32_t)forced;
    return path_slots >= (24u << 20) ? (uint32_t)RT_WF_WAVES_PER_CU : RT_WF_WAVES_SMALL;
}

hipError_t wf_beams(const DevScene& sc, const DevFrame& fr, const WfBuffers& wb, hipStream_t s) {
    if (!wb.beam_count || wb.n_blocks == 0) return hipSuccess;
    const hipError_t e = hipMemsetAsync(wb.beam_count + wb.n_blocks, 0, sizeof(uint32_t), s); // blocks without a list so far
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_wf_beams, dim3(wb.n_blocks), dim3(WAVE), 0, s, sc, fr, wb);
    return hipGetLastError();
}

hipError_t wf_generate(const DevScene&, const DevFrame& fr, const WfBuffers& wb, uint32_t first_sample, uint32_t n_samples, hipStream_t s) {
    hipError_t e = hipMemsetAsync(wb.counters, 0, WF_N_COUNTERS * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (wb.n_blocks == 0 || n_samples == 0) return hipSuccess;
    hipLaunchKernelGGL(k_wf_generate, dim3(wf_shading_blocks()), dim3(256), 0, s, fr, wb, first_sample, n_samples * wb.n_blocks);
    return hipGetLastError();
}

template <bool COUNT, bool ANY>
static void launch_trace(const DevScene& sc, const WfBuffers& wb, const uint32_t* q, uint32_t count_slot, uint32_t cursor_slot, uint32_t window_slot, hipStream_t s) {
    const size_t lds = (size_t)RT_WF8_LDS_STACK * WAVE * sizeof(uint2) + 2048;
    hipLaunchKernelGGL((k_wf_trace<COUNT, ANY>), dim3((uint32_t)cu_count() * trace_waves_per_cu(wb.capacity)), dim3(WAVE), lds, s, sc, wb, q, count_slot, cursor_slot, window_slot);
}

hipError_t wf_bounce(const DevScene& sc, const DevFrame& fr, const WfBuffers& wb, uint32_t iteration, uint32_t n_samples, bool counters, hipStream_t s, hipEvent_t* grid_events) {
    const dim3 sgrid(wf_shading_blocks()), sblock(256);
    uint32_t* cur_q = wb.q_ext[iteration & 1u];
    uint32_t* next_q = wb.q_ext[(iteration + 1u) & 1u];
    if (iteration == 0 && wb.beam_count && n_samples && wb.n_blocks) {
        // camera segments: every path slot of the batch holds one, sample-block sb = slots [64 sb, 64 sb + 64); from their block's leaf
        // triangle list (k_wf_trace_camera); the blocks without a list were queued by k_wf_generate for the persistent tree walk
        const dim3 lgrid(std::min<uint32_t>(wb.n_blocks * ((n_samples + RT_BEAM_SAMPLES_PER_WAVE - 1u) / RT_BEAM_SAMPLES_PER_WAVE), (uint32_t)cu_count() * 96u));
        const size_t list_lds = (size_t)RT_BEAM_CAP * (3 * sizeof(uint4) + sizeof(float)), walk_lds = (size_t)std::max(1u, sc.stack_entries) * WAVE * sizeof(uint2);
        if (counters) {
            hipLaunchKernelGGL((k_wf_trace_camera<true>), lgrid, dim3(WAVE), wb.probe == 2u ? (size_t)RT_BEAM_CAP * 64 + walk_lds : list_lds, s, sc, wb, n_samples);
            launch_trace<true, false>(sc, wb, wb.q_ext[1], WF_FB_COUNT, WF_FB_CURSOR, WF_SHADOW2_WINDOW, s);
        } else {
            hipLaunchKernelGGL((k_wf_trace_camera<false>), lgrid, dim3(WAVE), list_lds, s, sc, wb, n_samples);
            launch_trace<false, false>(sc, wb, wb.q_ext[1], WF_FB_COUNT, WF_FB_CURSOR, WF_SHADOW2_WINDOW, s);
        }
    } else if (counters) launch_trace<true, false>(sc, wb, cur_q, WF_EXT_COUNT, WF_EXT_CURSOR, WF_EXT_WINDOW, s);
    else launch_trace<false, false>(sc, wb, cur_q, WF_EXT_COUNT, WF_EXT_CURSOR, WF_EXT_WINDOW, s);
    hipLaunchKernelGGL(k_wf_shade, sgrid, sblock, 0, s, sc, fr, wb, (const uint32_t*)cur_q);
    if ((fr.flags & 2u) == 0) {
        if (wb.grids) { // the light grids answer what they can and hand the rest on to the traversal
            const dim3 ggrid((uint32_t)(cu_count() * RT_WF_GRID_BLOCKS_PER_CU));
            if (grid_events) (void)hipEventRecord(grid_events[0], s);
            if (counters) hipLaunchKernelGGL(k_wf_shadow_grid<true>, ggrid, dim3(256), 0, s, sc, wb, (const uint32_t*)cur_q);
            else hipLaunchKernelGGL(k_wf_shadow_grid<false>, ggrid, dim3(256), 0, s, sc, wb, (const uint32_t*)cur_q);
            if (grid_events) (void)hipEventRecord(grid_events[1], s);
            if (counters) launch_trace<true, true>(sc, wb, wb.q_shadow2, WF_SHADOW2_COUNT, WF_SHADOW2_CURSOR, WF_SHADOW2_WINDOW, s);
            else launch_trace<false, true>(sc, wb, wb.q_shadow2, WF_SHADOW2_COUNT, WF_SHADOW2_CURSOR, WF_SHADOW2_WINDOW, s);
        } else if (counters) launch_trace<true, true>(sc, wb, wb.q_shadow, WF_SHADOW_COUNT, WF_SHADOW_CURSOR, WF_SHADOW_WINDOW, s);
        else launch_trace<false, true>(sc, wb, wb.q_shadow, WF_SHADOW_COUNT, WF_SHADOW_CURSOR, WF_SHADOW_WINDOW, s);
    }
    hipLaunchKernelGGL(k_wf_finish, sgrid, sblock, 0, s, sc, fr, wb, (const uint32_t*)cur_q, next_q);
    hipLaunchKernelGGL(k_wf_advance, dim3(1), dim3(1), 0, s, wb);
    return hipGetLastError();
}

hipError_t wf_resolve(const DevFrame& fr, const WfBuffers& wb, const DevTargets& tg, uint32_t n_samples, bool first_batch, bool last_batch, hipStream_t s) {
    if (wb.n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_wf_resolve, dim3(wb.n_blocks), dim3(WAVE), 0, s, fr, wb, tg, n_samples, first_batch ? 1u : 0u, last_batch ? 1u : 0u);
    return hipGetLastError();
}

} // namespace rt
