// kernels.hip — gfx950 kernels of the ray-casting hot path.
//
// Replaces the reference's rust-gpu kernel `main_cs` (shader/src/lib.rs:25-89) and
// everything it calls (ray.rs, bvh.rs, intersection.rs, lighting.rs, material.rs,
// wavefront.rs).  One wavefront (64 lanes) = one 8x8 pixel block; each lane owns a
// pixel.  All three colour channels are produced in ONE pass: the trace and the
// lighting are channel independent, only the transmission term of
// shader/src/lib.rs:323-337 depends on the channel, so the reference's three
// dispatches per tile (src/compute.rs:184-190) collapse into one.
//
// Arithmetic that decides WHICH primitive is hit (ray generation, Möller–Trumbore,
// sphere quadratic) and the shading keep the reference's f32 operation order; the file
// is compiled with -ffp-contract=off so nothing is fused implicitly.  The boxes of the
// 8-wide tree are a conservative filter (device_common.h); culling by the closest hit.
#include "kernels.h"

#include <algorithm>

#include "device_common.h"

using namespace rtdev;

namespace {

// ------------------------------------------------------------------------------------
// k_render_reference: modes 0 and 1 (reference semantics): one pixel-centre ray per pixel,
// closest hit, direct lighting; miss = black (mode 0, lib.rs:77) or sky (mode 1,
// wavefront.rs:146-151).  Mode 1 only produces colour when current_bounce <= max_bounce
// (lib.rs:117-121), otherwise (0,0,0).
// ------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(WAVE) void k_render_reference(DevScene sc, DevFrame fr, DevTargets tg) {
    extern __shared__ uint2 s_stack[]; // DevScene::stack_entries * 64 64-bit entries
    PixelCoord px = block_pixel(fr);
    if (!px.valid) return;
    uint2* stack = s_stack + threadIdx.x;
    Counts cnt = {0u, 0u};
    const bool wavefront = fr.mode != 0;
    V3 color = v3(0.0f, 0.0f, 0.0f);
    Hit hit;
    hit.t = RT_F32_MAX;
    hit.prim = RT_PRIM_MISS;
    hit.slot = 0;
    const bool traced = !wavefront || fr.cur_bounce <= fr.max_bounce;
    if (traced) {
        V3 o, d;
        camera_ray(fr.cam, (float)px.x + 0.5f, (float)px.y + 0.5f, wavefront, o, d);
        hit = find_closest<COUNT>(sc, o, d, stack, cnt);
        if (hit.prim != RT_PRIM_MISS) color = shade_hit(sc, hit, o, d);
        else if (wavefront) color = v3(0.1f, 0.2f, 0.3f);
    }
    const size_t pix = (size_t)px.y * fr.width + px.x;
    if (tg.rgba32f) reinterpret_cast<float4*>(tg.rgba32f)[pix] = make_float4(color.x, color.y, color.z, 1.0f);
    if (tg.prim_id) tg.prim_id[pix] = hit.prim;
    if (tg.hit_t) tg.hit_t[pix] = hit.t;
    // channel texture c keeps only component c (filter_color_by_channel) and alpha 1
    if ((fr.channel_mask & 1u) && tg.chan[0]) reinterpret_cast<uint32_t*>(tg.chan[0])[pix] = unorm8(color.x) | 0xFF000000u;
    if ((fr.channel_mask & 2u) && tg.chan[1]) reinterpret_cast<uint32_t*>(tg.chan[1])[pix] = (unorm8(color.y) << 8) | 0xFF000000u;
    if ((fr.channel_mask & 4u) && tg.chan[2]) reinterpret_cast<uint32_t*>(tg.chan[2])[pix] = (unorm8(color.z) << 16) | 0xFF000000u;
    if (COUNT && tg.counters) {
        atomicAdd(&tg.counters[0], traced ? 1ull : 0ull);
        atomicAdd(&tg.counters[1], (unsigned long long)cnt.nodes);
        atomicAdd(&tg.counters[2], (unsigned long long)cnt.tris);
        atomicAdd(&tg.counters[3], traced ? 1ull : 0ull);
    }
}


// any hit in (1e-5, tmax)?  Spheres first, then the BVH with early exit.
template <bool COUNT>
__device__ __forceinline__ bool occluded(const DevScene& sc, V3 o, V3 d, float tmax, uint2* stack, Counts& cnt) {
    Hit h;
    h.t = tmax;
    h.prim = RT_PRIM_MISS;
    h.slot = 0;
    test_spheres(sc, o, d, h);
    if (h.prim != RT_PRIM_MISS) return true;
    traverse<COUNT, true>(sc, o, d, stack, h, cnt);
    return h.prim != RT_PRIM_MISS;
}

template <bool COUNT>
__device__ __forceinline__ V3 ext_direct(const DevScene& sc, const DevMaterial& m, V3 point, V3 normal, bool ambient, bool shadows,
                                         uint2* stack, Counts& cnt, SegCounts& seg) {
    V3 total = v3(0.0f, 0.0f, 0.0f);
    if (ambient) total = total + ld3(m.albedo) * 0.1f;
    for (uint32_t li = 0; li < sc.n_lights; li++) {
        V3 sdir;
        float sdist;
        V3 contrib = light_contribution(sc.lights[li], m, point, normal, sdir, sdist);
        if (!(contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f)) {
            total = total + contrib;
            continue;
        }
        if (shadows) {
            seg.shadow++;
            if (occluded<COUNT>(sc, point + normal * EXT_EPS, sdir, sdist, stack, cnt)) continue;
        }
        total = total + contrib;
    }
    return total + ld3(m.emission);
}

template <bool COUNT>
__device__ __forceinline__ V3 ext_trace_path(const DevScene& sc, const DevFrame& fr, uint32_t px, uint32_t py, uint32_t sample,
                                             uint2* stack, Counts& cnt, SegCounts& seg) {
    SimpleRng rng = rng_for(fr.frame_seed + px + py * fr.width, sample);
    float jx = 0.5f, jy = 0.5f;
    if (fr.spp > 1) {
        jx = rng.next_f32();
        jy = rng.next_f32();
    }
    V3 o, d;
    camera_ray(fr.cam, (float)px + jx, (float)py + jy, true, o, d);
    V3 radiance = v3(0.0f, 0.0f, 0.0f);
    V3 throughput = v3(1.0f, 1.0f, 1.0f);
    uint32_t channel = 3;
    const bool shadows = (fr.flags & 2u) == 0;
    for (uint32_t depth = 0;; depth++) {
        if (depth == 0) seg.camera++; else seg.continuation++;
        Hit hit = find_closest<COUNT>(sc, o, d, stack, cnt);
        if (hit.prim == RT_PRIM_MISS) {
            radiance = radiance + v3(0.1f, 0.2f, 0.3f) * throughput;
            break;
        }
        V3 point, normal;
        uint32_t material_id;
        hit_geometry(sc, hit, o, d, point, normal, material_id);
        if (material_id >= sc.n_materials) {
            radiance = radiance + v3(1.0f, 0.0f, 1.0f) * throughput;
            break;
        }
        const DevMaterial m = sc.materials[material_id];
        float tf = fminf(fmaxf(m.transmission, 0.0f), 1.0f);
        bool terminal = depth >= fr.max_bounce;
        V3 lighting = ext_direct<COUNT>(sc, m, point, normal, terminal, shadows, stack, cnt, seg);
        if (terminal) {
            V3 out = lighting;
            if (tf > 0.0f) out = transmission_mix(m, lighting, tf);
            radiance = radiance + out * throughput;
            break;
        }
        radiance = radiance + (lighting * (1.0f - tf)) * throughput;

        bool front = dot(normal, d) < 0.0f;
        V3 nf = front ? normal : -normal;
        bool transmit = false;
        if (tf > 0.0f) transmit = rng.next_f32() < tf;
        V3 ndir, norigin;
        V3 albedo = ld3(m.albedo);
        if (transmit) {
            if (channel == 3) {
                uint32_t c = (uint32_t)(rng.next_f32() * 3.0f);
                channel = c < 2 ? c : 2;
                throughput = v3(channel == 0 ? throughput.x * 3.0f : 0.0f, channel == 1 ? throughput.y * 3.0f : 0.0f,
                                channel == 2 ? throughput.z * 3.0f : 0.0f);
            }
            float offs = channel == 0 ? -0.018f : (channel == 1 ? 0.0f : 0.035f); // material.rs:47-52
            float ior_c = m.ior + offs;
            float eta = front ? (1.0f / ior_c) : ior_c;
            float cos_i = -dot(nf, d);
            float sin2_t = eta * eta * (1.0f - cos_i * cos_i);
            if (sin2_t > 1.0f) {
                ndir = d - nf * (2.0f * dot(d, nf));
                norigin = point + nf * EXT_EPS;
            } else {
                float cos_t = sqrtf(1.0f - sin2_t);
                ndir = d * eta + nf * (eta * cos_i - cos_t);
                norigin = point - nf * EXT_EPS;
            }
            ndir = normalize(ndir);
            throughput = throughput * albedo;
        } else if (m.metallic > 0.5f) {
            float u1 = rng.next_f32(), u2 = rng.next_f32();
            V3 r = d - nf * (2.0f * dot(d, nf));
            ndir = normalize(r + unit_vector(u1, u2) * m.roughness);
            if (!(dot(ndir, nf) > 0.0f)) break;
            norigin = point + nf * EXT_EPS;
            throughput = throughput * albedo;
        } else {
            float u1 = rng.next_f32(), u2 = rng.next_f32();
            V3 w = nf + unit_vector(u1, u2);
            if (dot(w, w) < 1e-12f) w = nf;
            ndir = normalize(w);
            norigin = point + nf * EXT_EPS;
            throughput = throughput * albedo;
        }
        if (depth >= 2) {
            float p = fminf(fmaxf(fmaxf(fmaxf(throughput.x, throughput.y), throughput.z), 0.05f), 1.0f);
            if (rng.next_f32() > p) break;
            throughput = v3(throughput.x / p, throughput.y / p, throughput.z / p);
        }
        o = norigin;
        d = ndir;
    }
    return radiance;
}

template <bool COUNT>
__global__ __launch_bounds__(WAVE) void k_render_extended(DevScene sc, DevFrame fr, DevTargets tg) {
    extern __shared__ uint2 s_stack[]; // DevScene::stack_entries * 64 64-bit entries
    PixelCoord px = block_pixel(fr);
    uint2* stack = s_stack + threadIdx.x;
    Counts cnt = {0u, 0u};
    SegCounts seg = {0u, 0u, 0u};
    if (px.valid) {
        V3 sum = v3(0.0f, 0.0f, 0.0f);
        for (uint32_t s = 0; s < fr.spp; s++) sum = sum + ext_trace_path<COUNT>(sc, fr, px.x, px.y, s, stack, cnt, seg);
        float n = (float)fr.spp;
        V3 color = v3(sum.x / n, sum.y / n, sum.z / n);
        const size_t pix = (size_t)px.y * fr.width + px.x;
        if (tg.rgba32f) reinterpret_cast<float4*>(tg.rgba32f)[pix] = make_float4(color.x, color.y, color.z, 1.0f);
        if (tg.chan[0]) reinterpret_cast<uint32_t*>(tg.chan[0])[pix] = unorm8(color.x) | 0xFF000000u;
        if (tg.chan[1]) reinterpret_cast<uint32_t*>(tg.chan[1])[pix] = (unorm8(color.y) << 8) | 0xFF000000u;
        if (tg.chan[2]) reinterpret_cast<uint32_t*>(tg.chan[2])[pix] = (unorm8(color.z) << 16) | 0xFF000000u;
    }
    // segment counts are part of the result (rt_stats.rays): one atomic per wave and counter
    unsigned long long c0 = wave_sum(seg.camera), c1 = wave_sum(seg.continuation), c2 = wave_sum(seg.shadow);
    unsigned long long n0 = COUNT ? wave_sum(cnt.nodes) : 0ull, n1 = COUNT ? wave_sum(cnt.tris) : 0ull;
    if (threadIdx.x == 0 && tg.counters) {
        atomicAdd(&tg.counters[0], c0 + c1 + c2);
        atomicAdd(&tg.counters[3], c0);
        atomicAdd(&tg.counters[4], c1);
        atomicAdd(&tg.counters[5], c2);
        if (COUNT) {
            atomicAdd(&tg.counters[1], n0);
            atomicAdd(&tg.counters[2], n1);
        }
    }
}


// ====================================================================================
// k_render_extended_sm — extended mode as a per-lane STATE MACHINE (v2).
//
// v1 (k_render_extended) nests the loops samples -> path vertices -> lights -> traversal, so a
// wave runs every traversal in lockstep and lanes whose segment ends early idle until the
// slowest lane is done (measured: ~17 % lane utilisation, profiles/r01_extended_*_v1).  Here
// every lane carries its own path state and the wave alternates between two phases:
//   traversal phase  - one BVH step at a time for every lane that has a segment in flight
//                      (closest-hit and any-hit shadow segments mixed); a lane whose segment
//                      finishes parks;
//   transition phase - entered when no lane is traversing or RT_SM_PARK_THRESHOLD lanes are
//                      parked (__ballot + popcount): parked lanes consume their result (shade,
//                      next light, next bounce, next sample) and set up their next segment.
// Per-lane arithmetic and its order are exactly those of v1 / the CPU statement; only the
// interleaving across lanes changes, so results stay bit-identical.
// ====================================================================================
#ifndef RT_SM_MIN_WAVES
#define RT_SM_MIN_WAVES 4
#endif
#ifndef RT_SM_PARK_THRESHOLD
#define RT_SM_PARK_THRESHOLD 8
#endif

enum : uint32_t { ST_NEW_SAMPLE = 0, ST_CLOSEST_DONE = 1, ST_SHADOW_DONE = 2, ST_LIGHTS = 3, ST_TRAVERSING = 4, ST_DONE = 5 };

template <bool COUNT>
__global__ __launch_bounds__(WAVE, RT_SM_MIN_WAVES) void k_render_extended_sm(DevScene sc, DevFrame fr, DevTargets tg) {
    extern __shared__ uint2 s_stack[]; // DevScene::stack_entries * 64 64-bit entries
    const PixelCoord px = block_pixel(fr);
    uint2* __restrict__ stack = s_stack + threadIdx.x;
    const uint4* __restrict__ nodes = reinterpret_cast<const uint4*>(sc.nodes);
    Counts cnt = {0u, 0u};
    SegCounts seg = {0u, 0u, 0u};
    const bool shadows = (fr.flags & 2u) == 0;

    // ---- per-lane path state ----
    uint32_t state = px.valid ? ST_NEW_SAMPLE : ST_DONE;
    uint32_t sample = 0, depth = 0, channel = 3, li = 0, material_id = 0;
    SimpleRng rng = {0u};
    V3 sum = v3(0.0f, 0.0f, 0.0f), radiance = sum, throughput = sum, lighting = sum, pending = sum;
    V3 point = sum, normal = sum, din = sum; // vertex position, geometric normal, incoming direction
    bool terminal = false;
    // ---- per-lane segment (traversal) state ----
    V3 o = sum, d = sum;
    FilterRay fray;
    fray.o = sum;
    fray.inv = sum;
    Hit hit;
    hit.t = RT_F32_MAX;
    hit.prim = RT_PRIM_MISS;
    hit.slot = 0;
    uint32_t g_base = 0, g_bits = 0; // inner children still to visit: (child_base, hits | imask << 8)
    uint32_t t_base = 0, t_bits = 0; // leaves of the last visited node still to test: (tri_base, hits | lmask << 8)
    uint32_t oct = 0;
    int sp = 0;
    bool anyhit = false;

    // start a segment: spheres are tested right away (wave-uniform loop), the BVH walk is deferred to the traversal phase
    auto begin_segment = [&](V3 so, V3 sd, float tmax, bool any) {
        o = so;
        d = sd;
        fray = make_filter_ray(so, sd);
        hit.t = tmax;
        hit.prim = RT_PRIM_MISS;
        hit.slot = 0;
        anyhit = any;
        oct = ray_octant(fray);
        test_spheres(sc, so, sd, hit);
        sp = 0;
        g_base = 0;
        g_bits = 1u | (1u << 8); // the root as the only child of a group
        t_bits = 0;
        const bool finished = sc.n_nodes == 0 || (any && hit.prim != RT_PRIM_MISS);
        state = finished ? (any ? ST_SHADOW_DONE : ST_CLOSEST_DONE) : ST_TRAVERSING;
    };

    // diagnostics of the counting variant (wave-level, lane 0 accumulates): [8] transition passes, [9] lanes served,
    // [10] node iterations, [11] lanes active in them, [12] leaf iterations, [13] lanes active, [14]/[15] cycles in
    // transition / traversal phases
    unsigned long long dg_tp = 0, dg_tl = 0, dg_ni = 0, dg_nl = 0, dg_li = 0, dg_ll = 0, dg_ct = 0, dg_cv = 0;
    for (;;) {
        // =========================== transition phase ===========================
        unsigned long long t_begin = 0;
        if (COUNT) {
            unsigned long long need = __ballot(state != ST_TRAVERSING && state != ST_DONE);
            if (need) {
                dg_tp++;
                dg_tl += __popcll(need);
            }
            t_begin = __builtin_readcyclecounter();
        }
        while (state != ST_TRAVERSING && state != ST_DONE) {
            if (state == ST_NEW_SAMPLE) {
                if (sample >= fr.spp) {
                    float n = (float)fr.spp;
                    V3 color = v3(sum.x / n, sum.y / n, sum.z / n);
                    const size_t pix = (size_t)px.y * fr.width + px.x;
                    if (tg.rgba32f) reinterpret_cast<float4*>(tg.rgba32f)[pix] = make_float4(color.x, color.y, color.z, 1.0f);
                    if (tg.chan[0]) reinterpret_cast<uint32_t*>(tg.chan[0])[pix] = unorm8(color.x) | 0xFF000000u;
                    if (tg.chan[1]) reinterpret_cast<uint32_t*>(tg.chan[1])[pix] = (unorm8(color.y) << 8) | 0xFF000000u;
                    if (tg.chan[2]) reinterpret_cast<uint32_t*>(tg.chan[2])[pix] = (unorm8(color.z) << 16) | 0xFF000000u;
                    state = ST_DONE;
                    break;
                }
                rng = rng_for(fr.frame_seed + px.x + px.y * fr.width, sample);
                float jx = 0.5f, jy = 0.5f;
                if (fr.spp > 1) {
                    jx = rng.next_f32();
                    jy = rng.next_f32();
                }
                V3 co, cd;
                camera_ray(fr.cam, (float)px.x + jx, (float)px.y + jy, true, co, cd);
                radiance = v3(0.0f, 0.0f, 0.0f);
                throughput = v3(1.0f, 1.0f, 1.0f);
                channel = 3;
                depth = 0;
                seg.camera++;
                begin_segment(co, cd, RT_F32_MAX, false);
                continue;
            }
            bool end_sample = false;
            if (state == ST_CLOSEST_DONE) {
                if (hit.prim == RT_PRIM_MISS) {
                    radiance = radiance + v3(0.1f, 0.2f, 0.3f) * throughput;
                    end_sample = true;
                } else {
                    hit_geometry(sc, hit, o, d, point, normal, material_id);
                    din = d;
                    if (material_id >= sc.n_materials) {
                        radiance = radiance + v3(1.0f, 0.0f, 1.0f) * throughput;
                        end_sample = true;
                    } else {
                        terminal = depth >= fr.max_bounce;
                        lighting = v3(0.0f, 0.0f, 0.0f);
                        if (terminal) lighting = lighting + ld3(sc.materials[material_id].albedo) * 0.1f;
                        li = 0;
                        state = ST_LIGHTS;
                    }
                }
            } else if (state == ST_SHADOW_DONE) {
                if (hit.prim == RT_PRIM_MISS) lighting = lighting + pending;
                li++;
                state = ST_LIGHTS;
            }
            if (!end_sample && state == ST_LIGHTS) {
                const DevMaterial m = sc.materials[material_id];
                bool launched = false;
                while (li < sc.n_lights) {
                    V3 sdir;
                    float sdist;
                    V3 contrib = light_contribution(sc.lights[li], m, point, normal, sdir, sdist);
                    if ((contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f) && shadows) {
                        pending = contrib;
                        seg.shadow++;
                        launched = true;
                        begin_segment(point + normal * EXT_EPS, sdir, sdist, true); // overwrites o/d: the incoming direction lives in `din`
                        break;
                    }
                    lighting = lighting + contrib;
                    li++;
                }
                if (launched) continue;
                // ---- all lights done: finish the vertex ----
                lighting = lighting + ld3(m.emission);
                float tf = fminf(fmaxf(m.transmission, 0.0f), 1.0f);
                if (terminal) {
                    V3 out = lighting;
                    if (tf > 0.0f) out = transmission_mix(m, lighting, tf);
                    radiance = radiance + out * throughput;
                    end_sample = true;
                } else {
                    radiance = radiance + (lighting * (1.0f - tf)) * throughput;
                    bool front = dot(normal, din) < 0.0f;
                    V3 nf = front ? normal : -normal;
                    bool transmit = false;
                    if (tf > 0.0f) transmit = rng.next_f32() < tf;
                    V3 ndir, norigin;
                    V3 albedo = ld3(m.albedo);
                    bool absorbed = false;
                    if (transmit) {
                        if (channel == 3) {
                            uint32_t c = (uint32_t)(rng.next_f32() * 3.0f);
                            channel = c < 2 ? c : 2;
                            throughput = v3(channel == 0 ? throughput.x * 3.0f : 0.0f, channel == 1 ? throughput.y * 3.0f : 0.0f,
                                            channel == 2 ? throughput.z * 3.0f : 0.0f);
                        }
                        float offs = channel == 0 ? -0.018f : (channel == 1 ? 0.0f : 0.035f);
                        float ior_c = m.ior + offs;
                        float eta = front ? (1.0f / ior_c) : ior_c;
                        float cos_i = -dot(nf, din);
                        float sin2_t = eta * eta * (1.0f - cos_i * cos_i);
                        if (sin2_t > 1.0f) {
                            ndir = din - nf * (2.0f * dot(din, nf));
                            norigin = point + nf * EXT_EPS;
                        } else {
                            float cos_t = sqrtf(1.0f - sin2_t);
                            ndir = din * eta + nf * (eta * cos_i - cos_t);
                            norigin = point - nf * EXT_EPS;
                        }
                        ndir = normalize(ndir);
                        throughput = throughput * albedo;
                    } else if (m.metallic > 0.5f) {
                        float u1 = rng.next_f32(), u2 = rng.next_f32();
                        V3 r = din - nf * (2.0f * dot(din, nf));
                        ndir = normalize(r + unit_vector(u1, u2) * m.roughness);
                        absorbed = !(dot(ndir, nf) > 0.0f);
                        norigin = point + nf * EXT_EPS;
                        if (!absorbed) throughput = throughput * albedo;
                    } else {
                        float u1 = rng.next_f32(), u2 = rng.next_f32();
                        V3 w = nf + unit_vector(u1, u2);
                        if (dot(w, w) < 1e-12f) w = nf;
                        ndir = normalize(w);
                        norigin = point + nf * EXT_EPS;
                        throughput = throughput * albedo;
                    }
                    if (!absorbed && depth >= 2) {
                        float p = fminf(fmaxf(fmaxf(fmaxf(throughput.x, throughput.y), throughput.z), 0.05f), 1.0f);
                        if (rng.next_f32() > p) absorbed = true;
                        else throughput = v3(throughput.x / p, throughput.y / p, throughput.z / p);
                    }
                    if (absorbed) {
                        end_sample = true;
                    } else {
                        depth++;
                        seg.continuation++;
                        begin_segment(norigin, ndir, RT_F32_MAX, false);
                    }
                }
            }
            if (end_sample) {
                sum = sum + radiance;
                sample++;
                state = ST_NEW_SAMPLE;
            }
        }
        if (COUNT) {
            unsigned long long t_mid = __builtin_readcyclecounter();
            dg_ct += t_mid - t_begin;
            t_begin = t_mid;
        }
        // =========================== anything left? ===========================
        if (__ballot(state == ST_TRAVERSING) == 0ull) {
            if (__ballot(state != ST_DONE) == 0ull) break;
            continue;
        }
        // =========================== traversal phase ===========================
        // while-while on groups: every lane that has an inner child to visit (and no leaves waiting) steps until none has,
        // then the lanes whose last visit entered leaves test them.  (A speculative variant that postponed one leaf per lane
        // and kept descending raised lane utilisation of the node steps from 26 % to 48 % in round 1 but was not faster here;
        // the queue pipeline's persistent kernels do postpone, wavefront.hip.)
        for (;;) {
            for (;;) {
                const bool trav = state == ST_TRAVERSING;
                if (trav && (t_bits & 0xFFu) == 0u && (g_bits & 0xFFu) == 0u && sp > 0) {
                    sp--;
                    const uint2 e = stack[sp * WAVE];
                    g_base = e.x;
                    g_bits = e.y;
                }
                const bool want = trav && (t_bits & 0xFFu) == 0u && (g_bits & 0xFFu) != 0u;
                const unsigned long long wmask = __ballot(want);
                if (wmask == 0ull) break;
                if (COUNT) {
                    dg_ni++;
                    dg_nl += __popcll(wmask);
                }
                if (want) {
                    const uint32_t i = first_slot(g_bits, oct);
                    g_bits ^= 1u << i;
                    const uint32_t node = g_base + (uint32_t)__popc(__builtin_amdgcn_ubfe(g_bits, 8u, i));
                    if (g_bits & 0xFFu) {
                        stack[sp * WAVE] = make_uint2(g_base, g_bits);
                        sp++;
                    }
                    uint32_t cb, tb, im, lm;
                    const uint32_t hm = visit_node8<COUNT>(nodes, node, fray, hit.t, cnt, cb, tb, im, lm);
                    g_base = cb;
                    g_bits = (hm & im) | (im << 8);
                    t_base = tb;
                    t_bits = (hm & lm) | (lm << 8);
                }
            }
            if (COUNT) {
                unsigned long long lm_ = __ballot(state == ST_TRAVERSING && (t_bits & 0xFFu) != 0u);
                if (lm_) {
                    dg_li++;
                    dg_ll += __popcll(lm_);
                }
            }
            if (state == ST_TRAVERSING) {
                // `anyhit` differs between lanes, so one loop serves both kinds (any-hit lanes leave it at their first hit)
                bool stop = false;
                while ((t_bits & 0xFFu) != 0u && !stop) {
                    const uint32_t sl = first_slot(t_bits, oct);
                    t_bits ^= 1u << sl;
                    const uint32_t start = t_base + RT_DEV_LEAF_STRIDE * (uint32_t)__popc(__builtin_amdgcn_ubfe(t_bits, 8u, sl));
                    uint32_t n_tri = 1;
                    for (uint32_t i = 0; i < n_tri; i++) {
                        if (COUNT) cnt.tris++;
                        const uint32_t lc = test_triangle(sc.tris, start + i, o, d, hit);
                        if (i == 0) n_tri = lc;
                        if (anyhit && hit.prim != RT_PRIM_MISS) {
                            stop = true;
                            break;
                        }
                    }
                }
                if (stop) state = ST_SHADOW_DONE;
                else if ((g_bits & 0xFFu) == 0u && sp == 0) state = anyhit ? ST_SHADOW_DONE : ST_CLOSEST_DONE;
            }
            const unsigned long long still = __ballot(state == ST_TRAVERSING);
            const unsigned long long parked = __ballot(state != ST_TRAVERSING && state != ST_DONE);
            if (still == 0ull || __popcll(parked) >= RT_SM_PARK_THRESHOLD) break;
        }
        if (COUNT) dg_cv += __builtin_readcyclecounter() - t_begin;
    }
    unsigned long long c0 = wave_sum(seg.camera), c1 = wave_sum(seg.continuation), c2 = wave_sum(seg.shadow);
    unsigned long long n0 = COUNT ? wave_sum(cnt.nodes) : 0ull, n1 = COUNT ? wave_sum(cnt.tris) : 0ull;
    if (threadIdx.x == 0 && tg.counters) {
        atomicAdd(&tg.counters[0], c0 + c1 + c2);
        atomicAdd(&tg.counters[3], c0);
        atomicAdd(&tg.counters[4], c1);
        atomicAdd(&tg.counters[5], c2);
        if (COUNT) {
            atomicAdd(&tg.counters[1], n0);
            atomicAdd(&tg.counters[2], n1);
            atomicAdd(&tg.counters[8], dg_tp);
            atomicAdd(&tg.counters[9], dg_tl);
            atomicAdd(&tg.counters[10], dg_ni);
            atomicAdd(&tg.counters[11], dg_nl);
            atomicAdd(&tg.counters[12], dg_li);
            atomicAdd(&tg.counters[13], dg_ll);
            atomicAdd(&tg.counters[14], dg_ct);
            atomicAdd(&tg.counters[15], dg_cv);
        }
    }
}

} // namespace

namespace rt {

// The one-ray-per-lane walks park only the group of siblings still to visit (a node's leaves are tested right after its visit),
// one entry per level: half of DevScene::stack_entries, which is sized for the queue pipeline's walk (it also parks postponed leaf
// groups).  At 512 bytes per entry and wave this decides how many waves fit a CU: depth 11 -> 6 KB, 26 waves per CU.
static size_t lds_bytes(const DevScene& sc) { return (size_t)(sc.stack_entries / 2u + 1u) * WAVE * sizeof(uint2); }

__global__ __launch_bounds__(256) void k_combine_rgba8(const uint32_t* __restrict__ red, const uint32_t* __restrict__ green, const uint32_t* __restrict__ blue,
                                                        uint32_t* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = (red[i] & 0x000000FFu) | (green[i] & 0x0000FF00u) | (blue[i] & 0x00FF0000u) | 0xFF000000u;
}
__global__ __launch_bounds__(256) void k_pack_rgb32f(const float4* __restrict__ rgba, float* __restrict__ rgb, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = rgba[i];
        rgb[3 * i + 0] = v.x;
        rgb[3 * i + 1] = v.y;
        rgb[3 * i + 2] = v.z;
    }
}

uint32_t blocks_per_tile(uint32_t tile_size) {
    uint32_t b = (tile_size + 7u) >> 3;
    return b * b;
}

hipError_t launch_combine_rgba8(const uint8_t* red, const uint8_t* green, const uint8_t* blue, uint8_t* out, size_t n_pixels, hipStream_t stream) {
    if (n_pixels == 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<size_t>((n_pixels + 255) / 256, 4096);
    hipLaunchKernelGGL(k_combine_rgba8, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(red), reinterpret_cast<const uint32_t*>(green),
                       reinterpret_cast<const uint32_t*>(blue), reinterpret_cast<uint32_t*>(out), n_pixels);
    return hipGetLastError();
}
hipError_t launch_pack_rgb32f(const float* rgba, float* rgb, size_t n_pixels, hipStream_t stream) {
    if (n_pixels == 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<size_t>((n_pixels + 255) / 256, 4096);
    hipLaunchKernelGGL(k_pack_rgb32f, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(rgba), rgb, n_pixels);
    return hipGetLastError();
}

hipError_t launch_render_reference(const DevScene& sc, const DevFrame& fr, const DevTargets& tg, bool counters, hipStream_t stream) {
    uint32_t n_tiles = fr.single_tile ? 1u : fr.n_owned_tiles;
    if (n_tiles == 0) return hipSuccess;
    dim3 grid(n_tiles * blocks_per_tile(fr.tile_size)), block(WAVE);
    if (counters)
        hipLaunchKernelGGL(k_render_reference<true>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
    else
        hipLaunchKernelGGL(k_render_reference<false>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
    return hipGetLastError();
}

hipError_t launch_render_extended(const DevScene& sc, const DevFrame& fr, const DevTargets& tg, bool counters, hipStream_t stream) {
    uint32_t n_tiles = fr.n_owned_tiles;
    if (n_tiles == 0) return hipSuccess;
    dim3 grid(n_tiles * blocks_per_tile(fr.tile_size)), block(WAVE);
    const bool v1 = (fr.flags & 4u) != 0; // RT_FLAG_KERNEL_V1: nested loops; otherwise (RT_FLAG_KERNEL_SM) the state machine
    if (v1) {
        if (counters)
            hipLaunchKernelGGL(k_render_extended<true>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
        else
            hipLaunchKernelGGL(k_render_extended<false>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
    } else {
        if (counters)
            hipLaunchKernelGGL(k_render_extended_sm<true>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
        else
            hipLaunchKernelGGL(k_render_extended_sm<false>, grid, block, lds_bytes(sc), stream, sc, fr, tg);
    }
    return hipGetLastError();
}

} // namespace rt
