// kernels.h — host-callable launchers of kernels.hip.
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>

#include "device_layout.h"

namespace rt {

uint32_t blocks_per_tile(uint32_t tile_size);

// Modes 0/1 (reference semantics).  Asynchronous on `stream`.
hipError_t launch_render_reference(const DevScene& sc, const DevFrame& fr, const DevTargets& tg, bool counters, hipStream_t stream);

// Mode 2 (extended: jittered spp, shadow rays, bounces).  counters[0] rays, [3] camera, [4] continuation, [5] shadow.
hipError_t launch_render_extended(const DevScene& sc, const DevFrame& fr, const DevTargets& tg, bool counters, hipStream_t stream);

// Read-back epilogues (single device owning the whole frame).  combine: main_fs of the reference (shader/src/lib.rs:383-388),
// out = (red_tex.x, green_tex.y, blue_tex.z, 255).  pack: rgba32f -> tightly packed rgb32f.
hipError_t launch_combine_rgba8(const uint8_t* red, const uint8_t* green, const uint8_t* blue, uint8_t* out, size_t n_pixels, hipStream_t stream);
hipError_t launch_pack_rgb32f(const float* rgba, float* rgb, size_t n_pixels, hipStream_t stream);

} // namespace rt
#endif
