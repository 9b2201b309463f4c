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

} // namespace rt
#endif
