// device_build.h — the acceleration structure built on the GPU (device_build.hip): Morton sort, PLOC, the 8-slot collapse program
// and the emission of DevNode8 / DevTri arrays, all on the device the scene will live on.
#ifndef RT_DEVICE_BUILD_H
#define RT_DEVICE_BUILD_H

#include <hip/hip_runtime.h>

#include "bvh_builder.h"

namespace rt {

struct DeviceBuild {
    DevNode8* nodes = nullptr; // device memory, owned by the caller after a successful build
    DevTri* tris = nullptr;
    uint32_t n_nodes = 0, n_tris = 0, depth = 0, n_leaves = 0;
};

// Builds on the current device; `tris` is host memory (copied).  Uses opt.ploc_radius, max_leaf, cost_traverse8, cost_intersect.
// Meant for scenes of at least a few hundred triangles (the caller leaves tiny ones to the host builder).  n_nodes == 0 on return
// with hipSuccess means that no triangle had finite coordinates.
hipError_t device_build(const BuildTri* tris, size_t n, const BvhBuildOptions& opt, hipStream_t stream, DeviceBuild* out);

} // namespace rt
#endif
