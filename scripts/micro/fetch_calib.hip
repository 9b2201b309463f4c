// FETCH_SIZE calibration for gathers.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a
// wide coalesced streaming read and is uncalibrated for other shapes.  The light-grid stage reads scattered 16-byte quads: a
// 128-byte cell block (its first 64 bytes, sometimes all of it), a 32-byte vertex record, 48-byte list entries.  Each kernel below
// makes N accesses of one shape at uniformly random places of a table far larger than L2 + Infinity Cache (4 GiB), so the bytes
// that must come from HBM are known per shape; run
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- build/fetch_calib
// and divide each kernel's FETCH_SIZE by its access count (scripts/fetch_calib_summary.py).
// build: hipcc --offload-arch=gfx950 -O3 -o build/fetch_calib scripts/micro/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint64_t rnd(uint64_t x) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return x;
}
__global__ void k_fill(uint4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
// streaming: every lane reads consecutive 16-byte quads (the guide's calibrated case: FETCH_SIZE = bytes / 2)
__global__ void k_stream(const uint4* __restrict__ p, size_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x;
    if (acc == 0x12345u) out[0] = acc;
}
// QUADS consecutive 16-byte quads starting at a random multiple of ALIGN quads
template <int QUADS, int ALIGN>
__global__ void k_gather(const uint4* __restrict__ p, size_t n16, uint32_t iters, uint32_t* out) {
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    const size_t slots = n16 / ALIGN - 1;
    for (uint32_t i = 0; i < iters; i++) {
        x = rnd(x);
        const size_t r = (x % slots) * ALIGN;
        uint4 q[QUADS];
#pragma unroll
        for (int k = 0; k < QUADS; k++) q[k] = p[r + k];
#pragma unroll
        for (int k = 0; k < QUADS; k++) acc += q[k].x;
    }
    if (acc == 0x12345u) out[0] = acc;
}
#define RUN(name, ...)                                                                     \
    do {                                                                                   \
        hipEvent_t a, b;                                                                   \
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);                                \
        (void)hipEventRecord(a);                                                           \
        __VA_ARGS__;                                                                       \
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);                             \
        float ms; (void)hipEventElapsedTime(&ms, a, b);                                    \
        printf("%-44s %8.3f ms\n", name, ms);                                              \
    } while (0)
int main() {
    const size_t n16 = 1ull << 28; // 4 GiB
    uint4* p; uint32_t* out;
    if (hipMalloc(&p, n16 * 16) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, p, n16);
    (void)hipDeviceSynchronize();
    const dim3 grid(256 * 16), block(256);
    const uint32_t iters = 64; // accesses per kernel = 4096 * 256 * 64 = 67,108,864
    printf("accesses per gather kernel: %llu, table 4 GiB\n", (unsigned long long)grid.x * block.x * iters);
    RUN("k_stream (4 GiB, 16 B per lane)", hipLaunchKernelGGL(k_stream, grid, block, 0, 0, p, n16, out));
    RUN("gather 1 quad (16 B), 16-B aligned", hipLaunchKernelGGL((k_gather<1, 1>), grid, block, 0, 0, p, n16, iters, out));
    RUN("gather 2 quads (32 B), 32-B aligned", hipLaunchKernelGGL((k_gather<2, 2>), grid, block, 0, 0, p, n16, iters, out));
    RUN("gather 3 quads (48 B), 48-B aligned", hipLaunchKernelGGL((k_gather<3, 3>), grid, block, 0, 0, p, n16, iters, out));
    RUN("gather 4 quads (64 B), 64-B aligned", hipLaunchKernelGGL((k_gather<4, 4>), grid, block, 0, 0, p, n16, iters, out));
    RUN("gather 4 quads (64 B), 128-B aligned", hipLaunchKernelGGL((k_gather<4, 8>), grid, block, 0, 0, p, n16, iters, out));
    RUN("gather 8 quads (128 B), 128-B aligned", hipLaunchKernelGGL((k_gather<8, 8>), grid, block, 0, 0, p, n16, iters, out));
    RUN("gather 7 quads (112 B), 128-B aligned", hipLaunchKernelGGL((k_gather<7, 8>), grid, block, 0, 0, p, n16, iters, out));
    (void)hipFree(p);
    return 0;
}
