// Random gathers on MI355X: reads per second by footprint and by shape (is a 1 GB gather slower than a 32 MB one?  do three
// 16-byte loads of one 48-byte record cost three requests or one line?).
// build: hipcc --offload-arch=gfx950 -O3 -o build/random_read scripts/micro/random_read.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_fill(uint4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
// mode 0: one 8-byte read per step; 1: one 16-byte read; 2: three 16-byte reads of one 48-byte record; 3: four 16-byte reads of one
// 64-byte aligned record; 4: an 8-byte read, then (dependent) three 16-byte reads of a record elsewhere
__global__ void k_gather(const uint4* __restrict__ p, size_t n16, uint32_t iters, uint32_t mode, unsigned long long* out) {
    uint64_t x = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        if (mode == 0) {
            acc += reinterpret_cast<const uint2*>(p)[(x + acc) % (n16 * 2)].x & 1u;
        } else if (mode == 1) {
            acc += p[(x + acc) % n16].x & 1u;
        } else if (mode == 2) {
            const size_t r = ((x + acc) % (n16 / 3)) * 3;
            const uint4 a = p[r], b = p[r + 1], c = p[r + 2];
            acc += (a.x ^ b.y ^ c.z) & 1u;
        } else if (mode == 3) {
            const size_t r = ((x + acc) % (n16 / 4)) * 4;
            const uint4 a = p[r], b = p[r + 1], c = p[r + 2], d = p[r + 3];
            acc += (a.x ^ b.y ^ c.z ^ d.w) & 1u;
        } else {
            const uint32_t e = reinterpret_cast<const uint2*>(p)[(x + acc) % (n16 * 2)].x;
            const size_t r = ((x * 31 + e) % (n16 / 3)) * 3;
            const uint4 a = p[r], b = p[r + 1], c = p[r + 2];
            acc += (a.x ^ b.y ^ c.z) & 1u;
        }
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
int main() {
    unsigned long long* out;
    (void)hipMalloc(&out, 8);
    const char* names[5] = {"8 B", "16 B", "3 x 16 B of one 48 B record", "4 x 16 B of one 64 B record", "8 B then 3 x 16 B elsewhere"};
    for (int lg = 21; lg <= 27; lg += 3) { // 16-byte units: 32 MB, 256 MB, 2 GB
        const size_t n = 1ull << lg;
        uint4* p;
        if (hipMalloc(&p, n * 16) != hipSuccess) { printf("alloc failed\n"); break; }
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, p, n);
        for (uint32_t mode = 0; mode < 5; mode++) {
            const uint32_t iters = 128; const dim3 grid(256 * 8), block(256);
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            hipLaunchKernelGGL(k_gather, grid, block, 0, 0, p, n, iters, mode, out);
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k_gather, grid, block, 0, 0, p, n, iters, mode, out);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf("%6.0f MB  %-30s %.1f G steps/s\n", n * 16 / 1048576.0, names[mode], (double)grid.x * block.x * iters / ms / 1e6);
        }
        (void)hipFree(p);
    }
    return 0;
}
