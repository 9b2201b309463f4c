// Micro-benchmark: random 64-byte record gather per lane, (A) four per-lane dwordx4 loads vs
// (B) quad-cooperative global_load_lds_dwordx4 + conflict-free ds_read_b128.  Development aid.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ uint32_t next_idx(uint32_t& s, uint32_t n) { s = s * 1664525u + 1013904223u; return (s >> 8) % n; }

__global__ __launch_bounds__(64) void k_direct(const float4* __restrict__ src, uint32_t n, int iters, float* out) {
    uint32_t s = blockIdx.x * 64 + threadIdx.x + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        uint32_t i = next_idx(s, n);
        const float4* p = src + (size_t)i * 4;
        float4 a = p[0], b = p[1], c = p[2], e = p[3];
        acc += a.x + b.y + c.z + e.w;
        s ^= __float_as_uint(acc) & 1u; // dependent chain like a traversal
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(64) void k_coop(const float4* __restrict__ src, uint32_t n, int iters, float* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
    const uint32_t lane = threadIdx.x;
    uint32_t s = blockIdx.x * 64 + threadIdx.x + 12345u;
    float acc = 0.f;
    const uint32_t sw = (lane >> 2) & 3u;
    for (int it = 0; it < iters; it++) {
        uint32_t i = next_idx(s, n);
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            uint32_t owner = 16u * k4 + (lane >> 2);
            uint32_t oidx = __shfl(i, owner, 64);
            uint32_t chunk = (lane & 3u) ^ ((owner >> 2) & 3u);
            const char* g = reinterpret_cast<const char*>(src) + (size_t)oidx * 64 + chunk * 16;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(lds + k4 * 256), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float4* row = reinterpret_cast<const float4*>(lds) + lane * 4;
        float4 a = row[0 ^ sw], b = row[1 ^ sw], c = row[2 ^ sw], e = row[3 ^ sw];
        acc += a.x + b.y + c.z + e.w;
        s ^= __float_as_uint(acc) & 1u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // reads done before the next DMA overwrites the stage
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const uint32_t N = argc > 1 ? (uint32_t)atoi(argv[1]) : 160000; // default 10 MB of 64-byte records, like the sponza-like node array
    float4* h = (float4*)malloc((size_t)N * 64);
    for (uint32_t i = 0; i < N * 4; i++) h[i] = make_float4(i % 7, i % 11, i % 13, i % 17);
    float4* d; float *o1, *o2;
    const int blocks = 256 * 16, iters = 2000;
    hipMalloc(&d, (size_t)N * 64); hipMalloc(&o1, blocks * 64 * 4); hipMalloc(&o2, blocks * 64 * 4);
    hipMemcpy(d, h, (size_t)N * 64, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("N=%u (%.1f KB)\n", N, N * 64 / 1024.0);
    for (int rep = 0; rep < 2; rep++) {
        float ms1, ms2;
        hipEventRecord(e0); hipLaunchKernelGGL(k_direct, dim3(blocks), dim3(64), 0, 0, d, N, iters, o1); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k_coop, dim3(blocks), dim3(64), 0, 0, d, N, iters, o2); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms2, e0, e1);
        double recs = (double)blocks * 64 * iters;
        printf("direct %.3f ms (%.1f Grec/s, %.2f TB/s)   coop %.3f ms (%.1f Grec/s, %.2f TB/s)\n", ms1, recs / ms1 / 1e6, recs * 64 / ms1 / 1e9, ms2, recs / ms2 / 1e6, recs * 64 / ms2 / 1e9);
    }
    float* r1 = (float*)malloc(blocks * 64 * 4); float* r2 = (float*)malloc(blocks * 64 * 4);
    hipMemcpy(r1, o1, blocks * 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, blocks * 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < blocks * 64; i++) bad += r1[i] != r2[i];
    printf("mismatch=%d\n", bad);
    return bad != 0;
}
