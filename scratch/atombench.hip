// Micro-benchmark: throughput of wave-level atomicAdd on K distinct addresses (device memory) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// every wave performs `iters` atomics (lane 0 only, result broadcast and used), on address (wave % K) * stride
__global__ __launch_bounds__(256) void k(unsigned *ctr, int K, int stride, int iters, unsigned *out) {
    unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        unsigned v = 0;
        if (lane == 0) v = atomicAdd(ctr + (size_t) ((wave + i) % K) * stride, 64u);
        acc += __builtin_amdgcn_readfirstlane(v);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// same, but the result is not used (fire and forget)
__global__ __launch_bounds__(256) void k_noret(unsigned *ctr, int K, int stride, int iters) {
    unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (int i = 0; i < iters; ++i) if (lane == 0) atomicAdd(ctr + (size_t) ((wave + i) % K) * stride, 64u);
}
int main() {
    unsigned *c, *o; CHK(hipMalloc(&c, 1 << 22)); CHK(hipMemset(c, 0, 1 << 22)); CHK(hipMalloc(&o, 64));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 64; unsigned blocks = 256 * 6;
    for (int K : {1, 2, 8, 32, 128, 1024}) for (int stride : {1, 64}) for (int ret = 0; ret < 2; ++ret) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (ret) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, c, K, stride, iters, o); else hipLaunchKernelGGL(k_noret, dim3(blocks), dim3(256), 0, 0, c, K, stride, iters);
            hipEventRecord(e1); CHK(hipEventSynchronize(e1)); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double n = (double) blocks * 4 * iters;
        printf("K %5d stride %3d B  %s: %8.3f ms for %.0f atomics  -> %6.2f ns per atomic\n", K, stride * 4, ret ? "returning" : "no-return", best, n, best * 1e6 / n);
    }
    return 0;
}
