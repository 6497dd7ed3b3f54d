// How fast can a kernel with the shade stage's access pattern stream the path pool?  8 float4 arrays read and 8 written per
// slot, blocks own contiguous runs of slots (as k_shade), no arithmetic.  usage: poolbench [slots]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Pool { float4 *a[10]; };
template <int NR, int NW, bool MASKED>
__global__ __launch_bounds__(256) void k_stream(Pool p, unsigned int n) {
    const unsigned int per_block = (((n + gridDim.x - 1) / gridDim.x) + 255u) & ~255u;
    const unsigned int lo = blockIdx.x * per_block, hi = (lo + per_block < n) ? lo + per_block : n;
    for (unsigned int s = lo + threadIdx.x; s < hi; s += 256u) {
        float4 v[NR]; float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NR; ++k) { v[k] = p.a[k][s]; acc += v[k].x; }
        const bool w = !MASKED || ((__float_as_uint(acc) ^ s) & 3u) != 0u;       // MASKED: a quarter of the lanes skip the stores
        if (w) {
#pragma unroll
            for (int k = 0; k < NW; ++k) { float4 o = v[k % NR]; o.w = acc; p.a[9 - k][s] = o; }
        }
    }
}
template <int NR, int NW, bool M> int run(const char *name, Pool p, unsigned int n, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<NR, NW, M>), dim3(blocks), dim3(256), 0, 0, p, n);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<NR, NW, M>), dim3(blocks), dim3(256), 0, 0, p, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-28s blocks %5d  %.3f ms  %.2f TB/s (%d B read + %d B written per slot)\n", name, blocks, ms, (double) n * 16.0 * (NR + NW) / (ms * 1e-3) / 1e12, 16 * NR, 16 * NW);
    return 0;
}
int main(int argc, char **argv) {
    const unsigned int n = argc > 1 ? (unsigned int) atol(argv[1]) : (8u << 20);
    Pool p;
    for (int k = 0; k < 10; ++k) { CK(hipMalloc(&p.a[k], (size_t) n * 16)); CK(hipMemset(p.a[k], 0, (size_t) n * 16)); }
    for (int blocks : {1536, 3072, 8192, 32768}) {
        if (run<8, 8, false>("read 8, write 8", p, n, blocks)) return 1;
        if (run<8, 8, true>("read 8, write 8 (3/4 lanes)", p, n, blocks)) return 1;
        if (run<8, 0, false>("read 8", p, n, blocks)) return 1;
        if (run<4, 2, false>("read 4, write 2 (trace)", p, n, blocks)) return 1;
    }
    return 0;
}
