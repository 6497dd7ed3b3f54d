// Probe: what FETCH_SIZE (rocprofv3, gfx950) reports for scattered reads of known size from a buffer far larger than the
// Infinity Cache, and how fast they go.  Settles whether a scattered 64-byte node read moves 64 or 128 bytes.
//   k<0>  streaming, 16 B per lane, coalesced (calibration: the guide says FETCH_SIZE = bytes / 2)
//   k<1>  one random 128-B line per lane, the lane reads its FIRST 64 B   (the 64-byte two-box node)
//   k<2>  one random 128-B line per lane, the lane reads all 128 B        (a 128-byte four-box node)
//   k<3>  one random 64-B-aligned node per lane (either half of a line), 64 B
//   k<4>  one random 128-B line per lane, the lane reads 16 B
// Every line is visited once per launch (odd multiplier modulo a power of two is a bijection).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4 *buf, unsigned long long n_lines, float *out) {
    const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines) return;
    float acc = 0.f;
    if (MODE == 0) {
        // each thread streams 8 x 16 B with a grid-wide stride: coalesced 16 B per lane
        for (int k8 = 0; k8 < 8; ++k8) { float4 v = buf[i + (unsigned long long) k8 * n_lines]; acc += v.x + v.w; }
    } else if (MODE == 3) {
        const unsigned long long node = (i * 0x9E3779B1ull) & (2 * n_lines - 1);
        const float4 *q = buf + node * 4; float4 a = q[0], b = q[1], c = q[2], d = q[3]; acc = a.x + b.y + c.z + d.w;
    } else {
        const unsigned long long line = (i * 0x9E3779B1ull) & (n_lines - 1);
        const float4 *q = buf + line * 8;
        if (MODE == 1) { float4 a = q[0], b = q[1], c = q[2], d = q[3]; acc = a.x + b.y + c.z + d.w; }
        if (MODE == 2) { float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6], h = q[7]; acc = a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w; }
        if (MODE == 4) { float4 a = q[0]; acc = a.x; }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char **argv) {
    const unsigned long long n_lines = 1ull << (argc > 1 ? atoi(argv[1]) : 24);      // 2^24 lines x 128 B = 2 GiB
    float4 *d; float *o; CHK(hipMalloc(&d, n_lines * 128)); CHK(hipMalloc(&o, 64)); CHK(hipMemset(d, 0, n_lines * 128));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned blocks = (unsigned) ((n_lines + 255) / 256);
    const double req[5] = {128.0, 64.0, 128.0, 64.0, 16.0};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 5; ++mode) {
        hipEventRecord(e0);
        switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, n_lines, o); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, n_lines, o); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, n_lines, o); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, n_lines, o); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, n_lines, o); break;
        }
        hipEventRecord(e1); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rep %d mode %d: %8.3f ms, %.0f MB requested, %7.1f GB/s requested, %.1f G reads/s\n", rep, mode, ms, req[mode] * n_lines / 1e6, req[mode] * n_lines / ms / 1e6, n_lines / ms / 1e6);
    }
    return 0;
}
