// Micro-benchmark: throughput of divergent per-lane node fetches (the access pattern of BVH traversal) on gfx950.
//   mode 0: each lane reads its own 64-byte node with 4 x dwordx4 (global)
//   mode 1: each lane reads 48 bytes (3 x dwordx4)
//   mode 2: each lane reads 16 bytes (1 x dwordx4)
//   mode 3: each lane reads 4 bytes (1 x dword)
//   mode 4: quad-cooperative: 4 adjacent lanes read the 4 x 16 bytes of ONE node (16 nodes per wave instruction)
//   mode 5: nodes in LDS (table <= 96 KB), each lane 4 x ds_read_b128
// The next index depends on the loaded data, as in a traversal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4 *tab, unsigned n_nodes, int iters, unsigned *out, unsigned lds_nodes) {
    extern __shared__ float4 lds[];
    if (MODE == 5) { for (unsigned i = threadIdx.x; i < lds_nodes * 4; i += blockDim.x) lds[i] = tab[i]; __syncthreads(); }
    unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned idx = (gid * 2654435761u) % n_nodes;
    if (MODE == 4) idx = ((gid >> 2) * 2654435761u) % n_nodes;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        unsigned next;
        if (MODE == 0) { const float4 *q = tab + 4 * (size_t) idx; float4 a = q[0], b = q[1], c = q[2], d = q[3]; acc += a.x + b.y + c.z; next = __float_as_uint(d.w); }
        else if (MODE == 1) { const float4 *q = tab + 4 * (size_t) idx; float4 a = q[0], b = q[1], d = q[3]; acc += a.x + b.y; next = __float_as_uint(d.w); }
        else if (MODE == 2) { const float4 *q = tab + 4 * (size_t) idx; float4 d = q[3]; acc += d.x; next = __float_as_uint(d.w); }
        else if (MODE == 3) { const float *q = (const float *) (tab + 4 * (size_t) idx); next = __float_as_uint(q[15]); }
        else if (MODE == 4) { const float4 *q = tab + 4 * (size_t) idx + (threadIdx.x & 3u); float4 a = *q; acc += a.x;
                              next = __float_as_uint(a.w); next = __shfl(next, (threadIdx.x & 63u) | 3u, 64); }
        else { const float4 *q = lds + 4 * (idx % lds_nodes); float4 a = q[0], b = q[1], c = q[2], d = q[3]; acc += a.x + b.y + c.z; next = __float_as_uint(d.w); }
        idx = (next + i * 40503u + gid) % n_nodes;
    }
    if (acc == 123.456f) out[0] = idx;
    out[1 + (gid & 1023)] = idx;
}

int main(int argc, char **argv) {
    unsigned n_nodes = argc > 1 ? atoi(argv[1]) : 3532;
    int iters = 2000;
    std::vector<float4> h(n_nodes * 4);
    srand(1);
    for (unsigned i = 0; i < n_nodes; ++i) { for (int k = 0; k < 4; ++k) h[i * 4 + k] = make_float4(1.f, 2.f, 3.f, 0.f); unsigned r = (unsigned) rand() * 7919u + i; h[i * 4 + 3].w = __builtin_bit_cast(float, r % n_nodes); }
    float4 *d; unsigned *o; CHK(hipMalloc(&d, h.size() * 16)); CHK(hipMalloc(&o, 4096 * 4)); CHK(hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc : {4, 6}) {
        unsigned blocks = 256 * wpc;
        for (int mode = 0; mode <= 5; ++mode) {
            unsigned lds_nodes = n_nodes < 1024 ? n_nodes : 1024; size_t shm = mode == 5 ? lds_nodes * 64 : 0;
            if (mode == 5 && wpc * shm > 160 * 1024) { shm = 0; }
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, iters, o, lds_nodes); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, iters, o, lds_nodes); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, iters, o, lds_nodes); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, iters, o, lds_nodes); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, iters, o, lds_nodes); break;
                    case 5: if (shm) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), shm, 0, d, n_nodes, iters, o, lds_nodes); break;
                }
                hipEventRecord(e1); CHK(hipEventSynchronize(e1)); CHK(hipGetLastError());
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            double lanes = (double) blocks * 256 * iters;
            double nodes = mode == 4 ? lanes / 4 : lanes;
            printf("nodes %u blocks/CU %d mode %d: %8.3f ms  %7.2f G node-fetches/s  (%.2f clk per node-fetch per CU at 2.4 GHz)\n", n_nodes, wpc, mode, best, nodes / best / 1e6, 2.4e9 * 256 * best * 1e-3 / nodes);
        }
    }
    return 0;
}
