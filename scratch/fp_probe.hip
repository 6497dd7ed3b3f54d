#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#pragma clang fp contract(off)
__global__ void k(const float *a, const float *b, float *s1, float *s2, float *d, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    s1[i] = __fsqrt_rn(a[i]); s2[i] = sqrtf(a[i]); d[i] = a[i] / b[i];
}
int main() {
    int n = 1 << 20; std::vector<float> a(n), b(n), s1(n), s2(n), d(n);
    std::mt19937 g(1); std::uniform_real_distribution<float> u(-12.f, 2.f);
    for (int i = 0; i < n; ++i) { a[i] = std::pow(10.f, u(g)); b[i] = std::pow(10.f, u(g)); }
    float *da, *db, *d1, *d2, *dd; hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&dd, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, d1, d2, dd, n);
    hipMemcpy(s1.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(s2.data(), d2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
    int e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; ++i) { volatile float r = std::sqrt(a[i]); volatile float q = a[i] / b[i]; e1 += s1[i] != r; e2 += s2[i] != r; e3 += d[i] != q; }
    printf("mismatch: __fsqrt_rn %d, sqrtf %d, div %d of %d\n", e1, e2, e3, n);
}
