// Device-side restatement of Nori's per-ray arithmetic for gfx950.
//
// Everything that feeds an integer-exact comparison (camera rays, slab test, Moller-Trumbore, sphere test,
// hit reconstruction) keeps the reference's operation ORDER and is compiled without FMA contraction, so that
// primary-hit primitive ids are bit-identical to the CPU renderer's (DESIGN.md "float parity").
// Reference: include/nori/{frame,bbox,ray}.h, src/common.cpp:274-314, src/mesh.cpp:84-168, src/sphere.cpp:42-141,
//            src/warp.cpp, src/{diffuse,microfacet,dielectric,mirror,arealight}.cpp, src/pathMisIntegrator.cpp,
//            ext/pcg32/pcg32.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nori_hip.h"

#pragma clang fp contract(off)

namespace nd {

#define ND_INF __builtin_huge_valf()
#define ND_PI 3.14159265358979323846f       /* include/nori/common.h:56 — a float literal in the reference */
#define ND_INV_PI 0.31830988618379067154f
#define ND_EPS 1e-4f

// Correctly rounded fp32 square root.  On ROCm 7.2 sqrtf() lowers to v_sqrt_f32 plus the one-ulp correction step
// (IEEE result), whereas __fsqrt_rn() lowers to the bare v_sqrt_f32 approximation: verified with scratch/fp_probe.hip.
__device__ __forceinline__ float nd_sqrt(float x) { return sqrtf(x); }

// ---- libm ---------------------------------------------------------------------------------------------------------------------
// The reference calls cos / sin / acos / atan2 unqualified on floats, which with this toolchain are glibc's float routines (oracle/
// orender.hpp, "unqualified <math.h> calls"), and their last bit decides comparisons further down a path (the z of a cosine-hemisphere
// sample against 0, a sphere's uv against a checkerboard edge).  The device's own sinf / cosf differ from glibc's in the last place for
// about one argument in five, so these restate glibc 2.35's algorithms instead:
//   sincosf   sysdeps/ieee754/flt-32/s_sincosf.{c,h} (the ARM optimized-routines code): reduction by pi/2 and two degree-8 / degree-7
//             polynomials, all in double, one rounding to float at the end
//   acosf     sysdeps/ieee754/flt-32/e_acosf.c, atan2f: e_atan2f.c + s_atanf.c (fdlibm's float versions): float arithmetic as written
// Checked bit for bit against the build container's libm on the host: every third float of [0, 2 pi] for sin and cos (362 M values),
// every third float of [-1, 1] for acos (710 M), 300 M random pairs for atan2 - zero differences (scratch/glibc_libm_check.c).
// glibc picks a variant of sincosf compiled with fused multiply-adds on CPUs that have them; the two variants differ before the final
// rounding by parts in 1e16, i.e. in the float result about once in 2^29 calls, which is what "bit-identical" means below.
struct NdSinCosTab { double c0, c1, c2, c3, c4; };
__device__ __forceinline__ void nd_sincosf(float y, float *sinp, float *cosp) {
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ffu;
    double x = (double) y; int n = 0;
    if (top < 0x3f4u) {                                                   // |y| < pi/4 (compared on the top 12 bits, as glibc does)
        if (top < 0x398u) { *sinp = y; *cosp = 1.0f; return; }            // |y| < 2^-12
    } else if (top < 0x42fu) {                                            // |y| < 120: one multiply-subtract reduction
        const double r = x * 0x1.45F306DC9C883p+23;                       // 2/pi * 2^24: the quadrant ends up in bits 24..31
        n = ((int32_t) r + 0x800000) >> 24;
        x = x - (double) n * 0x1.921FB54442D18p0;
        if (((n & 3) == 1) || ((n & 3) == 2)) x = -x;                     // sign[n & 3]
    } else { *sinp = sinf(y); *cosp = cosf(y); return; }                  // (never reached by the renderer: its arguments lie in [0, 2 pi])
    const double sg = (n & 2) ? -1.0 : 1.0;                               // the second table negates the cosine polynomial
    const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5, c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double x2 = x * x, x4 = x2 * x2, x3 = x2 * x;
    const double cc2 = c3 + x2 * c4, ss1 = s2 + x2 * s3, cc1 = c0 + x2 * c1;
    const double x5 = x3 * x2, x6 = x4 * x2;
    const double sv = x + x3 * s1, cv = cc1 + x4 * c2;
    const float sf = (float) (sv + x5 * ss1), cf = (float) (cv + x6 * cc2);
    if (n & 1) { *sinp = cf; *cosp = sf; } else { *sinp = sf; *cosp = cf; }
}
__device__ __forceinline__ float nd_acosf(float x) {
    const float pi = __uint_as_float(0x40490fdau), pio2_hi = __uint_as_float(0x3fc90fdau), pio2_lo = __uint_as_float(0x33a22168u),
        pS0 = __uint_as_float(0x3e2aaaabu), pS1 = __uint_as_float(0xbea6b090u), pS2 = __uint_as_float(0x3e4e0aa8u), pS3 = __uint_as_float(0xbd241146u),
        pS4 = __uint_as_float(0x3a4f7f04u), pS5 = __uint_as_float(0x3811ef08u), qS1 = __uint_as_float(0xc019d139u), qS2 = __uint_as_float(0x4001572du),
        qS3 = __uint_as_float(0xbf303361u), qS4 = __uint_as_float(0x3d9dc62eu);
    const int32_t hx = (int32_t) __float_as_uint(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {                                                // |x| < 0.5
        if (ix <= 0x23000000) return pio2_hi + pio2_lo;
        const float z = x * x, p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5))))), q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {                                                         // x < -0.5
        const float z = (1.0f + x) * 0.5f, p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5))))), q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float sq = sqrtf(z), r = p / q, w = r * sq - pio2_lo;
        return pi - 2.0f * (sq + w);
    }
    const float z = (1.0f - x) * 0.5f, sq = sqrtf(z), df = __uint_as_float(__float_as_uint(sq) & 0xfffff000u), c = (z - df * df) / (sq + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5))))), q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q, w = r * sq + c;
    return 2.0f * (df + w);
}
__device__ __forceinline__ float nd_atanf(float x) {
    const float hi[4] = {__uint_as_float(0x3eed6338u), __uint_as_float(0x3f490fdau), __uint_as_float(0x3f7b985eu), __uint_as_float(0x3fc90fdau)};
    const float lo[4] = {__uint_as_float(0x31ac3769u), __uint_as_float(0x33222168u), __uint_as_float(0x33140fb4u), __uint_as_float(0x33a22168u)};
    const float a0 = __uint_as_float(0x3eaaaaabu), a1 = __uint_as_float(0xbe4ccccdu), a2 = __uint_as_float(0x3e124925u), a3 = __uint_as_float(0xbde38e38u),
        a4 = __uint_as_float(0x3dba2e6eu), a5 = __uint_as_float(0xbd9d8795u), a6 = __uint_as_float(0x3d886b35u), a7 = __uint_as_float(0xbd6ef16bu),
        a8 = __uint_as_float(0x3d4bda59u), a9 = __uint_as_float(0xbd15a221u), a10 = __uint_as_float(0x3c8569d7u);
    const int32_t hx = (int32_t) __float_as_uint(x), ix = hx & 0x7fffffff; int id;
    if (ix >= 0x4c000000) { if (ix > 0x7f800000) return x + x; return hx > 0 ? hi[3] + lo[3] : -hi[3] - lo[3]; }
    if (ix < 0x3ee00000) { if (ix < 0x31000000) return x; id = -1; }
    else {
        x = fabsf(x);
        if (ix < 0x3f980000) { if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); } else { id = 1; x = (x - 1.0f) / (x + 1.0f); } }
        else { if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); } else { id = 3; x = -1.0f / x; } }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10))))), s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0) return x - x * (s1 + s2);
    const float hv = id == 0 ? hi[0] : id == 1 ? hi[1] : id == 2 ? hi[2] : hi[3], lv = id == 0 ? lo[0] : id == 1 ? lo[1] : id == 2 ? lo[2] : lo[3];
    const float r = hv - ((x * (s1 + s2) - lv) - x);
    return hx < 0 ? -r : r;
}
__device__ __forceinline__ float nd_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_2 = __uint_as_float(0x3fc90fdbu), pi = __uint_as_float(0x40490fdbu), pi_lo = __uint_as_float(0xb3bbbd2eu);
    const int32_t hx = (int32_t) __float_as_uint(x), hy = (int32_t) __float_as_uint(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return nd_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : m == 2 ? pi + tiny : -pi - tiny;
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000 || iy == 0x7f800000) return atan2f(y, x);        // (infinite coordinates: not a case a hit point produces)
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo; else if (hx < 0 && k < -60) z = 0.0f; else z = nd_atanf(fabsf(y / x));
    return m == 0 ? z : m == 1 ? __uint_as_float(__float_as_uint(z) ^ 0x80000000u) : m == 2 ? pi - (z - pi_lo) : (z - pi_lo) - pi;
}

// logf and expf: sysdeps/ieee754/flt-32/e_logf.c and e_expf.c (the ARM optimized-routines code): a table of 16 (1/c, log c) pairs or of
// 32 values of 2^(i/32), a short polynomial, everything in double, one rounding to float at the end.  The multiply-adds are fused where
// the compiler fused them in the variant glibc picks on CPUs with the instruction.  Checked against the build container's libm for EVERY
// float: logf on all 2^31 non-negative ones, expf on all 2^32 - zero differences (scratch/glibc_logf_expf_check.c; without the fused
// z - kd of expf, two floats differ).
static __device__ const double ND_LOGF_TAB[16][2] = {
    { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 }, { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 }, { 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2 },
    { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 }, { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 }, { 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3 },
    { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 }, { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 }, { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 },
    { 0x1p+0, 0x0p+0 }, { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 }, { 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4 },
    { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 }, { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3 }, { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 },
    { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 } };
static __device__ const unsigned long long ND_EXP2F_TAB[32] = {                      // bits of 2^(i/32) minus i << 47
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };
__device__ __forceinline__ float nd_logf(float x) {
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {                  // x < 2^-126, infinite or NaN
        if (ix * 2u == 0u) return -__uint_as_float(0x7f800000u);
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return (x - x) / (x - x);
        ix = __float_as_uint(x * 0x1p23f); ix -= 23u << 23;               // subnormal: normalised first
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int) ((tmp >> 19) & 15u), k = (int32_t) tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = ND_LOGF_TAB[i][0], logc = ND_LOGF_TAB[i][1], z = (double) __uint_as_float(iz);
    const double r = fma(z, invc, -1.0), y0 = fma((double) k, 0x1.62e42fefa39efp-1, logc), r2 = r * r;
    double y = fma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = fma(-0x1.00ea348b88334p-2, r2, y);
    y = fma(y, r2, y0 + r);
    return (float) y;
}
__device__ __forceinline__ float nd_expf(float x) {
    const double xd = (double) x;
    const uint32_t abstop = (__float_as_uint(x) >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {                                               // |x| >= 88 or NaN
        if (__float_as_uint(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    const double inv = 0x1.71547652b82fep+0 * 32, shift = 0x1.8p+52;
    const double z = inv * xd;
    double kd = z + shift; const unsigned long long ki = (unsigned long long) __double_as_longlong(kd); kd -= shift;
    const double r = fma(inv, xd, -kd);
    const double s = __longlong_as_double((long long) (ND_EXP2F_TAB[ki & 31ull] + (ki << 47)));
    const double c0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, c1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, c2 = 0x1.62e42ff0c52d6p-1 / 32;
    const double zz = fma(c0, r, c1), r2 = r * r;
    double y = fma(c2, r, 1.0);
    y = fma(zz, r2, y);
    return (float) (y * s);
}

struct F3 { float x, y, z; };
struct F2 { float x, y; };

__device__ __forceinline__ F3 f3(float a, float b, float c) { F3 r; r.x = a; r.y = b; r.z = c; return r; }
__device__ __forceinline__ F3 f3(float a) { return f3(a, a, a); }
__device__ __forceinline__ F3 f3(const float *p) { return f3(p[0], p[1], p[2]); }
__device__ __forceinline__ F2 f2(float a, float b) { F2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return f3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator/(F3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
// Eigen's unrolled 3-term reduction: x0 + (x1 + x2)
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ float norm(F3 a) { return nd_sqrt(dot(a, a)); }
__device__ __forceinline__ F3 normalized(F3 a) { float n = norm(a); return a / n; }
// reductions over dynamic-size Eigen expressions are sequential (src/mesh.cpp:69-73,154-157)
__device__ __forceinline__ F3 normalized_seq(F3 a) { float n = nd_sqrt((a.x * a.x + a.y * a.y) + a.z * a.z); return a / n; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ bool is_zero(F3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
__device__ __forceinline__ bool color_valid(F3 c) {      // Color3f::isValid, src/common.cpp:224-231
    return !(c.x < 0 || !isfinite(c.x) || c.y < 0 || !isfinite(c.y) || c.z < 0 || !isfinite(c.z));
}

// src/common.cpp:274-283, include/nori/frame.h:49-63
struct Frame {
    F3 s, t, n;
    __device__ __forceinline__ void set(F3 a) {
        n = a; F3 c;
        if (fabsf(a.x) > fabsf(a.y)) { float invLen = 1.0f / nd_sqrt(a.x * a.x + a.z * a.z); c = f3(a.z * invLen, 0.0f, -a.x * invLen); }
        else { float invLen = 1.0f / nd_sqrt(a.y * a.y + a.z * a.z); c = f3(0.0f, a.z * invLen, -a.y * invLen); }
        t = c; s = cross(c, a);
    }
    __device__ __forceinline__ F3 toLocal(F3 v) const { return f3(dot(v, s), dot(v, t), dot(v, n)); }
    __device__ __forceinline__ F3 toWorld(F3 v) const { return (s * v.x + t * v.y) + n * v.z; }
};

// ext/pcg32/pcg32.h:51-66,101-110
struct Pcg32 {
    uint64_t state, inc;
    __device__ __forceinline__ uint32_t next_uint() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t) (old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31));
    }
    __device__ __forceinline__ void seed(uint64_t initstate, uint64_t initseq) {
        state = 0U; inc = (initseq << 1u) | 1u; next_uint(); state += initstate; next_uint();
    }
    __device__ __forceinline__ float next1D() { return __uint_as_float((next_uint() >> 9) | 0x3f800000u) - 1.0f; }
    // Independent::next2D (src/independent.cpp:62-67): the reference's compiler evaluates the second constructor
    // argument first, so y receives the earlier draw (pinned in oracle/make_golden.py).
    __device__ __forceinline__ F2 next2D() { float a = next1D(); float b = next1D(); return f2(b, a); }
};

struct DevScene {
    const NoriHipNode *nodes;
    const float4 *prims;               // 3 x float4 per NoriHipPrim
    const NoriHipShadePrim *shade;
    const NoriHipShape *shapes;
    const NoriHipEmitter *emitters;
    const float *cdf;
    const NoriHipShadePrim *light_prims;
    const NoriHipTexture *textures;    // image tables (null when the scene has none)
    const uint8_t *tex_data;
    const float4 *nodes4;              // the same tree as 128-byte four-box nodes (8 x float4 per node, see Trav::box_step4), or null
    uint32_t root4;                    // child reference of the root in nodes4 numbering
    const uint4 *nodes4q;              // device-built trees beyond the caches: the four-box nodes with quantised boxes, 64 bytes each (4 x uint4 per
                                       // node, same numbering as nodes4; Trav::box_step4q), or null
    const uint4 *nodes8;               // device-built trees beyond the caches: the tree as 128-byte eight-box nodes with quantised boxes (Trav::box_step8;
                                       // root = node 0), or null.  When set, every traversal of a render and nori_hip_trace walk these.
    uint32_t root, n_emitters;
    int ordered;                       // bit 0 clear: the reference's child order (left first; decides ties, bit-exact hits).  set: nearer child first —
                                       // fewer box tests, same hit except where two primitives tie within rounding (NORI_HIP_ORDER=near)
                                       // bit 1 set (default): any-hit rays enter the nearer child first whatever bit 0 says (same answer for every order)
    int cull_zero;                     // emitter samples whose term is exactly zero get no shadow ray (shade_finish)
    int prim_lanes;                    // run-to-completion traversal: lanes that must wait at a leaf before a record step is issued
    int leaf2;                         // leaves hold at most two records, placed inside one 128-byte line (Trav::prim_step2 may be used)
    uint32_t n_top;                    // nodes [0, n_top) are the top of the tree in breadth-first order (LDS cache candidates)
    float root_min[3], root_max[3];
    NoriHipCamera cam;
    float inv_w, inv_h;
    int border, film_cols;
    NoriHipEnvMap env;                 // environment emitter (env.texture = 0: none)
    const float *env_cdf;              // normalised CDF over the env_h x env_w (u, v) cells of the environment map, env_h * env_w + 1 entries
    uint32_t env_w, env_h;
    const float *filt_table;           // null: the fork's single-pixel film add.  Else the 33-entry tabulated reconstruction filter
    float filt_radius, filt_lookup;    //       of ImageBlock::init (src/block.cpp:54-64) for the filter-weighted splat (:113-128)
};

struct Ray { F3 o, d; float mint, maxt; };
struct Hit { uint32_t pos; float t, u, v; };       // pos = record index in leaf order; 0xffffffff = miss

// include/nori/bbox.h:336-363, written without branches: every lane of a wave runs the same instructions and the
// per-axis rejections are accumulated in `ok` (nearT only grows and farT only shrinks, so an early "nearT > farT"
// of the reference also fails every later check; the values that reach `near_out` on a hit are the reference's).
// `near_out` is the entry distance; the caller re-checks it against a shrunken maxt when the node is popped, which
// is exactly what re-running the reference's test then would decide.
#define ND_SLAB_AXIS(O, D, R, MN, MX) { \
    float t1 = ((MN) - (O)) * (R), t2 = ((MX) - (O)) * (R); \
    bool sw = t1 > t2; float lo = sw ? t2 : t1, hi = sw ? t1 : t2; \
    float nn = (lo < nearT) ? nearT : lo, ff = (farT < hi) ? farT : hi; \
    bool z = ((D) == 0); \
    nearT = z ? nearT : nn; farT = z ? farT : ff; \
    ok = ok & (z ? !((O) < (MN) || (O) > (MX)) : (nearT <= farT)); }
__device__ __forceinline__ bool slab(const float *bmin, const float *bmax, F3 o, F3 d, F3 rcp, float mint, float maxt, float &near_out) {
    float nearT = -ND_INF, farT = ND_INF; bool ok = true;
    ND_SLAB_AXIS(o.x, d.x, rcp.x, bmin[0], bmax[0])
    ND_SLAB_AXIS(o.y, d.y, rcp.y, bmin[1], bmax[1])
    ND_SLAB_AXIS(o.z, d.z, rcp.z, bmin[2], bmax[2])
    near_out = nearT;
    return ok & (mint <= farT) & (nearT <= maxt);
}

// Where box steps fetch their 64-byte node from.  GlobalNodes: always HBM/L2 through the vector memory path.
// CachedNodes: the first `n` nodes (the top of the tree, breadth-first, see nori_hip_upload_scene) live in the
// workgroup's LDS, whose random-access rate and latency are several times better than the L1/TA path's; deeper
// nodes still come from memory.
typedef float nd_v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) nd_v4f lds_f4;
__device__ __forceinline__ float4 nd_f4(nd_v4f v) { return make_float4(v.x, v.y, v.z, v.w); }
struct GlobalNodes {
    const NoriHipNode *nodes;
    __device__ __forceinline__ void fetch(uint32_t cur, float4 &n0, float4 &n1, float4 &n2, float4 &n3) const {
        const float4 *q = reinterpret_cast<const float4 *>(nodes + cur);
        n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3];
    }
};
struct CachedNodes {
    const NoriHipNode *nodes; lds_f4 *cache; uint32_t n;
    __device__ __forceinline__ void fetch(uint32_t cur, float4 &n0, float4 &n1, float4 &n2, float4 &n3) const {
        if (cur < n) { const lds_f4 *q = cache + 4 * cur; n0 = nd_f4(q[0]); n1 = nd_f4(q[1]); n2 = nd_f4(q[2]); n3 = nd_f4(q[3]); }
        else { const float4 *q = reinterpret_cast<const float4 *>(nodes + cur); n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3]; }
    }
    // cooperative fill by the whole workgroup; the caller synchronises afterwards
    __device__ __forceinline__ void fill() const {
        const float4 *g = reinterpret_cast<const float4 *>(nodes);
        for (uint32_t i = threadIdx.x; i < 4 * n; i += blockDim.x) { float4 v = g[i]; nd_v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; cache[i] = w; }
    }
};

#ifndef ND_STACK
#define ND_STACK 64        /* the reference's traversal stack holds 64 entries (src/bvh.cpp:405) */
#endif
#define ND_STACK8 96       /* eight-box nodes leave up to seven entries per level; the build reports what a tree needs (lbvh::Result::stack8) */

// BVH::rayIntersect (src/bvh.cpp:404-463) over 64-byte two-box nodes, in the reference's visiting order
// (left subtree first, right pushed), so closest-hit ties resolve to the same primitive.
// Stack entries are (child reference, entry distance); the stack type decides where they live (LDS / scratch).
//
// Trav is the traversal as a resumable state machine whose steps contain no data-dependent loops: box_step()
// visits one 64-byte node, prim_step() tests ONE leaf record, pop_step() takes one entry off the stack.  A kernel
// runs the steps for the lanes that are in the matching state; the queue kernels choose per trip which step is
// worth running for the wave (see k_trace_pool).  `cur` encodes the state:
//   < ND_POP            inner node index            (box_step)
//   ND_POP              needs an entry of the stack (pop_step)
//   ND_DONE             traversal finished
//   NORI_HIP_LEAF_BIT|p next leaf record to test    (prim_step)
#define ND_Q8(W, K) ((float) (((W) >> (8 * (K))) & 0xffu))       /* byte K of word W as a float (one v_cvt_f32_ubyteK) */
#define ND_POP  0x7ffffffeu
#define ND_DONE 0x7fffffffu

template <bool ANY> struct Trav {
    F3 o, d, rcp; float mint, maxt; uint32_t cur; int sp; Hit hit; bool found;
    float cur_near;    // entry distance of the box of `cur` (read by the speculative trace stage only, dead code elsewhere)
    bool special;      // a direction component is zero, denormal or NaN: box tests take the general path

    __device__ __forceinline__ bool at_inner() const { return cur < ND_POP; }
    __device__ __forceinline__ bool at_leaf() const { return (cur & NORI_HIP_LEAF_BIT) != 0u; }
    __device__ __forceinline__ bool at_pop() const { return cur == ND_POP; }
    __device__ __forceinline__ bool done() const { return cur == ND_DONE; }

    // false: the ray ends without entering the tree
    __device__ __forceinline__ bool begin(const DevScene &sc, const Ray &r, uint32_t &n_boxes) {
        o = r.o; d = r.d; mint = r.mint; maxt = r.maxt; sp = 0; found = false; cur = ND_DONE;
        special = !(fabsf(d.x) >= 1.17549435e-38f && fabsf(d.y) >= 1.17549435e-38f && fabsf(d.z) >= 1.17549435e-38f);
        hit.pos = 0xffffffffu; hit.t = ND_INF; hit.u = hit.v = 0.f;
        if (mint == ND_EPS) {                       // adaptive epsilon, src/bvh.cpp:411-412
            float m = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z)));
            float e = mint * m; mint = (mint < e) ? e : mint;
        }
        if (maxt < mint) return false;
        rcp = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        float nr; n_boxes++;
        bool in;
        if (!special) {
            // all direction components are normal numbers: the same decision by min/max, as in box_compute_fast
            float x1 = (sc.root_min[0] - o.x) * rcp.x, x2 = (sc.root_max[0] - o.x) * rcp.x;
            float y1 = (sc.root_min[1] - o.y) * rcp.y, y2 = (sc.root_max[1] - o.y) * rcp.y;
            float z1 = (sc.root_min[2] - o.z) * rcp.z, z2 = (sc.root_max[2] - o.z) * rcp.z;
            nr = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x1, x2), __builtin_fminf(y1, y2)), __builtin_fminf(z1, z2));
            float fr = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x1, x2), __builtin_fmaxf(y1, y2)), __builtin_fmaxf(z1, z2));
            in = (nr <= fr) & (mint <= fr) & (nr <= maxt);
        } else in = slab(sc.root_min, sc.root_max, o, d, rcp, mint, maxt, nr);
        if (!in) return false;
        cur = sc.root; cur_near = nr;        // (a scene of one leaf has no inner node: the root reference is that leaf)
        return true;
    }
    // an entry is skipped when its box no longer passes with the current maxt (then the lane stays in ND_POP)
    template <typename StackT> __device__ __forceinline__ void pop_step(StackT &stk) {
        if (sp == 0) { cur = ND_DONE; return; }
        uint2 e = stk.pop(--sp);
        const float en = __uint_as_float(e.y); const bool ok = en <= maxt;
        cur = ok ? e.x : ND_POP; cur_near = ok ? en : cur_near;      // (conditional on purpose: as a plain copy of e the pair {cur, cur_near} ends up in scratch memory)
    }
    // The same node visit for rays whose direction components are all normal numbers (`special` false).  Then every
    // reciprocal is finite, no product can be NaN, and the reference's compare-and-select chains (include/nori/
    // bbox.h:345-358) return exactly min/max of their operands; its per-axis "nearT > farT" exits are implied by the
    // final comparison because nearT only grows and farT only shrinks.  Same decisions and entry distances (up to
    // the sign of a zero, which no comparison sees), a third of the instructions.
    template <typename StackT, typename NodesT> __device__ __forceinline__ void box_step_fast(const NodesT &nodes, StackT &stk, uint32_t &n_boxes, const bool ordered = false) {
        float4 n0, n1, n2, n3; nodes.fetch(cur, n0, n1, n2, n3);
        box_compute_fast(n0, n1, n2, n3, stk, n_boxes, ordered);
    }
    // the arithmetic of box_step_fast on a node that has been fetched already (kernels that issue the node and the
    // record fetches of a trip together, before either is used)
    template <typename StackT> __device__ __forceinline__ void box_compute_fast(const float4 n0, const float4 n1, const float4 n2, const float4 n3, StackT &stk, uint32_t &n_boxes, const bool ordered) {
        uint32_t left = __float_as_uint(n3.x), right = __float_as_uint(n3.y);
        n_boxes += 2;
        float ax1 = (n0.x - o.x) * rcp.x, ax2 = (n0.w - o.x) * rcp.x;
        float ay1 = (n0.y - o.y) * rcp.y, ay2 = (n1.x - o.y) * rcp.y;
        float az1 = (n0.z - o.z) * rcp.z, az2 = (n1.y - o.z) * rcp.z;
        float bx1 = (n1.z - o.x) * rcp.x, bx2 = (n2.y - o.x) * rcp.x;
        float by1 = (n1.w - o.y) * rcp.y, by2 = (n2.z - o.y) * rcp.y;
        float bz1 = (n2.x - o.z) * rcp.z, bz2 = (n2.w - o.z) * rcp.z;
        float nl = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax1, ax2), __builtin_fminf(ay1, ay2)), __builtin_fminf(az1, az2));
        float fl = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax1, ax2), __builtin_fmaxf(ay1, ay2)), __builtin_fmaxf(az1, az2));
        float nrr = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(bx1, bx2), __builtin_fminf(by1, by2)), __builtin_fminf(bz1, bz2));
        float fr = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(bx1, bx2), __builtin_fmaxf(by1, by2)), __builtin_fmaxf(bz1, bz2));
        bool hl = (nl <= fl) & (mint <= fl) & (nl <= maxt);
        bool hr = (nrr <= fr) & (mint <= fr) & (nrr <= maxt) & (right != NORI_HIP_NO_CHILD);
        const bool sw = ordered & hl & hr & (nrr < nl);                  // ordered mode: the right child is entered first
        if (hl & hr) stk.push(sp++, make_uint2(sw ? left : right, __float_as_uint(sw ? nl : nrr)));
        cur = hl ? (sw ? right : left) : (hr ? right : ND_POP); cur_near = (hl & !sw) ? nl : nrr;
    }
    template <typename StackT, typename NodesT> __device__ __forceinline__ void box_step(const NodesT &nodes, StackT &stk, uint32_t &n_boxes, const bool ordered = false) {
        float4 n0, n1, n2, n3; nodes.fetch(cur, n0, n1, n2, n3);
        box_compute(n0, n1, n2, n3, stk, n_boxes, ordered);
    }
    template <typename StackT> __device__ __forceinline__ void box_compute(const float4 n0, const float4 n1, const float4 n2, const float4 n3, StackT &stk, uint32_t &n_boxes, const bool ordered) {
        float lmin[3] = {n0.x, n0.y, n0.z}, lmax[3] = {n0.w, n1.x, n1.y};
        float rmin[3] = {n1.z, n1.w, n2.x}, rmax[3] = {n2.y, n2.z, n2.w};
        uint32_t left = __float_as_uint(n3.x), right = __float_as_uint(n3.y);
        float nl, nrr;
        n_boxes += 2;
        bool hl = slab(lmin, lmax, o, d, rcp, mint, maxt, nl);
        bool hr = slab(rmin, rmax, o, d, rcp, mint, maxt, nrr) & (right != NORI_HIP_NO_CHILD);
        const bool sw = ordered & hl & hr & (nrr < nl);
        if (hl & hr) stk.push(sp++, make_uint2(sw ? left : right, __float_as_uint(sw ? nl : nrr)));
        cur = hl ? (sw ? right : left) : (hr ? right : ND_POP); cur_near = (hl & !sw) ? nl : nrr;
    }
    // ---- four-box nodes -------------------------------------------------------------------------------------------
    // A 128-byte node holds the boxes and references of up to four children: an inner node of the binary tree with its
    // inner children opened, i.e. its grandchildren in depth-first order (left-left, left-right, right-left, right-right;
    // a child that is a leaf stays as it is), valid slots first.  Layout, 8 x float4:
    //   0 min.x of the four children | 1 min.y | 2 min.z | 3 max.x | 4 max.y | 5 max.z | 6 child references | 7 unused
    // One aligned 128-byte fetch - a whole HBM line, which a 64-byte node fetch moves as well (scratch/fetchprobe.hip) -
    // decides four descents, and the chain of dependent fetches per ray is about half as long.
    // Exactness: the boxes are bit-copies of the binary tree's.  A child's box lies inside its parent's and the slab test is
    // monotone in the box (subtraction, multiplication by the reciprocal, min and max all are), so a box that passes implies
    // that every box above it passes with the larger maxt it would have been tested with: skipping the opened node's own test
    // changes no decision below it.  Visiting order: `ordered` = by entry distance (any-hit rays, device-built trees); otherwise
    // slot order, which is the binary tree's depth-first order (left first, src/bvh.cpp:425-431), so ties in t fall as before.
    // The children that pass are pushed far to near (or last to first), the first one becomes `cur`.
#define ND_CE4(A, B) { const bool sw_ = key[B] < key[A]; const float tk_ = sw_ ? key[B] : key[A]; key[B] = sw_ ? key[A] : key[B]; key[A] = tk_; \
                       const float tn_ = sw_ ? nr[B] : nr[A]; nr[B] = sw_ ? nr[A] : nr[B]; nr[A] = tn_; \
                       const uint32_t tr_ = sw_ ? ref[B] : ref[A]; ref[B] = sw_ ? ref[A] : ref[B]; ref[A] = tr_; }
    template <typename StackT> __device__ __forceinline__ void box4_finish(float nr[4], const bool hit[4], uint32_t ref[4], StackT &stk, const bool ordered, const bool far = false) {
        float key[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) key[k] = hit[k] ? (ordered ? (far ? -nr[k] : nr[k]) : (float) k) : ND_INF;
        ND_CE4(0, 1) ND_CE4(2, 3) ND_CE4(0, 2) ND_CE4(1, 3) ND_CE4(1, 2)
        const int n = (int) hit[0] + (int) hit[1] + (int) hit[2] + (int) hit[3];
        if (n > 3) stk.push(sp++, make_uint2(ref[3], __float_as_uint(nr[3])));
        if (n > 2) stk.push(sp++, make_uint2(ref[2], __float_as_uint(nr[2])));
        if (n > 1) stk.push(sp++, make_uint2(ref[1], __float_as_uint(nr[1])));
        cur = n ? ref[0] : ND_POP; cur_near = n ? nr[0] : cur_near;
    }
    // the same when no lane of the wave wants its children by distance: slot order is visiting order, nothing to sort
    template <typename StackT> __device__ __forceinline__ void box4_finish_inorder(const float nr[4], const bool hit[4], const uint32_t ref[4], StackT &stk) {
        if (hit[3] & (hit[0] | hit[1] | hit[2])) stk.push(sp++, make_uint2(ref[3], __float_as_uint(nr[3])));
        if (hit[2] & (hit[0] | hit[1])) stk.push(sp++, make_uint2(ref[2], __float_as_uint(nr[2])));
        if (hit[1] & hit[0]) stk.push(sp++, make_uint2(ref[1], __float_as_uint(nr[1])));
        const bool any_hit = hit[0] | hit[1] | hit[2] | hit[3];
        cur = hit[0] ? ref[0] : hit[1] ? ref[1] : hit[2] ? ref[2] : hit[3] ? ref[3] : ND_POP;
        cur_near = hit[0] ? nr[0] : hit[1] ? nr[1] : hit[2] ? nr[2] : any_hit ? nr[3] : cur_near;
    }
    // direction components all normal numbers: min/max form, as box_compute_fast
    // (both forms return the number of boxes the node holds, for the statistics)
    // `sort` (wave-uniform): false when no lane of the wave is `ordered`
    template <typename StackT> __device__ __forceinline__ uint32_t box_step4_fast(const float4 *nodes4, StackT &stk, const bool ordered, const bool sort = true, const bool far = false) {
        const float4 *q = nodes4 + 8 * (size_t) cur;
        const float4 mnx = q[0], mny = q[1], mnz = q[2], mxx = q[3], mxy = q[4], mxz = q[5]; const float4 cr = q[6];
        const float lx[4] = {mnx.x, mnx.y, mnx.z, mnx.w}, ly[4] = {mny.x, mny.y, mny.z, mny.w}, lz[4] = {mnz.x, mnz.y, mnz.z, mnz.w};
        const float hx[4] = {mxx.x, mxx.y, mxx.z, mxx.w}, hy[4] = {mxy.x, mxy.y, mxy.z, mxy.w}, hz[4] = {mxz.x, mxz.y, mxz.z, mxz.w};
        uint32_t ref[4] = {__float_as_uint(cr.x), __float_as_uint(cr.y), __float_as_uint(cr.z), __float_as_uint(cr.w)};
        float nr[4]; bool hit[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x1 = (lx[k] - o.x) * rcp.x, x2 = (hx[k] - o.x) * rcp.x;
            const float y1 = (ly[k] - o.y) * rcp.y, y2 = (hy[k] - o.y) * rcp.y;
            const float z1 = (lz[k] - o.z) * rcp.z, z2 = (hz[k] - o.z) * rcp.z;
            const float nn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x1, x2), __builtin_fminf(y1, y2)), __builtin_fminf(z1, z2));
            const float ff = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x1, x2), __builtin_fmaxf(y1, y2)), __builtin_fmaxf(z1, z2));
            nr[k] = nn; hit[k] = (nn <= ff) & (mint <= ff) & (nn <= maxt) & (ref[k] != NORI_HIP_NO_CHILD);
        }
        const uint32_t n_valid = (uint32_t) (ref[0] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[1] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[2] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[3] != NORI_HIP_NO_CHILD);
        if (sort) box4_finish(nr, hit, ref, stk, ordered, far); else box4_finish_inorder(nr, hit, ref, stk);
        return n_valid;
    }
    // any ray: the reference's test as written (zero, denormal or NaN direction components)
    template <typename StackT> __device__ __forceinline__ uint32_t box_step4(const float4 *nodes4, StackT &stk, const bool ordered, const bool far = false) {
        const float4 *q = nodes4 + 8 * (size_t) cur;
        const float4 mnx = q[0], mny = q[1], mnz = q[2], mxx = q[3], mxy = q[4], mxz = q[5]; const float4 cr = q[6];
        const float lx[4] = {mnx.x, mnx.y, mnx.z, mnx.w}, ly[4] = {mny.x, mny.y, mny.z, mny.w}, lz[4] = {mnz.x, mnz.y, mnz.z, mnz.w};
        const float hx[4] = {mxx.x, mxx.y, mxx.z, mxx.w}, hy[4] = {mxy.x, mxy.y, mxy.z, mxy.w}, hz[4] = {mxz.x, mxz.y, mxz.z, mxz.w};
        uint32_t ref[4] = {__float_as_uint(cr.x), __float_as_uint(cr.y), __float_as_uint(cr.z), __float_as_uint(cr.w)};
        float nr[4]; bool hit[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float bmin[3] = {lx[k], ly[k], lz[k]}, bmax[3] = {hx[k], hy[k], hz[k]};
            float nn; const bool h = slab(bmin, bmax, o, d, rcp, mint, maxt, nn);
            nr[k] = nn; hit[k] = h & (ref[k] != NORI_HIP_NO_CHILD);
        }
        const uint32_t n_valid = (uint32_t) (ref[0] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[1] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[2] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[3] != NORI_HIP_NO_CHILD);
        box4_finish(nr, hit, ref, stk, ordered, far);
        return n_valid;
    }
    // ---- four-box nodes with quantised boxes, 64 bytes (layout: lbvh::k_quantize4): the arithmetic of the eight-box step below on
    // four children - slab distances t = q * (unit * rcp) + (p - o) * rcp, one byte-to-float conversion and one multiply-add per
    // plane, near and far plane picked by the sign of the direction - and the visiting order of the four-box step above.
    template <typename StackT> __device__ __forceinline__ uint32_t box_step4q_fast(const uint4 *nodes4q, StackT &stk, const bool ordered, const bool sort = true, const bool far = false) {
        const uint4 *q = nodes4q + 4 * (size_t) cur;
        const uint4 h = q[0], a = q[1], b = q[2], r = q[3];
        const float ax = __uint_as_float(h.w) * rcp.x, ay = __uint_as_float(b.z) * rcp.y, az = __uint_as_float(b.w) * rcp.z;
        const float bx = (__uint_as_float(h.x) - o.x) * rcp.x, by = (__uint_as_float(h.y) - o.y) * rcp.y, bz = (__uint_as_float(h.z) - o.z) * rcp.z;
        const bool sx = rcp.x < 0.f, sy = rcp.y < 0.f, sz = rcp.z < 0.f;
        const uint32_t nx = sx ? a.w : a.x, fx = sx ? a.x : a.w, ny = sy ? b.x : a.y, fy = sy ? a.y : b.x, nz = sz ? b.y : a.z, fz = sz ? a.z : b.y;
        uint32_t ref[4] = {r.x, r.y, r.z, r.w};
        float nr[4]; bool hit[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float tnx = __builtin_fmaf(ND_Q8(nx, k), ax, bx), tfx = __builtin_fmaf(ND_Q8(fx, k), ax, bx);
            const float tny = __builtin_fmaf(ND_Q8(ny, k), ay, by), tfy = __builtin_fmaf(ND_Q8(fy, k), ay, by);
            const float tnz = __builtin_fmaf(ND_Q8(nz, k), az, bz), tfz = __builtin_fmaf(ND_Q8(fz, k), az, bz);
            const float nn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, mint));
            const float ff = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, maxt));
            nr[k] = nn; hit[k] = (nn <= ff) & (ref[k] != NORI_HIP_NO_CHILD);
        }
        const uint32_t n_valid = (uint32_t) (ref[0] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[1] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[2] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[3] != NORI_HIP_NO_CHILD);
        if (sort) box4_finish(nr, hit, ref, stk, ordered, far); else box4_finish_inorder(nr, hit, ref, stk);
        return n_valid;
    }
    // any ray (a zero, denormal or NaN direction component): the boxes decoded, then the reference's test as written
    template <typename StackT> __device__ __forceinline__ uint32_t box_step4q(const uint4 *nodes4q, StackT &stk, const bool ordered, const bool far = false) {
        const uint4 *q = nodes4q + 4 * (size_t) cur;
        const uint4 h = q[0], a = q[1], b = q[2], r = q[3];
        const float u[3] = {__uint_as_float(h.w), __uint_as_float(b.z), __uint_as_float(b.w)};
        const float p[3] = {__uint_as_float(h.x), __uint_as_float(h.y), __uint_as_float(h.z)};
        const uint32_t lo[3] = {a.x, a.y, a.z}, hi[3] = {a.w, b.x, b.y};
        uint32_t ref[4] = {r.x, r.y, r.z, r.w};
        float nr[4]; bool hit[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float bmin[3], bmax[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { bmin[c] = p[c] + ND_Q8(lo[c], k) * u[c]; bmax[c] = p[c] + ND_Q8(hi[c], k) * u[c]; }
            float nn; const bool hh = slab(bmin, bmax, o, d, rcp, mint, maxt, nn);
            nr[k] = nn < mint ? mint : nn; hit[k] = hh & (ref[k] != NORI_HIP_NO_CHILD);
        }
        box4_finish(nr, hit, ref, stk, ordered, far);
        return (uint32_t) (ref[0] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[1] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[2] != NORI_HIP_NO_CHILD) + (uint32_t) (ref[3] != NORI_HIP_NO_CHILD);
    }
    // ---- eight-box nodes with quantised boxes (device-built trees that do not fit the caches; layout: lbvh::k_wide_emit8) -----------
    // A child's box is p + q * unit per axis with 8-bit q, low corners rounded down and high corners up, so it contains the box of
    // the binary tree: a ray reaches every record it reached before (and a few boxes more), closest hits are the same primitive
    // except on ties in t, any-hit answers the same.  The slab distances are t = q * (unit * rcp) + (p - o) * rcp: two products per
    // axis and node, then one multiply-add per plane; which plane is the near one follows from the sign of the direction.  The
    // children that pass are visited nearest first: each finds its rank among the others by comparisons (no sorting network: the
    // rank is the stack position, the payload never moves), the nearest becomes `cur`, the rest go on the stack far to near.
    // Entry distances are clamped to mint from below (the pop test `entry <= maxt` is unchanged by that).
    // Returns the number of children in use (statistics).
    template <typename StackT> __device__ __forceinline__ void box8_finish(const float nr[8], const bool hit[8], const uint32_t ref[8], StackT &stk) {
        float key[8]; int rank[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { key[k] = hit[k] ? nr[k] : ND_INF; rank[k] = 0; }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = j + 1; k < 8; ++k) { const bool c = key[j] <= key[k]; rank[k] += c ? 1 : 0; rank[j] += c ? 0 : 1; }
        int n = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) n += hit[k] ? 1 : 0;
        uint32_t c0 = ND_POP; float n0 = cur_near;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (hit[k] && rank[k] > 0) stk.push(sp + (n - 1 - rank[k]), make_uint2(ref[k], __float_as_uint(nr[k])));
            const bool first = hit[k] && rank[k] == 0;
            c0 = first ? ref[k] : c0; n0 = first ? nr[k] : n0;
        }
        sp += n > 1 ? n - 1 : 0;
        cur = c0; cur_near = n0;
    }
    template <typename StackT> __device__ __forceinline__ uint32_t box_step8_fast(const uint4 *nodes8, StackT &stk) {
        const uint4 *q = nodes8 + 8 * (size_t) cur;
        const uint4 h = q[0], qa = q[1], qb = q[2], qc = q[3], r0 = q[4], r1 = q[5];
        const float ux = __uint_as_float((h.w & 0xffu) << 23), uy = __uint_as_float(((h.w >> 8) & 0xffu) << 23), uz = __uint_as_float(((h.w >> 16) & 0xffu) << 23);
        const float ax = ux * rcp.x, ay = uy * rcp.y, az = uz * rcp.z;
        const float bx = (__uint_as_float(h.x) - o.x) * rcp.x, by = (__uint_as_float(h.y) - o.y) * rcp.y, bz = (__uint_as_float(h.z) - o.z) * rcp.z;
        // near and far planes per axis: the low corner is the near one when the direction is positive
        const bool sx = rcp.x < 0.f, sy = rcp.y < 0.f, sz = rcp.z < 0.f;
        const uint32_t nx[2] = {sx ? qb.z : qa.x, sx ? qb.w : qa.y}, fx[2] = {sx ? qa.x : qb.z, sx ? qa.y : qb.w};
        const uint32_t ny[2] = {sy ? qc.x : qa.z, sy ? qc.y : qa.w}, fy[2] = {sy ? qa.z : qc.x, sy ? qa.w : qc.y};
        const uint32_t nz[2] = {sz ? qc.z : qb.x, sz ? qc.w : qb.y}, fz[2] = {sz ? qb.x : qc.z, sz ? qb.y : qc.w};
        const uint32_t ref[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        float nr[8]; bool hit[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float tnx = __builtin_fmaf(ND_Q8(nx[k >> 2], k & 3), ax, bx), tfx = __builtin_fmaf(ND_Q8(fx[k >> 2], k & 3), ax, bx);
            const float tny = __builtin_fmaf(ND_Q8(ny[k >> 2], k & 3), ay, by), tfy = __builtin_fmaf(ND_Q8(fy[k >> 2], k & 3), ay, by);
            const float tnz = __builtin_fmaf(ND_Q8(nz[k >> 2], k & 3), az, bz), tfz = __builtin_fmaf(ND_Q8(fz[k >> 2], k & 3), az, bz);
            const float nn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, mint));
            const float ff = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, maxt));
            nr[k] = nn; hit[k] = (nn <= ff) & (ref[k] != NORI_HIP_NO_CHILD);
        }
        box8_finish(nr, hit, ref, stk);
        return h.w >> 24;
    }
    // any ray (a zero, denormal or NaN direction component): the boxes decoded, then the reference's test as written
    template <typename StackT> __device__ __forceinline__ uint32_t box_step8(const uint4 *nodes8, StackT &stk) {
        const uint4 *q = nodes8 + 8 * (size_t) cur;
        const uint4 h = q[0], qa = q[1], qb = q[2], qc = q[3], r0 = q[4], r1 = q[5];
        const float u[3] = {__uint_as_float((h.w & 0xffu) << 23), __uint_as_float(((h.w >> 8) & 0xffu) << 23), __uint_as_float(((h.w >> 16) & 0xffu) << 23)};
        const float p[3] = {__uint_as_float(h.x), __uint_as_float(h.y), __uint_as_float(h.z)};
        const uint32_t lo[3][2] = {{qa.x, qa.y}, {qa.z, qa.w}, {qb.x, qb.y}}, hi[3][2] = {{qb.z, qb.w}, {qc.x, qc.y}, {qc.z, qc.w}};
        const uint32_t ref[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        float nr[8]; bool hit[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float bmin[3], bmax[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { bmin[a] = p[a] + ND_Q8(lo[a][k >> 2], k & 3) * u[a]; bmax[a] = p[a] + ND_Q8(hi[a][k >> 2], k & 3) * u[a]; }
            float nn; const bool hh = slab(bmin, bmax, o, d, rcp, mint, maxt, nn);
            nr[k] = nn < mint ? mint : nn; hit[k] = hh & (ref[k] != NORI_HIP_NO_CHILD);
        }
        box8_finish(nr, hit, ref, stk);
        return h.w >> 24;
    }
    // one record of the current leaf; afterwards the lane is at the next record, in ND_POP (leaf finished) or done
    __device__ __forceinline__ void prim_step(const DevScene &sc, uint32_t &n_prims) { prim_step(sc, n_prims, ANY); }
    // `any`: stop at the first hit (a per-lane flag where one kernel traces shadow and extend rays together)
    __device__ __forceinline__ void prim_step(const DevScene &sc, uint32_t &n_prims, const bool any) {
        float4 A, B, C; prim_fetch(sc, A, B, C);
        prim_compute(A, B, C, n_prims, any);
    }
    // the same for a query restricted to the records of shape `mask` (BVH::rayIntersect's mask argument, src/bvh.cpp:438;
    // 0xffffffff: no restriction): a record of another shape is stepped over.  Returns whether the record was tested.
    __device__ __forceinline__ bool prim_step_masked(const DevScene &sc, uint32_t &n_prims, const bool any, const uint32_t mask) {
        float4 A, B, C; prim_fetch(sc, A, B, C);
        const uint32_t fl = __float_as_uint(C.z);
        if (mask == 0xffffffffu || (fl & 0xffffffu) == mask) { prim_compute(A, B, C, n_prims, any); return true; }
        cur = (fl & 0x80000000u) ? ND_POP : (NORI_HIP_LEAF_BIT | ((cur & 0x7fffffffu) + 1u));
        return false;
    }
    // Both records of a leaf in ONE step, for trees whose leaves hold at most two records placed inside one 128-byte line (device-built
    // trees of a million records or more: lbvh `align_leaves`): the six 16-byte loads of the line are issued together and the lane
    // waits for memory once per leaf instead of once per record.  The second record is tested after the first with the maxt the first
    // left behind, exactly as two single steps would, so hits and the count of record tests are unchanged.  Returns whether the second
    // record was tested (statistics).  (The position after the last record of the array exists: the placement pads to eight.)
    __device__ __forceinline__ bool prim_step2(const DevScene &sc, uint32_t &n_prims, const bool any) {
        const float4 *r = sc.prims + 3 * (size_t) (cur & 0x7fffffffu);
        const float4 A0 = r[0], B0 = r[1], C0 = r[2], A1 = r[3], B1 = r[4], C1 = r[5];
        prim_compute(A0, B0, C0, n_prims, any);
        const bool second = at_leaf();
        if (second) prim_compute(A1, B1, C1, n_prims, any);
        return second;
    }
    __device__ __forceinline__ void prim_fetch(const DevScene &sc, float4 &A, float4 &B, float4 &C) const {
        const float4 *r = sc.prims + 3 * (size_t) (cur & 0x7fffffffu);
        A = r[0]; B = r[1]; C = r[2];
    }
    __device__ __forceinline__ void prim_compute(const float4 A, const float4 B, const float4 C, uint32_t &n_prims, const bool any) {
        const uint32_t pos = cur & 0x7fffffffu;
        const uint32_t flags = __float_as_uint(C.z);
        n_prims++;
        float t, u, v; bool h;
        {
            // Mesh::rayIntersect, src/mesh.cpp:84-121: same arithmetic, the early returns folded into one predicate.
            // Evaluated for every record (a sphere record just yields a value that is discarded below), so that the
            // three 16-byte loads of the record are issued together instead of two of them behind a branch on the first.
            F3 p0 = f3(A.x, A.y, A.z), edge1 = f3(A.w, B.x, B.y), edge2 = f3(B.z, B.w, C.x);
            F3 pvec = cross(d, edge2);
            float det = dot(edge1, pvec);
            bool ok = !(det > -1e-8f && det < 1e-8f);
            float inv_det = 1.0f / det;
            F3 tvec = o - p0;
            u = dot(tvec, pvec) * inv_det;
            ok = ok & !(u < 0.0f || u > 1.0f);
            F3 qvec = cross(tvec, edge1);
            v = dot(d, qvec) * inv_det;
            ok = ok & !(v < 0.0f || u + v > 1.0f);
            t = dot(edge2, qvec) * inv_det;
            h = ok & (t >= mint) & (t <= maxt);
        }
        if (flags & 0x40000000u) {
            // Sphere::rayIntersect, src/sphere.cpp:42-113
            t = 0.f; u = 0.f; v = 0.f; h = false;
            F3 c = f3(A.x, A.y, A.z); float rad = A.w;
            float Bq = 2 * dot(o - c, d);
            float Cq = ((-rad * rad + dot(c, c)) + dot(o, o)) - 2 * dot(c, o);
            float discr = Bq * Bq - 4 * Cq;
            if (discr > 0) {
                float sq = nd_sqrt(discr);
                float t1 = (-Bq + sq) * 0.5f, t2 = (-Bq - sq) * 0.5f;
                bool v1 = t1 >= mint && t1 <= maxt, v2 = t2 >= mint && t2 <= maxt;
                if (v1 && v2) { t = (t2 < t1) ? t2 : t1; h = true; }
                else if (v1) { t = t1; h = true; }
                else if (v2) { t = t2; h = true; }
            } else if (discr == 0) {
                float tt = -Bq * 0.5f;
                if (tt >= mint && tt < maxt) { t = tt; h = true; }
            }
        }
        cur = (flags & 0x80000000u) ? ND_POP : (NORI_HIP_LEAF_BIT | (pos + 1u));
        if (h) {
            found = true;
            if (any) { cur = ND_DONE; return; }
            maxt = t; hit.pos = pos; hit.t = t; hit.u = u; hit.v = v;
        }
    }
};

// run one ray to completion
// Run one ray per lane to completion.  Every trip serves the lanes at an inner node, then the lanes at a leaf
// record once at least sc.prim_lanes of them have gathered (or no lane is at an inner node), then the pops.  A lane
// that waits at a leaf costs nothing; per lane the sequence of steps is the same as in a private loop.
template <bool ANY, typename StackT, typename NodesT>
__device__ __forceinline__ bool trace(const DevScene &sc, const NodesT &nodes, const Ray &ray_in, Hit &hit, StackT &stk, uint32_t &n_nodes, uint32_t &n_prims) {
    Trav<ANY> tv;
    // any-hit: maxt never shrinks, so the set of records reached and with it the answer is the same for every child
    // order; the nearer child first finds an occluder sooner.  closest-hit: the reference's order unless sc.ordered
    const bool ordered = (ANY && (sc.ordered & 2)) || (sc.ordered & 1);
    tv.begin(sc, ray_in, n_nodes);
    for (;;) {
        const unsigned long long m_in = __ballot(tv.at_inner());
        if (m_in) {
            if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) tv.box_step_fast(nodes, stk, n_nodes, ordered); }
            else if (tv.at_inner()) { if (tv.special) tv.box_step(nodes, stk, n_nodes, ordered); else tv.box_step_fast(nodes, stk, n_nodes, ordered); }
        }
        const unsigned long long m_lf = __ballot(tv.at_leaf());
        if (m_lf && (__popcll(m_lf) >= sc.prim_lanes || __ballot(tv.at_inner()) == 0ull)) { if (tv.at_leaf()) tv.prim_step(sc, n_prims); }
        if (tv.at_pop()) tv.pop_step(stk);
        if (__ballot(!tv.done()) == 0ull) break;
    }
    hit = tv.hit;
    return tv.found;
}

template <bool ANY, typename StackT>
__device__ __forceinline__ bool trace(const DevScene &sc, const Ray &ray_in, Hit &hit, StackT &stk, uint32_t &n_nodes, uint32_t &n_prims) {
    GlobalNodes g; g.nodes = sc.nodes;
    return trace<ANY>(sc, g, ray_in, hit, stk, n_nodes, n_prims);
}

// the same loop over the four-box nodes (sc.nodes4 must not be null)
template <bool ANY, typename StackT>
__device__ __forceinline__ bool trace4(const DevScene &sc, const Ray &ray_in, Hit &hit, StackT &stk, uint32_t &n_nodes, uint32_t &n_prims) {
    Trav<ANY> tv;
    const bool ordered = (ANY && (sc.ordered & 2)) || (sc.ordered & 1);
    const bool far = ANY && (sc.ordered & 4);
    if (tv.begin(sc, ray_in, n_nodes)) tv.cur = sc.root4;
    for (;;) {
        const unsigned long long m_in = __ballot(tv.at_inner());
        if (m_in) {
            if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) n_nodes += tv.box_step4_fast(sc.nodes4, stk, ordered, true, far); }
            else if (tv.at_inner()) { if (tv.special) n_nodes += tv.box_step4(sc.nodes4, stk, ordered, far); else n_nodes += tv.box_step4_fast(sc.nodes4, stk, ordered, true, far); }
        }
        const unsigned long long m_lf = __ballot(tv.at_leaf());
        if (m_lf && (__popcll(m_lf) >= sc.prim_lanes || __ballot(tv.at_inner()) == 0ull)) { if (tv.at_leaf()) tv.prim_step(sc, n_prims); }
        if (tv.at_pop()) tv.pop_step(stk);
        if (__ballot(!tv.done()) == 0ull) break;
    }
    hit = tv.hit;
    return tv.found;
}

// the same loop over the 64-byte quantised four-box nodes (sc.nodes4q must not be null), as the trace stage of big device-built trees
// walks them; two records of a leaf per trip where the leaves were placed for it
template <bool ANY, typename StackT>
__device__ __forceinline__ bool trace4q(const DevScene &sc, const Ray &ray_in, Hit &hit, StackT &stk, uint32_t &n_nodes, uint32_t &n_prims) {
    Trav<ANY> tv;
    const bool ordered = (ANY && (sc.ordered & 2)) || (sc.ordered & 1);
    const bool far = ANY && (sc.ordered & 4);
    if (tv.begin(sc, ray_in, n_nodes)) tv.cur = sc.root4;
    for (;;) {
        const unsigned long long m_in = __ballot(tv.at_inner());
        if (m_in) {
            if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) n_nodes += tv.box_step4q_fast(sc.nodes4q, stk, ordered, true, far); }
            else if (tv.at_inner()) { if (tv.special) n_nodes += tv.box_step4q(sc.nodes4q, stk, ordered, far); else n_nodes += tv.box_step4q_fast(sc.nodes4q, stk, ordered, true, far); }
        }
        const unsigned long long m_lf = __ballot(tv.at_leaf());
        if (m_lf && (__popcll(m_lf) >= sc.prim_lanes || __ballot(tv.at_inner()) == 0ull)) {
            if (tv.at_leaf()) { if (sc.leaf2) (void) tv.prim_step2(sc, n_prims, ANY); else tv.prim_step(sc, n_prims); }
        }
        if (tv.at_pop()) tv.pop_step(stk);
        if (__ballot(!tv.done()) == 0ull) break;
    }
    hit = tv.hit;
    return tv.found;
}

// the same loop over the quantised eight-box nodes (sc.nodes8 must not be null; the stack must hold ND_STACK8 entries)
template <bool ANY, typename StackT>
__device__ __forceinline__ bool trace8(const DevScene &sc, const Ray &ray_in, Hit &hit, StackT &stk, uint32_t &n_nodes, uint32_t &n_prims) {
    Trav<ANY> tv;
    if (tv.begin(sc, ray_in, n_nodes)) tv.cur = sc.root4;
    for (;;) {
        const unsigned long long m_in = __ballot(tv.at_inner());
        if (m_in) {
            if (__ballot(tv.at_inner() && tv.special) == 0ull) { if (tv.at_inner()) n_nodes += tv.box_step8_fast(sc.nodes8, stk); }
            else if (tv.at_inner()) { if (tv.special) n_nodes += tv.box_step8(sc.nodes8, stk); else n_nodes += tv.box_step8_fast(sc.nodes8, stk); }
        }
        const unsigned long long m_lf = __ballot(tv.at_leaf());
        if (m_lf && (__popcll(m_lf) >= sc.prim_lanes || __ballot(tv.at_inner()) == 0ull)) { if (tv.at_leaf()) tv.prim_step(sc, n_prims); }
        if (tv.at_pop()) tv.pop_step(stk);
        if (__ballot(!tv.done()) == 0ull) break;
    }
    hit = tv.hit;
    return tv.found;
}

// per-lane stack in private (scratch) memory
template <int SIZE> struct ScratchStackT {
    uint2 e[SIZE];
    __device__ __forceinline__ void push(int i, uint2 v) { e[i] = v; }
    __device__ __forceinline__ uint2 pop(int i) const { return e[i]; }
};
typedef ScratchStackT<ND_STACK> ScratchStack;
// ... whose entries lose the low 16 bits of the entry distance, as every LDS level of LdsStack6T does (k_trace over quantised nodes:
// nori_hip_trace then answers with the traversal the trace stage runs, and the parity tests compare exactly that with the oracle)
template <int SIZE> struct ScratchStackCutT {
    uint2 e[SIZE];
    __device__ __forceinline__ void push(int i, uint2 v) { e[i] = make_uint2(v.x, ((v.y << 1) >> 16) << 15); }
    __device__ __forceinline__ uint2 pop(int i) const { return e[i]; }
};

// Per-lane traversal stack with its first LDS_DEPTH entries in LDS ([entry][lane] layout: the 64 lanes of a wave
// touch 64 consecutive 8-byte words, conflict-free for ds_read/write_b64) and the rare deeper entries in scratch.
#define LDS_DEPTH 12
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
// (the spill array is a separate object, handed in by pointer: as a member it drags the whole struct, the LDS base
// included, into scratch memory, and every push and pop then starts with a scratch load of that base)
template <int DEPTH> struct LdsStackT {
    static constexpr int WAVE_BYTES = DEPTH * 64 * 8;      // LDS of one wave's stacks
    lds_u64 *base;                      // this lane's column: base[entry * 64] (LDS address space, so ds_read/ds_write)
    uint2 *spill;                       // ND_STACK - DEPTH entries of private memory for the deeper levels
    __device__ __forceinline__ void attach(void *lds, unsigned int wave, unsigned int lane) { base = (lds_u64 *) lds + wave * (DEPTH * 64) + lane; }
    __device__ __forceinline__ void push(int i, uint2 v) { if (i < DEPTH) base[i * 64] = ((unsigned long long) v.y << 32) | v.x; else spill[i - DEPTH] = v; }
    __device__ __forceinline__ uint2 pop(int i) const {
        if (i < DEPTH) { unsigned long long w = base[i * 64]; return make_uint2((unsigned int) w, (unsigned int) (w >> 32)); }
        return spill[i - DEPTH];
    }
};
typedef LdsStackT<LDS_DEPTH> LdsStack;
// The same with 6-byte entries in LDS: the reference as it is, the entry distance cut to its upper 16 bits without the sign (8 exponent and
// 8 mantissa bits, rounded towards zero).  Half as many again levels fit the same LDS, and the levels beyond go to scratch memory, i.e. to HBM
// (the 10 M-triangle scene: 8 levels 319 ms a step, 9 levels 305).  Entry distances are never negative (they are clamped to mint), so the
// distance read back is at most the one stored: the test at the pop, entry <= maxt, lets a few more entries through and never drops one -
// such an entry's boxes or records are then tested against the exact maxt, so no result changes.  Deeper levels keep all 32 bits.
// Precondition: mint >= 0 (the sign bit is not stored).  Every ray of a render over quantised nodes satisfies it - camera rays start at the
// near clip, bounce and shadow rays at 1e-3; the probe rays of subsurface shapes (mint = -1e-3) belong to scenes that keep the float nodes
// and 8-byte entries.
typedef __attribute__((address_space(3))) unsigned int lds_u32;
typedef __attribute__((address_space(3))) unsigned short lds_u16;
template <int DEPTH> struct LdsStack6T {
    static constexpr int WAVE_BYTES = DEPTH * 64 * 6;
    lds_u32 *ref; lds_u16 *dist;        // this lane's columns: ref[entry * 64], dist[entry * 64]
    uint2 *spill;
    __device__ __forceinline__ void attach(void *lds, unsigned int wave, unsigned int lane) {
        lds_u32 *w = (lds_u32 *) lds + wave * (WAVE_BYTES / 4);
        ref = w + lane; dist = (lds_u16 *) (w + DEPTH * 64) + lane;
    }
    __device__ __forceinline__ void push(int i, uint2 v) {
        if (i < DEPTH) { ref[i * 64] = v.x; dist[i * 64] = (unsigned short) ((v.y << 1) >> 16); } else spill[i - DEPTH] = v;
    }
    __device__ __forceinline__ uint2 pop(int i) const {
        if (i < DEPTH) return make_uint2(ref[i * 64], (unsigned int) dist[i * 64] << 15);
        return spill[i - DEPTH];
    }
};

struct Surface { F3 p; F2 uv; Frame sh; F3 geo_n; uint32_t shape; uint32_t prim; };

// ---- images (cuda/nori/imagetexture.cpp.cu:12-26, bumpMap.cpp.cu:10-27, normalMap.cpp.cu:10-24; src/imagetexture.cpp:49-77)
// rescale, shift, keep the fractional part
__device__ __forceinline__ F2 tex_coords(const NoriHipTexture &t, F2 uv) {
    float x = (uv.x / t.scale[0]) - t.delta[0], y = (uv.y / t.scale[1]) - t.delta[1];
    return f2(x - floorf(x), y - floorf(y));
}
// One filtered lookup of the 4-byte texels as four floats; `cu` runs along the columns, `rv` along the rows, both in
// [0, 1).  BILINEAR restates tex2D of the CUDA fork's texture objects (normalised coordinates, wrap addressing, linear
// filter: texel centres at i + 0.5, blend weights held with 8 fractional bits); NEAREST is the CPU renderer's
// truncating lookup.  `dec` turns a texel word into four floats.
template <typename Dec>
__device__ __forceinline__ float4 tex_lookup(const DevScene &sc, const NoriHipTexture &t, float cu, float rv, Dec dec) {
    const uint32_t *tx = reinterpret_cast<const uint32_t *>(sc.tex_data + t.offset);
    const int W = (int) t.width, H = (int) t.height;
    if (t.lookup == NORI_HIP_LOOKUP_NEAREST) {
        int col = (int) (cu * (float) W), row = (int) (rv * (float) H);
        col = col < W ? col : W - 1; row = row < H ? row : H - 1;
        return dec(tx[(size_t) row * W + col]);
    }
    float xb = cu * (float) W - 0.5f, yb = rv * (float) H - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float a = floorf((xb - fi) * 256.0f + 0.5f) * (1.0f / 256.0f), b = floorf((yb - fj) * 256.0f + 0.5f) * (1.0f / 256.0f);
    int i0 = (int) fi, j0 = (int) fj;
    i0 = i0 < 0 ? i0 + W : i0; j0 = j0 < 0 ? j0 + H : j0;
    int i1 = i0 + 1 == W ? 0 : i0 + 1, j1 = j0 + 1 == H ? 0 : j0 + 1;
    float4 t00 = dec(tx[(size_t) j0 * W + i0]), t10 = dec(tx[(size_t) j0 * W + i1]), t01 = dec(tx[(size_t) j1 * W + i0]), t11 = dec(tx[(size_t) j1 * W + i1]);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return make_float4(((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x, ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y,
                       ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z, ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w);
}
struct DecRgba8 { __device__ __forceinline__ float4 operator()(uint32_t w) const {      // cudaReadModeNormalizedFloat of uchar4
    return make_float4((float) (w & 0xffu) / 255.0f, (float) ((w >> 8) & 0xffu) / 255.0f, (float) ((w >> 16) & 0xffu) / 255.0f, (float) (w >> 24) / 255.0f); } };
struct DecS16x2 { __device__ __forceinline__ float4 operator()(uint32_t w) const {      // ... of short: v / 32767, clamped at -1
    float x = (float) (int16_t) (w & 0xffffu) / 32767.0f, y = (float) (int16_t) (w >> 16) / 32767.0f;
    return make_float4(x < -1.0f ? -1.0f : x, y < -1.0f ? -1.0f : y, 0.f, 0.f); } };
// Color3f::toLinearRGB (src/common.cpp:208-222)
__device__ __forceinline__ float srgb_to_linear(float v) { return v <= 0.04045f ? v * (1.0f / 12.92f) : powf((v + 0.055f) * (1.0f / 1.055f), 2.4f); }
// Checkerboard<T>::eval (src/checkerboard.cpp:32-55): which of the two values uv falls on.  fmodf is exact, so this is the
// reference's decision bit for bit.
__device__ __forceinline__ bool checker_first(const NoriHipTexture &t, F2 uv) {
    float x = (uv.x / t.scale[0]) - t.delta[0], y = (uv.y / t.scale[1]) - t.delta[1];
    float modx = fmodf(x, 2.0f), mody = fmodf(y, 2.0f);
    if (x < 0) modx = fmodf(fabsf(x) + 1, 2.0f);
    if (y < 0) mody = fmodf(fabsf(y) + 1, 2.0f);
    return (modx <= 1 && mody <= 1) || (modx >= 1 && mody >= 1);
}
// ImageTexture<Color3f>::eval / Checkerboard<Color3f>::eval
__device__ __forceinline__ F3 tex_color(const DevScene &sc, const NoriHipTexture &t, F2 uv) {
    if (t.kind == NORI_HIP_TEX_CHECKER_COLOR) {
        const float *v = reinterpret_cast<const float *>(sc.tex_data + t.offset) + (checker_first(t, uv) ? 0 : 3);
        return f3(v[0], v[1], v[2]);
    }
    F2 c = tex_coords(t, uv);
    float4 rgb = tex_lookup(sc, t, c.y, c.x, DecRgba8());                  // tex2D(image_tex, y, x) / row = x*rows, col = y*cols
    if (t.lookup == NORI_HIP_LOOKUP_NEAREST) return f3(rgb.x, rgb.y, rgb.z);
    return f3(srgb_to_linear(rgb.x), srgb_to_linear(rgb.y), srgb_to_linear(rgb.z));
}
// BumpMap::eval / NormalMap::eval: the new shading normal
__device__ __forceinline__ F3 tex_normal(const DevScene &sc, const NoriHipTexture &t, F2 uv, const Frame &fr) {
    // the CPU renderer's bump_map over a Texture<float> (src/bumpMap.cpp:44-50): the value is assigned to the normal as it is
    if (t.kind == NORI_HIP_TEX_CHECKER_FLOAT) return f3(reinterpret_cast<const float *>(sc.tex_data + t.offset)[checker_first(t, uv) ? 0 : 1]);
    if (t.kind == NORI_HIP_TEX_IMAGE_FLOAT) {                              // ImageTexture<float>::eval, src/imagetexture.cpp:26-46
        F2 q = tex_coords(t, uv);
        const int row = (int) (q.x * (float) t.height), col = (int) (q.y * (float) t.width);
        const uint32_t w = reinterpret_cast<const uint32_t *>(sc.tex_data + t.offset)[(size_t) row * t.width + col];
        return f3((float) (w & 0xffu) / 255.0f);
    }
    F2 c = tex_coords(t, uv);
    if (t.kind == NORI_HIP_TEX_BUMP) {
        float4 g = tex_lookup(sc, t, c.x, c.y, DecS16x2());                // tex2D(image_tex_x / _y, x, y)
        F3 n = (fr.n + g.x * cross(fr.n, fr.s)) - g.y * cross(fr.n, fr.t);
        return normalized(n);
    }
    float4 rgb = tex_lookup(sc, t, c.y, c.x, DecRgba8());
    F3 n = normalized(f3((2.0f * rgb.x) - 1.0f, (2.0f * rgb.y) - 1.0f, rgb.z));
    return fr.toWorld(n);
}

// Mesh::setHitInformation (src/mesh.cpp:123-168) / Sphere::setHitInformation (src/sphere.cpp:115-141); EXT adds the
// CUDA fork's normal modifier (cuda/nori/mesh.cpp.cu:181-185, sphere.cpp.cu:145-149)
template <bool EXT = false>
__device__ __forceinline__ void surface_from_hit(const DevScene &sc, const Ray &ray, const Hit &hit, Surface &s, bool want_geo) {
    const float4 *r = sc.prims + 3 * (size_t) hit.pos;
    float4 C = r[2];
    uint32_t flags = __float_as_uint(C.z);
    s.shape = flags & 0xffffffu; s.prim = __float_as_uint(C.y);
    const NoriHipShape &shp = sc.shapes[s.shape];
    if (flags & 0x40000000u) {
        F3 c = f3(shp.center);
        s.p = ray.o + hit.t * ray.d;
        F3 n = normalized(s.p - c);
        s.sh.set(n); s.geo_n = n;
        F3 pl = s.p - c;
        float frac = pl.z / shp.radius;
        if (frac < -1) frac = -1; else if (frac > 1) frac = 1;
        float theta = nd_acosf(frac) / ND_PI;
        float phi = (nd_atan2f(pl.y, pl.x) + ND_PI) / (2 * ND_PI);
        s.uv = f2(phi, theta);
    } else {
        const NoriHipShadePrim &sp = sc.shade[hit.pos];
        float b0 = 1 - (hit.u + hit.v), b1 = hit.u, b2 = hit.v;
        F3 p0 = f3(sp.p0), p1 = f3(sp.p1), p2 = f3(sp.p2);
        s.p = (b0 * p0 + b1 * p1) + b2 * p2;
        if (shp.flags & 2u) s.uv = f2((b0 * sp.uv0[0] + b1 * sp.uv1[0]) + b2 * sp.uv2[0], (b0 * sp.uv0[1] + b1 * sp.uv1[1]) + b2 * sp.uv2[1]);
        else s.uv = f2(hit.u, hit.v);
        if (shp.flags & 1u) {
            F3 n = normalized_seq((b0 * f3(sp.n0) + b1 * f3(sp.n1)) + b2 * f3(sp.n2));
            s.sh.set(n);
            if (want_geo) s.geo_n = normalized(cross(p1 - p0, p2 - p0));
        } else {
            F3 g = normalized(cross(p1 - p0, p2 - p0));
            s.sh.set(g); s.geo_n = g;
        }
    }
    if (EXT && shp.normal_tex) {
        const NoriHipTexture &nt = sc.textures[shp.normal_tex - 1];
        // the float kinds are the CPU renderer's modifier: meshes with vertex normals only (src/mesh.cpp:151-164)
        const bool cpu_kind = nt.kind == NORI_HIP_TEX_CHECKER_FLOAT || nt.kind == NORI_HIP_TEX_IMAGE_FLOAT;
        if (!cpu_kind || (!(flags & 0x40000000u) && (shp.flags & 1u))) { F3 n = tex_normal(sc, nt, s.uv, s.sh); s.sh.set(n); }
    }
}

// PerspectiveCamera::sampleRay (src/perspective.cpp:90-111); Transform::operator* (include/nori/transform.h:68-81)
__device__ __forceinline__ F3 xform_point(const float *m, F3 p) {
    float r0 = ((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3] * 1.0f;
    float r1 = ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7] * 1.0f;
    float r2 = ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11] * 1.0f;
    float r3 = ((m[12] * p.x + m[13] * p.y) + m[14] * p.z) + m[15] * 1.0f;
    return f3(r0 / r3, r1 / r3, r2 / r3);
}
__device__ __forceinline__ F3 xform_vector(const float *m, F3 v) {
    return f3((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z, (m[8] * v.x + m[9] * v.y) + m[10] * v.z);
}
__device__ __forceinline__ void camera_ray(const DevScene &sc, F2 ps, Ray &ray) {
    F3 nearP = xform_point(sc.cam.sample_to_camera, f3(ps.x * sc.inv_w, ps.y * sc.inv_h, 0.0f));
    F3 d = normalized(nearP);
    float invZ = 1.0f / d.z;
    ray.o = xform_point(sc.cam.camera_to_world, f3(0.f, 0.f, 0.f));
    ray.d = xform_vector(sc.cam.camera_to_world, d);
    ray.mint = sc.cam.near_clip * invZ;
    ray.maxt = sc.cam.far_clip * invZ;
}

// ---- warps (src/warp.cpp:59-63,221-244,263-267) ----------------------------------------------------------
__device__ __forceinline__ F3 square_to_cosine_hemisphere(F2 s) {
    float r = nd_sqrt(s.x);
    float theta = 2 * ND_PI * s.y;
    float sn, cs; nd_sincosf(theta, &sn, &cs);
    float dx = r * cs, dy = r * sn;
    return f3(dx, dy, nd_sqrt(1.0f - dx * dx - dy * dy));
}
__device__ __forceinline__ F3 square_to_beckmann(F2 s, float alpha) {
    float inner = 1.0f / (1.0f - alpha * alpha * nd_logf(1.0f - s.x));
    float theta = nd_acosf(nd_sqrt(inner));
    float phi = s.y * 2.0f * ND_PI;
    float st, ct, sp, cp; nd_sincosf(theta, &st, &ct); nd_sincosf(phi, &sp, &cp);
    return f3(st * cp, st * sp, ct);
}
__device__ __forceinline__ F3 square_to_uniform_triangle(F2 s) {
    float su1 = nd_sqrt(s.x);
    float u = 1.f - su1, v = s.y * su1;
    return f3(u, v, 1.f - u - v);
}

// src/common.cpp:285-314
__device__ __forceinline__ float fresnel(float cosThetaI, float extIOR, float intIOR) {
    float etaI = extIOR, etaT = intIOR;
    if (extIOR == intIOR) return 0.0f;
    if (cosThetaI < 0.0f) { float s = etaI; etaI = etaT; etaT = s; cosThetaI = -cosThetaI; }
    float eta = etaI / etaT, sinThetaTSqr = eta * eta * (1 - cosThetaI * cosThetaI);
    if (sinThetaTSqr > 1.0f) return 1.0f;
    float cosThetaT = nd_sqrt(1.0f - sinThetaTSqr);
    float Rs = (etaI * cosThetaI - etaT * cosThetaT) / (etaI * cosThetaI + etaT * cosThetaT);
    float Rp = (etaT * cosThetaI - etaI * cosThetaT) / (etaT * cosThetaI + etaI * cosThetaT);
    return (Rs * Rs + Rp * Rp) / 2.0f;
}
__device__ __forceinline__ float tan_theta(F3 v) { float temp = 1 - v.z * v.z; if (temp <= 0.0f) return 0.0f; return nd_sqrt(temp) / v.z; }

// ---- BSDFs ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float beckmann_D(float alpha, F3 m) {          // src/microfacet.cpp:55-61
    float temp = tan_theta(m) / alpha, ct = m.z, ct2 = ct * ct;
    return nd_expf(-temp * temp) / (ND_PI * alpha * alpha * ct2 * ct2);
}
__device__ __forceinline__ float smith_g1(float alpha, F3 v, F3 m) {      // src/microfacet.cpp:64-84
    float tanTheta = tan_theta(v);
    if (tanTheta == 0.0f) return 1.0f;
    if (dot(m, v) * v.z <= 0) return 0.0f;
    float a = 1.0f / (alpha * tanTheta);
    if (a >= 1.6f) return 1.0f;
    float a2 = a * a;
    return (3.535f * a + 2.181f * a2) / (1.0f + 2.276f * a + 2.577f * a2);
}
// Subsurface::fresnelTransmission (src/subsurface.cpp:173-178)
__device__ __forceinline__ float ss_fresnel_transmission(const NoriHipShape &b, F3 wi, F3 wo) {
    float Ft_i = 1.0f - fresnel(wi.z, b.ext_ior, b.int_ior);
    float Ft_o = 1.0f - fresnel(wo.z, b.ext_ior, b.int_ior);
    return Ft_i * Ft_o;
}
// `alb`: the albedo at the hit's uv (constant colour or image; diffuse albedo | microfacet kd | subsurface albedo)
__device__ __forceinline__ F3 bsdf_eval(const NoriHipShape &b, F3 alb, F3 wi, F3 wo) {
    if (b.bsdf == NORI_HIP_BSDF_DIFFUSE) {                                // src/diffuse.cpp:75-86
        if (wi.z <= 0 || wo.z <= 0) return f3(0.f);
        return alb * ND_INV_PI;
    }
    if (b.bsdf == NORI_HIP_BSDF_SUBSURFACE)                               // src/subsurface.cpp:99-104 (no hemisphere test)
        return (ss_fresnel_transmission(b, wi, wo) * alb) * ND_INV_PI;
    if (b.bsdf == NORI_HIP_BSDF_MICROFACET) {                             // src/microfacet.cpp:87-100
        F3 wh = normalized(wi + wo);
        float G = smith_g1(b.alpha, wi, wh) * smith_g1(b.alpha, wo, wh);
        float ci = wi.z, co = wo.z;
        if (co > 0 && ci > 0) {
            float spec = (b.ks * beckmann_D(b.alpha, wh) * fresnel(dot(wh, wi), b.ext_ior, b.int_ior) * G) / (4 * ci * co);
            return alb / ND_PI + f3(spec);
        }
        return f3(0.f);
    }
    return f3(0.f);
}
__device__ __forceinline__ float bsdf_pdf(const NoriHipShape &b, F3 wi, F3 wo) {
    if (b.bsdf == NORI_HIP_BSDF_DIFFUSE || b.bsdf == NORI_HIP_BSDF_SUBSURFACE) {   // src/diffuse.cpp:89-105, src/subsurface.cpp:181-197
        if (wi.z <= 0 || wo.z <= 0) return 0.0f;
        return ND_INV_PI * wo.z;
    }
    if (b.bsdf == NORI_HIP_BSDF_MICROFACET) {                             // src/microfacet.cpp:103-117
        F3 wh = normalized(wi + wo);
        float ch = wh.z, co = wo.z, ci = wi.z;
        float Jh = 1.0f / (4 * dot(wh, wo));
        if (co > 0 && ci > 0) return b.ks * beckmann_D(b.alpha, wh) * ch * Jh + (1 - b.ks) * co / ND_PI;
        return 0.f;
    }
    return 0.0f;
}
// returns the importance weight; wo_valid=false marks the reference's early returns that leave wo unset
// (`sample2` is the second of the two samples the integrator hands over; only the subsurface model uses it)
__device__ __forceinline__ F3 bsdf_sample(const NoriHipShape &b, F3 alb, F3 wi, F2 sample, F2 sample2, F3 &wo, bool &is_delta, bool &wo_valid) {
    wo_valid = true; is_delta = false; wo = f3(0.f);
    switch (b.bsdf) {
    case NORI_HIP_BSDF_DIFFUSE:                                           // src/diffuse.cpp:109-126
        if (wi.z <= 0) { wo_valid = false; return f3(0.f); }
        wo = square_to_cosine_hemisphere(sample);
        return alb;
    case NORI_HIP_BSDF_SUBSURFACE:                                        // src/subsurface.cpp:253-281
        wo = square_to_cosine_hemisphere(sample2);
        if (wi.z <= 0 || wo.z <= 0) return f3(0.f);
        return ss_fresnel_transmission(b, wi, wo) * alb;
    case NORI_HIP_BSDF_MICROFACET: {                                      // src/microfacet.cpp:121-161
        if (sample.x < b.ks) {
            sample.x /= b.ks;
            F3 wh = square_to_beckmann(sample, b.alpha);
            wh = wh / norm(wh);
            wo = -wi + (2 * dot(wh, wi)) * wh;
            if (wo.z <= 0) return f3(0.f);
        } else {
            sample.x = (sample.x - b.ks) / (1.0f - b.ks);
            wo = square_to_cosine_hemisphere(sample);
        }
        float co = wo.z;
        if (co > 0) {
            float pdf_val = bsdf_pdf(b, wi, wo);
            if (fabsf(pdf_val) > 1e-6) return (bsdf_eval(b, alb, wi, wo) * co) / pdf_val;
        }
        return f3(0.f);
    }
    case NORI_HIP_BSDF_DIELECTRIC: {                                      // src/dielectric.cpp:48-103
        float ci = wi.z;
        float extIOR = b.ext_ior, intIOR = b.int_ior;
        if (ci < 0) { extIOR = b.int_ior; intIOR = b.ext_ior; }
        float F = fresnel(fabsf(ci), extIOR, intIOR);
        is_delta = true;
        if (sample.x <= F) { wo = f3(-wi.x, -wi.y, wi.z); return f3(1.0f); }
        F3 n = f3(0, 0, 1.0f); if (ci < 0) n = f3(0, 0, -1.0f);
        float win = dot(wi, n);
        float IORrel = extIOR / intIOR;
        float root = (float) sqrt(1.0 - (double) (IORrel * IORrel) * (1.0 - (double) (win * win)));   // double in the reference
        F3 w = (-IORrel) * (wi - win * n) - n * root;
        wo = w / norm(w);
        float inv = 1.0f / IORrel;
        return f3(inv * inv);
    }
    default:                                                              // mirror, src/mirror.cpp:41-58
        if (wi.z <= 0) { wo_valid = false; return f3(0.f); }
        wo = f3(-wi.x, -wi.y, wi.z); is_delta = true;
        return f3(1.0f);
    }
}

// ---- area emitter (src/arealight.cpp:41-116, src/mesh.cpp:40-61, include/nori/dpdf.h:119-157) -------------
__device__ __forceinline__ float emitter_pdf(float area_pdf, F3 ref, F3 p, F3 n, F3 wi) {
    F3 dd = ref - p; float dist2 = dot(dd, dd);
    float costheta = -dot(normalized(wi), normalized(n));
    if (costheta <= 0) return 0.0f;
    return area_pdf * dist2 / fabsf(costheta);
}
__device__ __forceinline__ float shape_area_pdf(const DevScene &sc, const NoriHipShape &shp) {
    if (shp.type == NORI_HIP_SHAPE_SPHERE) { float a = 1.f / shp.radius; return (a * a) * (1.0f / (4.0f * ND_PI)); }   // src/sphere.cpp:150-152
    return sc.emitters[shp.emitter].area_pdf;
}
// Emitter::sample: fills p, n, wi, shadow ray (from the light to ref), returns radiance / pdf
__device__ __forceinline__ F3 emitter_sample(const DevScene &sc, const NoriHipEmitter &em, F3 ref, F2 sample, F3 &wi, F3 &ln, F3 &lp, Ray &shadow) {
    const NoriHipShape &shp = sc.shapes[em.shape];
    F3 p, n; float apdf;
    if (shp.type == NORI_HIP_SHAPE_SPHERE) {                              // src/sphere.cpp:143-149, src/warp.cpp:26-31,198-203
        float phi = 2 * ND_PI * sample.x, sn, cs; nd_sincosf(phi, &sn, &cs);
        float z = 2.0f * sample.y - 1.0f; float r = nd_sqrt(1.0f - (z * z));
        F3 q = f3(cs * r, sn * r, z);
        p = f3(shp.center) + shp.radius * q; n = q;
        apdf = (float) (1.0 / (4.0 * ND_PI * shp.radius * shp.radius));
    } else {
        const float *cdf = sc.cdf + em.cdf_offset;
        // std::lower_bound over n_tris+1 entries, then DiscretePDF::sample's clamps
        uint32_t lo = 0, len = em.n_tris + 1;
        while (len > 0) { uint32_t half = len >> 1; if (cdf[lo + half] < sample.x) { lo += half + 1; len -= half + 1; } else len = half; }
        int idx = (int) lo - 1; if (idx < 0) idx = 0; if (idx > (int) em.n_tris - 1) idx = (int) em.n_tris - 1;
        float sx = (sample.x - cdf[idx]) / (cdf[idx + 1] - cdf[idx]);
        F3 bc = square_to_uniform_triangle(f2(sx, sample.y));
        const NoriHipShadePrim &lt = sc.light_prims[em.light_prim_offset + idx];
        F3 p0 = f3(lt.p0), p1 = f3(lt.p1), p2 = f3(lt.p2);
        p = (bc.x * p0 + bc.y * p1) + bc.z * p2;
        if (shp.flags & 1u) n = normalized_seq((bc.x * f3(lt.n0) + bc.y * f3(lt.n1)) + bc.z * f3(lt.n2));
        else n = normalized(cross(p1 - p0, p2 - p0));
        apdf = em.area_pdf;
    }
    F3 dir = p - ref;
    float dist = norm(dir);
    wi = normalized(dir); ln = n; lp = p;
    shadow.o = p; shadow.d = -wi; shadow.mint = 1e-3f; shadow.maxt = dist - 1e-3f;
    float pdfv = emitter_pdf(shape_area_pdf(sc, shp), ref, p, n, wi);
    (void) apdf;
    if (fabsf(pdfv) > 1e-6) return f3(em.radiance) / pdfv;
    return f3(0.f);
}

// ---- environment emitter (cuda/nori/spheremap.cpp.cu; sampling and density are this library's, see NoriHipEnvMap) -------------
__device__ __forceinline__ F2 env_uv(const DevScene &sc, F3 dir) {        // SphereMap::eval's mapping, :64-67
    const float *f = sc.env.frame;
    F3 p = f3(dot(dir, f3(f[0], f[1], f[2])), dot(dir, f3(f[3], f[4], f[5])), dot(dir, f3(f[6], f[7], f[8])));
    float z = p.z < -1.f ? -1.f : p.z > 1.f ? 1.f : p.z;
    return f2(nd_acosf(z) / ND_PI, (nd_atan2f(p.y, p.x) + ND_PI) / (2 * ND_PI));
}
__device__ __forceinline__ F3 env_eval(const DevScene &sc, F3 dir) {
    F3 c = tex_color(sc, sc.textures[sc.env.texture - 1], env_uv(sc, dir));
    return c * sc.env.brightness;
}
// solid-angle density of env_sample at `dir`: cell probability / (cell size in (theta, phi) * sin theta)
__device__ __forceinline__ float env_pdf(const DevScene &sc, F3 dir) {
    F2 uv = env_uv(sc, dir);
    int i = (int) (uv.x * (float) sc.env_h), j = (int) (uv.y * (float) sc.env_w);
    i = i < 0 ? 0 : i > (int) sc.env_h - 1 ? (int) sc.env_h - 1 : i; j = j < 0 ? 0 : j > (int) sc.env_w - 1 ? (int) sc.env_w - 1 : j;
    const uint32_t k = (uint32_t) i * sc.env_w + (uint32_t) j;
    const float p = sc.env_cdf[k + 1] - sc.env_cdf[k];
    const float st = sinf(uv.x * ND_PI);
    if (!(st > 1e-6f)) return 0.f;
    return p * (float) sc.env_w * (float) sc.env_h / (2.0f * ND_PI * ND_PI * st);
}
// a direction drawn from the table: the cell by the CDF (DiscretePDF::sampleReuse), uniform inside the cell
__device__ __forceinline__ F3 env_sample(const DevScene &sc, F2 sample, float &pdf) {
    const uint32_t n = sc.env_w * sc.env_h;
    uint32_t lo = 0, len = n + 1;
    while (len > 0) { uint32_t half = len >> 1; if (sc.env_cdf[lo + half] < sample.x) { lo += half + 1; len -= half + 1; } else len = half; }
    int idx = (int) lo - 1; if (idx < 0) idx = 0; if (idx > (int) n - 1) idx = (int) n - 1;
    const float c0 = sc.env_cdf[idx], c1 = sc.env_cdf[idx + 1];
    float sx = (c1 > c0) ? (sample.x - c0) / (c1 - c0) : 0.5f; sx = sx < 0.f ? 0.f : sx > 0.999999f ? 0.999999f : sx;
    const uint32_t i = (uint32_t) idx / sc.env_w, j = (uint32_t) idx % sc.env_w;
    const float u = ((float) i + sx) / (float) sc.env_h, v = ((float) j + sample.y) / (float) sc.env_w;
    const float theta = u * ND_PI, phi = v * 2 * ND_PI - ND_PI;
    float st, ct, sp, cp; nd_sincosf(theta, &st, &ct); nd_sincosf(phi, &sp, &cp);
    const float *f = sc.env.frame;
    F3 l = f3(st * cp, st * sp, ct);
    F3 d = (f3(f[0], f[1], f[2]) * l.x + f3(f[3], f[4], f[5]) * l.y) + f3(f[6], f[7], f[8]) * l.z;
    pdf = (st > 1e-6f) ? (c1 - c0) * (float) sc.env_w * (float) sc.env_h / (2.0f * ND_PI * ND_PI * st) : 0.f;
    return d;
}

// ---- one path, PathMisIntegrator::Li (src/pathMisIntegrator.cpp:18-168) ---------------------------------
// State kept between bounces (what the reference keeps in locals across iterations of its while(true)).
struct PathState {
    Ray ray; F3 t, li; float last_bsdf_pdf; bool last_delta;
};
__device__ __forceinline__ void path_init(PathState &ps, const Ray &r) {
    ps.ray = r; ps.t = f3(1.0f); ps.li = f3(0.f); ps.last_bsdf_pdf = 1.0f; ps.last_delta = true;
}
// A ray that left the scene: the environment's radiance with the MIS weight of the BSDF sample that made the ray
// (cuda/nori/pathMisIntegrator.cpp.cu:46-75).  Zero without an environment emitter.
template <bool EXT>
__device__ __forceinline__ F3 env_miss(const DevScene &sc, F3 dir, F3 t, float last_bsdf_pdf, bool last_delta) {
    if (!EXT || !sc.env.texture) return f3(0.f);
    F3 flight = env_eval(sc, dir);
    float light_pdf = env_pdf(sc, dir);
    float wmat = last_bsdf_pdf / (light_pdf + last_bsdf_pdf);
    if (last_delta) wmat = 1.0f;
    return (wmat > 0.0f) ? (wmat * t) * flight : f3(0.f);
}
// What the shading of one hit leaves behind for the shadow stage
struct ShadeOut { bool end; bool has_shadow; Ray shadow; F3 contrib; bool culled; };    // culled: an emitter sample whose term is exactly zero (no shadow ray)

// ---- subsurface: where the path leaves the object (src/subsurface.cpp:106-150,199-249, src/pathMisIntegrator.cpp:86-100)
__device__ __forceinline__ float ss_cdf(float r, float cs, float cd) {
    float rsd = -r * cs / cd;
    return (1.0f - 0.25f * nd_expf(rsd)) - 0.75f * nd_expf(rsd / 3.0f);
}
__device__ __forceinline__ float ss_cdf_derivative(float r, float cs, float cd) {
    float sd = cs / cd, rsd = -r * sd;
    return (sd * 0.25f) * (nd_expf(rsd) + nd_expf(rsd / 3.0f));
}
// Subsurface::diffuseScatterPDFsample: radius by three Newton steps on the CDF, uniform angle
__device__ __forceinline__ F2 ss_sample_disk(F2 s2, float cs, float cd) {
    float sample = s2.x, sdInv = cd / cs;
    float r0 = (float) (((-log((double) (1.0f - sample) * 4.0 / 3.0)) * (double) 3.0f) * (double) sdInv);
    for (int i = 0; i < 3; ++i) {
        float f = ss_cdf(r0, cs, cd) - sample, df = ss_cdf_derivative(r0, cs, cd);
        r0 = r0 - f / df;
    }
    float theta = 2.0f * ND_PI * s2.y;
    float sn, c; nd_sincosf(theta, &sn, &c);
    return f2(r0 * c, r0 * sn);
}
// a closest-hit query that only sees the records of shape `mask` (BVH::rayIntersect's `mask` argument, src/bvh.cpp:404,438),
// run to completion by one lane with a private stack
__device__ __forceinline__ bool trace_masked(const DevScene &sc, const Ray &ray, uint32_t mask, Hit &hit, uint32_t &n_nodes, uint32_t &n_prims) {
    ScratchStack stk; GlobalNodes nodes; nodes.nodes = sc.nodes;
    Trav<false> tv; tv.begin(sc, ray, n_nodes);
    while (!tv.done()) {
        if (tv.at_inner()) { if (tv.special) tv.box_step(nodes, stk, n_nodes, (sc.ordered & 1) != 0); else tv.box_step_fast(nodes, stk, n_nodes, (sc.ordered & 1) != 0); }
        else if (tv.at_leaf()) {
            float4 A, B, C; tv.prim_fetch(sc, A, B, C);
            const uint32_t fl = __float_as_uint(C.z);
            if ((fl & 0xffffffu) == mask) tv.prim_compute(A, B, C, n_prims, false);
            else tv.cur = (fl & 0x80000000u) ? ND_POP : (NORI_HIP_LEAF_BIT | ((tv.cur & 0x7fffffffu) + 1u));
        } else tv.pop_step(stk);
    }
    hit = tv.hit;
    return tv.found;
}
#define ND_PROBE_TRIES 4096      /* the reference retries without bound; here a bounce that never finds its way out ends the path, which keeps (and splats) the
                                    radiance gathered so far; counted in NoriHipStats::ss_paths_dropped */

struct Counters { uint32_t closest, shadow, nodes, prims, culled, dropped; };     // dropped: subsurface bounces that ran out of probes (ND_PROBE_TRIES)

// shade_hit in pieces, so that the wavefront pipeline can put a trace stage between drawing a subsurface probe and using its hit:
//   shade_ctx     what every piece needs about the hit: surface record, local incident direction, albedo at the hit's uv
//   shade_begin   emission + MIS weight of the previous BSDF sample, Russian roulette (:42-77); false: the path ends here
//   ss_probe_ray  Subsurface::sampleDx (src/subsurface.cpp:199-249): one probe ray; consumes next1D + next2D
//   shade_finish  BSDF sample, emitter sample, next ray (:103-164), continuing from (op, osh) = its_out
struct ShadeCtx { Surface sf; F3 wi_l, alb; };
template <bool EXT = false>
__device__ __forceinline__ void shade_ctx(const DevScene &sc, const Ray &ray, const Hit &hit, ShadeCtx &cx) {
    surface_from_hit<EXT>(sc, ray, hit, cx.sf, false);
    const NoriHipShape &shp = sc.shapes[cx.sf.shape];
    cx.wi_l = cx.sf.sh.toLocal(-ray.d);                                   // :81-87
    cx.alb = f3(shp.albedo);
    if (EXT && shp.albedo_tex) cx.alb = tex_color(sc, sc.textures[shp.albedo_tex - 1], cx.sf.uv);
}
__device__ __forceinline__ bool shade_begin(const DevScene &sc, PathState &ps, Pcg32 &rng, const ShadeCtx &cx, ShadeOut &out) {
    out.end = false; out.has_shadow = false; out.contrib = f3(0.f); out.culled = false;
    const Surface &sf = cx.sf;
    const NoriHipShape &shp = sc.shapes[sf.shape];
    if (shp.emitter >= 0) {                                               // :42-63
        const NoriHipEmitter &em = sc.emitters[shp.emitter];
        F3 wi = normalized(sf.p - ps.ray.o);
        F3 flight = (dot(sf.sh.n, wi) < 0) ? f3(em.radiance) : f3(0.f);
        float light_pdf = emitter_pdf(shape_area_pdf(sc, shp), ps.ray.o, sf.p, sf.sh.n, wi);
        float wmat = ps.last_bsdf_pdf / (light_pdf + ps.last_bsdf_pdf);
        if (ps.last_delta) wmat = 1.0f;
        if (wmat > 0.0f) ps.li = ps.li + (wmat * ps.t) * flight;
    }
    const float successProb = 0.95f;                                      // :67-77
    if (rng.next1D() > successProb) { out.end = true; return false; }
    ps.t = ps.t / successProb;
    return true;
}
__device__ __forceinline__ void ss_probe_ray(const DevScene &sc, const ShadeCtx &cx, Pcg32 &rng, Ray &probe) {
    const Surface &sf = cx.sf; const NoriHipShape &shp = sc.shapes[sf.shape];
    float thres = rng.next1D();
    F3 d = thres < 0.5f ? -sf.sh.n : thres < 0.625f ? sf.sh.t : thres < 0.75f ? -sf.sh.t : thres < 0.875f ? sf.sh.s : -sf.sh.s;
    F2 dx = ss_sample_disk(rng.next2D(), shp.ss_cs, shp.ss_cd);
    Frame fr; fr.set(-d);
    probe.d = d; probe.o = sf.p + fr.toWorld(f3(dx.x, dx.y, 0.f));
    probe.mint = -1e-3f; probe.maxt = nd_sqrt(dx.x * dx.x + dx.y * dx.y) + 1e-3f;
}
template <bool EXT = false>
__device__ __forceinline__ void shade_finish(const DevScene &sc, PathState &ps, Pcg32 &rng, const ShadeCtx &cx, F3 op, const Frame &osh, ShadeOut &out) {
    const NoriHipShape &shp = sc.shapes[cx.sf.shape];
    const F3 wi_l = cx.wi_l, alb = cx.alb;
    F2 s2 = rng.next2D(); F2 s1 = rng.next2D();                           // second argument is drawn first
    F3 wo; bool is_delta, wo_valid;
    F3 f = bsdf_sample(shp, alb, wi_l, s1, s2, wo, is_delta, wo_valid);
    ps.last_bsdf_pdf = wo_valid ? bsdf_pdf(shp, wi_l, wo) : 0.0f;
    ps.last_delta = is_delta;
    F3 next_dir = wo_valid ? normalized(osh.toWorld(wo)) : f3(0.f);

    float pick = rng.next1D();                                            // :114 (drawn for delta BSDFs too)
    if (!is_delta) {
        uint32_t n = sc.n_emitters;
        // with an environment emitter there is one more light, picked with probability 1 / lights by a draw of its own
        // (cuda/nori/pathMisIntegrator.cpp.cu:22-24, 176-184)
        const bool has_env = EXT && sc.env.texture != 0u;
        const uint32_t lights = n + (has_env ? 1u : 0u);
        bool use_env = false;
        if (has_env) { const float c = rng.next1D(); use_env = n == 0u || c > (float) (lights - 1u) / (float) lights; }
        F3 lwi, liLoc; float lpdf;
        if (use_env) {
            F2 s = rng.next2D();
            lwi = env_sample(sc, s, lpdf);
            liLoc = (lpdf > 1e-6f) ? env_eval(sc, lwi) / lpdf : f3(0.f);
            out.shadow.o = op; out.shadow.d = lwi; out.shadow.mint = ND_EPS; out.shadow.maxt = ND_INF;      // Ray3f(ref, wi, Epsilon, INFINITY), spheremap.cpp.cu:84
        } else {
            uint32_t index = (uint32_t) floorf((float) n * pick); if (index > n - 1) index = n - 1;
            const NoriHipEmitter &em = sc.emitters[index];
            F3 ln, lp;
            liLoc = emitter_sample(sc, em, op, rng.next2D(), lwi, ln, lp, out.shadow);
            lpdf = emitter_pdf(shape_area_pdf(sc, sc.shapes[em.shape]), op, lp, ln, lwi);
        }
        out.has_shadow = true;
        F3 wo_l = osh.toLocal(lwi);
        float bpdf = bsdf_pdf(shp, wi_l, wo_l);
        float wem = lpdf / (lpdf + bpdf);
        float cosS = dot(normalized(lwi), normalized(osh.n));
        if (wem > 0) out.contrib = ((((ps.t * wem) * liLoc) * bsdf_eval(shp, alb, wi_l, wo_l)) * fabsf(cosS)) * (float) lights;
        // The reference asks for visibility first and multiplies afterwards (src/pathMisIntegrator.cpp:125-143).  When the term is
        // exactly zero - the light behind the surface's horizon (eval = 0), the emitter facing away (liLoc = 0), zero throughput -
        // the answer cannot change Li (x + 0 = x in every rounding mode; a NaN term is not zero and keeps its ray), so the ray is
        // not traced.  Counted apart (NoriHipStats::shadow_rays_culled), never as a ray in Mrays/s.
        if (sc.cull_zero && out.contrib.x == 0.f && out.contrib.y == 0.f && out.contrib.z == 0.f) { out.has_shadow = false; out.culled = true; }
    }
    if (!wo_valid || is_zero(f)) { out.end = true; return; }
    ps.ray.d = next_dir; ps.ray.o = op; ps.ray.mint = 1e-3f; ps.ray.maxt = ND_INF;
    ps.t = ps.t * f;
}

// The part of one bounce between the closest-hit query and the shadow query (src/pathMisIntegrator.cpp:42-164):
// emission with the MIS weight of the previous BSDF sample, Russian roulette, BSDF sampling, emitter sampling.
// Updates ps (radiance so far, throughput, next ray, last pdf); out.contrib is what the NEE adds if the shadow ray
// is unoccluded.  out.end: the path stops after this bounce (roulette, or a zero BSDF weight — the reference keeps
// bouncing with zero throughput, which adds exact zeros).
// EXT: scenes with images (albedo textures, bump / normal maps) or subsurface shapes; the probe rays of the latter
// are traced here, by the shading lane (`ctr`, if given, counts them as closest-hit rays).
// SS (EXT only): the scene may hold subsurface shapes.  False for the single kernel of scenes that only have images: the probe
// loop and sampleDx's double arithmetic then do not weigh on the register allocation of a kernel that never runs them.
template <bool EXT = false, bool SS = EXT>
__device__ __forceinline__ void shade_hit(const DevScene &sc, PathState &ps, Pcg32 &rng, const Hit &hit, ShadeOut &out, Counters *ctr = nullptr) {
    ShadeCtx cx; shade_ctx<EXT>(sc, ps.ray, hit, cx);
    if (!shade_begin(sc, ps, rng, cx, out)) return;
    // its_out: where the path continues from.  The hit itself, except for subsurface shapes (:86-100)
    F3 op = cx.sf.p; Frame osh = cx.sf.sh;
    if (EXT && SS && sc.shapes[cx.sf.shape].bsdf == NORI_HIP_BSDF_SUBSURFACE) {
        bool found = false;
        for (int tries = 0; tries < ND_PROBE_TRIES && !found; ++tries) {
            Ray probe; ss_probe_ray(sc, cx, rng, probe);
            Hit ph; uint32_t nn = 0, np = 0;
            found = trace_masked(sc, probe, cx.sf.shape, ph, nn, np);
            if (ctr) { ctr->closest++; ctr->nodes += nn; ctr->prims += np; }
            if (found) { Surface so; surface_from_hit<EXT>(sc, probe, ph, so, false); op = so.p; osh = so.sh; }
        }
        if (!found) { out.end = true; if (ctr) ctr->dropped++; return; }
    }
    shade_finish<EXT>(sc, ps, rng, cx, op, osh, out);
}

// A bounce that the wavefront pipeline left between its two halves: the hit `hit0` of ray `ray0` was shaded up to the roulette and
// the probe `probe` (attempt number `tries`, 0-based) was traced with result `ph` (pos = 0xffffffff: it missed).  Finishes the
// bounce in one go, with further probes traced by this lane if that one missed (the tail kernel's way).
template <bool EXT = true>
__device__ __forceinline__ void shade_resume(const DevScene &sc, PathState &ps, Pcg32 &rng, const Ray &ray0, const Hit &hit0, Ray probe, Hit ph, int tries,
                                             ShadeOut &out, Counters *ctr) {
    out.end = false; out.has_shadow = false; out.contrib = f3(0.f); out.culled = false;
    ShadeCtx cx; shade_ctx<EXT>(sc, ray0, hit0, cx);
    bool found = ph.pos != 0xffffffffu;
    for (++tries; tries < ND_PROBE_TRIES && !found; ++tries) {
        ss_probe_ray(sc, cx, rng, probe);
        uint32_t nn = 0, np = 0;
        found = trace_masked(sc, probe, cx.sf.shape, ph, nn, np);
        if (ctr) { ctr->closest++; ctr->nodes += nn; ctr->prims += np; }
    }
    if (!found) { out.end = true; if (ctr) ctr->dropped++; return; }
    Surface so; surface_from_hit<EXT>(sc, probe, ph, so, false);
    shade_finish<EXT>(sc, ps, rng, cx, so.p, so.sh, out);
}

// One iteration of the reference's bounce loop, run to completion by one lane.  Returns false when the path ended.
template <bool EXT = false, bool SS = EXT, typename StackT, typename NodesT>
__device__ __forceinline__ bool path_bounce(const DevScene &sc, const NodesT &nodes, PathState &ps, Pcg32 &rng, StackT &stk, Counters &ctr) {
    Hit hit;
    ctr.closest++;
    if (!trace<false>(sc, nodes, ps.ray, hit, stk, ctr.nodes, ctr.prims)) { ps.li = ps.li + env_miss<EXT>(sc, ps.ray.d, ps.t, ps.last_bsdf_pdf, ps.last_delta); return false; }
    ShadeOut so;
    shade_hit<EXT, SS>(sc, ps, rng, hit, so, &ctr);
    if (so.culled) ctr.culled++;
    if (so.has_shadow) {
        Hit sh_hit;
        ctr.shadow++;
        if (!trace<true>(sc, nodes, so.shadow, sh_hit, stk, ctr.nodes, ctr.prims)) ps.li = ps.li + so.contrib;
    }
    return !so.end;
}
// One iteration of PathMatsIntegrator::Li's loop (src/pathMatsIntegrator.cpp:32-73): emission without MIS, roulette with
// the throughput's largest component, BSDF sampling; no emitter sampling, no shadow ray.  The next ray keeps the camera
// ray's maxt and starts at mint = 1e-4 (which switches the adaptive epsilon of the traversal on), as the reference does.
template <bool EXT = false, typename StackT, typename NodesT>
__device__ __forceinline__ bool path_bounce_mats(const DevScene &sc, const NodesT &nodes, PathState &ps, Pcg32 &rng, StackT &stk, Counters &ctr) {
    Hit hit;
    ctr.closest++;
    if (!trace<false>(sc, nodes, ps.ray, hit, stk, ctr.nodes, ctr.prims)) { ps.li = ps.li + env_miss<EXT>(sc, ps.ray.d, ps.t, 1.0f, true); return false; }   // (no emitter sampling: weight 1)
    Surface sf;
    surface_from_hit<EXT>(sc, ps.ray, hit, sf, false);
    const NoriHipShape &shp = sc.shapes[sf.shape];
    if (shp.emitter >= 0) {
        F3 wi = normalized(sf.p - ps.ray.o);
        if (dot(sf.sh.n, wi) < 0) ps.li = ps.li + ps.t * f3(sc.emitters[shp.emitter].radiance);
    }
    const float m = fmaxf(ps.t.x, fmaxf(ps.t.y, ps.t.z));
    const float successProb = m < 0.99f ? m : 0.99f;
    if (rng.next1D() > successProb) return false;
    ps.t = ps.t / successProb;
    F3 wi_l = sf.sh.toLocal(-ps.ray.d);
    F3 alb = f3(shp.albedo);
    if (EXT && shp.albedo_tex) alb = tex_color(sc, sc.textures[shp.albedo_tex - 1], sf.uv);
    F2 s2 = rng.next2D(); F2 s1 = rng.next2D();
    F3 wo; bool is_delta, wo_valid;
    F3 f = bsdf_sample(shp, alb, wi_l, s1, s2, wo, is_delta, wo_valid);
    if (!wo_valid || is_zero(f)) return false;
    ps.ray.d = normalized(sf.sh.toWorld(wo)); ps.ray.o = sf.p; ps.ray.mint = 1e-4f;
    ps.t = ps.t * f;
    return true;
}

// DirectMisIntegrator::Li (src/directMisIntegrator.cpp:19-141) as one "bounce" that always ends the path: every emitter
// sampled once, then one BSDF sample, both MIS-weighted; quirks kept as written (see the oracle's direct_mis_li).
template <bool EXT = false, typename StackT, typename NodesT>
__device__ __forceinline__ bool path_direct_mis(const DevScene &sc, const NodesT &nodes, PathState &ps, Pcg32 &rng, StackT &stk, Counters &ctr) {
    Hit hit;
    ctr.closest++;
    if (!trace<false>(sc, nodes, ps.ray, hit, stk, ctr.nodes, ctr.prims)) return false;
    Surface sf;
    surface_from_hit<EXT>(sc, ps.ray, hit, sf, false);
    const NoriHipShape &shp = sc.shapes[sf.shape];
    if (shp.emitter >= 0) {
        F3 wi = normalized(sf.p - ps.ray.o);
        if (dot(sf.sh.n, wi) < 0) ps.li = ps.li + f3(sc.emitters[shp.emitter].radiance);
    }
    const F3 wi_l = sf.sh.toLocal(-ps.ray.d);
    F3 alb = f3(shp.albedo);
    if (EXT && shp.albedo_tex) alb = tex_color(sc, sc.textures[shp.albedo_tex - 1], sf.uv);
    for (uint32_t e = 0; e < sc.n_emitters; ++e) {
        const NoriHipEmitter &em = sc.emitters[e];
        F3 lwi, ln, lp; Ray shadow;
        F3 liLoc = emitter_sample(sc, em, sf.p, rng.next2D(), lwi, ln, lp, shadow);
        Hit sh; ctr.shadow++;
        if (trace<true>(sc, nodes, shadow, sh, stk, ctr.nodes, ctr.prims)) continue;
        F3 wo_l = sf.sh.toLocal(lwi);
        F3 f = bsdf_eval(shp, alb, wi_l, wo_l);
        float bpdf = bsdf_pdf(shp, wi_l, wo_l);
        float lpdf = emitter_pdf(shape_area_pdf(sc, sc.shapes[em.shape]), sf.p, lp, ln, lwi);
        float wem = lpdf / (lpdf + bpdf);
        float cosS = dot(normalized(lwi), normalized(sf.sh.n));
        if (wem > 0) ps.li = ps.li + ((wem * liLoc) * f) * cosS;
    }
    F2 s2 = rng.next2D(); F2 s1 = rng.next2D();
    F3 wo; bool is_delta, wo_valid;
    F3 f = bsdf_sample(shp, alb, wi_l, s1, s2, wo, is_delta, wo_valid);
    if (!wo_valid) return false;
    Ray refl; refl.o = sf.p; refl.d = sf.sh.toWorld(wo); refl.mint = 1e-4f; refl.maxt = ND_INF;
    Hit h2; ctr.closest++;
    if (trace<false>(sc, nodes, refl, h2, stk, ctr.nodes, ctr.prims)) {
        Surface s2f; surface_from_hit<EXT>(sc, refl, h2, s2f, false);
        const NoriHipShape &shp2 = sc.shapes[s2f.shape];
        if (shp2.emitter >= 0) {
            F3 wi = normalized(s2f.p - refl.o);
            F3 flight = (dot(s2f.sh.n, wi) < 0) ? f3(sc.emitters[shp2.emitter].radiance) : f3(0.f);
            float bpdf = bsdf_pdf(shp, wi_l, wo);
            float lpdf = emitter_pdf(shape_area_pdf(sc, shp2), refl.o, s2f.p, s2f.sh.n, wi);
            float wmat = bpdf / (lpdf + bpdf);
            if (wmat > 0) ps.li = ps.li + (wmat * f) * flight;
        }
    }
    return false;
}

template <bool EXT = false, typename StackT>
__device__ __forceinline__ bool path_bounce(const DevScene &sc, PathState &ps, Pcg32 &rng, StackT &stk, Counters &ctr) {
    GlobalNodes g; g.nodes = sc.nodes;
    return path_bounce<EXT>(sc, g, ps, rng, stk, ctr);
}

} // namespace nd
