/* glibc 2.35's logf and expf (sysdeps/ieee754/flt-32/e_logf.c, e_expf.c, after ARM's optimized routines) restated, and compared with the
 * libm of this machine over every float in [lo, hi) of the bit patterns.  usage: check <log|exp> <first bits, hex> <last bits, hex>
 * The multiply-adds are written as fma(): glibc picks, on CPUs that have the instruction, a variant of these functions compiled with
 * -mfma, in which the compiler contracted exactly these (without them: logf equal on all 2^31 non-negative floats, expf different for 2 of
 * the 2^32 floats; with them: no difference at all).  csrc/nori_device.h (nd_logf, nd_expf) is this code.
 * build: gcc -O2 -mfma -ffp-contract=off -o /tmp/lm/check scratch/glibc_logf_expf_check.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

static const struct { double invc, logc; } LT[16] = {
  { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 }, { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 },
  { 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2 },  { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 },
  { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 }, { 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3 },
  { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 }, { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 },
  { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 }, { 0x1p+0, 0x0p+0 },
  { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 },  { 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4 },
  { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 },  { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3 },
  { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 },  { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 } };
static const double LN2 = 0x1.62e42fefa39efp-1, LA[3] = { -0x1.00ea348b88334p-2, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2 };
static float r_logf(float x) {
    uint32_t ix = asuint(x);
    if (ix == 0x3f800000) return 0;
    if (ix - 0x00800000 >= 0x7f800000 - 0x00800000) {
        if (ix * 2 == 0) return -INFINITY;
        if (ix == 0x7f800000) return x;
        if ((ix & 0x80000000) || ix * 2 >= 0xff000000) return NAN;
        ix = asuint(x * 0x1p23f); ix -= 23 << 23;
    }
    uint32_t tmp = ix - 0x3f330000;
    int i = (tmp >> (23 - 4)) % 16;
    int k = (int32_t) tmp >> 23;
    uint32_t iz = ix - (tmp & 0x1ff << 23);
    double invc = LT[i].invc, logc = LT[i].logc, z = (double) asfloat(iz);
    double r = fma(z, invc, -1.0), y0 = fma((double) k, LN2, logc);
    double r2 = r * r, y = fma(LA[1], r, LA[2]);
    y = fma(LA[0], r2, y);
    y = fma(y, r2, y0 + r);
    return (float) y;
}
static uint64_t ET[32];
static const double E_SHIFT = 0x1.8p+52, E_INVLN2N = 0x1.71547652b82fep+0 * 32,
    EC[3] = { 0x1.c6af84b912394p-5 / 32 / 32 / 32, 0x1.ebfce50fac4f3p-3 / 32 / 32, 0x1.62e42ff0c52d6p-1 / 32 };
static float r_expf(float x) {
    double xd = (double) x;
    uint32_t abstop = (asuint(x) >> 20) & 0x7ff;
    if (abstop >= (asuint(88.0f) >> 20)) {
        if (asuint(x) == asuint(-INFINITY)) return 0.0f;
        if (abstop >= (asuint(INFINITY) >> 20)) return x + x;
        if (x > 0x1.62e42ep6f) return INFINITY;
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    double z = E_INVLN2N * xd;
    double kd = z + E_SHIFT; uint64_t ki = asuint64(kd); kd -= E_SHIFT;
    double r = fma(E_INVLN2N, xd, -kd);        /* z - kd with z = InvLn2N * xd contracted: the one that decides the 2 floats */
    uint64_t t = ET[ki % 32]; t += ki << (52 - 5);
    double s = asdouble(t);
    z = fma(EC[0], r, EC[1]);
    double r2 = r * r, y = fma(EC[2], r, 1.0);
    y = fma(z, r2, y); y = y * s;
    return (float) y;
}
int main(int argc, char **argv) {
    /* 2^(i/32), correctly rounded, minus i << 47: exp2 of libm is correctly rounded here for these arguments (checked against the table's
       published first entries) */
    for (int i = 0; i < 32; ++i) ET[i] = asuint64(exp2((double) i / 32.0)) - ((uint64_t) i << 47);
    int is_log = !strcmp(argv[1], "log");
    uint64_t lo = strtoull(argv[2], 0, 16), hi = strtoull(argv[3], 0, 16), bad = 0, n = 0;
    for (uint64_t b = lo; b < hi; ++b) {
        float x = asfloat((uint32_t) b), a = is_log ? r_logf(x) : r_expf(x), c = is_log ? logf(x) : expf(x);
        if (asuint(a) != asuint(c) && !(a != a && c != c)) { if (bad < 5) printf("  x=%a (%08x): restated %a libm %a\n", x, (uint32_t) b, a, c); ++bad; }
        ++n;
    }
    printf("%s [%llx, %llx): %llu values, %llu mismatches\n", argv[1], (unsigned long long) lo, (unsigned long long) hi, (unsigned long long) n, (unsigned long long) bad);
    return 0;
}
