/* Host-side check of the libm restatements in csrc/nori_device.h (nd_acosf, nd_atanf, nd_atan2f) against the container's glibc:
   gcc -O2 -ffp-contract=off scratch/glibc_libm_check.c -lm && ./a.out   -> 0 differences.  The sincosf check is scratch/glibc_sincosf_check.c. */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static inline int32_t fw(float x){ int32_t u; memcpy(&u,&x,4); return u; }
static inline float wf(int32_t u){ float x; memcpy(&x,&u,4); return x; }
#define H(x) wf((int32_t)(x))
static float my_acosf(float x) {
  const float one = 1.0f, pi = H(0x40490fda), pio2_hi = H(0x3fc90fda), pio2_lo = H(0x33a22168),
    pS0 = H(0x3e2aaaab), pS1 = H(0xbea6b090), pS2 = H(0x3e4e0aa8), pS3 = H(0xbd241146), pS4 = H(0x3a4f7f04), pS5 = H(0x3811ef08),
    qS1 = H(0xc019d139), qS2 = H(0x4001572d), qS3 = H(0xbf303361), qS4 = H(0x3d9dc62e);
  float z,p,q,r,w,s,c,df; int32_t hx = fw(x), ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) { if (hx > 0) return 0.0f; else return pi + 2.0f * pio2_lo; }
  else if (ix > 0x3f800000) return (x-x)/(x-x);
  if (ix < 0x3f000000) {
    if (ix <= 0x23000000) return pio2_hi + pio2_lo;
    z = x*x; p = z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q = one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); r = p/q;
    return pio2_hi - (x - (pio2_lo - x*r));
  } else if (hx < 0) {
    z = (one+x)*0.5f; p = z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q = one+z*(qS1+z*(qS2+z*(qS3+z*qS4)));
    s = sqrtf(z); r = p/q; w = r*s - pio2_lo; return pi - 2.0f*(s+w);
  } else {
    z = (one-x)*0.5f; s = sqrtf(z); df = wf(fw(s) & 0xfffff000); c = (z-df*df)/(s+df);
    p = z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q = one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); r = p/q; w = r*s+c;
    return 2.0f*(df+w);
  }
}
static float my_atanf(float x) {
  const float atanhi[4] = {H(0x3eed6338), H(0x3f490fda), H(0x3f7b985e), H(0x3fc90fda)}, atanlo[4] = {H(0x31ac3769), H(0x33222168), H(0x33140fb4), H(0x33a22168)};
  const float aT[11] = {H(0x3eaaaaab), H(0xbe4ccccd), H(0x3e124925), H(0xbde38e38), H(0x3dba2e6e), H(0xbd9d8795), H(0x3d886b35), H(0xbd6ef16b), H(0x3d4bda59), H(0xbd15a221), H(0x3c8569d7)};
  const float one = 1.0f;
  float w,s1,s2,z; int32_t hx = fw(x), ix = hx & 0x7fffffff, id;
  if (ix >= 0x4c000000) { if (ix > 0x7f800000) return x+x; if (hx > 0) return atanhi[3]+atanlo[3]; else return -atanhi[3]-atanlo[3]; }
  if (ix < 0x3ee00000) { if (ix < 0x31000000) return x; id = -1; }
  else { x = fabsf(x);
    if (ix < 0x3f980000) { if (ix < 0x3f300000) { id = 0; x = (2.0f*x-one)/(2.0f+x); } else { id = 1; x = (x-one)/(x+one); } }
    else { if (ix < 0x401c0000) { id = 2; x = (x-1.5f)/(one+1.5f*x); } else { id = 3; x = -1.0f/x; } } }
  z = x*x; w = z*z;
  s1 = z*(aT[0]+w*(aT[2]+w*(aT[4]+w*(aT[6]+w*(aT[8]+w*aT[10])))));
  s2 = w*(aT[1]+w*(aT[3]+w*(aT[5]+w*(aT[7]+w*aT[9]))));
  if (id < 0) return x - x*(s1+s2);
  z = atanhi[id] - ((x*(s1+s2) - atanlo[id]) - x);
  return (hx < 0) ? -z : z;
}
static float my_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_2 = H(0x3fc90fdb), pi = H(0x40490fdb), pi_lo = H(0xb3bbbd2e);
  float z; int32_t k,m,hx = fw(x),hy = fw(y),ix = hx&0x7fffffff,iy = hy&0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x+y;
  if (hx == 0x3f800000) return my_atanf(y);
  m = ((hy>>31)&1)|((hx>>30)&2);
  if (iy == 0) { switch (m) { case 0: case 1: return y; case 2: return pi+tiny; case 3: return -pi-tiny; } }
  if (ix == 0) return (hy < 0) ? -pi_o_2-tiny : pi_o_2+tiny;
  if (ix == 0x7f800000 || iy == 0x7f800000) return atan2f(y, x);   /* infinities: not needed here */
  k = (iy-ix)>>23;
  if (k > 60) z = pi_o_2 + 0.5f*pi_lo; else if (hx < 0 && k < -60) z = 0.0f; else z = my_atanf(fabsf(y/x));
  switch (m) { case 0: return z; case 1: return wf(fw(z) ^ 0x80000000); case 2: return pi-(z-pi_lo); default: return (z-pi_lo)-pi; }
}
int main() {
  uint64_t bad = 0, n = 0;
  for (uint32_t u = 0; u <= 0x3f800010u; u += 3) { for (int sgn = 0; sgn < 2; ++sgn) { float x = wf((int32_t)(u | (sgn ? 0x80000000u : 0u))); float a = my_acosf(x), b = acosf(x); if (fw(a) != fw(b) && !(isnan(a) && isnan(b))) { if (bad < 5) printf("acosf(%a): %a vs %a\n", x, a, b); bad++; } n++; } }
  printf("acosf: n=%llu bad=%llu\n", (unsigned long long) n, (unsigned long long) bad);
  bad = 0; n = 0; uint64_t st = 88172645463325252ull;
  for (long i = 0; i < 300000000; ++i) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    float y = ((int32_t)(st & 0xffffffff)) * (1.0f/2147483648.0f), x = ((int32_t)(st >> 32)) * (1.0f/2147483648.0f);
    if ((i & 7) == 0) { y *= 1e-3f; } if ((i & 15) == 1) { x *= 1e-4f; }
    float a = my_atan2f(y, x), b = atan2f(y, x);
    if (fw(a) != fw(b)) { if (bad < 5) printf("atan2f(%a,%a): %a vs %a\n", y, x, a, b); bad++; } n++;
  }
  printf("atan2f: n=%llu bad=%llu\n", (unsigned long long) n, (unsigned long long) bad);
  return 0;
}
