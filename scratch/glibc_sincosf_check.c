#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef struct { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } sincos_t;
static const sincos_t T[2] = {
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
#ifdef USE_FMA
#define MA(a,b,c) fma((b),(c),(a))
#else
#define MA(a,b,c) ((a)+(b)*(c))
#endif
static inline uint32_t top12(float x){ uint32_t u; memcpy(&u,&x,4); return (u>>20)&0x7ff; }
static void my_sincosf(float y, float *sp, float *cp) {
  double x = y, s; int n = 0; const sincos_t *p = &T[0];
  if (top12(y) < top12(0x1.921FB6p-1f)) {
    double x2 = x*x;
    if (top12(y) < top12(0x1p-12f)) { *sp = y; *cp = 1.0f; return; }
    s = x;
    n = 0;
    x = s;
    goto poly;
  } else if (top12(y) < top12(120.0f)) {
    double r = x * p->hpi_inv;
    n = ((int32_t) r + 0x800000) >> 24;
    x = x - n * p->hpi;
    s = p->sign[n & 3];
    if (n & 2) p = &T[1];
    x = x * s;
  } else { *sp = sinf(y); *cp = cosf(y); return; }
poly: {
    double x2 = x*x, x3, x4, x5, x6, ss, c, c1, c2, s1;
    x4 = x2*x2; x3 = x2*x; c2 = MA(p->c3, x2, p->c4); s1 = MA(p->s2, x2, p->s3);
    c1 = MA(p->c0, x2, p->c1); x5 = x3*x2; x6 = x4*x2;
    ss = MA(x, x3, p->s1); c = MA(c1, x4, p->c2);
    float sv = (float) MA(ss, x5, s1), cv = (float) MA(c, x6, c2);
    if (n & 1) { *sp = cv; *cp = sv; } else { *sp = sv; *cp = cv; }
  }
}
int main() {
  uint64_t bad_s = 0, bad_c = 0, bad_sc = 0, n = 0;
  // every float in [0, 2*pi] stepping 7 ulps, plus negatives
  float lim = 6.2831855f; uint32_t ul; memcpy(&ul, &lim, 4);
  for (uint32_t u = 0; u <= ul + 64; u += 3) {
    float y; memcpy(&y, &u, 4);
    float s, c; my_sincosf(y, &s, &c);
    float s0 = sinf(y), c0 = cosf(y), s1, c1; sincosf(y, &s1, &c1);
    if (memcmp(&s,&s0,4)) bad_s++; if (memcmp(&c,&c0,4)) bad_c++;
    if (memcmp(&s1,&s0,4) || memcmp(&c1,&c0,4)) bad_sc++;
    n++;
  }
  printf("n=%llu bad_sin=%llu bad_cos=%llu libm sincosf!=sinf/cosf: %llu\n", (unsigned long long) n, (unsigned long long) bad_s, (unsigned long long) bad_c, (unsigned long long) bad_sc);
  return 0;
}
