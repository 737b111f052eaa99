/* Host check of the restated glibc sinf / cosf arithmetic (the form the descriptor kernel uses) against
 * the C library for every float in [2^-15, 2*pi]:
 *   gcc -O2 -fopenmp tools/check_sincos_host.c -lm && ./a.out            (separate multiply and add)
 *   gcc -O2 -fopenmp -DUSE_FMA -mfma tools/check_sincos_host.c -lm       (fused)
 * Both print 0 mismatches on glibc 2.35. */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>
typedef struct { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } sincos_t;
static const sincos_t T[2] = {
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
#ifdef USE_FMA
#define MADD(a,b,c) __builtin_fma((a),(b),(c))
#else
#define MADD(a,b,c) ((a)*(b)+(c))
#endif
static inline float sinf_poly(double x, double x2, const sincos_t* p, int n) {
  double x3,x4,x6,x7,s,c,c1,c2,s1;
  if ((n & 1) == 0) { x3 = x*x2; s1 = MADD(x2, p->s3, p->s2); x7 = x3*x2; s = MADD(x3, p->s1, x); return (float)MADD(x7, s1, s); }
  else { x4 = x2*x2; c2 = MADD(x2, p->c4, p->c3); c1 = MADD(x2, p->c1, p->c0); x6 = x4*x2; c = MADD(x4, p->c2, c1); return (float)MADD(x6, c2, c); }
}
static inline double reduce_fast(double x, const sincos_t* p, int* np) {
  double r = x * p->hpi_inv; int n = ((int32_t)r + 0x800000) >> 24; *np = n;
#ifdef USE_FMA
  return __builtin_fma(-(double)n, p->hpi, x);
#else
  return x - n * p->hpi;
#endif
}
static float my_sinf(float y) { double x = y; const sincos_t* p = &T[0]; int n; if (fabsf(y) < 0x1.921FB6p-1f) { double x2 = x*x; return sinf_poly(x, x2, p, 0);} x = reduce_fast(x, p, &n); double s = p->sign[n&3]; if (n & 2) p = &T[1]; return sinf_poly(x*s, x*x, p, n); }
static float my_cosf(float y) { double x = y; const sincos_t* p = &T[0]; int n; if (fabsf(y) < 0x1.921FB6p-1f) { double x2 = x*x; return sinf_poly(x, x2, p, 1);} x = reduce_fast(x, p, &n); double s = p->sign[n&3]; if (n & 2) p = &T[1]; return sinf_poly(x*s, x*x, p, n ^ 1); }
int main() {
  uint32_t hi; float lim = 6.2831860f; memcpy(&hi, &lim, 4);
  long bad_s = 0, bad_c = 0; uint32_t first_s = 0, first_c = 0;
  #pragma omp parallel for reduction(+:bad_s,bad_c)
  for (uint32_t u = 0x38000000u; u <= hi; u++) { float x; memcpy(&x, &u, 4); float a = sinf(x), b = my_sinf(x), c = cosf(x), d = my_cosf(x);
    if (memcmp(&a,&b,4)) { bad_s++; if (!first_s) first_s = u; } if (memcmp(&c,&d,4)) { bad_c++; if (!first_c) first_c = u; } }
  printf("sin mismatches %ld (first %08x) cos mismatches %ld (first %08x)\n", bad_s, first_s, bad_c, first_c);
  return 0; }
