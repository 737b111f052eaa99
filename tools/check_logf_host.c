/* Host check of the restated glibc logf arithmetic (the form the frustum kernel uses) against the C
 * library for every positive normal float:
 *   gcc -O2 -fopenmp tools/check_logf_host.c -lm && ./a.out              (separate multiply and add)
 *   gcc -O2 -fopenmp -DUSE_FMA -mfma tools/check_logf_host.c -lm         (fused)
 * Both print 0 mismatches on glibc 2.35. */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static const struct { double invc, logc; } T[16] = {
  { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 },
  { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 },
  { 0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2 },
  { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 },
  { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 },
  { 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3 },
  { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 },
  { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 },
  { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 },
  { 0x1p+0, 0x0p+0 },
  { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 },
  { 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4 },
  { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 },
  { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3 },
  { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 },
  { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 },
};
static const double Ln2 = 0x1.62e42fefa39efp-1;
static const double A[3] = { -0x1.00ea348b88334p-2, 0x1.5575b0be00b6ap-2, -0x1.ffffef20a4123p-2 };
#ifdef USE_FMA
#define MADD(a,b,c) __builtin_fma((a),(b),(c))
#else
#define MADD(a,b,c) ((a)*(b)+(c))
#endif
static float my_logf(float x) {
  uint32_t ix; memcpy(&ix, &x, 4);
  if (ix == 0x3f800000) return 0;
  uint32_t tmp = ix - 0x3f330000u;
  int i = (tmp >> 19) % 16;
  int k = (int32_t)tmp >> 23;
  uint32_t iz = ix - (tmp & 0xff800000u);
  float zf; memcpy(&zf, &iz, 4);
  double z = zf;
  double r = MADD(z, T[i].invc, -1.0);
  double y0 = MADD((double)k, Ln2, T[i].logc);
  double r2 = r * r;
  double y = MADD(A[1], r, A[2]);
  y = MADD(A[0], r2, y);
  y = MADD(y, r2, y0 + r);
  return (float)y;
}
int main() {
  long bad = 0; uint32_t first = 0;
  #pragma omp parallel for reduction(+:bad)
  for (uint32_t u = 0x00800000u; u < 0x7f800000u; u++) { float x; memcpy(&x, &u, 4); float a = logf(x), b = my_logf(x);
    if (memcmp(&a,&b,4)) { bad++; if (!first) first = u; } }
  printf("log mismatches %ld (first %08x)\n", bad, first);
  return 0; }
