// exhaustive-ish check: Markstein two-step division with a correctly rounded reciprocal == IEEE division
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
static inline float div2(float v, float d, float y) {
  float q0 = v * y;
  float r0 = fmaf(-d, q0, v);
  float q1 = fmaf(r0, y, q0);
  float r1 = fmaf(-d, q1, v);
  return fmaf(r1, y, q1);
}
static inline float div1(float v, float d, float y) {
  float q0 = v * y;
  float r0 = fmaf(-d, q0, v);
  return fmaf(r0, y, q0);
}
static inline uint32_t rng(uint64_t* s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return (uint32_t)(*s >> 16); }
int main() {
  long bad2 = 0, bad1 = 0, n = 0;
#pragma omp parallel for reduction(+ : bad2, bad1, n)
  for (int t = 0; t < 64; ++t) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (t + 1);
    for (long i = 0; i < 40000000L; ++i) {
      uint32_t a = rng(&s), b = rng(&s);
      // v: exponent in [2^-80, 2^60], any mantissa, any sign; d: [2^-40, 2^64], |v| <= d * 1.0 not enforced (wider than needed)
      uint32_t ev = 127 - 80 + (a >> 23) % 141, ed = 127 - 40 + (b >> 23) % 105;
      uint32_t vb = (a & 0x807fffffu) | (ev << 23), db = (b & 0x007fffffu) | (ed << 23);
      if (i & 1) { vb = (vb & ~0x7fffffu) | ((a >> 3) & 0x7) | ((a & 0x40) ? 0x7ffff8 : 0); }       // mantissas near 1.0 / 2.0
      if (i & 2) { db = (db & ~0x7fffffu) | ((b >> 3) & 0x7) | ((b & 0x40) ? 0x7ffff8 : 0); }
      float v, d;
      memcpy(&v, &vb, 4);
      memcpy(&d, &db, 4);
      float y = 1.0f / d;
      float ref = v / d;
      float g2 = div2(v, d, y), g1 = div1(v, d, y);
      if (isinf(ref) || fabsf(ref) < 1e-37f) continue;
      n++;
      if (memcmp(&g2, &ref, 4)) bad2++;
      if (memcmp(&g1, &ref, 4)) bad1++;
    }
  }
  printf("tested %ld  two-step mismatches %ld  one-step mismatches %ld\n", n, bad2, bad1);
  return 0;
}
