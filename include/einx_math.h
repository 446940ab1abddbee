/* einx_math.h -- the numeric contract shared by the HIP kernels (csrc/) and the CPU
 * oracle (oracle/).  Every transcendental on the hot path is written here in terms of
 * IEEE-754 exact primitives (fmaf, +, *, /, sqrtf, rintf, bit moves), so the same source
 * compiled by hipcc for gfx950 and by gcc for x86-64 (+FMA) gives the SAME BITS.  That is
 * what lets the GPU path be compared bit-for-bit with the oracle, while each function stays
 * within a few ulp of the libm/ATen function the reference calls (tolerance 1e-4 there).
 *
 * Build rules for both sides: -ffp-contract=off, no fast-math, f32 denormals preserved.
 */
#ifndef EINX_MATH_H
#define EINX_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define EINX_HD __host__ __device__ __forceinline__
#else
#define EINX_HD static inline
#endif

EINX_HD float einx_u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
EINX_HD uint32_t einx_f2u(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

/* exp(x), <= ~1 ulp.  x = n ln2 + r, |r| <= ln2/2, degree-7 Horner, two-step power-of-two
 * scaling so results in the subnormal range round once.  Replaces torch.exp / the exp inside
 * softmax (reference: core/modules/utils/detector_util.py:36-39, matchers/MNN.py:97,
 * matchers/lightglue.py:369-371). */
EINX_HD float einx_expf(float x) {
  if (x > 88.72283f) return einx_u2f(0x7f800000u);
  if (x < -103.97208f) return 0.0f;
  const float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693145751953125f, x);           /* ln2 hi (12 bits) */
  r = fmaf(n, -1.42860682030941723212e-6f, r);         /* ln2 lo */
  float p = 1.98412698412698413e-4f;                   /* 1/5040 */
  p = fmaf(p, r, 1.38888888888888894e-3f);             /* 1/720 */
  p = fmaf(p, r, 8.33333333333333322e-3f);             /* 1/120 */
  p = fmaf(p, r, 4.16666666666666644e-2f);             /* 1/24 */
  p = fmaf(p, r, 1.66666666666666657e-1f);             /* 1/6 */
  p = fmaf(p, r, 0.5f);
  p = fmaf(p, r, 1.0f);
  p = fmaf(p, r, 1.0f);
  const int ni = (int)n;
  const int n1 = ni / 2, n2 = ni - n1;
  const float s1 = einx_u2f((uint32_t)(n1 + 127) << 23);
  const float s2 = einx_u2f((uint32_t)(n2 + 127) << 23);
  return (p * s1) * s2;
}

/* natural log for x > 0 (normal or subnormal), ~1-2 ulp.  Used by log-softmax / logsigmoid
 * (reference: matchers/MNN.py:97, matchers/lightglue.py:365-377). */
EINX_HD float einx_logf(float x) {
  if (x == 0.0f) return -einx_u2f(0x7f800000u);
  uint32_t u = einx_f2u(x);
  int e = 0;
  if (u < 0x00800000u) { /* subnormal: scale up by 2^23 */
    x = x * 8388608.0f;
    u = einx_f2u(x);
    e = -23;
  }
  e += (int)(u >> 23) - 127;
  uint32_t mu = (u & 0x007fffffu) | 0x3f800000u;
  float m = einx_u2f(mu); /* [1,2) */
  if (m > 1.41421356237f) {
    m = m * 0.5f;
    e += 1;
  }
  const float f = m - 1.0f; /* [-0.2929, 0.4142] */
  const float s = f / (2.0f + f);
  const float z = s * s;
  /* log(1+f) = 2 atanh(s) = 2s (1 + z/3 + z^2/5 + z^3/7 + z^4/9 + z^5/11) */
  float q = 0.0909090909090909f;
  q = fmaf(q, z, 0.111111111111111f);
  q = fmaf(q, z, 0.142857142857143f);
  q = fmaf(q, z, 0.2f);
  q = fmaf(q, z, 0.333333333333333f);
  q = q * z;
  const float two_s = 2.0f * s;
  const float lg = fmaf(two_s, q, two_s);
  const float fe = (float)e;
  return fmaf(fe, 0.693145751953125f, fmaf(fe, 1.42860682030941723212e-6f, lg));
}

/* sin and cos with Cody-Waite reduction by pi/2 (3 constants), |x| up to ~1e4 keeps ~1e-6
 * absolute error.  Reference: matchers/lightglue.py:171 (torch.cos / torch.sin of Wr x). */
EINX_HD void einx_sincosf(float x, float* sn, float* cs) {
  const float k = rintf(x * 0.636619772367581343f); /* 2/pi */
  float r = fmaf(k, -1.5703125f, x);
  r = fmaf(k, -4.83751296997070312e-4f, r);
  r = fmaf(k, -7.54978995489188e-8f, r);
  const float z = r * r;
  /* sin(r), |r| <= pi/4 */
  float ps = -1.9515295891e-4f;
  ps = fmaf(ps, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  const float s = fmaf(ps * z, r, r);
  /* cos(r) */
  float pc = 2.443315711809948e-5f;
  pc = fmaf(pc, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  const float c = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
  const int q = ((int)k) & 3;
  float so, co;
  if (q == 0) { so = s; co = c; }
  else if (q == 1) { so = c; co = -s; }
  else if (q == 2) { so = -s; co = -c; }
  else { so = -c; co = s; }
  *sn = so;
  *cs = co;
}

/* erf(x), absolute error < 5e-7.  |x|<0.84375: Maclaurin series (9 terms); else 1-erfc via
 * exp(-x^2) * rational (Numerical-Recipes style Chebyshev fit evaluated with fmaf).
 * Reference: nn.GELU() exact form in matchers/lightglue.py:254,297. */
EINX_HD float einx_erff(float x) {
  const float ax = fabsf(x);
  float r;
  if (ax < 0.84375f) {
    const float z = x * x;
    /* plain Maclaurin series of erf, 9 terms: 2/sqrt(pi) * sum (-1)^n x^(2n+1)/(n!(2n+1)) */
    float t = 1.0f / 685440.0f;           /* 1/(8!*17) */
    t = fmaf(t, -z, 1.0f / 75600.0f);     /* 1/(7!*15) */
    t = fmaf(t, -z, 1.0f / 9360.0f);      /* 1/(6!*13) */
    t = fmaf(t, -z, 1.0f / 1320.0f);      /* 1/(5!*11) */
    t = fmaf(t, -z, 1.0f / 216.0f);       /* 1/(4!*9)  */
    t = fmaf(t, -z, 1.0f / 42.0f);        /* 1/(3!*7)  */
    t = fmaf(t, -z, 0.1f);                /* 1/(2!*5)  */
    t = fmaf(t, -z, 1.0f / 3.0f);         /* 1/(1!*3)  */
    t = fmaf(t, -z, 1.0f);
    r = 1.12837916709551257f * x * t;
    return r;
  }
  if (ax > 4.0f) return x > 0.0f ? 1.0f : -1.0f;
  /* erfc(ax) = t * exp(-ax^2 + poly(t)), t = 1/(1+ax/2)  (W. J. Cody-style Chebyshev fit) */
  const float t = 1.0f / fmaf(0.5f, ax, 1.0f);
  float q = 0.17087277f;
  q = fmaf(q, t, -0.82215223f);
  q = fmaf(q, t, 1.48851587f);
  q = fmaf(q, t, -1.13520398f);
  q = fmaf(q, t, 0.27886807f);
  q = fmaf(q, t, -0.18628806f);
  q = fmaf(q, t, 0.09678418f);
  q = fmaf(q, t, 0.37409196f);
  q = fmaf(q, t, 1.00002368f);
  q = fmaf(q, t, -1.26551223f);
  const float erfc = t * einx_expf(fmaf(-ax, ax, q));
  r = 1.0f - erfc;
  return x > 0.0f ? r : -r;
}

/* logistic sigmoid exactly as the reference writes it: 1 / (1 + exp(-x))
 * (core/modules/utils/detector_util.py:37). */
EINX_HD float einx_sigmoidf(float x) { return 1.0f / (1.0f + einx_expf(-x)); }

/* logsigmoid(x) = min(x,0) - log(1 + exp(-|x|))  (ATen log_sigmoid formula). */
EINX_HD float einx_logsigmoidf(float x) {
  const float mn = x < 0.0f ? x : 0.0f;
  return mn - einx_logf(1.0f + einx_expf(-fabsf(x)));
}

/* exact GELU (erf form): 0.5 x (1 + erf(x / sqrt 2)). */
EINX_HD float einx_geluf(float x) { return 0.5f * x * (1.0f + einx_erff(x * 0.707106781186547524f)); }

/* acos(x) (cephes-style asin polynomial), ~2 ulp; |x| > 1 gives NaN like libm.
 * Reference: torch.acos in core/metrics/keypoints_metrics.py:241-247. */
EINX_HD float einx_asin_poly(float z, float x) {
  /* asin(x) = x + x z P(z), z = x^2, |x| <= 0.5 */
  float p = 4.2163199048e-2f;
  p = fmaf(p, z, 2.4181311049e-2f);
  p = fmaf(p, z, 4.5470025998e-2f);
  p = fmaf(p, z, 7.4953002686e-2f);
  p = fmaf(p, z, 1.6666752422e-1f);
  return fmaf(x * z, p, x);
}
EINX_HD float einx_acosf(float x) {
  const float ax = fabsf(x);
  if (!(ax <= 1.0f)) return einx_u2f(0x7fc00000u);
  if (ax <= 0.5f) return 1.57079632679489662f - einx_asin_poly(x * x, x);
  const float z = 0.5f * (1.0f - ax);
  const float s = sqrtf(z);
  const float r = 2.0f * einx_asin_poly(z, s);
  return x > 0.0f ? r : 3.14159265358979324f - r;
}

/* order-preserving map float -> uint32 (for radix selection / packed arg-max keys). */
EINX_HD uint32_t einx_ordered_key(float f) {
  uint32_t u = einx_f2u(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
EINX_HD float einx_ordered_unkey(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return einx_u2f(u);
}

#endif /* EINX_MATH_H */
