/* mcs_math.h -- normative deterministic elementary functions of the transport path.
 *
 * The per-particle history of the hot path (reference: src/particle_loop.jl,
 * src/scattering.jl, src/transformers.jl, src/prob_return.jl, src/get_psd_bins.jl)
 * is chaotic: a 1-ulp difference in one sin() changes which zone boundary a
 * particle crosses a few hundred steps later.  Bit-for-bit agreement between
 * the CPU oracle and the gfx950 kernels is therefore only possible when both
 * evaluate the *same* sequence of correctly-rounded IEEE-754 operations.
 * Everything here is built from  + - * / sqrt fma rint floor  only (all
 * correctly rounded on x86-64 and on gfx950) and must be compiled with
 * -ffp-contract=off so that no compiler forms an fma that is not written.
 *
 * Julia's Base.sin/cos/asin/acos/atan/log10/mod2pi (what the reference calls)
 * are themselves pure-software <1 ulp implementations; these are <2 ulp
 * (tests/test_math.py measures it against libm).  The coefficients come from
 * tools/gen_math_coeffs.py (mpmath, from scratch).
 *
 * Usable from g++ (host) and hipcc (host + device).
 */
#ifndef MCS_MATH_H
#define MCS_MATH_H

#include <stdint.h>
#include "mcs_math_coeffs.inc"

#if defined(__HIPCC__)
#define MCS_HD __host__ __device__ __forceinline__
#else
#define MCS_HD static inline __attribute__((always_inline))
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

/* MCS_SC(c): a coefficient at its point of use.  Plain literal: the device build passes
 * `-mllvm -disable-machine-licm`, which is what keeps hipcc from hoisting ~150 VGPRs of
 * polynomial coefficients out of the particle loop (and then spilling); the constant is
 * rematerialised where it is consumed.  (Round 1 first used inline-asm s_mov pairs for this;
 * the compiler had to pad those with s_nop and v_mov, 5-6 instructions per constant.) */
#define MCS_SC(c) (c)

#define MCS_PI      MCS_PI_DD_0
#define MCS_TWOPI   MCS_TWOPI_DD_0
#define MCS_HALFPI  MCS_PIO2_DD_0
/* prevfloat(1.0): reference src/scattering.jl:3 */
#define MCS_SIN_UPPER_LIMIT 0x1.fffffffffffffp-1

namespace mcsm {

MCS_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
/* sqrt_: correctly rounded fp64 square root.  On gfx950 hipcc expands __builtin_sqrt to
 * v_rsq_f64 + two Goldschmidt/Newton steps wrapped in an exponent rescale (for arguments
 * below 2^-767) and a class test: 17 instructions.  With MCS_DEVICE_FAST_SQRT the same
 * iteration is used WITHOUT the rescale (the path never takes roots of subnormal-range
 * numbers): 10 instructions, identical bits for every normal argument and for +0. */
#if defined(__HIP_DEVICE_COMPILE__) && defined(MCS_DEVICE_FAST_SQRT)
MCS_HD double sqrt_(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return (x == 0.0 || x == __builtin_inf()) ? x : g;
}
/* The same for an argument known to be finite and >= 0 (1 - c^2 of a cosine, (1 - |x|) / 2 of an |x| <= 1): the
 * reciprocal square root is taken of max(x, DBL_MIN), so that x = 0 comes out as 0 through the arithmetic itself (g = x * y
 * = 0 and every correction term vanishes) instead of through a class test and two selects.  For every normal x the
 * operations and their results are those of sqrt_ above. */
MCS_HD double sqrt_nn_(double x) {
  const double y = __builtin_amdgcn_rsq(__builtin_fmax(x, 2.2250738585072014e-308));
  double g = x * y;
  double h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return g;
}
#else
MCS_HD double sqrt_(double a) { return __builtin_sqrt(a); }
MCS_HD double sqrt_nn_(double a) { return __builtin_sqrt(a); }
#endif
MCS_HD double abs_(double a) { return __builtin_fabs(a); }
MCS_HD double copysign_(double a, double b) { return __builtin_copysign(a, b); }

MCS_HD uint64_t bits_(double x) { union { double d; uint64_t u; } v; v.d = x; return v.u; }
MCS_HD double from_bits_(uint64_t u) { union { double d; uint64_t u; } v; v.u = u; return v.d; }

/* ---- polynomial cores (Horner, fma) ------------------------------------ */
MCS_HD double sin_core(double z) {
  double p = MCS_SC(MCS_SIN_5);
  p = fma_(p, z, MCS_SC(MCS_SIN_4)); p = fma_(p, z, MCS_SC(MCS_SIN_3)); p = fma_(p, z, MCS_SC(MCS_SIN_2));
  p = fma_(p, z, MCS_SC(MCS_SIN_1)); p = fma_(p, z, MCS_SC(MCS_SIN_0));
  return p;
}
MCS_HD double cos_core(double z) {
  double p = MCS_SC(MCS_COS_5);
  p = fma_(p, z, MCS_SC(MCS_COS_4)); p = fma_(p, z, MCS_SC(MCS_COS_3)); p = fma_(p, z, MCS_SC(MCS_COS_2));
  p = fma_(p, z, MCS_SC(MCS_COS_1)); p = fma_(p, z, MCS_SC(MCS_COS_0));
  return p;
}
/* sin and cos of a reduced argument |r| <= pi/4 */
MCS_HD double ksin(double r) { double z = r * r; return fma_(r * z, sin_core(z), r); }
MCS_HD double kcos(double r) {
  double z = r * r;
  return fma_(z * z, cos_core(z), fma_(-0.5, z, 1.0));
}

/* Cody-Waite reduction by pi/2; valid for |x| < ~1e5 (k*PIO2_0 exact). */
MCS_HD double reduce_pio2(double x, int* n) {
  double k = __builtin_rint(x * MCS_TWO_OVER_PI);
  double r = fma_(-k, MCS_SC(MCS_PIO2_0), x);
  r = fma_(-k, MCS_SC(MCS_PIO2_1), r);
  r = fma_(-k, MCS_SC(MCS_PIO2_2), r);
  *n = (int)k & 3;
  return r;
}

MCS_HD void sincos(double x, double* s, double* c) {
  int n; double r = reduce_pio2(x, &n);
  double sr = ksin(r), cr = kcos(r);
  double a = (n & 1) ? cr : sr;      /* sin candidate */
  double b = (n & 1) ? sr : cr;      /* cos candidate */
  *s = (n & 2) ? -a : a;
  *c = ((n + 1) & 2) ? -b : b;
}
MCS_HD double sin(double x) { double s, c; sincos(x, &s, &c); return s; }
MCS_HD double cos(double x) { double s, c; sincos(x, &s, &c); return c; }

/* Base.mod2pi for the bounded phase angles of the path (|x| < ~1e5). */
MCS_HD double mod2pi(double x) {
  /* Straight-line on purpose: for x already in [0, 2 pi) k is 0 and both fma return x bit for bit, so the early-out of
   * the first version ("if x is in range return it") bought nothing -- and on the GPU it was a conditional region that
   * some lane of a wave took in every other pass (the phase leaves [0, 2 pi) once per gyration). */
#ifdef MCS_MOD2PI_EARLY_OUT      /* A/B builds only: the first version */
  if (__builtin_expect(x >= 0.0 && x < MCS_TWOPI, 1)) return x;
#endif
  const double k = __builtin_floor(x * MCS_INV_TWOPI);
  double r = fma_(-k, MCS_SC(MCS_TWOPI_DD_0), x);
  r = fma_(-k, MCS_SC(MCS_TWOPI_DD_1), r);
  r = r < 0.0 ? r + MCS_TWOPI : r;
  r = r >= MCS_TWOPI ? r - MCS_TWOPI : r;
  return r;
}

MCS_HD double asin_core(double z) {
  double p = MCS_SC(MCS_ASIN_12);
  p = fma_(p, z, MCS_SC(MCS_ASIN_11)); p = fma_(p, z, MCS_SC(MCS_ASIN_10)); p = fma_(p, z, MCS_SC(MCS_ASIN_9));
  p = fma_(p, z, MCS_SC(MCS_ASIN_8));  p = fma_(p, z, MCS_SC(MCS_ASIN_7));  p = fma_(p, z, MCS_SC(MCS_ASIN_6));
  p = fma_(p, z, MCS_SC(MCS_ASIN_5));  p = fma_(p, z, MCS_SC(MCS_ASIN_4));  p = fma_(p, z, MCS_SC(MCS_ASIN_3));
  p = fma_(p, z, MCS_SC(MCS_ASIN_2));  p = fma_(p, z, MCS_SC(MCS_ASIN_1));  p = fma_(p, z, MCS_SC(MCS_ASIN_0));
  return p;
}

/* asin for |x| <= 1 (callers clamp; |x| > 1 returns NaN through sqrt). Branch-free. */
MCS_HD double asin(double x) {
  double ax = abs_(x);
  bool small = ax < 0.5;
  double z = small ? x * x : (1.0 - ax) * 0.5;
  double s = small ? ax : sqrt_(z);
  double t = fma_(s * z, asin_core(z), s);            /* asin(s) */
  double big = MCS_SC(MCS_PIO2_DD_0) - (2.0 * t - MCS_PIO2_DD_1);
  return copysign_(small ? t : big, x);
}

/* ---- the same sincos / asin with the polynomial coefficients taken from a caller-held
 * table instead of literals.  The HIP transport kernel keeps the 25 hot coefficients in
 * VGPRs for the whole loop (no per-use constant materialisation); arithmetic, order and
 * results are identical to sincos()/asin() above. */
struct HotCoef {
  double S0, S1, S2, S3, S4, S5;          /* MCS_SIN_0..5 */
  double C0, C1, C2, C3, C4, C5;          /* MCS_COS_0..5 */
  double A0, A1, A2, A3, A4, A5, A6, A7, A8, A9, A10, A11, A12;   /* MCS_ASIN_0..12 */
  double R0, R1, R2, R3;                  /* 2/pi and the three parts of pi/2 (Cody-Waite reduction) */
};
MCS_HD void sincos_t(double x, double* s, double* c, const HotCoef& k) {
  /* reduce_pio2 with its four constants from the table: same operations, same order */
  const double kk = __builtin_rint(x * k.R0);
  double r = fma_(-kk, k.R1, x);
  r = fma_(-kk, k.R2, r);
  r = fma_(-kk, k.R3, r);
  const int n = (int)kk & 3;
  const double z = r * r;
  double ps = k.S5;
  ps = fma_(ps, z, k.S4); ps = fma_(ps, z, k.S3); ps = fma_(ps, z, k.S2); ps = fma_(ps, z, k.S1); ps = fma_(ps, z, k.S0);
  double pc = k.C5;
  pc = fma_(pc, z, k.C4); pc = fma_(pc, z, k.C3); pc = fma_(pc, z, k.C2); pc = fma_(pc, z, k.C1); pc = fma_(pc, z, k.C0);
  const double sr = fma_(r * z, ps, r);
  const double cr = fma_(z * z, pc, fma_(-0.5, z, 1.0));
  double a = (n & 1) ? cr : sr;
  double b = (n & 1) ? sr : cr;
  /* the two negations as sign-bit flips (bit 1 of the quadrant number moved to bit 63): same bits as
   * (n & 2) ? -a : a  and  ((n + 1) & 2) ? -b : b  in sincos() above, without the compares and selects */
  *s = from_bits_(bits_(a) ^ ((uint64_t)((uint32_t)n & 2u) << 62));
  *c = from_bits_(bits_(b) ^ ((uint64_t)((uint32_t)(n + 1) & 2u) << 62));
}
MCS_HD double asin_t(double x, const HotCoef& k) {
  double ax = abs_(x);
  bool small = ax < 0.5;
  double z = small ? x * x : (1.0 - ax) * 0.5;
  double s = small ? ax : sqrt_nn_(z);
  double p = k.A12;
  p = fma_(p, z, k.A11); p = fma_(p, z, k.A10); p = fma_(p, z, k.A9); p = fma_(p, z, k.A8); p = fma_(p, z, k.A7);
  p = fma_(p, z, k.A6); p = fma_(p, z, k.A5); p = fma_(p, z, k.A4); p = fma_(p, z, k.A3); p = fma_(p, z, k.A2);
  p = fma_(p, z, k.A1); p = fma_(p, z, k.A0);
  double t = fma_(s * z, p, s);
  double big = MCS_SC(MCS_PIO2_DD_0) - (2.0 * t - MCS_SC(MCS_PIO2_DD_1));
  return copysign_(small ? t : big, x);
}

MCS_HD double acos(double x) {
  double ax = abs_(x);
  if (ax < 0.5) {
    double z = x * x;
    double t = fma_(x * z, asin_core(z), x);
    return MCS_SC(MCS_PIO2_DD_0) - (t - MCS_PIO2_DD_1);
  }
  if (ax >= 1.0) return (x > 0.0) ? 0.0 : ((x < 0.0) ? MCS_PI : x);
  double z = (1.0 - ax) * 0.5;
  double s = sqrt_(z);
  double t = 2.0 * fma_(s * z, asin_core(z), s);
  return (x > 0.0) ? t : (MCS_PI_DD_0 - (t - MCS_PI_DD_1));
}

MCS_HD double atan_core(double z) {
  double p = MCS_SC(MCS_ATAN_10);
  p = fma_(p, z, MCS_SC(MCS_ATAN_9)); p = fma_(p, z, MCS_SC(MCS_ATAN_8)); p = fma_(p, z, MCS_SC(MCS_ATAN_7));
  p = fma_(p, z, MCS_SC(MCS_ATAN_6)); p = fma_(p, z, MCS_SC(MCS_ATAN_5)); p = fma_(p, z, MCS_SC(MCS_ATAN_4));
  p = fma_(p, z, MCS_SC(MCS_ATAN_3)); p = fma_(p, z, MCS_SC(MCS_ATAN_2)); p = fma_(p, z, MCS_SC(MCS_ATAN_1));
  p = fma_(p, z, MCS_SC(MCS_ATAN_0));
  return p;
}

/* Base.atan(y, x) for finite arguments. */
MCS_HD double atan2(double y, double x) {
  double ax = abs_(x), ay = abs_(y);
  double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
  double r;
  if (mx == 0.0) {
    r = 0.0;
  } else {
    double a = mn / mx;
    double t = a, bh = 0.0, bl = 0.0;
    if (a > 0x1.a827999fcef32p-2 /* tan(pi/8) */) {
      t = (a - 1.0) / (a + 1.0); bh = MCS_SC(MCS_PIO4_DD_0); bl = MCS_SC(MCS_PIO4_DD_1);
    }
    double z = t * t;
    double p = fma_(t * z, atan_core(z), t);
    r = bh + (p + bl);
    if (ay > ax) r = MCS_SC(MCS_PIO2_DD_0) - (r - MCS_PIO2_DD_1);
  }
  if (x < 0.0 || (x == 0.0 && (bits_(x) >> 63))) r = MCS_SC(MCS_PI_DD_0) - (r - MCS_PI_DD_1);
  return copysign_(r, y);
}

MCS_HD double log_core(double z) {
  double p = MCS_SC(MCS_LOG_6);
  p = fma_(p, z, MCS_SC(MCS_LOG_5)); p = fma_(p, z, MCS_SC(MCS_LOG_4)); p = fma_(p, z, MCS_SC(MCS_LOG_3));
  p = fma_(p, z, MCS_SC(MCS_LOG_2)); p = fma_(p, z, MCS_SC(MCS_LOG_1)); p = fma_(p, z, MCS_SC(MCS_LOG_0));
  return p;
}

/* log10 for positive normal x (the path only bins momenta/angles >= a positive floor). */
MCS_HD double log10(double x) {
  uint64_t u = bits_(x);
  int e = (int)((u >> 52) & 0x7ff) - 1022;
  double m = from_bits_((u & 0x000fffffffffffffULL) | 0x3fe0000000000000ULL); /* [0.5,1) */
  if (m < 0x1.6a09e667f3bcdp-1 /* sqrt(1/2) */) { m *= 2.0; e -= 1; }
  double f = m - 1.0;
  double s = f / (m + 1.0);
  double z = s * s;
  double s2 = 2.0 * s;
  double lnm = fma_(s2 * z, log_core(z), s2);
  double de = (double)e;
  double lo = fma_(lnm, MCS_SC(MCS_INVLN10_DD_0), de * MCS_LOG10_2_1);
  return fma_(de, MCS_SC(MCS_LOG10_2_0), lo);
}

/* hypot(1, t) as the path uses it (t = p/mc, 1e-6 .. 1e12): no scaling needed. */
MCS_HD double hypot1(double t) { return sqrt_(1.0 + t * t); }
/* LinearAlgebra.norm of a 3-vector of momenta (no over/underflow in range). */
MCS_HD double norm3(double x, double y, double z) { return sqrt_(x * x + y * y + z * z); }

}  /* namespace mcsm */
#endif
