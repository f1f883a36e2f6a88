/* mcs_math_f32.h -- the fp32 primitives of the fp32-state variant in their EXACT form: built only from correctly rounded
 * fp32 + - * / sqrt, rint / floor and conversions, so that the device kernel compiled with them (mcs_k_transport_f32_loop_exact,
 * MCS_F32_EXACT=1) and the CPU restatement (oracle/mcs_oracle_f32.inc) agree BIT FOR BIT -- the bit-level check the
 * hardware-primitive kernels (v_rcp_f32, v_sqrt_f32, v_sin_f32, v_cos_f32: ~1 ulp, not reproducible off the device) cannot have.
 * The hardware build is then compared with this one.  Mirrors include/mcs_math.h (the fp64 normative header) in float.
 * Shared by csrc/mcs_transport_f32.inc (namespace mcsf, EXACT = true) and the oracle; no fma is formed (-ffp-contract=off on
 * both sides; hipcc's `/` and sqrt are correctly rounded for fp32: -fhip-fp32-correctly-rounded-divide-sqrt is its default). */
#ifndef MCS_MATH_F32_H
#define MCS_MATH_F32_H

#if defined(__HIPCC__)
#define MCSF_FN __host__ __device__ __forceinline__
#else
#define MCSF_FN static inline
#endif

namespace mcsf {

#define MCSF_TWOPI 6.2831855f          /* float(2 pi) */
#define MCSF_INV_TWOPI 0.15915494f     /* float(1 / 2 pi) */

MCSF_FN float div_(float a, float b) { return a / b; }
MCSF_FN float rcp_(float b) { return 1.0f / b; }
MCSF_FN float sqrt_(float x) { return __builtin_sqrtf(x); }

/* sin and cos of 2 pi r, r in revolutions: quadrant by rint(4 r) (exact), the remainder |f| <= 1/8 revolution times float(2 pi),
 * then the classic single-precision minimax pair on [-pi/4, pi/4] (Cephes sinf / cosf coefficients). */
MCSF_FN void sincos_rev_(float r, float* s_out, float* c_out) {
  const float q = __builtin_rintf(r * 4.0f);
  const float f = r - q * 0.25f;
  const float t = f * MCSF_TWOPI;
  const float z = t * t;
  float ps = -1.9515295891e-4f;
  ps = ps * z + 8.3321608736e-3f; ps = ps * z + -1.6666654611e-1f;
  const float s = t + t * z * ps;
  float pc = 2.443315711809948e-5f;
  pc = pc * z + -1.388731625493765e-3f; pc = pc * z + 4.166664568298827e-2f;
  const float c = (1.0f - 0.5f * z) + z * z * pc;
  const int n = (int)q & 3;
  *s_out = n == 0 ? s : (n == 1 ? c : (n == 2 ? -s : -c));
  *c_out = n == 0 ? c : (n == 1 ? -s : (n == 2 ? -c : s));
}
MCSF_FN float cos_rad_(float x) { float s, c; sincos_rev_(x * MCSF_INV_TWOPI, &s, &c); return c; }

}  // namespace mcsf

#endif
