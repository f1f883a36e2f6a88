/* mcs_synch.h -- the synchrotron fold of the reference's photon post-processing, shared by the device
 * kernel (csrc/mcs_consumers.hip: mcs_k_photon_synch) and its CPU twin (oracle/mcs_consumers.cpp: orc_photon_synch).
 *
 * Reference: src/synch_emission.jl:27-171 (`synch_emission`, `synch_emission!`), called through src/photon_synch.jl:28-72 from
 * src/photon_calcs.jl:27-113 for every grid zone of an electron species.  The photon stack is DEAD code in the reference (none
 * of its files is `include`d by the module, the one call site passes 16 of 26 arguments: SURVEY.md section 2 row 25), so these
 * files are followed as specification text.  The first synchrotron function F(x) = x int_x^inf K_{5/3}(t) dt comes from the
 * third-party package SynchrotronKernel.jl (`synchrotron_intensity`; Project.toml:31,63 pins 0.2.2; not in the reference
 * tree): restated from its definition as two Chebyshev fits (tools/gen_synch_coeffs.py, fit error <= 1e-13). */
#ifndef MCS_SYNCH_H
#define MCS_SYNCH_H

#include <math.h>
#include "mcs.h"
#include "mcs_synch_coeffs.inc"

#if defined(__HIPCC__)
#define MCS_SYNCH_FN __host__ __device__ inline
#else
#define MCS_SYNCH_FN static inline
#endif

#define MCS_HBAR 1.054571817e-27          /* erg s (CODATA 2018, Unitful's hbar) */
#define MCS_MEV_ERG 1.602176634e-6        /* 1 MeV in erg */
#define MCS_SYNCH_X_MAX 30.0              /* synch_emission.jl:120 xxx_max_set */

MCS_SYNCH_FN double mcs_cheb_eval(const double* c, int n, double t) {
  double b1 = 0.0, b2 = 0.0;
  for (int j = n - 1; j >= 1; --j) { const double b0 = 2 * t * b1 - b2 + c[j]; b2 = b1; b1 = b0; }
  return t * b1 - b2 + 0.5 * c[0];
}

/* F(x) = x * int_x^inf K_{5/3}(t) dt for 0 < x <= 32 (the fold never evaluates it at x >= 30) */
MCS_SYNCH_FN double mcs_synch_F(double x) {
  if (x <= MCS_SYNCH_X0) {
    const double v = cbrt(x);
    const double w = v * v;
    return v * mcs_cheb_eval(MCS_SYNCH_SMALL, MCS_SYNCH_SMALL_N, 2 * w / MCS_SYNCH_WMAX - 1);
  }
  const double s = 1 / x, a = MCS_SYNCH_SMIN, b = 1 / MCS_SYNCH_X0;
  return sqrt(3.141592653589793 * x / 2) * exp(-x) * mcs_cheb_eval(MCS_SYNCH_LARGE, MCS_SYNCH_LARGE_N, (2 * s - (a + b)) / (b - a));
}

/* photon energy of bin j (0-based) [erg]: exp10.(range(start = log10(E_min), step = 1 / bins_per_dec, length = n)), synch_emission.jl:40-43 */
MCS_SYNCH_FN double mcs_synch_energy(double log_emin_erg, double bins_per_dec, int j) { return pow(10.0, log_emin_erg + j * (1.0 / bins_per_dec)); }

/* synch_emission! (synch_emission.jl:112-171) for ONE photon energy: the sum over the electron momentum bins 0..nbins, in
 * order.  dN[i] = electrons in bin i (photon_synch.jl:45-52), p_edge[i], p_edge[i+1] its edges (cgs), mc of the species. */
MCS_SYNCH_FN double mcs_synch_fold_one(double acc, const double* dNdp, const double* p_edge, int nbins, double B, double mc, double E_erg) {
  if (B < 1.0e-20) return acc;
  const double p_fac = sqrt(3.0) / (2 * 3.141592653589793) * (MCS_QCGS * MCS_QCGS * MCS_QCGS * B / (MCS_ME * MCS_C * MCS_C));   /* :60 */
  const double w_g = E_erg / MCS_HBAR;
  for (int i = 0; i <= nbins; ++i) {
    const double d = dNdp[i];
    const double xN = d <= 1.0e-99 ? 1.0e-99 : d * (p_edge[i + 1] - p_edge[i]);       /* photon_synch.jl:45-52 */
    if (xN <= 1.0e-60) continue;
    const double p = sqrt(p_edge[i] * p_edge[i + 1]);
    if (p * MCS_C < 3 * MCS_MEV_ERG) continue;                                          /* :130 */
    const double ge = hypot(p / mc, 1.0);
    const double w_c = 3 * ge * ge * MCS_QCGS * B / (2 * mc);                           /* R&L 6.17c without sin(alpha) */
    if (w_c < 1.0e-55) continue;
    const double x = w_g / w_c;
    if (x >= MCS_SYNCH_X_MAX || x < 1.0e-15) continue;
    const double add = xN * w_g * p_fac * mcs_synch_F(x);
    if (add > 1.0e-55) acc += add;
  }
  return acc;
}

#endif /* MCS_SYNCH_H */
