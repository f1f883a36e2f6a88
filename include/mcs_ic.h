/* mcs_ic.h -- the inverse-Compton fold of the reference's photon post-processing, shared by the device
 * kernel (csrc/mcs_consumers.hip: mcs_k_photon_ic) and its CPU twin (oracle/mcs_consumers.cpp: orc_photon_ic).
 *
 * Reference: src/inverse_compton.jl:191-311 (`IC_emission_FCJ`: equation (9) of F. C. Jones, Phys. Rev. 167, 1159 (1968), summed
 * over the electron momentum bins of the explosion-frame d2N/dp dcos inside the jet cone and over the bins of the incoming photon
 * field), called through `photon_IC` (:36-188) from src/photon_calcs.jl:116-138 for every grid zone of the last (electron)
 * species; the photon field is the CMB of `photon_field!` (:313-383), a 60-bin table made on the host.  The photon stack is dead
 * code in the reference and is followed as specification text; where it cannot run as written:
 *   I1  `n_nu` is a local of photon_field! and undefined in IC_emission_FCJ (:246): the 60 bins photon_field! fills;
 *   I2  `findfirst(>(2 f - 1), cos_bounds)` is `nothing` for a full sphere (f = 1) and can point past the last angle bin: every
 *       angle bin then (clamped to num_psd_tht_bins);
 *   I3  d2N_slice is allocated (0:nmom, 0:ntht) and indexed [angle, momentum] (:54-61): indexed as meant. */
#ifndef MCS_IC_H
#define MCS_IC_H

#include <math.h>
#include "mcs.h"

#if defined(__HIPCC__)
#define MCS_IC_FN __host__ __device__ inline
#else
#define MCS_IC_FN static inline
#endif

#define MCS_IC_NNU 60                     /* photon_field!: n_nu (inverse_compton.jl:330) */
#define MCS_IC_MEV_ERG 1.602176634e-6

/* Lorentz factor of the electrons of one momentum bin (inverse_compton.jl:240-241) */
MCS_IC_FN double mcs_ic_gamma(double p_lo, double p_hi, double mc) {
  const double t = sqrt(p_lo * p_hi) / mc;
  return t < MCS_E_REL_PT ? 1.0 : sqrt(t * t + 1.0);
}
/* outgoing photon energy k (0-based) in units of m_e c^2 (:202-208) */
MCS_IC_FN double mcs_ic_alpha_out(double log_min_rm, double bins_per_dec, int k) { return pow(10.0, log_min_rm + k * (1.0 / bins_per_dec)); }

/* d2N/(dt d alpha) at ONE outgoing photon energy: the loops of :232-282 in the reference's order (electron bins outermost, then the
 * incoming photon bins).  xnum[i] <= 0 marks an electron bin the reference skips (:235).  Starts from the 1e-99 fill of :230. */
MCS_IC_FN double mcs_ic_fold_one(const double* xnum, const double* gam, int nbins, const double* alpha_in, const double* n_in, int n_nu,
                                 double alpha_out) {
  const double r0 = MCS_QCGS * MCS_QCGS / (MCS_ME * MCS_C * MCS_C);       /* :217 */
  double acc = 1.0e-99;
  for (int i = 0; i < nbins; ++i) {
    const double xn = xnum[i];
    if (!(xn > 0.0)) continue;
    const double g = gam[i];
    if (alpha_out >= g) continue;                                         /* :262 (does not depend on the incoming photon) */
    for (int j = 0; j < n_nu; ++j) {
      const double a1 = alpha_in[j];
      const double norm_fac = n_in[j] * 2 * 3.141592653589793 * (r0 * r0) * MCS_C / (a1 * (g * g));     /* :252 */
      const double q = alpha_out / (4 * a1 * (g * g) * (1 - alpha_out / g));                               /* :265 */
      const double t = a1 * g * q;
      const double cur = norm_fac * xn * (2 * q * log(q) + (1 + 2 * q) * (1 - q) + 8 * (t * t) * (1 - q) / (1 + 4 * a1 * g * q));   /* :268-271 */
      if (cur > 1.0e-60) acc += cur;                                      /* :275-277 */
    }
  }
  return acc;
}
/* observed energy flux per d(ln E) at Earth from d2N/(dt d alpha) (:285-308) */
MCS_IC_FN double mcs_ic_emis(double d2n, double alpha_out, double beam_area) {
  const double mec2 = MCS_ME * MCS_C * MCS_C;
  const double e = alpha_out * mec2;
  const double v = d2n / beam_area / mec2 * (e * e);
  return v <= 1.0e-55 ? 1.0e-99 : v;
}

#endif
