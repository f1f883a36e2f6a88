/* mcs_pion.h -- the pion-decay gamma-ray fold of the reference's photon post-processing, shared by the device
 * kernel (csrc/mcs_consumers.hip: mcs_k_photon_pion) and its CPU twin (oracle/mcs_consumers.cpp: orc_photon_pion).
 *
 * Reference: src/pion_kafexhiu.jl:37-245 (`pion_kafexhiu`: for every momentum bin of a nucleus species' plasma-frame dN/dp, the
 * gamma-ray production cross section d sigma / d ln E of Kafexhiu, Aharonian, Taylor & Vila, Phys. Rev. D 90, 123014 (2014) times
 * target density, particle count and speed), with src/KATV2014.jl:16-297 (`get_sigma_pi`: sections 4, eqs. 1-7; `get_Amax`: eq. 12,
 * table VII; `get_Ffunc`: eqs. 9, 11, 14, 15, table V), called through `photon_pion_decay` (src/photon_pion_decay.jl:40-183) from
 * src/photon_calcs.jl:66-88 for every grid zone of a species with aa >= 1.  The photon stack is dead code in the reference (SURVEY.md
 * section 2 row 25) and is followed as specification text, AS WRITTEN where it differs from the paper:
 *   P1  Y_gamma = E + m_pi^2 / E (KATV2014.jl:139-140; the paper's eq. 9 and the function's own docstring have m_pi^2 / (4 E));
 *   P2  the nucleon rest energy enters as `mp` next to GeV quantities (Unitful would refuse): read as m_p c^2 = 0.93827 GeV;
 *   P3  i_data = 1 (GEANT 4) is hard-wired (pion_kafexhiu.jl:79): the other three parametrisations are kept behind the argument.
 * Energies in GeV, cross sections in mb, as in the paper. */
#ifndef MCS_PION_H
#define MCS_PION_H

#include <math.h>
#include "mcs.h"

#if defined(__HIPCC__)
#define MCS_PION_FN __host__ __device__ inline
#else
#define MCS_PION_FN static inline
#endif

#define MCS_PION_TTH 0.2797            /* constants.jl:16 threshold kinetic energy [GeV] */
#define MCS_PION_MRES 1.1883           /* :18 resonance mass [GeV] */
#define MCS_PION_GRES 0.2264           /* :20 resonance width [GeV] */
#define MCS_PION_E0 0.134976           /* :22 neutral pion rest energy [GeV] */
#define MCS_PION_MPC2 0.93827208816    /* m_p c^2 [GeV] (CODATA 2018) */
#define MCS_PION_GEV_ERG 1.602176634e-3

/* total inelastic pp cross section, eq. (1) */
MCS_PION_FN double mcs_pion_sigma_inel(double Tp) {
  const double ratio = Tp / MCS_PION_TTH;
  const double lr = log(ratio);
  const double t = 1 - pow(ratio, -1.9);
  return (30.7 - 0.96 * lr + 0.18 * lr * lr) * (t * t * t);
}

/* inclusive pi0 production cross section [mb] (KATV2014.jl:22-111) */
MCS_PION_FN double mcs_pion_sigma_pi(double Tp, int i_data, double s) {
  const double mp = MCS_PION_MPC2, Mr = MCS_PION_MRES, Gr = MCS_PION_GRES, E0 = MCS_PION_E0;
  if (Tp < 2) {
    const double g4 = Mr * hypot(Mr, Gr);
    const double K = sqrt(8.0) * Mr * Gr * g4 / (3.141592653589793 * sqrt(Mr * Mr + g4));
    const double rs = sqrt(s);
    const double d = (rs - mp) * (rs - mp) - Mr * Mr;
    const double fBW = mp * K / (d * d + Mr * Mr * Gr * Gr);
    const double a = s - E0 * E0 - 4 * mp * mp, b = 4 * E0 * mp;
    const double eta = sqrt(a * a - b * b) / (2 * E0 * rs);
    const double s1 = 7.66e-3 * pow(eta, 1.95) * (1 + eta + pow(eta, 5.0)) * pow(fBW, 1.86);     /* eq. (2) */
    const double s2 = Tp < 2 * MCS_PION_TTH ? 0.0 : 5.7 / (1 + exp(-9.3 * (Tp - 1.4)));      /* eq. (5) */
    return s1 + s2;
  }
  if (Tp < 5) {
    const double Q = (Tp - MCS_PION_TTH) / mp;
    const double n_pi0 = -6.0e-3 + 0.237 * Q - 0.023 * Q * Q;                               /* eq. (6) */
    return n_pi0 * mcs_pion_sigma_inel(Tp);
  }
  double a1, a2, a3, a4, a5;
  if (i_data == 2 && Tp > 5.0e1) { a1 = 0.652; a2 = 0.0016; a3 = 0.488; a4 = 0.1928; a5 = 0.483; }
  else if (i_data == 3 && Tp > 1.0e2) { a1 = 5.436; a2 = 0.254; a3 = 0.072; a4 = 0.075; a5 = 0.166; }
  else if (i_data == 4 && Tp > 1.0e2) { a1 = 0.908; a2 = 0.0009; a3 = 6.089; a4 = 0.176; a5 = 0.448; }
  else { a1 = 0.728; a2 = 0.596; a3 = 0.491; a4 = 0.2503; a5 = 0.117; }
  const double xi = (Tp - 3) / mp;
  const double n_pi0 = a1 * pow(xi, a4) * (1 + exp(-a2 * pow(xi, a5))) * (1 - exp(-a3 * pow(xi, 0.25)));      /* eq. (7) */
  return n_pi0 * mcs_pion_sigma_inel(Tp);
}

/* E_gamma^max and A_max(Tp) (KATV2014.jl:229-296) */
MCS_PION_FN void mcs_pion_amax(double Tp, int i_data, double s, double sigma_pi, double* Eg_max, double* Amax) {
  const double mp = MCS_PION_MPC2, E0 = MCS_PION_E0;
  const double rs = sqrt(s);
  const double E_pi_CM = (s - 4 * mp * mp + E0 * E0) / (2 * rs);
  const double g_CM = (Tp + 2 * mp) / rs;
  const double b_CM = sqrt(1 - 1 / (g_CM * g_CM));
  const double P_pi_CM = sqrt(E_pi_CM * E_pi_CM - E0 * E0);
  const double Emax_pi_LAB = g_CM * (E_pi_CM + P_pi_CM * b_CM);
  const double g_LAB = Emax_pi_LAB / E0;
  const double b_LAB = sqrt(1 - 1 / (g_LAB * g_LAB));
  *Eg_max = E0 / 2 * g_LAB * (1 + b_LAB);
  if (Tp < 1) { *Amax = 5.9 * sigma_pi / Emax_pi_LAB; return; }
  double b1, b2, b3;
  if (i_data == 1 && Tp < 5) { b1 = 9.53; b2 = 0.52; b3 = 0.054; }
  else if (i_data == 2 && Tp > 50) { b1 = 9.06; b2 = 0.3795; b3 = 0.01105; }
  else if (i_data == 3 && Tp > 100) { b1 = 10.77; b2 = 0.412; b3 = 0.01264; }
  else if (i_data == 4 && Tp > 100) { b1 = 13.16; b2 = 0.4419; b3 = 0.01439; }
  else { b1 = 9.13; b2 = 0.35; b3 = 0.0097; }
  const double th = Tp / mp;
  const double lt = log(th);
  *Amax = b1 * pow(th, -b2) * sigma_pi / mp * exp(b3 * lt * lt);
}

/* F(Tp, E_gamma) (KATV2014.jl:136-212; P1) */
MCS_PION_FN double mcs_pion_F(double Tp, double Eg, int i_data, double Eg_max) {
  const double mp = MCS_PION_MPC2, E0 = MCS_PION_E0;
  const double Y = Eg + E0 * E0 / Eg;
  const double Ymax = Eg_max + E0 * E0 / Eg_max;
  const double X = (Y - E0) / (Ymax - E0);
  if (X < 0 || X > 1) return 0.0;
  if (Tp < 1) {
    const double th = Tp / mp;
    const double kappa = 3.29 - 0.2 * pow(th, -1.5);
    return pow(1 - X, kappa);
  }
  double lam, al, be, ga;
  if (Tp < 4) {
    const double q = (Tp - 1) / mp;
    const double mu = 1.25 * pow(q, 1.25) * exp(-1.25 * q);
    lam = 3.0; al = 1.0; be = mu + 2.45; ga = mu + 1.45;
  } else if (Tp < 20) {
    const double q = (Tp - 1) / mp;
    const double mu = 1.25 * pow(q, 1.25) * exp(-1.25 * q);
    lam = 3.0; al = 1.0; be = 1.5 * mu + 4.95; ga = mu + 1.5;
  } else if (i_data == 1 && Tp > 100) { lam = 3.0; al = 0.5; be = 4.9; ga = 1.0; }
  else if (i_data == 2 && Tp > 50) { lam = 3.5; al = 0.5; be = 4.0; ga = 1.0; }
  else if (i_data == 3 && Tp > 100) { lam = 3.55; al = 0.5; be = 3.6; ga = 1.0; }
  else if (i_data == 4 && Tp > 100) { lam = 3.55; al = 0.5; be = 4.5; ga = 1.0; }
  else { lam = 3.0; al = 0.5; be = 4.2; ga = 1.0; }
  const double Cc = lam * E0 / Ymax;
  return pow(1 - pow(X, al), be) / pow(1 + X / Cc, ga);
}

/* what pion_kafexhiu needs of one momentum bin and does not depend on the photon energy (pion_kafexhiu.jl:171-193): returns 0
 * when the bin is skipped (empty, or below the production threshold).  p_lo, p_hi: the bin's edges (cgs); mc = A m_p c; aa = A. */
MCS_PION_FN int mcs_pion_bin(double count, double p_lo, double p_hi, double mc, double aa, int i_data, double* Tp_out, double* vel_out,
                              double* Egmax_out, double* Amax_out) {
  if (count <= 1.0e-99) return 0;
  const double p2 = p_lo * p_hi;
  const double gam = sqrt(1 + p2 / (mc * mc));
  const double E0_gev = aa * MCS_PION_MPC2;                /* ustrip(GeV, E0) */
  double Tp = (gam - 1) * E0_gev;
  Tp /= aa;
  const double vel = sqrt(p2) / (gam * aa * MCS_MP);
  if (Tp < MCS_PION_TTH) return 0;
  const double s = 2 * MCS_PION_MPC2 * (Tp + 2 * MCS_PION_MPC2);
  const double sig = mcs_pion_sigma_pi(Tp, i_data, s);
  mcs_pion_amax(Tp, i_data, s, sig, Egmax_out, Amax_out);
  *Tp_out = Tp; *vel_out = vel;
  return 1;
}

/* dP/d(ln E) [erg/s] at ONE photon energy: the cosmic-ray loop of pion_kafexhiu.jl:171-229 over the bins prepared by mcs_pion_bin,
 * in order; `pref[i]` = target_density * count * vel (<= 0: skipped); floor and species scaling of :235-241. */
MCS_PION_FN double mcs_pion_fold_one(const double* pref, const double* Tp, const double* Egmax, const double* Amax, int nbins, int i_data,
                                     double e_erg, double scaling) {
  const double Eg = e_erg / MCS_PION_GEV_ERG;
  double acc = 1.0e-99;
  for (int i = 0; i < nbins; ++i) {
    if (!(pref[i] > 0.0)) continue;
    const double sig_tot = Amax[i] * mcs_pion_F(Tp[i], Eg, i_data, Egmax[i]) * Eg;
    const double rate = pref[i] * (sig_tot * 1.0e-27);
    acc += rate * e_erg;
  }
  return acc < 1.0e-99 ? 1.0e-99 : acc * scaling;
}

#endif
