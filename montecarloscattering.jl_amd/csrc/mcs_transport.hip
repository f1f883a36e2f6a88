// mcs_transport.hip -- K1: the per-particle transport kernel for gfx950 (MI355X).
//
// Replaces the `for i_prt` loop of the reference (src/main_loops.jl:228-292):
// particle_loop (src/particle_loop.jl:1-508) + its callees scattering
// (src/scattering.jl:29-101), transform_p_PS / transform_p_PSP
// (src/transformers.jl:440-607), all_flux! / F_stream! (src/all_flux.jl:45-259),
// get_psd_bin_* (src/get_psd_bins.jl), prob_return / retro_time
// (src/prob_return.jl:36-344), tcut_track! (src/cuts.jl:149-162) and
// particle_finish! (src/particle_finish.jl:46-107).
//
// Execution model (CDNA4):
//  * one wavefront lane = one live particle; all particle state lives in VGPRs;
//  * persistent lanes: a lane whose particle finished claims the next unclaimed
//    particle (wave-aggregated atomic on one counter), so the 64 lanes of a wave
//    stay busy although histories last 1 .. 10^4 steps;
//  * the grid tables (9 x (n_grid+2) fp64) plus per-zone sin/cos(theta_B) and
//    1/(qB) sit in LDS; the three flux vectors and num_crossings are staged in
//    LDS (ds_add_f64) and flushed with one global atomic per entry per block;
//    the 22 MB psd and the escape spectra take global_atomic_add_f64 directly;
//  * RNG: Philox4x32-10 keyed by the reference's iseed_mod, counter = draw
//    number; no RNG state in memory.
//  * no MFMA: scalar fp64 per-particle arithmetic.
//
// Numerics: every fp64 expression is evaluated in the same order as the CPU
// oracle (which follows the Julia source), with include/mcs_math.h for the
// transcendentals and -ffp-contract=off, so per-particle results are
// bit-identical to the oracle; only the order of the atomic tally adds differs.
#include "mcs_device.h"
#include "../../include/mcs_math.h"

#pragma clang fp contract(off)

namespace {

constexpr double PI_ = 3.141592653589793;
constexpr double TWOPI_ = 6.283185307179586;
constexpr double SIN_UL = 0x1.fffffffffffffp-1;
constexpr double MP_ = MCS_MP, CC_ = MCS_C;

// ---- Philox4x32-10 ------------------------------------------------------------
__device__ __forceinline__ void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                             uint32_t k1, uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

__device__ __forceinline__ double u64_to_unit(uint32_t lo, uint32_t hi) {
  const unsigned long long u = ((unsigned long long)hi << 32) | lo;
  return (double)(u >> 11) * 0x1.0p-53;
}

// Per-particle stream: draw j -> words (2*(j&1), 2*(j&1)+1) of block j>>1.  The odd
// draw of a block is kept in a register so that two consecutive draws cost one block.
struct Rng {
  uint32_t k0, k1;
  uint32_t n;        // draws so far (a history makes < 2^32 draws: <= 10^4 steps + retro)
  double spare;
  __device__ __forceinline__ double rand() {
    const uint32_t j = n++;
    if (j & 1u) return spare;
    uint32_t o0, o1, o2, o3;
    philox_block(j >> 1, 0u, 0u /*STREAM_PARTICLE*/, 0u, k0, k1, o0, o1, o2, o3);
    spare = u64_to_unit(o2, o3);
    return u64_to_unit(o0, o1);
  }
};

// ---- LDS view -------------------------------------------------------------------
struct Lds {
  double *x, *ux, *uz, *ut, *gsf, *gef, *bt, *bsin, *bcos, *gd;   // n_grid+2 each
  double *fl;                                                   // 3*n_grid flux staging
  int* nc;                                                      // n_grid crossings staging
};

// ---- particle state (registers) ---------------------------------------------------
struct Pt {
  double weight, ptot_pf, pb_pf, p_perp, gam_pf, x, x_old, phi, prp, acctime, xn_per;
  double gyro_denom, gyro_rad, gyro_rad_tot, gyro_period, t_step;
  double ux, uz, ut, gsf, gef, bsin, bcos;
  int i_grid, i_grid_old, helix, tcut, i_return, n_retro;
  bool downstream, inj;
};

// Tally atomics with the address space spelled out, so that the ISA is
// global_atomic_add_f64 / ds_add_f64 (no-return forms) and never a flat atomic or a CAS loop.
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(3))) double ldouble;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(3))) int lint;
__device__ __forceinline__ void gadd_f64(double* p, double v) {
  (void)__builtin_amdgcn_global_atomic_fadd_f64((gdouble*)p, v);
}
__device__ __forceinline__ void ladd_f64(double* p, double v) {
  (void)__builtin_amdgcn_ds_atomic_fadd_f64((ldouble*)p, v);
}
__device__ __forceinline__ void gadd_u64(unsigned long long* p, unsigned long long v) {
  (void)__hip_atomic_fetch_add((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ladd_i32(int* p, int v) {
  (void)__hip_atomic_fetch_add((lint*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void cnt(const KArgs& a, int which, unsigned long long v = 1ull) {
  gadd_u64(&a.I[a.P.n_grid + which], v);
}
__device__ __forceinline__ void tadd(const KArgs& a, long long off, double v) { gadd_f64(&a.T[off], v); }

// src/get_psd_bins.jl:16-39
__device__ __forceinline__ int bin_momentum(const KArgs& a, double ptot_sk) {
  int bin;
  if (ptot_sk < a.P.psd_mom_min) bin = 0;
  else bin = (int)__builtin_trunc(mcsm::log10(ptot_sk / a.P.psd_mom_min) * a.P.psd_bins_per_dec_mom) + 1;
  if (bin > a.P.num_psd_mom_bins) { cnt(a, MCS_IC_MOMBIN_CLAMP); bin = a.P.num_psd_mom_bins; }
  return bin;
}
// src/get_psd_bins.jl:73-97
__device__ __forceinline__ int bin_angle(const KArgs& a, double px_sk, double ptot_sk) {
  if (ptot_sk == 0.0) return 0;
  const double p_cos = -px_sk / ptot_sk;
  int bin;
  if (p_cos < a.P.psd_cos_fine) {
    bin = a.P.num_psd_tht_bins - (int)__builtin_trunc((p_cos + 1) / a.P.psd_dcos);
  } else {
    const double th = mcsm::acos(p_cos);
    bin = th < a.P.psd_tht_min ? 0 : (int)__builtin_trunc(mcsm::log10(th / a.P.psd_tht_min) * a.P.psd_bins_per_dec_tht) + 1;
  }
  return bin < a.P.num_psd_tht_bins ? bin : a.P.num_psd_tht_bins;
}

// src/transformers.jl:440-476
__device__ __forceinline__ void transform_p_PS(double aa, double pb_pf, double p_perp, double gam_pf, double phi,
                                               double ux, double gsf, double bcos, double bsin, double& ptot_sk,
                                               double& px, double& py, double& pz, double& gam_sk) {
  const double m = aa * MP_;
  const double mc = m * CC_;
  const double phi_p = phi + PI_ / 2;
  double s, c;
  mcsm::sincos(phi_p, &s, &c);
  const double p_p_cos = p_perp * c;
  const double fx = pb_pf * bcos - p_p_cos * bsin;
  const double fy = p_perp * s;
  const double fz = pb_pf * bsin + p_p_cos * bcos;
  const double dpx = (gsf - 1) * fx + gsf * gam_pf * m * ux;
  px = fx + dpx; py = fy; pz = fz;
  ptot_sk = mcsm::norm3(px, py, pz);
  gam_sk = mcsm::hypot1(ptot_sk / mc);
}

// src/transformers.jl:523-607
__device__ __forceinline__ void transform_p_PSP(const KArgs& a, Pt& p, double ux_o, double uz_o, double ut_o, double gsf_o,
                                             double bcos_o, double bsin_o) {
  const double aa = a.aa;
  double phi_p = p.phi + PI_ / 2;
  const double m = aa * MP_;
  const double mc = m * CC_;
  double s, c;
  mcsm::sincos(phi_p, &s, &c);
  const double p_p_cos = p.p_perp * c;
  double fx = p.pb_pf * bcos_o - p_p_cos * bsin_o;
  double fy = p.p_perp * s;
  double fz = p.pb_pf * bsin_o + p_p_cos * bcos_o;
  double kx, ky, kz;
  {
    const double qx = ux_o / ut_o, qz = uz_o / ut_o;
    kx = ((gsf_o - 1) * (qx * qx) + 1) * fx + (gsf_o - 1) * (ux_o * uz_o / (ut_o * ut_o)) * fz + gsf_o * p.gam_pf * m * ux_o;
    ky = fy;
    kz = (gsf_o - 1) * (ux_o * uz_o / (ut_o * ut_o)) * fx + ((gsf_o - 1) * (qz * qz) + 1) * fz + gsf_o * p.gam_pf * m * uz_o;
  }
  const double ptot_sk = mcsm::norm3(kx, ky, kz);
  double pb_sk = kx * p.bcos + kz * p.bsin;
  if (ptot_sk < __builtin_fabs(pb_sk)) cnt(a, MCS_IC_PSP_CLAMP);   // the clamped shock-frame pair is never used again
  const double gam_sk = mcsm::hypot1(ptot_sk / mc);
  {
    const double ux = p.ux, uz = p.uz, ut = p.ut, gsf = p.gsf;
    const double qx = ux / ut, qz = uz / ut;
    fx = ((gsf - 1) * (qx * qx) + 1) * kx + (gsf - 1) * (ux * uz / (ut * ut)) * kz - gsf * gam_sk * m * ux;
    fy = ky;
    fz = (gsf - 1) * (ux * uz / (ut * ut)) * kx + ((gsf - 1) * (qz * qz) + 1) * kz - gsf * gam_sk * m * uz;
  }
  p.ptot_pf = mcsm::norm3(fx, fy, fz);
  p.pb_pf = fx * p.bcos + fz * p.bsin;
  if (p.ptot_pf < __builtin_fabs(p.pb_pf)) {
    p.p_perp = 1.0e-6 * p.ptot_pf;
    p.pb_pf = __builtin_copysign(__builtin_sqrt(p.ptot_pf * p.ptot_pf - p.p_perp * p.p_perp), p.pb_pf);
    cnt(a, MCS_IC_PSP_CLAMP);
  } else {
    p.p_perp = __builtin_sqrt(p.ptot_pf * p.ptot_pf - p.pb_pf * p.pb_pf);
  }
  p.gam_pf = mcsm::hypot1(p.ptot_pf / mc);
  phi_p = mcsm::atan2(fy, -fx * p.bsin + fz * p.bcos);
  p.phi = phi_p - PI_ / 2;
}

// src/scattering.jl:29-101
__device__ __forceinline__ void scattering(const KArgs& a, Rng& rng, Pt& p) {
  const double aa = a.aa;
  const double mc = aa * MP_ * CC_;
  double grt;
  if (aa < 1 && p.ptot_pf < a.P.pe_crit) {
    grt = a.P.pe_crit * CC_ * p.gyro_denom;
    p.gyro_period = TWOPI_ * a.P.game_crit * mc * p.gyro_denom;
  } else {
    grt = p.ptot_pf * CC_ * p.gyro_denom;
    p.gyro_period = TWOPI_ * p.gam_pf * mc * p.gyro_denom;
  }
  const double vp_tg = TWOPI_ * grt;
  const double lam = a.P.eta_mfp * grt;
  const double cos_max = mcsm::cos(__builtin_sqrt(6 * vp_tg / (p.xn_per * lam)));

  const double cos_old = p.pb_pf / p.ptot_pf;
  const double sin_old = p.p_perp / p.ptot_pf;
  const double cos_d = 1 - rng.rand() * (1 - cos_max);
  const double sin_d = __builtin_sqrt(1 - cos_d * cos_d);
  const double phi_scat = rng.rand() * TWOPI_ - PI_;
  double s_ps, c_ps;
  mcsm::sincos(phi_scat, &s_ps, &c_ps);
  const double cos_new = cos_old * cos_d + sin_old * sin_d * c_ps;
  double arg = 1 - cos_new * cos_new;
  if (arg < 0) arg = 0;
  const double sin_new = __builtin_sqrt(arg);
  p.pb_pf = p.ptot_pf * cos_new;
  p.p_perp = p.ptot_pf * sin_new;
  const double phi_p_old = p.phi + PI_ / 2;
  double phi_p_new = phi_p_old;
  if (sin_new != 0) {
    double sd = s_ps * sin_d / sin_new;
    if (__builtin_fabs(sd) > SIN_UL) sd = __builtin_copysign(SIN_UL, sd);
    phi_p_new += mcsm::asin(sd);
  }
  p.phi = phi_p_new - PI_ / 2;
}

// src/particle_loop.jl:639-650
__device__ __forceinline__ double perpendicular_momentum(const KArgs& a, double ptot, double pb) {
  if (ptot < __builtin_fabs(pb)) { cnt(a, MCS_IC_PPERP_CLAMP); return 1.0e-6 * ptot; }
  return __builtin_sqrt(ptot * ptot - pb * pb);
}

// src/particle_loop.jl:578-592
__device__ __forceinline__ double radiation_loss(double B2, double pp, double dt) {
  const double dlnp = MCS_RAD_LOSS_FAC * B2 * pp * dt;
  if (dlnp > 1.0e-2) pp /= 1 + dlnp; else pp *= 1 - dlnp;
  return pp;
}

// src/cuts.jl:149-162
__device__ __forceinline__ void tcut_track(const KArgs& a, int tcut_curr, double weight, double ptot_pf) {
  const int ion = a.i_ion - 1;
  tadd(a, a.L.weight_coupled + (tcut_curr - 1) + (long long)MCS_NA_C * ion, weight);
  const int i_pt = bin_momentum(a, ptot_pf);
  tadd(a, a.L.spectra_coupled + i_pt + (long long)(MCS_PSD_MAX + 1) * ((tcut_curr - 1) + (long long)MCS_NA_C * ion), weight);
}

// src/all_flux.jl:45-259 (all_flux!, calculate_x_spec_spectra!, F_stream!).  false: zone search failed.
__device__ __forceinline__ bool all_flux(const KArgs& a, const Lds& s, Pt& p) {
  const mcs_params& P = a.P;
  p.i_grid_old = p.i_grid;
  const int ne = P.n_grid + 2;
  int found = -1;
  if (p.x > p.x_old) {
    for (int j = p.i_grid + 1; j < ne; ++j) if (s.x[j] > p.x) { found = j - 1; break; }
  } else {
    for (int j = p.i_grid; j >= 0; --j) if (s.x[j] <= p.x) { found = j; break; }
  }
  if (found < 0) { cnt(a, MCS_IC_ZONE_FAIL); return false; }
  p.i_grid = found;
  const int n_xspec = a.tb.n_xspec;
  if (p.i_grid == p.i_grid_old && p.i_grid > P.i_grid_feb && n_xspec == 0) return true;

  const double aa = a.aa;
  double ptot_sk, px, py, pz, gam_sk;
  transform_p_PS(aa, p.pb_pf, p.p_perp, p.gam_pf, p.phi, p.ux, p.gsf, p.bcos, p.bsin, ptot_sk, px, py, pz, gam_sk);
  const double m = aa * MP_;
  double pt_o_px_sk, abs_inv_vx;
  if (ptot_sk > __builtin_fabs(px * MCS_SPIKE_AWAY)) {
    pt_o_px_sk = MCS_SPIKE_AWAY;
    abs_inv_vx = __builtin_fabs(MCS_SPIKE_AWAY / p.ux);
  } else {
    pt_o_px_sk = ptot_sk / px;
    abs_inv_vx = __builtin_fabs(gam_sk * aa * MP_ / px);
  }
  double pt_o_px_pf = __builtin_fabs(p.ptot_pf / p.pb_pf);
  if (!(pt_o_px_pf < MCS_SPIKE_AWAY)) pt_o_px_pf = MCS_SPIKE_AWAY;
  double eadd;
  if ((gam_sk - 1) > MCS_E_REL_PT) eadd = (gam_sk - 1) * m * (CC_ * CC_) * p.weight;
  else eadd = ptot_sk * ptot_sk / (2 * m) * p.weight;

  if (n_xspec > 0) {   // all_flux.jl:164-190
    const int i_pt = bin_momentum(a, ptot_sk);
    const int i_pt_pf = bin_momentum(a, p.ptot_pf);
    for (int i = 0; i < n_xspec; ++i) {
      const double xs = a.tb.x_spec[i];
      if ((p.x_old < xs && p.x >= xs) || (p.x <= xs && p.x_old > xs)) {
        tadd(a, a.L.spectra_sf + i_pt + (long long)(MCS_PSD_MAX + 1) * i, p.weight * pt_o_px_sk);
        const double Fw = __builtin_fabs(p.pb_pf / px) * (gam_sk / p.gam_pf);
        tadd(a, a.L.spectra_pf + i_pt_pf + (long long)(MCS_PSD_MAX + 1) * i, p.weight * pt_o_px_pf * Fw);
      }
    }
  }

  // F_stream! (all_flux.jl:197-259)
  const bool down = p.x > p.x_old;
  const int i_first = down ? p.i_grid_old + 1 : p.i_grid_old;
  const int i_last = down ? p.i_grid : p.i_grid + 1;
  const int step = down ? 1 : -1;
  const int sign_fac = down ? 1 : -1;
  const bool inj_check = !down;
  const int ng = P.n_grid;
  int i_pt = 0, jth = 0;
  if (p.inj) { i_pt = bin_momentum(a, ptot_sk); jth = bin_angle(a, px, ptot_sk); }
  const double f_pxx = sign_fac * px * p.weight * P.gam0 * P.u0;
  const double f_pxz = __builtin_fabs(pz) * p.weight * P.gam0 * P.u0;
  const double f_en = sign_fac * eadd * P.gam0 * P.u0;
  const double tw = p.weight * abs_inv_vx;
  int k_sf = 0, j_sf = 0;
  bool have_sf = false;
  for (int i = i_first; down ? i <= i_last : i >= i_last; i += step) {
    if (inj_check && p.inj && i <= P.i_grid_feb) continue;
    ladd_f64(&s.fl[i - 1], f_pxx);
    ladd_f64(&s.fl[ng + i - 1], f_pxz);
    ladd_f64(&s.fl[2 * ng + i - 1], f_en);
    if (p.inj) {
      tadd(a, a.L.psd + i_pt + a.L.psd_stride_tht * jth + a.L.psd_stride_zone * (long long)(i - 1), tw);
    } else {
      if (P.track_thermal) {   // A9: bin the thermal crossing instead of appending to a list
        if (!have_sf) { k_sf = bin_momentum(a, ptot_sk); j_sf = bin_angle(a, px, ptot_sk); have_sf = true; }
        tadd(a, a.L.therm_sf + k_sf + a.L.psd_stride_tht * j_sf + a.L.psd_stride_zone * (long long)(i - 1), tw);
        const double gam = s.gsf[i];
        const double beta = s.ux[i] / CC_;
        const double E0 = a.m * CC_ * CC_;
        const double pc = ptot_sk * CC_;
        const double etot = __builtin_sqrt(pc * pc + E0 * E0);
        double px_Xf = gam * (px - beta * etot / CC_);
        const double pt_Xf = __builtin_sqrt((ptot_sk * ptot_sk - px * px) + px_Xf * px_Xf);
        if (__builtin_fabs(px_Xf) > pt_Xf) px_Xf = __builtin_copysign(pt_Xf, px_Xf);
        const int k_pf = bin_momentum(a, pt_Xf);
        const int j_pf = bin_angle(a, px_Xf, pt_Xf);
        tadd(a, a.L.therm_pf + k_pf + a.L.psd_stride_tht * j_pf + a.L.psd_stride_zone * (long long)(i - 1), tw);
      }
      ladd_i32(&s.nc[i - 1], 1);
    }
  }
  if (p.inj && p.x < P.feb_upstream && p.x_old >= P.feb_upstream) {
    tadd(a, a.L.scalars + 3, eadd * P.gam0 * P.u0);
    tadd(a, a.L.scalars + 2, -(px * p.weight * P.gam0 * P.u0));
  }
  return true;
}

// src/prob_return.jl:217-344 (with D1: the scattered pitch is kept)
__device__ __forceinline__ void retro_time(const KArgs& a, const Lds& s, Rng& rng, Pt& p, bool& lose_pt) {
  const mcs_params& P = a.P;
  const int ng = P.n_grid;
  const double aa = a.aa;
  const double xn_per = MCS_RETRO_XN_PER;
  const double phi_step = TWOPI_ / xn_per;
  const double t_step_fac = TWOPI_ * aa * MP_ * CC_ * p.gyro_denom / xn_per;
  const double ux_sk = -s.ux[ng];
  const double gsf = s.gsf[ng];
  const double gef = s.gef[ng];
  double B = s.bt[ng];
  if (P.use_custom_epsB) B *= __builtin_sqrt(P.x_grid_stop / p.prp);
  const double bcos = s.bcos[ng], bsin = s.bsin[ng];
  const double B_CMB_loc = P.B_CMBz * gef;
  double B2_tot = B * B + B_CMB_loc * B_CMB_loc;
  lose_pt = false;
  double x_PT = p.prp;
  p.phi = rng.rand() * TWOPI_;
  const int n_tcuts = a.tb.n_tcuts;
  const double mc = aa * MP_ * CC_;
  while (true) {
    ++p.n_retro;
    const double x_PT_old = x_PT;
    const double phi_old = p.phi;
    if (P.use_custom_epsB) {
      B = s.bt[ng] * __builtin_sqrt(P.x_grid_stop / x_PT);
      B2_tot = B * B + B_CMB_loc * B_CMB_loc;
      p.gyro_denom = 1 / (a.zzq * B);
    }
    const double gyro_rad = p.p_perp * CC_ * p.gyro_denom;
    p.phi = mcsm::mod2pi(phi_old + phi_step);
    const double t_step = t_step_fac * p.gam_pf;
    const double x_move = p.pb_pf * t_step_fac / (aa * MP_);
    double gyr = 0.0;
    if (bsin != 0.0) gyr = gyro_rad * bsin * (mcsm::cos(p.phi) - mcsm::cos(phi_old));
    x_PT = x_PT_old + gsf * (x_move * bcos - gyr + ux_sk * t_step);
    p.acctime += t_step * gef;
    if (P.do_tcuts) {
      if (p.tcut > n_tcuts) cnt(a, MCS_IC_TCUT_OVERRUN);
      else if (p.acctime >= a.tb.tcuts[p.tcut - 1]) { tcut_track(a, p.tcut, p.weight, p.ptot_pf); p.tcut += 1; }
    }
    p.phi = TWOPI_ * rng.rand();
    const double ptot_old = p.ptot_pf;
    p.pb_pf = (2 * rng.rand() - 1) * p.ptot_pf;
    double arg = p.ptot_pf * p.ptot_pf - p.pb_pf * p.pb_pf;
    if (arg < 0) arg = 0;
    p.p_perp = __builtin_sqrt(arg);
    const double cos_new = p.pb_pf / ptot_old;
    const double sin_new = p.p_perp / ptot_old;
    if (P.do_rad_losses && aa < 1) p.ptot_pf = radiation_loss(B2_tot, p.ptot_pf, t_step);
    if (p.ptot_pf <= 0) {
      p.ptot_pf = MCS_FLOOR; p.gam_pf = 1.0; lose_pt = true;
      break;
    } else {
      p.pb_pf = p.ptot_pf * cos_new;
      p.p_perp = p.ptot_pf * sin_new;
      p.gam_pf = mcsm::hypot1(p.ptot_pf / mc);
    }
    if (x_PT < p.prp) break;
  }
}

// src/prob_return.jl:36-173
__device__ __forceinline__ void prob_return(const KArgs& a, const Lds& s, Rng& rng, Pt& p, bool& lose_pt) {
  const mcs_params& P = a.P;
  const double aa = a.aa;
  p.i_return = 2;
  lose_pt = false;
  if (p.x < P.x_grid_stop) {
  } else if (p.x_old < P.x_grid_stop && P.x_grid_stop <= p.x) {
    double gyro_tmp;
    if (P.use_custom_epsB && p.x > P.x_grid_stop) gyro_tmp = __builtin_sqrt(P.x_grid_stop / p.x); else gyro_tmp = 1.0;
    const double grt = p.ptot_pf * CC_ * gyro_tmp / (MCS_QCGS * P.bmag2);
    const double L_diff = P.eta_mfp / 3 * grt * p.ptot_pf / (aa * MP_ * p.gam_pf * P.u2);
    p.prp = p.x + 3 * L_diff;
  } else if (p.x_old < p.prp && p.x >= p.prp) {
    const double vt = p.ptot_pf / (p.gam_pf * aa * MP_);
    const double q = (vt - P.u2) / (vt + P.u2);
    const double prob_ret = q * q;
    if (vt < P.u2 || rng.rand() > prob_ret) {
      p.i_return = 0;
    } else {
      p.i_return = 1;
      retro_time(a, s, rng, p, lose_pt);
      if (lose_pt) p.i_return = 0;
      p.x = p.prp;
    }
  } else {
    if (aa < 1 && p.ptot_pf < a.pcut_prev && p.helix % 1000 == 0) {
      const double grt = p.ptot_pf * CC_ * p.gyro_denom;
      const double L_diff = P.eta_mfp / 3 * grt * p.ptot_pf / (aa * MP_ * p.gam_pf * P.u2);
      if (p.x > 2.0e3 * L_diff) {
        p.prp = 0.8 * p.x;
      } else {
        const double r = a.pcut_prev / p.ptot_pf;
        const double r2 = r * r;
        const double alt = P.x_grid_stop + L_diff * (r2 * r2 * r);
        p.prp = p.prp < alt ? p.prp : alt;
      }
    }
  }
}

// src/particle_loop.jl:652-723
__device__ __forceinline__ void do_energy_transfer(const KArgs& a, Pt& p) {
  const mcs_params& P = a.P;
  const int i_start = p.i_grid_old;
  const int i_stop = p.i_grid < P.i_shock ? p.i_grid : P.i_shock;
  bool scale = false;
  const double m = a.aa * MP_;
  const double E0 = m * (CC_ * CC_);
  double gam_f = p.gam_pf;
  double eps_max = -1e300, recv_max = 0.0;
  for (int i = i_start + 1; i <= i_stop; ++i) {
    if (i < 1 || i > P.n_grid) continue;
    const double e = a.tb.eps_target[i - 1];
    if (e > eps_max) eps_max = e;
    const double r = a.T[a.L.energy_recv_pool + (i - 1)];
    if (r > recv_max) recv_max = r;
  }
  if (a.aa >= 1 && eps_max > 0) {
    const double gam_i = mcsm::hypot1(p.ptot_pf / a.mc);
    const double eps_stop = a.tb.eps_target[i_stop - 1];
    const double eps_start = i_start >= 1 ? a.tb.eps_target[i_start - 1] : 0.0;
    gam_f = 1 + (gam_i - 1) * (1 - eps_stop) / (1 - eps_start);
    int n_split = 0;
    for (int i = i_start + 1; i <= i_stop; ++i) if (a.tb.eps_target[i - 1] > 0) ++n_split;
    const double inc = (gam_i - gam_f) * E0 * p.weight / n_split;
    for (int i = i_start + 1; i <= i_stop; ++i)
      if (a.tb.eps_target[i - 1] > 0) tadd(a, a.L.energy_transfer_pool + (i - 1), inc);
    scale = true;
  } else if (recv_max > 0) {
    double sum = 0.0;
    for (int i = i_start + 1; i <= i_stop; ++i) sum += a.T[a.L.energy_recv_pool + (i - 1)];
    const double e_tr = sum * a.ewf;
    const double gam_i = mcsm::hypot1(p.ptot_pf / a.mc);
    gam_f = gam_i + e_tr / E0;
    scale = true;
  }
  if (scale) {
    const double ptot_f = a.mc * __builtin_sqrt(gam_f * gam_f - 1);
    const double sf = ptot_f / p.ptot_pf;
    p.pb_pf *= sf;
    p.p_perp *= sf;
    p.ptot_pf = ptot_f;
    p.gam_pf = gam_f;
  }
}

// src/particle_finish.jl:46-107 (with D2)
__device__ __forceinline__ void particle_finish(const KArgs& a, const Pt& p, int i_reason) {
  const double aa = a.aa;
  const double m = aa * MP_;
  const double E0 = m * (CC_ * CC_);
  double ptot_sk, px, py, pz, gam_sk;
  transform_p_PS(aa, p.pb_pf, p.p_perp, p.gam_pf, p.phi, p.ux, p.gsf, p.bcos, p.bsin, ptot_sk, px, py, pz, gam_sk);
  const int ip = bin_momentum(a, ptot_sk);
  const int jth = bin_angle(a, px, ptot_sk);
  double wf;
  if (ptot_sk > __builtin_fabs(MCS_SPIKE_AWAY * px)) wf = gam_sk * m * MCS_SPIKE_AWAY / ptot_sk;
  else wf = gam_sk * (m / __builtin_fabs(px));
  const long long pm = MCS_PSD_MAX + 1;
  const int ion = a.i_ion - 1, iter = a.i_iter - 1;
  if (i_reason == 1) {
    tadd(a, a.L.esc_psd_down + ip + pm * jth, p.weight * wf);
  } else if (i_reason == 2) {
    tadd(a, a.L.esc_flux + ion, p.weight);
    tadd(a, a.L.esc_psd_up + ip + pm * jth, p.weight * wf);
    const bool rel = (gam_sk - 1) >= MCS_E_REL_PT;
    const double E_kin = rel ? (gam_sk - 1) * E0 : ptot_sk * ptot_sk / (2 * m);
    const double eadd = E_kin * p.weight;
    tadd(a, a.L.px_esc_feb + ion + (long long)a.P.n_ions * iter, __builtin_fabs(px) * p.weight);
    tadd(a, a.L.energy_esc_feb + ion + (long long)a.P.n_ions * iter, eadd);
    tadd(a, a.L.esc_energy_eff + ip + pm * ion, eadd);
    tadd(a, a.L.esc_num_eff + ip + pm * ion, p.weight);
  }
}

// load a particle and run the prologue of particle_loop (src/particle_loop.jl:44-153)
__device__ __forceinline__ void load_particle(const KArgs& a, const Lds& s, long long k, Pt& p, Rng& rng) {
  p.weight = a.in.weight[k];
  p.ptot_pf = a.in.ptot_pf[k];
  p.pb_pf = a.in.pb_pf[k];
  p.x = a.in.x_PT_cm[k];
  p.xn_per = a.in.xn_per[k];
  p.prp = a.in.prp_x_cm[k];
  p.acctime = a.in.acctime_sec[k];
  p.phi = a.in.phi_rad[k];
  const uint32_t meta = a.in.meta[k];
  p.i_grid = (int)(meta & 0xffffu);
  p.tcut = (int)((meta >> 16) & 0xffu);
  p.downstream = (meta >> 24) & 1u;
  p.inj = (meta >> 25) & 1u;
  p.i_grid_old = p.i_grid;
  p.helix = 0; p.n_retro = 0;
  const unsigned long long key = a.seed_base + (unsigned long long)(a.i_prt_offset + k + 1);
  rng.k0 = (uint32_t)key; rng.k1 = (uint32_t)(key >> 32); rng.n = 0; rng.spare = 0.0;

  p.gam_pf = mcsm::hypot1(p.ptot_pf / a.mc);
  p.gyro_denom = s.gd[p.i_grid];
  if (a.P.use_custom_epsB && p.x > a.P.x_grid_stop) p.gyro_denom *= __builtin_sqrt(p.x / a.P.x_grid_stop);
  p.gyro_rad_tot = p.ptot_pf * CC_ * p.gyro_denom;
  p.gyro_period = TWOPI_ * p.gam_pf * a.m * CC_ * p.gyro_denom;
  p.ux = s.ux[p.i_grid]; p.uz = s.uz[p.i_grid]; p.ut = s.ut[p.i_grid];
  p.gsf = s.gsf[p.i_grid]; p.gef = s.gef[p.i_grid];
  p.bsin = s.bsin[p.i_grid]; p.bcos = s.bcos[p.i_grid];
  p.i_return = -1;
  p.t_step = 0.0;
  p.p_perp = perpendicular_momentum(a, p.ptot_pf, p.pb_pf);
  p.gyro_rad = p.p_perp * CC_ * p.gyro_denom;
  p.x_old = 0.0;
}

// One pass of the helix loop (src/particle_loop.jl:154-499).  Returns -1 while the
// particle lives, else the end code: 0 = saved for the next pcut, 1..4 = i_reason.
__device__ __forceinline__ int helix_step(const KArgs& a, const Lds& s, Rng& rng, Pt& p) {
  const mcs_params& P = a.P;
  const double aa = a.aa;
  p.helix += 1;
  if (p.helix > MCS_HELIX_CAP) { cnt(a, MCS_IC_HELIX_CAP); return 1; }

  if (p.i_return == 1) {
    p.p_perp = perpendicular_momentum(a, p.ptot_pf, p.pb_pf);
    p.gyro_rad = p.p_perp * CC_ * p.gyro_denom;
  } else {
    // ---- Code Block 3
    const double ux_o = p.ux, uz_o = p.uz, ut_o = p.ut, gsf_o = p.gsf, bsin_o = p.bsin, bcos_o = p.bcos;
    const int ig = p.i_grid;
    p.ux = s.ux[ig]; p.uz = s.uz[ig]; p.ut = s.ut[ig]; p.gsf = s.gsf[ig]; p.gef = s.gef[ig];
    p.bsin = s.bsin[ig]; p.bcos = s.bcos[ig];
    double bmag = s.bt[ig];
    if (P.use_custom_epsB && p.x > P.x_grid_stop) {
      bmag = s.bt[P.n_grid] * __builtin_sqrt(P.x_grid_stop / p.x);
      p.gyro_denom = 1 / (a.zzq * bmag);
    } else {
      p.gyro_denom = s.gd[ig];            // == 1/(zz*btot[ig]), tabulated per zone
    }
    if (p.ux != ux_o) {
      transform_p_PSP(a, p, ux_o, uz_o, ut_o, gsf_o, bcos_o, bsin_o);
      p.gyro_rad = p.p_perp * CC_ * p.gyro_denom;
      p.gyro_rad_tot = p.ptot_pf * CC_ * p.gyro_denom;
    }
    if (P.energy_transfer_frac > 0 && !p.inj && p.x_old <= 0 && p.i_grid_old != p.i_grid) do_energy_transfer(a, p);

    if (P.dont_scatter && p.x > 10 * p.gyro_rad) { p.i_return = 0; return 1; }
    if (p.ptot_pf > a.pmax_cutoff) {
      double ptot_sk, px, py, pz, gam_sk;
      transform_p_PS(aa, p.pb_pf, p.p_perp, p.gam_pf, p.phi, p.ux, p.gsf, p.bcos, p.bsin, ptot_sk, px, py, pz, gam_sk);
      if (ptot_sk > a.pmax_cutoff) return 2;
    }
    if (p.inj && p.x < P.feb_upstream) return 2;
    if (P.age_max > 0 && p.acctime > P.age_max) return 3;

    if (P.do_rad_losses && aa < 1) {
      const double ptot_old = p.ptot_pf;
      const double B_CMB_loc = P.B_CMBz * p.gef;
      p.ptot_pf = radiation_loss(bmag * bmag + B_CMB_loc * B_CMB_loc, p.ptot_pf, p.t_step);
      if (p.ptot_pf <= 0) {
        p.ptot_pf = MCS_FLOOR; p.pb_pf = MCS_FLOOR; p.p_perp = MCS_FLOOR; p.gam_pf = 1;
        return 4;
      }
      p.gam_pf = mcsm::hypot1(p.ptot_pf / a.mc);
      p.pb_pf *= p.ptot_pf / ptot_old;
      p.p_perp *= p.ptot_pf / ptot_old;
      p.gyro_rad_tot = p.ptot_pf * CC_ * p.gyro_denom;
      p.gyro_rad = p.p_perp * CC_ * p.gyro_denom;
    }

    if (!P.dont_scatter) scattering(a, rng, p);

    if (p.downstream) {
      p.acctime += p.t_step * p.gef;
      if (P.do_tcuts) {
        if (p.tcut > a.tb.n_tcuts) cnt(a, MCS_IC_TCUT_OVERRUN);
        else if (p.acctime >= a.tb.tcuts[p.tcut - 1]) { tcut_track(a, p.tcut, p.weight, p.ptot_pf); p.tcut += 1; }
      }
      if (p.ptot_pf > a.pcut) return 0;   // saved for the next pcut (particle_loop.jl:361-380)
    }
    p.xn_per = p.x > p.gyro_rad_tot ? P.xn_per_coarse : P.xn_per_fine;
  }

  // ---- Code Block 2
  p.x_old = p.x;
  const double phi_old = p.phi;
  p.t_step = p.gyro_period / p.xn_per;
  // no_DSA_loop (particle_loop.jl:510-571)
  {
    const double m = aa * MP_;
    while (true) {
      p.phi = mcsm::mod2pi(p.phi + TWOPI_ / p.xn_per);
      const double x_move = p.pb_pf * p.t_step / (p.gam_pf * m);
      double gyr = 0.0;   // gyro_rad*b_sin*(...) is exactly +-0 for a parallel field (b_sin == 0)
      if (p.bsin != 0.0) gyr = p.gyro_rad * p.bsin * (mcsm::cos(p.phi) - mcsm::cos(phi_old));
      const double dx = p.gsf * (x_move * p.bcos - gyr + p.ux * p.t_step);
      p.x = p.x_old + dx;
      if (p.x <= 0 && p.x_old > 0 && !p.inj && (P.dont_DSA || a.inj_frac < 1)) {
        if (P.dont_DSA || (rng.rand() > a.inj_frac)) {
          if (p.pb_pf < 0) p.pb_pf = -p.pb_pf; else p.phi = rng.rand() * TWOPI_;
        } else break;
      } else break;
    }
  }
  if (p.x_old < 0 && p.x >= 0) {
    p.downstream = true;
    const double L_diff = P.eta_mfp / 3 * p.gyro_rad_tot * p.ptot_pf / (a.m * p.gam_pf * P.u2);
    p.prp = p.prp > L_diff ? p.prp : L_diff;
  }
  if (p.downstream && p.x < 0) p.inj = true;

  if (!all_flux(a, s, p)) return 3;

  // downstream_test (particle_loop.jl:595-637)
  bool do_prob_ret = true;
  if (P.feb_downstream > 0 && p.x > P.feb_downstream) {
    p.i_return = 0; do_prob_ret = false;
  } else if (p.x > 1.1 * p.prp) {
    const double m = aa * MP_;
    double v_fac;
    if (aa < 1 && p.ptot_pf < P.pe_crit) {
      const double gyro_fac = P.pe_crit * CC_ * p.gyro_denom;
      v_fac = gyro_fac * P.pe_crit / (m * P.game_crit * P.u2);
    } else {
      v_fac = p.gyro_rad_tot * p.ptot_pf / (m * p.gam_pf * P.u2);
    }
    const double L_diff = P.eta_mfp / 3 * v_fac;
    if (p.x > 6.91 * L_diff) { p.i_return = 0; do_prob_ret = false; }
  }
  bool lose_pt = false;
  if (do_prob_ret) prob_return(a, s, rng, p, lose_pt);

  if (p.i_return == 0) {
    double vel = p.ptot_pf / a.m;
    if ((p.gam_pf - 1) >= MCS_E_REL_PT) vel /= p.gam_pf;
    tadd(a, a.L.scalars + 0, p.ptot_pf / 3 * vel * p.weight * a.density);
    tadd(a, a.L.scalars + 1, (p.gam_pf - 1) * a.m * (CC_ * CC_) * p.weight * a.density);
    return lose_pt ? 4 : 1;
  }
  return -1;
}

}  // namespace

extern "C" __global__ void __launch_bounds__(256)
mcs_k_transport(KArgs a) {
  extern __shared__ double smem[];
  const int ne = a.P.n_grid + 2, ng = a.P.n_grid;
  Lds s;
  s.x = smem; s.ux = s.x + ne; s.uz = s.ux + ne; s.ut = s.uz + ne; s.gsf = s.ut + ne; s.gef = s.gsf + ne;
  s.bt = s.gef + ne; s.bsin = s.bt + ne; s.bcos = s.bsin + ne; s.gd = s.bcos + ne;
  s.fl = s.gd + ne;
  s.nc = (int*)(s.fl + 3 * ng);
  for (int i = threadIdx.x; i < ne; i += blockDim.x) {
    s.x[i] = a.tb.x_grid[i]; s.ux[i] = a.tb.ux[i]; s.uz[i] = a.tb.uz[i]; s.ut[i] = a.tb.utot[i];
    s.gsf[i] = a.tb.gsf[i]; s.gef[i] = a.tb.gef[i];
    const double bt = a.tb.btot[i], th = a.tb.theta[i];
    s.bt[i] = bt;
    double sn, cs;
    mcsm::sincos(th, &sn, &cs);
    s.bsin[i] = sn; s.bcos[i] = cs;
    s.gd[i] = 1 / (a.zzq * bt);
  }
  for (int i = threadIdx.x; i < 3 * ng; i += blockDim.x) s.fl[i] = 0.0;
  for (int i = threadIdx.x; i < ng; i += blockDim.x) s.nc[i] = 0;
  __syncthreads();

  Pt p;
  Rng rng;
  bool active = false, exhausted = false;
  long long k = -1;
  unsigned long long c_helix = 0, c_retro = 0, c_draws = 0;
  const unsigned lane = __lane_id();

  for (;;) {
    // ---- refill idle lanes (wave-aggregated claim)
    const unsigned long long idle = __ballot(!active);
    if (idle != 0ull && !exhausted) {
      const int nidle = __popcll(idle);
      const int leader = __ffsll((long long)idle) - 1;
      unsigned long long base = 0;
      if ((int)lane == leader) base = atomicAdd(a.work_counter, (unsigned long long)nidle);
      base = __shfl(base, leader);
      if (base >= (unsigned long long)a.n) {
        exhausted = true;
      } else if (!active) {
        const int rank = __popcll(idle & ((1ull << lane) - 1ull));
        const unsigned long long idx = base + (unsigned long long)rank;
        if (idx < (unsigned long long)a.n) {
          k = (long long)idx;
          load_particle(a, s, k, p, rng);
          active = true;
        }
      }
    }
    if (__ballot(active) == 0ull) {
      if (exhausted) break;
      continue;
    }
    if (active) {
      const int end = helix_step(a, s, rng, p);
      if (end >= 0) {
        const int steps = p.helix > MCS_HELIX_CAP ? MCS_HELIX_CAP : p.helix;
        c_helix += (unsigned long long)steps; c_retro += (unsigned long long)p.n_retro; c_draws += rng.n;
        if (end == 0) {
          a.l_save[k] = 1;
          a.sv.weight[k] = p.weight; a.sv.ptot_pf[k] = p.ptot_pf; a.sv.pb_pf[k] = p.pb_pf; a.sv.x_PT_cm[k] = p.x;
          a.sv.xn_per[k] = p.xn_per;
          a.sv.prp_x_cm[k] = p.x < p.prp ? p.prp : p.x * 1.1;   // quirk Q7
          a.sv.acctime_sec[k] = p.acctime; a.sv.phi_rad[k] = p.phi;
          a.sv.meta[k] = mcs_pack_meta(p.i_grid, p.tcut, p.downstream, p.inj);
          gadd_u64(a.n_saved, 1ull);
        } else {
          particle_finish(a, p, end);
        }
        cnt(a, MCS_IC_REASON0 + end);
        if (a.f_reason) {
          a.f_reason[k] = end; a.f_helix[k] = p.helix; a.f_retro[k] = p.n_retro; a.f_ptot[k] = p.ptot_pf; a.f_x[k] = p.x;
        }
        active = false;
      }
    }
  }

  // ---- flush per-lane counters (wave reduce) and the LDS staging
  for (int off = 32; off > 0; off >>= 1) {
    c_helix += __shfl_down(c_helix, off);
    c_retro += __shfl_down(c_retro, off);
    c_draws += __shfl_down(c_draws, off);
  }
  if (lane == 0) {
    if (c_helix) cnt(a, MCS_IC_STEPS_HELIX, c_helix);
    if (c_retro) cnt(a, MCS_IC_STEPS_RETRO, c_retro);
    if (c_draws) cnt(a, MCS_IC_RNG_DRAWS, c_draws);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ng; i += blockDim.x) {
    const double v0 = s.fl[i], v1 = s.fl[ng + i], v2 = s.fl[2 * ng + i];
    if (v0 != 0.0) gadd_f64(&a.T[a.L.pxx_flux + i], v0);
    if (v1 != 0.0) gadd_f64(&a.T[a.L.pxz_flux + i], v1);
    if (v2 != 0.0) gadd_f64(&a.T[a.L.energy_flux + i], v2);
    const int c = s.nc[i];
    if (c) gadd_u64(&a.I[MCS_I_NUM_CROSSINGS + i], (unsigned long long)c);
  }
}

extern "C" size_t mcs_transport_smem_bytes(int n_grid) {
  const int ne = n_grid + 2;
  return (size_t)(10 * ne + 3 * n_grid) * sizeof(double) + (size_t)n_grid * sizeof(int);
}

extern "C" hipError_t mcs_launch_transport(const KArgs* a, int blocks, int threads, hipStream_t st) {
  const size_t sm = mcs_transport_smem_bytes(a->P.n_grid);
  hipLaunchKernelGGL(mcs_k_transport, dim3(blocks), dim3(threads), sm, st, *a);
  return hipGetLastError();
}
