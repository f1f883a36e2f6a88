// mcs_consumers.hip -- K4: the immediate consumers of the PSD tallies, on the device.
//
//   mcs_k_dndp_cr  : get_dNdp_cr (src/particle_counter.jl:29-306) incl. get_transform_dN /
//                    triangular_distribution! / transform_psd_corners (src/transformers.jl:29-312,
//                    634-682) and identify_corners (src/identify_corners.jl), followed by the CR
//                    normalisation of get_normalized_dNdp (src/particle_counter.jl:733-790);
//   mcs_k_thermo   : thermo_calcs (src/thermo_calcs.jl:30-352).
//
// Both read the 22 MB psd where it already is (HBM, written by K1) and hand back O(n_grid x bins)
// numbers, so the 66 MB tally buffer never crosses PCIe for them.  One workgroup owns one grid
// zone: the zone's psd slab (nmom+2)(ntht+2) fp64 is streamed once, momentum-fastest =
// coalesced; the dN(p) histograms of the three frames are accumulated in LDS (ds_add_f64) and
// written once.  HBM-bound: 8 B per PSD cell per kernel (+ 16 B per cell for the thermo scratch).
// The reference's consumer quirks C1-C6 are handled as DESIGN.md section 3b says.
#include "mcs_device.h"
#include "../../include/mcs_math.h"
#include "../../include/mcs_synch.h"
#include "../../include/mcs_ic.h"
#include "../../include/mcs_pion.h"

#pragma clang fp contract(off)

namespace {

#define KC_MAXB 208            // >= psd_max + 2 bins per axis

__device__ __forceinline__ int c_bin_mom(const mcs_params& P, double p) {
  int b = p < P.psd_mom_min ? 0 : (int)trunc(mcsm::log10(p / P.psd_mom_min) * P.psd_bins_per_dec_mom) + 1;
  return b > P.num_psd_mom_bins ? P.num_psd_mom_bins : b;
}
__device__ __forceinline__ int c_bin_ang(const mcs_params& P, double px, double pt) {
  if (pt == 0.0) return 0;
  const double c = -px / pt;
  int b;
  if (c < P.psd_cos_fine) b = P.num_psd_tht_bins - (int)trunc((c + 1) / P.psd_dcos);
  else {
    const double th = mcsm::acos(c);
    b = th < P.psd_tht_min ? 0 : (int)trunc(mcsm::log10(th / P.psd_tht_min) * P.psd_bins_per_dec_tht) + 1;
  }
  return b < P.num_psd_tht_bins ? b : P.num_psd_tht_bins;
}

// one transformed corner (src/transformers.jl:662-676): log10 of the momentum and the cosine
__device__ __forceinline__ void c_corner(double gam, double beta, double E0, double p_edge, double cos_edge, double& lpt, double& ct) {
  const double px = p_edge * cos_edge;
  const double pc = p_edge * MCS_C;
  const double etot = __builtin_sqrt(pc * pc + E0 * E0);
  const double pxt = gam * (px - beta * etot / MCS_C);
  const double ptt = __builtin_sqrt(p_edge * p_edge + pxt * pxt - px * px);
  lpt = mcsm::log10(ptt);
  ct = pxt / ptt;
}

// identify_corners (src/identify_corners.jl): lowest / highest momentum corner, then the lower /
// higher cosine of the other two, with the tie rules; false on the reference's error() paths.
__device__ bool c_identify(const double pts[4], const double cts[4], double& lo_pt, double& hi_pt, double& clo_pt, double& chi_pt) {
  int i_lo = 0, i_hi = 0;
  for (int q = 1; q < 4; ++q) { if (pts[q] < pts[i_lo]) i_lo = q; if (pts[q] > pts[i_hi]) i_hi = q; }
  int n_lo = 0, n_hi = 0;
  for (int q = 0; q < 4; ++q) { n_lo += pts[q] == pts[i_lo]; n_hi += pts[q] == pts[i_hi]; }
  unsigned mask = 0xFu & ~(1u << i_lo) & ~(1u << i_hi);
  int j_hi = -1, j_lo = -1;
  for (int q = 0; q < 4; ++q) if (((mask >> q) & 1u) && (j_hi < 0 || cts[q] > cts[j_hi])) j_hi = q;
  if (j_hi < 0) return false;
  mask &= ~(1u << j_hi);
  for (int q = 0; q < 4; ++q) if (((mask >> q) & 1u) && (j_lo < 0 || cts[q] < cts[j_lo])) j_lo = q;
  if (j_lo < 0) return false;
  double a_lo_pt = pts[i_lo], a_lo_ct = cts[i_lo], a_hi_pt = pts[i_hi], a_hi_ct = cts[i_hi];
  double b_hi_pt = pts[j_hi], b_hi_ct = cts[j_hi], b_lo_pt = pts[j_lo], b_lo_ct = cts[j_lo];
  if (b_hi_ct == b_lo_ct) {
    if (b_hi_pt < b_lo_pt) { b_hi_pt = pts[j_lo]; b_hi_ct = cts[j_lo]; b_lo_pt = pts[j_hi]; b_lo_ct = cts[j_hi]; }
    else if (!(b_hi_pt > b_lo_pt)) return false;
  }
  if (n_lo > 1) {
    if (a_lo_pt == b_lo_pt) {
      if (a_lo_ct > b_lo_ct) { a_lo_pt = pts[j_lo]; a_lo_ct = cts[j_lo]; b_lo_pt = pts[i_lo]; b_lo_ct = cts[i_lo]; }
      else if (!(a_lo_ct < b_lo_ct)) return false;
    } else if (a_lo_pt == b_hi_pt) {
      if (a_lo_ct > b_hi_ct) { a_lo_pt = pts[j_hi]; a_lo_ct = cts[j_hi]; b_hi_pt = pts[i_lo]; b_hi_ct = cts[i_lo]; }
      else if (!(a_lo_ct < b_hi_ct)) return false;
    } else return false;
  }
  if (n_hi > 1) {
    if (a_hi_pt == b_lo_pt) {
      if (a_hi_ct > b_lo_ct) { a_hi_pt = pts[j_lo]; a_hi_ct = cts[j_lo]; b_lo_pt = pts[i_hi]; b_lo_ct = cts[i_hi]; }
      else if (!(a_hi_ct < b_lo_ct)) return false;
    } else if (a_hi_pt == b_hi_pt) {
      if (a_hi_ct > b_hi_ct) { a_hi_pt = pts[j_hi]; a_hi_ct = cts[j_hi]; b_hi_pt = pts[i_hi]; b_hi_ct = cts[i_hi]; }
      else if (!(a_hi_ct < b_hi_ct)) return false;
    } else return false;
  }
  lo_pt = a_lo_pt; hi_pt = a_hi_pt; clo_pt = b_lo_pt; chi_pt = b_hi_pt;
  return true;
}

typedef __attribute__((address_space(3))) double ldbl;
typedef __attribute__((address_space(1))) double gdbl;
__device__ __forceinline__ void c_ladd(double* p, double v) { (void)__builtin_amdgcn_ds_atomic_fadd_f64((ldbl*)p, v); }
__device__ __forceinline__ void c_gadd(double* p, double v) { (void)__builtin_amdgcn_global_atomic_fadd_f64((gdbl*)p, v); }

// triangular_distribution! with i_approx = 2 (src/transformers.jl:209-312), adds into LDS dN
__device__ void c_triangular(double* dN, double p_hi, double p_lo, double clo_pt, double chi_pt, double w, int l_lo, int l_hi,
                             const double* lb, int nmax1) {
  const double length_tot = 1 / (p_hi - p_lo);
  const double ct_height = 2 * w / length_tot;
  double p_bottom = p_lo;
  const double p_peak = (clo_pt + chi_pt) / 2;
  const double p_denom_lo = 1 / (p_peak - p_lo);
  const double p_denom_hi = 1 / (p_hi - p_peak);
  double fractional_area = 0;
  for (int l = l_lo; l <= l_hi; ++l) {
    if (l + 1 > nmax1) break;
    if (p_hi < lb[l_lo + 1]) { c_ladd(&dN[l], w); break; }
    const double top = lb[l + 1];
    if (top <= p_peak) {
      const double p_base = top - p_bottom;
      const double rh = (top - p_lo) * p_denom_lo * ct_height;
      const double lh = p_bottom == p_lo ? 0.0 : (p_bottom - p_lo) * p_denom_lo * ct_height;
      const double part = p_base / 2 * (lh + rh);
      c_ladd(&dN[l], part);
      p_bottom = top;
      fractional_area += part;
      continue;
    }
    if (top < p_hi) {
      const double p_base = p_hi - top;
      const double lh = p_base * p_denom_hi * ct_height;
      const double missing = p_base / 2 * lh;
      const double part = (w - fractional_area) - missing;
      c_ladd(&dN[l], part);
      p_bottom = top;
      fractional_area += part;
      continue;
    }
    c_ladd(&dN[l], w - fractional_area);
    break;
  }
}

struct ConsArgs {
  mcs_params P;
  const double* psd;        // tallies + L.psd
  const double* therm_pf;   // tallies + L.therm_pf
  const unsigned long long* num_crossings;
  const double *gam_sf, *ux;            // device grid tables, n_grid+2
  const double *mom_log, *mom_edge, *cos_edge, *cos_center, *pt_center, *zone_pop, *density_loc, *cold_pressure;
  double rest_energy, mc, n0, gam0;
  int therm_from_hist;
  double* out_dndp;         // [3][n_grid][nmom+2]
  unsigned long long* diag; // [2]
  double* scratch;          // [n_grid][ntht+2][nmom+2]
  double *out_par, *out_perp, *out_edens;
};

__global__ void __launch_bounds__(256) mcs_k_dndp_cr(ConsArgs a) {
  __shared__ double s_lb[KC_MAXB], s_pe[KC_MAXB], s_ce[KC_MAXB];
  __shared__ double s_dn[3][KC_MAXB];
  __shared__ double s_norm[3];
  const mcs_params& P = a.P;
  const int nm = P.num_psd_mom_bins, nt = P.num_psd_tht_bins, ng = P.n_grid;
  const int NM = nm + 2, NT = nt + 2;
  const int k = blockIdx.x + 1;                 // zone, 1-based
  const double* psd = a.psd + (long long)NM * NT * (k - 1);
  for (int i = threadIdx.x; i < NM; i += blockDim.x) { s_lb[i] = a.mom_log[i]; s_pe[i] = a.mom_edge[i]; s_dn[1][i] = 0.0; s_dn[2][i] = 0.0; }
  for (int i = threadIdx.x; i < NT; i += blockDim.x) s_ce[i] = a.cos_edge[i];
  // shock frame (particle_counter.jl:81-85): one thread per momentum bin, theta ascending
  for (int i = threadIdx.x; i < NM; i += blockDim.x) {
    double acc = 0.0;
    for (int j = 0; j < NT; ++j) { const double v = psd[i + NM * j]; if (v > 0) acc += v; }
    s_dn[0][i] = acc;
  }
  __syncthreads();
  // plasma and ISM frames
  const double gam2 = a.gam_sf[k], gam3 = a.gam0;
  const int ncell = (nm + 1) * (nt + 1);
  for (int q = threadIdx.x; q < ncell; q += blockDim.x) {
    const int i = q % (nm + 1), j = q / (nm + 1);
    const double v = psd[i + NM * j];
    if (v < 1.0e-66) continue;
    for (int m = 2; m <= 3; ++m) {
      const double gam = m == 2 ? gam2 : gam3;
      const double beta = gam >= 1.000001 ? __builtin_sqrt(1 - 1 / (gam * gam)) : 0.0;
      const double w = v / gam;
      double pts[4], cts[4];
      c_corner(gam, beta, a.rest_energy, s_pe[i], s_ce[j], pts[0], cts[0]);
      c_corner(gam, beta, a.rest_energy, s_pe[i + 1], s_ce[j], pts[1], cts[1]);
      c_corner(gam, beta, a.rest_energy, s_pe[i], s_ce[j + 1], pts[2], cts[2]);
      c_corner(gam, beta, a.rest_energy, s_pe[i + 1], s_ce[j + 1], pts[3], cts[3]);
      double p_lo, p_hi, clo, chi;
      if (!c_identify(pts, cts, p_lo, p_hi, clo, chi)) { atomicAdd(&a.diag[0], 1ull); continue; }
      // first edge above p_lo, minus one (the reference scans; the edges increase, so a bisection finds the same index in 8
      // dependent LDS reads instead of up to 174: the scan was most of this kernel's time)
      int l_lo = -1;
      {
        int lo = 0, hi = NM;                       // smallest l in [0, NM) with s_lb[l] > p_lo, NM if there is none
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_lb[mid] > p_lo) hi = mid; else lo = mid + 1; }
        if (lo < NM) l_lo = lo - 1;
      }
      if (l_lo < 0) { l_lo = nm; atomicAdd(&a.diag[1], 1ull); }
      int l_hi = -1;
      for (int l = l_lo; l < NM; ++l) if (s_lb[l] >= p_hi) { l_hi = l; break; }
      if (l_hi < 0) { l_hi = nm; atomicAdd(&a.diag[1], 1ull); }
      c_triangular(s_dn[m - 1], p_hi, p_lo, clo, chi, w, l_lo, l_hi, s_lb, nm + 1);
    }
  }
  __syncthreads();
  // dN(p) -> dN/dp (particle_counter.jl:295-304)
  for (int l = threadIdx.x; l <= nm; l += blockDim.x)
    for (int m = 0; m < 3; ++m) {
      const double d = s_dn[m][l];
      s_dn[m][l] = d < 1.0e-66 ? 1.0e-99 : d / (s_pe[l + 1] - s_pe[l]);
    }
  __syncthreads();
  // normalisation (particle_counter.jl:733-790, thermal area == 0: quirk C4); serial = reference order
  if (threadIdx.x < 3) {
    const int m = threadIdx.x;
    double area = 0.0;
    for (int j = 0; j <= nm; ++j) if (s_dn[m][j] > 1.0e-99) area += s_dn[m][j] * (s_pe[j + 1] - s_pe[j]);
    double area_tot;
    if (area > 0) {
      const double density_pf = a.n0 * a.gam0 * a.ux[1] / (a.gam_sf[k] * a.ux[k]);
      area_tot = density_pf / a.ux[k] + area;
    } else area_tot = 0.0 + area;
    s_norm[m] = area_tot > 0 ? a.zone_pop[k - 1] / area_tot : 0.0;
  }
  __syncthreads();
  for (int l = threadIdx.x; l < NM; l += blockDim.x)
    for (int m = 0; m < 3; ++m) {
      double d = s_dn[m][l];
      if (l <= nm && d > 1.0e-99) d *= s_norm[m];
      a.out_dndp[((long long)m * ng + (k - 1)) * NM + l] = d;
    }
}

// block-wide sum / max of one double per thread (256 threads)
__device__ double c_block_sum(double v, double* s_red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[w] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
__device__ double c_block_max(double v, double* s_red) {
  for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(v, off); v = o > v ? o : v; }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[w] = v;
  __syncthreads();
  double m = s_red[0];
  for (int q = 1; q < 4; ++q) m = s_red[q] > m ? s_red[q] : m;
  return m;
}

__global__ void __launch_bounds__(256) mcs_k_thermo(ConsArgs a) {
  __shared__ double s_red[4];
  const mcs_params& P = a.P;
  const int nm = P.num_psd_mom_bins, nt = P.num_psd_tht_bins;
  const int NM = nm + 2, NT = nt + 2;
  const int i = blockIdx.x + 1;
  const long long slab = (long long)NM * NT;
  const double* psd = a.psd + slab * (i - 1);
  const double* thp = a.therm_pf + slab * (i - 1);
  double* d2 = a.scratch + slab * (i - 1);
  const unsigned long long ncross = a.num_crossings[i - 1];
  const double E0 = a.rest_energy, mc = a.mc;
  // d2N_pf = 1e-99 (+ thermal crossings, already binned in the plasma frame by K1: A9)
  for (long long q = threadIdx.x; q < slab; q += blockDim.x) d2[q] = a.therm_from_hist ? 1.0e-99 + thp[q] : 1.0e-99;
  __threadfence(); __syncthreads();
  // CR cells: centre-point rebinning into the plasma frame (thermo_calcs.jl:179-211)
  const double gam = a.gam_sf[i], beta = a.ux[i] / MCS_C;
  const int ncell = (nm + 1) * (nt + 1);
  for (int q = threadIdx.x; q < ncell; q += blockDim.x) {
    const int k = q % (nm + 1), j = q / (nm + 1);
    const double w = psd[k + NM * j];
    if (w <= 1.0e-66) continue;
    const double cs = a.cos_center[j], pt = a.pt_center[k];
    const double px = pt * cs;
    const double pc = pt * MCS_C;
    const double et = __builtin_sqrt(pc * pc + E0 * E0);
    const double pxX = gam * (px - beta * et / MCS_C);
    const double ptX = __builtin_sqrt(pt * pt - px * px + pxX * pxX);
    c_gadd(&d2[c_bin_mom(P, ptX) + NM * c_bin_ang(P, pxX, ptX)], w);
  }
  __threadfence(); __syncthreads();
  // normalisation to the zone population (thermo_calcs.jl:213-232)
  double part = 0.0;
  for (long long q = threadIdx.x; q < slab; q += blockDim.x) { const double v = d2[q]; if (v > 1.0e-66) part += v; }
  double norm_fac = c_block_sum(part, s_red);
  if (ncross == 0ull && norm_fac > 0) norm_fac += a.n0 / a.ux[i];
  if (norm_fac > 0) norm_fac = a.zone_pop[i - 1] / norm_fac;
  double ppart = 0.0, mpart = 0.0;
  for (long long q = threadIdx.x; q < slab; q += blockDim.x) {
    double v = d2[q];
    if (v > 1.0e-66) { v *= norm_fac; d2[q] = v; }
    if (v > 1.0e-66) ppart += v;
    mpart = v > mpart ? v : mpart;
  }
  const double pop = c_block_sum(ppart, s_red);
  const double dmax = c_block_max(mpart, s_red);
  __threadfence(); __syncthreads();
  // pressure and energy density (thermo_calcs.jl:258-347)
  const double density_loc = a.density_loc[i - 1];
  double pressure_loc = a.cold_pressure[i - 1];
  double pp0 = 0.0, pq0 = 0.0, ed0 = 0.0, nf = 0.0;
  bool cold_only = false;
  if (dmax < 1.0e-66 && ncross == 0ull) {
    pp0 = 1.0 / 3 * pressure_loc; pq0 = 2.0 / 3 * pressure_loc; ed0 = 1.5 * pressure_loc;
    cold_only = true;
  } else if (ncross == 0ull) {
    pressure_loc *= 1 - pop / a.zone_pop[i - 1];
    pp0 = 1.0 / 3 * pressure_loc; pq0 = 2.0 / 3 * pressure_loc; ed0 = 1.5 * pressure_loc;
    nf = density_loc / a.zone_pop[i - 1];
  } else {
    nf = density_loc / a.zone_pop[i - 1];
  }
  double sp = 0.0, sq = 0.0, se = 0.0;
  if (!cold_only) {
    for (int q = threadIdx.x; q < ncell; q += blockDim.x) {
      const int k = q % (nm + 1), j = q / (nm + 1);
      const double c = d2[k + NM * j];
      if (c < 1.0e-66) continue;
      const double pt = a.pt_center[k];
      const double t = pt / mc;
      const double gtmp = __builtin_sqrt(1 + t * t);
      const double vel = pt * MCS_C / (mc * gtmp);
      const double pfac = 1.0 / 3 * pt * vel * nf;
      const double efac = (gtmp - 1) * E0;
      const double cs = a.cos_center[j];
      sp += c * pfac * (cs * cs);
      sq += c * pfac * (1 - cs * cs);
      se += efac * c * nf;
    }
  }
  sp = c_block_sum(sp, s_red); sq = c_block_sum(sq, s_red); se = c_block_sum(se, s_red);
  if (threadIdx.x == 0) { a.out_par[i - 1] = pp0 + sp; a.out_perp[i - 1] = pq0 + sq; a.out_edens[i - 1] = ed0 + se; }
}



// ---- K6: get_dNdp_2D (src/particle_counter.jl:343-627, called at src/ion_finalize.jl:50-59) and the inverse-Compton fold
// (src/inverse_compton.jl:36-311, called at src/photon_calcs.jl:116-138) -- SURVEY.md 8(f-4) --------------------------------------
// mcs_k_dndp_2d: one workgroup per grid zone.  d2N/dp dcos of the zone in the shock frame = the thermal crossings (binned on the
// fly by K1 into therm_sf: A9) + the psd cells above 1e-66, divided by dp, normalised to the zone population (:376-519); then the
// centre-point rebin of every cell into the frame moving with (gam_x, beta_x) against the shock frame -- the ISM frame, m = 2 of
// :531-598 (the plasma-frame array of the same loop is never read and is not built).  `sf`: the zone's slab of a scratch buffer,
// `ef`: of the output; both [ntht+2][nmom+2], momentum fastest (the psd's own layout).
__global__ void __launch_bounds__(256) mcs_k_dndp_2d(ConsArgs a, const double* __restrict__ therm_sf, double gam_x, double beta_x, double* __restrict__ ef_out) {
  __shared__ double s_red[4];
  __shared__ double s_dp[KC_MAXB];
  const mcs_params& P = a.P;
  const int nm = P.num_psd_mom_bins, nt = P.num_psd_tht_bins;
  const int NM = nm + 2, NT = nt + 2;
  const int i = blockIdx.x + 1;
  const long long slab = (long long)NM * NT;
  const double* psd = a.psd + slab * (i - 1);
  const double* ths = therm_sf + slab * (i - 1);
  double* sf = a.scratch + slab * (i - 1);
  double* ef = ef_out + slab * (i - 1);
  const unsigned long long ncross = a.num_crossings[i - 1];
  const double E0 = a.rest_energy;
  for (int k = threadIdx.x; k <= nm; k += blockDim.x) s_dp[k] = a.mom_edge[k + 1] - a.mom_edge[k];      // Delta p (:376-380; C1: cgs edges)
  __syncthreads();
  // thermal crossings (:430-451), CR cells (:466-472), dN -> dN/dp (:475-480)
  double part = 0.0;
  for (long long q = threadIdx.x; q < slab; q += blockDim.x) {
    const int k = (int)(q % NM), j = (int)(q / NM);
    double v = 1.0e-99;
    if (ncross != 0ull && a.therm_from_hist) v += ths[q];
    if (k <= nm && j <= nt) { const double w = psd[q]; if (w > 1.0e-66) v += w; }
    if (k <= nm && v > 1.0e-66) v /= s_dp[k];
    sf[q] = v;
    ef[q] = 1.0e-99;
    if (v > 1.0e-66) part += v;
  }
  // density of the array and the rescaling to the zone population (:487-519)
  double dens = c_block_sum(part, s_red);
  if (ncross == 0ull && dens > 0) dens += a.n0;
  const double norm = dens > 0 ? a.zone_pop[i - 1] / dens : 0.0;
  __threadfence(); __syncthreads();
  // the centre-point rebin (:548-598)
  const int ncell = (nm + 1) * (nt + 1);
  for (int q = threadIdx.x; q < ncell; q += blockDim.x) {
    const int k = q % (nm + 1), j = q / (nm + 1);
    double v = sf[k + NM * j];
    v = (v > 1.0e-99 && norm > 0) ? v * norm : 1.0e-99;
    if (v <= 1.0e-66) continue;
    const double w = v * s_dp[k];
    const double cs = a.cos_center[j], pt = a.pt_center[k];
    const double px = pt * cs;
    const double pc = pt * MCS_C;
    const double et = __builtin_sqrt(pc * pc + E0 * E0);                  // hypot(ptot c, E0)
    const double pxX = gam_x * (px - beta_x * et / MCS_C);
    const double ptX = __builtin_sqrt(pt * pt - px * px + pxX * pxX);
    const int kX = c_bin_mom(P, ptX), jX = c_bin_ang(P, pxX, ptX);
    c_gadd(&ef[kX + NM * jX], w / s_dp[kX]);
  }
}

// mcs_k_photon_ic: one workgroup per grid zone, one thread per outgoing photon energy; every thread walks the electron momentum
// bins and the incoming photon bins in the reference's order (include/mcs_ic.h), so a spectrum is the same sum in the same order
// on the CPU twin.  The electrons of a momentum bin inside the jet cone are counted once per zone (one thread per bin, angle bins in
// order: photon_IC's conversion to particle counts, inverse_compton.jl:54-61, and the sums of :235-238).
__global__ void __launch_bounds__(256) mcs_k_photon_ic(const double* __restrict__ ef /*[n_grid][NT][NM]*/, const double* __restrict__ p_edge /*[NM]*/,
                                                      const double* __restrict__ field /*alpha_in[n_nu] | n_in[n_nu]*/, int NM, int NT, int j_max, int n_nu,
                                                      int n_photon, double log_min_rm, double bins_per_dec, double mc_e, double beam_area,
                                                      double* __restrict__ out /*[n_grid][n_photon]*/) {
  __shared__ double s_x[KC_MAXB], s_g[KC_MAXB], s_a[MCS_IC_NNU], s_n[MCS_IC_NNU];
  const int zone = blockIdx.x + 1, nm = NM - 2;
  const double* d2 = ef + (long long)(zone - 1) * NM * NT;
  for (int i = threadIdx.x; i <= nm; i += blockDim.x) {
    const double dp = p_edge[i + 1] - p_edge[i];
    double mx = 0.0, sum = 0.0;
    for (int j = 0; j <= j_max; ++j) {
      const double v = d2[i + NM * j];
      const double c = v <= 1.0e-99 ? 1.0e-99 : v * dp;
      mx = c > mx ? c : mx;
      sum += c;
    }
    s_x[i] = mx <= 1.0e-99 ? 0.0 : sum;
    s_g[i] = mcs_ic_gamma(p_edge[i], p_edge[i + 1], mc_e);
  }
  for (int j = threadIdx.x; j < n_nu; j += blockDim.x) { s_a[j] = field[j]; s_n[j] = field[n_nu + j]; }
  __syncthreads();
  for (int k = threadIdx.x; k < n_photon; k += blockDim.x) {
    const double ao = mcs_ic_alpha_out(log_min_rm, bins_per_dec, k);
    out[(long long)(zone - 1) * n_photon + k] = mcs_ic_emis(mcs_ic_fold_one(s_x, s_g, nm + 1, s_a, s_n, n_nu, ao), ao, beam_area);
  }
}

}  // namespace

// Host-callable launchers (pointers are device pointers; tables were uploaded by the caller)
extern "C" hipError_t mcs_launch_dndp_cr(const mcs_params* P, const double* psd, const double* gam_sf, const double* ux,
                                         const double* tabs /*mom_log|mom_edge|cos_edge|zone_pop*/, double rest_energy, double n0,
                                         double gam0, double* out_dndp, unsigned long long* diag, hipStream_t st) {
  ConsArgs a{};
  a.P = *P;
  const int NM = P->num_psd_mom_bins + 2, NT = P->num_psd_tht_bins + 2;
  a.psd = psd; a.gam_sf = gam_sf; a.ux = ux;
  a.mom_log = tabs; a.mom_edge = tabs + NM; a.cos_edge = tabs + 2 * NM; a.zone_pop = tabs + 2 * NM + NT;
  a.rest_energy = rest_energy; a.n0 = n0; a.gam0 = gam0;
  a.out_dndp = out_dndp; a.diag = diag;
  hipLaunchKernelGGL(mcs_k_dndp_cr, dim3(P->n_grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

extern "C" hipError_t mcs_launch_thermo(const mcs_params* P, const double* psd, const double* therm_pf,
                                        const unsigned long long* num_crossings, const double* gam_sf, const double* ux,
                                        const double* tabs /*cos_center|pt_center|zone_pop|density_loc|cold_pressure*/,
                                        double rest_energy, double mc, double n0, int therm_from_hist, double* scratch,
                                        double* out3 /*par|perp|edens, n_grid each*/, hipStream_t st) {
  ConsArgs a{};
  a.P = *P;
  const int NM = P->num_psd_mom_bins + 2, NT = P->num_psd_tht_bins + 2, ng = P->n_grid;
  a.psd = psd; a.therm_pf = therm_pf; a.num_crossings = num_crossings; a.gam_sf = gam_sf; a.ux = ux;
  a.cos_center = tabs; a.pt_center = tabs + NT; a.zone_pop = tabs + NT + NM; a.density_loc = tabs + NT + NM + ng;
  a.cold_pressure = tabs + NT + NM + 2 * ng;
  a.rest_energy = rest_energy; a.mc = mc; a.n0 = n0; a.therm_from_hist = therm_from_hist;
  a.scratch = scratch; a.out_par = out3; a.out_perp = out3 + ng; a.out_edens = out3 + 2 * ng;
  hipLaunchKernelGGL(mcs_k_thermo, dim3(P->n_grid), dim3(256), 0, st, a);
  return hipGetLastError();
}


// ---- K5: the synchrotron fold of the photon post-processing (include/mcs_synch.h) ------------------------
// One workgroup per grid zone, one thread per photon energy; every thread walks the electron momentum bins in the reference's
// order (synch_emission.jl:124-169), so a spectrum is the same sum in the same order on the CPU twin.  O(n_grid x n_photon x nmom)
// evaluations of F(x): 3e6 for the stock binning -- microseconds of the chip; it runs where the dN/dp it reads was made.
__global__ void __launch_bounds__(256) mcs_k_photon_synch(const double* __restrict__ dndp_pf /*[n_grid][NM]*/, const double* __restrict__ p_edge /*[NM]*/,
                                                         const double* __restrict__ btot /*[n_grid+2]*/, int NM, int n_photon, double log_emin_erg,
                                                         double bins_per_dec, double mc, double* __restrict__ out /*[n_grid][n_photon]*/) {
  __shared__ double s_d[KC_MAXB], s_p[KC_MAXB];
  const int zone = blockIdx.x + 1;
  for (int i = threadIdx.x; i < NM; i += blockDim.x) { s_d[i] = dndp_pf[(long long)(zone - 1) * NM + i]; s_p[i] = p_edge[i]; }
  __syncthreads();
  const double B = btot[zone];
  for (int j = threadIdx.x; j < n_photon; j += blockDim.x) {
    const double E = mcs_synch_energy(log_emin_erg, bins_per_dec, j);
    out[(long long)(zone - 1) * n_photon + j] = mcs_synch_fold_one(1.0e-99, s_d, s_p, NM - 2, B, mc, E);     // fill(1e-99), :64
  }
}

extern "C" hipError_t mcs_launch_photon_synch(const double* dndp_pf, const double* p_edge, const double* btot, int n_grid, int NM, int n_photon,
                                              double log_emin_erg, double bins_per_dec, double mc, double* out, hipStream_t st) {
  hipLaunchKernelGGL(mcs_k_photon_synch, dim3((unsigned)n_grid), dim3(256), 0, st, dndp_pf, p_edge, btot, NM, n_photon, log_emin_erg, bins_per_dec, mc, out);
  return hipGetLastError();
}

// ---- K7: the pion-decay fold of the photon post-processing (include/mcs_pion.h) --------------------------------
// One workgroup per grid zone.  First one thread per momentum bin prepares what does not depend on the photon energy (T_p, E_gamma^max,
// A_max, target density x count x speed: pion_kafexhiu.jl:171-193); then one thread per photon energy walks the bins in the reference's
// order, so a spectrum is the same sum in the same order on the CPU twin.  O(n_grid x n_photon x nmom) evaluations of F(T_p, E).
__global__ void __launch_bounds__(256) mcs_k_photon_pion(const double* __restrict__ dndp_pf /*[n_grid][NM]*/, const double* __restrict__ p_edge /*[NM]*/,
                                                        const double* __restrict__ target /*[n_grid]*/, int NM, int n_photon, double log_emin_erg,
                                                        double bins_per_dec, double mc, double aa, double scaling, int i_data,
                                                        double* __restrict__ out /*[n_grid][n_photon]*/) {
  __shared__ double s_pref[KC_MAXB], s_T[KC_MAXB], s_E[KC_MAXB], s_A[KC_MAXB];
  const int zone = blockIdx.x;
  const double n_t = target[zone];
  for (int i = threadIdx.x; i < NM - 1; i += blockDim.x) {
    const double d = dndp_pf[(long long)zone * NM + i];
    const double lo = p_edge[i], hi = p_edge[i + 1];
    const double cnt = d <= 1.0e-99 ? 1.0e-99 : d * (hi - lo);                                   // photon_pion_decay.jl:86-92
    double T = 0, v = 0, E = 1, A = 0;
    const int ok = mcs_pion_bin(cnt, lo, hi, mc, aa, i_data, &T, &v, &E, &A);
    s_pref[i] = ok ? n_t * cnt * v : 0.0; s_T[i] = T; s_E[i] = E; s_A[i] = A;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n_photon; j += blockDim.x) {
    const double e = pow(10.0, log_emin_erg + j * (1.0 / bins_per_dec));
    out[(long long)zone * n_photon + j] = mcs_pion_fold_one(s_pref, s_T, s_E, s_A, NM - 1, i_data, e, scaling);
  }
}
extern "C" hipError_t mcs_launch_photon_pion(const double* dndp_pf, const double* p_edge, const double* target, int n_grid, int NM, int n_photon,
                                             double log_emin_erg, double bins_per_dec, double mc, double aa, double scaling, int i_data, double* out,
                                             hipStream_t st) {
  hipLaunchKernelGGL(mcs_k_photon_pion, dim3((unsigned)n_grid), dim3(256), 0, st, dndp_pf, p_edge, target, NM, n_photon, log_emin_erg, bins_per_dec, mc,
                     aa, scaling, i_data, out);
  return hipGetLastError();
}

extern "C" hipError_t mcs_launch_dndp_2d(const mcs_params* P, const double* psd, const double* therm_sf, const unsigned long long* num_crossings,
                                         const double* tabs /*mom_edge[NM] | cos_center[NT] | pt_center[NM] | zone_pop[ng]*/, double rest_energy, double n0,
                                         int therm_from_hist, double gam_x, double beta_x, double* scratch, double* ef, hipStream_t st) {
  ConsArgs a{};
  a.P = *P;
  const int NM = P->num_psd_mom_bins + 2, NT = P->num_psd_tht_bins + 2;
  a.psd = psd; a.num_crossings = num_crossings;
  a.mom_edge = tabs; a.cos_center = tabs + NM; a.pt_center = tabs + NM + NT; a.zone_pop = tabs + 2 * NM + NT;
  a.rest_energy = rest_energy; a.n0 = n0; a.therm_from_hist = therm_from_hist; a.scratch = scratch;
  hipLaunchKernelGGL(mcs_k_dndp_2d, dim3(P->n_grid), dim3(256), 0, st, a, therm_sf, gam_x, beta_x, ef);
  return hipGetLastError();
}
extern "C" hipError_t mcs_launch_photon_ic(const double* ef, const double* p_edge, const double* field, int n_grid, int NM, int NT, int j_max, int n_nu,
                                           int n_photon, double log_min_rm, double bins_per_dec, double mc_e, double beam_area, double* out, hipStream_t st) {
  hipLaunchKernelGGL(mcs_k_photon_ic, dim3((unsigned)n_grid), dim3(256), 0, st, ef, p_edge, field, NM, NT, j_max, n_nu, n_photon, log_min_rm, bins_per_dec,
                     mc_e, beam_area, out);
  return hipGetLastError();
}
