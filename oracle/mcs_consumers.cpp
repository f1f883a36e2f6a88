// mcs_consumers.cpp -- CPU restatement of the immediate consumers of the PSD tallies
// (SURVEY.md section 8(f-3)): get_dNdp_cr + the CR part of get_normalized_dNdp
// (src/particle_counter.jl:29-306, 674-790), get_transform_dN / triangular_distribution! /
// transform_psd_corners (src/transformers.jl:29-312, 634-682), identify_corners
// (src/identify_corners.jl) and thermo_calcs (src/thermo_calcs.jl:30-352).
//
// *** TEST INFRASTRUCTURE ONLY *** (same rules as mcs_oracle.cpp).  PARITY UNPINNED BY THE
// REFERENCE: it has no vectors for these functions and, as committed, they cannot run at
// all (see C1-C6); the restatement follows the evident design and is pinned by invariants
// (tests/test_consumers.py) and by an independent numpy restatement on small inputs.
//
// Consumer quirks:
//   C1 psd_mom_bounds holds log10(p / m_p c) (src/initializers.jl:220) but every consumer
//      takes exp10(bound) as a cgs momentum.  Here the caller passes the edges in cgs.
//   C2 set_psd_angle_bins sort!s its table (src/initializers.jl:281), which destroys the
//      "theta first, then cosine" order all consumers index by.  The caller passes the true
//      cos(theta) of every edge in the intended (unsorted) order.
//   C3 identify_corners calls maxloc/minloc (Fortran intrinsics; undefined in Julia,
//      src/identify_corners.jl:95,100).  Fortran semantics are used: first index of the
//      max/min among the masked entries.
//   C4 get_dNdp_therm returns early with all-1e-99 arrays ("XXX Early return for debugging",
//      src/particle_counter.jl:991): area_tot_therm == 0, so get_normalized_dNdp always takes
//      the fast-push branch (:755-757).  Replicated.
//   C5 the thermal crossing list is inert (A9); thermo_calcs gets the crossings from the
//      plasma-frame histogram `therm_pf` the transport path fills (what the loop at
//      src/thermo_calcs.jl:136-162 computes), or nothing at all with therm_from_hist = 0.
//   C6 error() paths of identify_corners (identical corners) and cells whose corners leave the
//      table skip the cell and are counted in diag[] instead of aborting.
//   (replicated as written: ct_height = 2*w/length_tot with length_tot = 1/(p_hi - p_lo),
//    src/transformers.jl:213-214, i.e. the triangle height is off by (p_hi-p_lo)^2; the last
//    bin absorbs the remainder so the cell weight is conserved.)
#include "../include/mcs.h"
#include "../include/mcs_synch.h"
#include "../include/mcs_ic.h"
#include "../include/mcs_pion.h"

#include <cmath>
#include <cstdint>
#include <vector>

#ifdef ORC_LIBM
namespace MC {
static inline double acos(double x) { return x >= 1.0 ? 0.0 : std::acos(x); }
static inline double log10(double x) { return std::log10(x); }
static inline double sqrt(double x) { return std::sqrt(x); }
}
#else
#include "../include/mcs_math.h"
namespace MC {
using mcsm::acos; using mcsm::log10;
static inline double sqrt(double x) { return __builtin_sqrt(x); }
}
#endif

namespace {

// src/get_psd_bins.jl:16-39, 73-97 (no counters here)
int bin_mom(const mcs_params& P, double p) {
  int b = p < P.psd_mom_min ? 0 : (int)std::trunc(MC::log10(p / P.psd_mom_min) * P.psd_bins_per_dec_mom) + 1;
  return b > P.num_psd_mom_bins ? P.num_psd_mom_bins : b;
}
int bin_ang(const mcs_params& P, double px, double pt) {
  if (pt == 0.0) return 0;
  const double c = -px / pt;
  int b;
  if (c < P.psd_cos_fine) b = P.num_psd_tht_bins - (int)std::trunc((c + 1) / P.psd_dcos);
  else {
    const double th = MC::acos(c);
    b = th < P.psd_tht_min ? 0 : (int)std::trunc(MC::log10(th / P.psd_tht_min) * P.psd_bins_per_dec_tht) + 1;
  }
  return b < P.num_psd_tht_bins ? b : P.num_psd_tht_bins;
}

// one corner of transform_psd_corners (src/transformers.jl:662-676)
inline void corner(double gam, double beta, double E0, double p_edge, double cos_edge, double& lpt, double& ct) {
  const double px = p_edge * cos_edge;
  const double pc = p_edge * MCS_C;
  const double etot = MC::sqrt(pc * pc + E0 * E0);
  const double pxt = gam * (px - beta * etot / MCS_C);
  const double ptt = MC::sqrt(p_edge * p_edge + pxt * pxt - px * px);
  lpt = MC::log10(ptt);
  ct = pxt / ptt;
}

struct Corners { double pt_lo_pt, pt_hi_pt, ct_lo_pt, ct_hi_pt; };

// src/identify_corners.jl:30-254.  Returns false on the error() paths (C6).
bool identify_corners(const double pts[4], const double cts[4], Corners& out) {
  bool mask[4] = {true, true, true, true};
  int i_lo = 0, i_hi = 0;
  for (int q = 1; q < 4; ++q) { if (pts[q] < pts[i_lo]) i_lo = q; if (pts[q] > pts[i_hi]) i_hi = q; }
  double pt_lo_pt = pts[i_lo], pt_lo_ct = cts[i_lo];
  double pt_hi_pt = pts[i_hi], pt_hi_ct = cts[i_hi];
  mask[i_lo] = false;
  int n_lo = 0, n_hi = 0;
  for (int q = 0; q < 4; ++q) { n_lo += pts[q] == pt_lo_pt; n_hi += pts[q] == pt_hi_pt; }
  int lo_tied = n_lo > 1 ? 1 : 0, hi_tied = n_hi > 1 ? 1 : 0;
  mask[i_hi] = false;
  int j_hi = -1;
  for (int q = 0; q < 4; ++q) if (mask[q] && (j_hi < 0 || cts[q] > cts[j_hi])) j_hi = q;
  if (j_hi < 0) return false;           // all four corners identical in momentum
  double ct_hi_pt = pts[j_hi], ct_hi_ct = cts[j_hi];
  mask[j_hi] = false;
  int j_lo = -1;
  for (int q = 0; q < 4; ++q) if (mask[q] && (j_lo < 0 || cts[q] < cts[j_lo])) j_lo = q;
  if (j_lo < 0) return false;
  double ct_lo_pt = pts[j_lo], ct_lo_ct = cts[j_lo];
  if (ct_hi_ct == ct_lo_ct) {
    if (ct_hi_pt > ct_lo_pt) {
    } else if (ct_hi_pt < ct_lo_pt) {
      ct_hi_pt = pts[j_lo]; ct_hi_ct = cts[j_lo]; ct_lo_pt = pts[j_hi]; ct_lo_ct = cts[j_hi];
    } else return false;
  }
  if (lo_tied == 1) {
    if (pt_lo_pt == ct_lo_pt) lo_tied = 2; else if (pt_lo_pt == ct_hi_pt) lo_tied = 3; else return false;
    if (lo_tied == 2) {
      if (pt_lo_ct > ct_lo_ct) { pt_lo_pt = pts[j_lo]; pt_lo_ct = cts[j_lo]; ct_lo_pt = pts[i_lo]; ct_lo_ct = cts[i_lo]; }
      else if (!(pt_lo_ct < ct_lo_ct)) return false;
    } else {
      if (pt_lo_ct > ct_hi_ct) { pt_lo_pt = pts[j_hi]; pt_lo_ct = cts[j_hi]; ct_hi_pt = pts[i_lo]; ct_hi_ct = cts[i_lo]; }
      else if (!(pt_lo_ct < ct_hi_ct)) return false;
    }
  }
  if (hi_tied == 1) {
    if (pt_hi_pt == ct_lo_pt) hi_tied = 2; else if (pt_hi_pt == ct_hi_pt) hi_tied = 3; else return false;
    if (hi_tied == 2) {
      if (pt_hi_ct > ct_lo_ct) { pt_hi_pt = pts[j_lo]; pt_hi_ct = cts[j_lo]; ct_lo_pt = pts[i_hi]; ct_lo_ct = cts[i_hi]; }
      else if (!(pt_hi_ct < ct_lo_ct)) return false;
    } else {
      if (pt_hi_ct > ct_hi_ct) { pt_hi_pt = pts[j_hi]; pt_hi_ct = cts[j_hi]; ct_hi_pt = pts[i_hi]; ct_hi_ct = cts[i_hi]; }
      else if (!(pt_hi_ct < ct_hi_ct)) return false;
    }
  }
  (void)pt_lo_ct; (void)pt_hi_ct; (void)ct_lo_ct; (void)ct_hi_ct;
  out.pt_lo_pt = pt_lo_pt; out.pt_hi_pt = pt_hi_pt; out.ct_lo_pt = ct_lo_pt; out.ct_hi_pt = ct_hi_pt;
  return true;
}

// src/transformers.jl:209-312 with i_approx = 2 (src/particle_counter.jl:72); lb = log10 edges [0..nmom+1]
void triangular(double* dN, double p_hi, double p_lo, double ct_lo_pt, double ct_hi_pt, double w, int l_lo, int l_hi,
                const double* lb, int nmax1) {
  const double length_tot = 1 / (p_hi - p_lo);
  const double ct_height = 2 * w / length_tot;
  double p_bottom = p_lo;
  const double p_peak = (ct_lo_pt + ct_hi_pt) / 2;
  const double p_denom_lo = 1 / (p_peak - p_lo);
  const double p_denom_hi = 1 / (p_hi - p_peak);
  double fractional_area = 0;
  for (int l = l_lo; l <= l_hi; ++l) {
    if (l + 1 > nmax1) break;                      // (never reached: lb[l_hi] >= p_hi ends the walk at l_hi-1)
    if (p_hi < lb[l_lo + 1]) { dN[l] += w; break; }
    if (lb[l + 1] <= p_peak) {
      const double p_base = lb[l + 1] - p_bottom;
      const double rh = (lb[l + 1] - p_lo) * p_denom_lo * ct_height;
      const double lh = p_bottom == p_lo ? 0.0 : (p_bottom - p_lo) * p_denom_lo * ct_height;
      const double part = p_base / 2 * (lh + rh);
      dN[l] += part;
      p_bottom = lb[l + 1];
      fractional_area += part;
      continue;
    }
    if (lb[l + 1] < p_hi) {
      const double p_base = p_hi - lb[l + 1];
      const double lh = p_base * p_denom_hi * ct_height;
      const double missing = p_base / 2 * lh;
      const double part = (w - fractional_area) - missing;
      dN[l] += part;
      p_bottom = lb[l + 1];
      fractional_area += part;
      continue;
    }
    if (lb[l + 1] >= p_hi) { dN[l] += w - fractional_area; break; }
  }
}

}  // namespace

extern "C" {

// get_dNdp_cr + CR normalisation of get_normalized_dNdp.  out: [3][n_grid][nmom+2] (frame, zone, bin).
// diag[0] cells skipped by C6 (corner errors), diag[1] cells whose l_lo/l_hi search ran off the table.
int orc_dndp_cr(const mcs_params* Pp, const double* T, const mcs_consumer_in* in, const double* gam_sf /*[n_grid+2]*/,
                const double* ux /*[n_grid+2]*/, double* out, int64_t* diag) {
  const mcs_params& P = *Pp;
  mcs_layout L; mcs_tally_layout(Pp, &L);
  const int nm = P.num_psd_mom_bins, nt = P.num_psd_tht_bins, ng = P.n_grid;
  const int NM = nm + 2;
  const double* lb = in->mom_log_cgs; const double* pe = in->mom_edge_cgs; const double* ce = in->cos_edge;
  diag[0] = diag[1] = 0;
  for (int64_t q = 0; q < 3LL * ng * NM; ++q) out[q] = 0.0;
  for (int k = 1; k <= ng; ++k) {
    const double* psd = T + L.psd + L.psd_stride_zone * (int64_t)(k - 1);
    double* d1 = out + (0LL * ng + (k - 1)) * NM;
    // shock frame: src/particle_counter.jl:81-85 (j outer, i inner)
    for (int j = 0; j < nt + 2; ++j)
      for (int i = 0; i < NM; ++i) { const double v = psd[i + L.psd_stride_tht * j]; if (v > 0) d1[i] += v; }
    for (int m = 2; m <= 3; ++m) {
      const double gam = m == 2 ? gam_sf[k] : in->gam0;
      const double beta = gam >= 1.000001 ? MC::sqrt(1 - 1 / (gam * gam)) : 0.0;   // transformers.jl:639
      double* dN = out + ((int64_t)(m - 1) * ng + (k - 1)) * NM;
      for (int j = 0; j <= nt; ++j)
        for (int i = 0; i <= nm; ++i) {
          const double v = psd[i + L.psd_stride_tht * j];
          if (v < 1.0e-66) continue;
          const double w = v / gam;
          double pts[4], cts[4];
          corner(gam, beta, in->rest_energy, pe[i], ce[j], pts[0], cts[0]);
          corner(gam, beta, in->rest_energy, pe[i + 1], ce[j], pts[1], cts[1]);
          corner(gam, beta, in->rest_energy, pe[i], ce[j + 1], pts[2], cts[2]);
          corner(gam, beta, in->rest_energy, pe[i + 1], ce[j + 1], pts[3], cts[3]);
          Corners c;
          if (!identify_corners(pts, cts, c)) { diag[0]++; continue; }
          const double p_lo = c.pt_lo_pt, p_hi = c.pt_hi_pt;
          int l_lo = -1;
          for (int l = 0; l < NM; ++l) if (lb[l] > p_lo) { l_lo = l - 1; break; }   // findfirst(>(p_cell_lo)) - 1
          if (l_lo < 0) { l_lo = nm; diag[1]++; }
          int l_hi = -1;
          for (int l = l_lo; l < NM; ++l) if (lb[l] >= p_hi) { l_hi = l; break; }   // findnext(>=(p_cell_hi), ., l_lo)
          if (l_hi < 0) { l_hi = nm; diag[1]++; }                                   // transformers.jl:86-92
          triangular(dN, p_hi, p_lo, c.ct_lo_pt, c.ct_hi_pt, w, l_lo, l_hi, lb, nm + 1);
        }
    }
  }
  // dN(p) -> dN/dp (src/particle_counter.jl:295-304)
  for (int m = 0; m < 3; ++m)
    for (int k = 0; k < ng; ++k) {
      double* d = out + ((int64_t)m * ng + k) * NM;
      for (int l = 0; l <= nm; ++l) {
        if (d[l] < 1.0e-66) { d[l] = 1.0e-99; continue; }
        d[l] /= pe[l + 1] - pe[l];
      }
    }
  // normalisation (src/particle_counter.jl:733-790) with area_tot_therm == 0 (C4)
  for (int m = 0; m < 3; ++m)
    for (int i = 1; i <= ng; ++i) {
      double* d = out + ((int64_t)m * ng + (i - 1)) * NM;
      double area_cr = 0.0;
      for (int j = 0; j <= nm; ++j) if (d[j] > 1.0e-99) area_cr += d[j] * (pe[j + 1] - pe[j]);
      double area_tot;
      if (area_cr > 0) {
        const double density_pf = in->n0 * in->gam0 * ux[1] / (gam_sf[i] * ux[i]);
        area_tot = density_pf / ux[i] + area_cr;
      } else area_tot = 0.0 + area_cr;
      const double norm = area_tot > 0 ? in->zone_pop[i - 1] / area_tot : 0.0;
      for (int j = 0; j <= nm; ++j) if (d[j] > 1.0e-99) d[j] *= norm;
    }
  return 0;
}

// thermo_calcs (src/thermo_calcs.jl:30-352).  Outputs [n_grid] each.
int orc_thermo_calcs(const mcs_params* Pp, const double* T, const int64_t* I, const mcs_consumer_in* in,
                     const double* gam_sf, const double* ux, double* P_par, double* P_perp, double* e_dens) {
  const mcs_params& P = *Pp;
  mcs_layout L; mcs_tally_layout(Pp, &L);
  const int nm = P.num_psd_mom_bins, nt = P.num_psd_tht_bins, ng = P.n_grid;
  const int NM = nm + 2, NT = nt + 2;
  const double E0 = in->rest_energy, mc = in->mc;
  std::vector<double> d2(NM * (size_t)NT);
  for (int i = 1; i <= ng; ++i) {
    const double* psd = T + L.psd + L.psd_stride_zone * (int64_t)(i - 1);
    const double* thp = T + L.therm_pf + L.psd_stride_zone * (int64_t)(i - 1);
    const int64_t ncross = I[MCS_I_NUM_CROSSINGS + (i - 1)];
    for (size_t q = 0; q < d2.size(); ++q) d2[q] = 1.0e-99;           // :44
    if (in->therm_from_hist) for (size_t q = 0; q < d2.size(); ++q) d2[q] += thp[q];   // :136-162 via A9 (C5)
    const double gam = gam_sf[i], beta = ux[i] / MCS_C;
    for (int j = 0; j <= nt; ++j)
      for (int k = 0; k <= nm; ++k) {                                  // :187-211
        const double w = psd[k + NM * j];
        if (w <= 1.0e-66) continue;
        const double cs = in->cos_center[j], pt = in->pt_center[k];
        const double px = pt * cs;
        const double pc = pt * MCS_C;
        const double et = MC::sqrt(pc * pc + E0 * E0);
        const double pxX = gam * (px - beta * et / MCS_C);
        const double ptX = MC::sqrt(pt * pt - px * px + pxX * pxX);
        d2[bin_mom(P, ptX) + NM * bin_ang(P, pxX, ptX)] += w;
      }
    double norm_fac = 0.0;                                             // :213-221
    for (size_t q = 0; q < d2.size(); ++q) if (d2[q] > 1.0e-66) norm_fac += d2[q];
    if (ncross == 0 && norm_fac > 0) norm_fac += in->n0 / ux[i];
    if (norm_fac > 0) norm_fac = in->zone_pop[i - 1] / norm_fac;
    for (size_t q = 0; q < d2.size(); ++q) if (d2[q] > 1.0e-66) d2[q] *= norm_fac;   // :224-228
    double pop = 0.0, dmax = 0.0;
    for (size_t q = 0; q < d2.size(); ++q) { if (d2[q] > 1.0e-66) pop += d2[q]; if (d2[q] > dmax) dmax = d2[q]; }
    // pressure (:258-347)
    double pp = 0.0, pq = 0.0, ed = 0.0;
    const double density_loc = in->density_loc[i - 1];
    double pressure_loc = in->cold_pressure[i - 1];
    double nf;
    if (dmax < 1.0e-66 && ncross == 0) {
      pp += 1.0 / 3 * pressure_loc; pq += 2.0 / 3 * pressure_loc; ed += 1.5 * pressure_loc;
      P_par[i - 1] = pp; P_perp[i - 1] = pq; e_dens[i - 1] = ed;
      continue;
    } else if (ncross == 0) {
      pressure_loc *= 1 - pop / in->zone_pop[i - 1];
      pp += 1.0 / 3 * pressure_loc; pq += 2.0 / 3 * pressure_loc;
      nf = density_loc / in->zone_pop[i - 1];
      ed += 1.5 * pressure_loc;
    } else {
      nf = density_loc / in->zone_pop[i - 1];
    }
    for (int k = 0; k <= nm; ++k) {
      const double pt = in->pt_center[k];
      const double t = pt / mc;
      const double gtmp = MC::sqrt(1 + t * t);
      const double vel = pt * MCS_C / (mc * gtmp);                      // vel_ptot (:246)
      const double pfac = 1.0 / 3 * pt * vel * nf;
      const double efac = (gtmp - 1) * E0;
      for (int j = 0; j <= nt; ++j) {
        const double c = d2[k + NM * j];
        if (c < 1.0e-66) continue;
        const double cs = in->cos_center[j];
        pp += c * pfac * (cs * cs);
        pq += c * pfac * (1 - cs * cs);
        ed += efac * c * nf;
      }
    }
    P_par[i - 1] = pp; P_perp[i - 1] = pq; e_dens[i - 1] = ed;
  }
  return 0;
}

// The synchrotron fold (src/synch_emission.jl:27-171 through src/photon_synch.jl:28-72), CPU twin of
// mcs_photon_synch: the same header (include/mcs_synch.h), glibc math.
int orc_photon_synch(const mcs_params* Pp, const double* dNdp_pf, const double* mom_edge_cgs, const double* btot /*[n_grid+2]*/, double mc,
                     int n_photon, double emin_mev, double bins_per_dec, double* energy_erg, double* emis) {
  const int NM = Pp->num_psd_mom_bins + 2, ng = Pp->n_grid;
  const double log_emin = std::log10(emin_mev * MCS_MEV_ERG);
  for (int zone = 1; zone <= ng; ++zone)
    for (int j = 0; j < n_photon; ++j) {
      const double E = mcs_synch_energy(log_emin, bins_per_dec, j);
      if (zone == 1 && energy_erg) energy_erg[j] = E;
      emis[(size_t)(zone - 1) * n_photon + j] = mcs_synch_fold_one(1.0e-99, dNdp_pf + (size_t)(zone - 1) * NM, mom_edge_cgs, NM - 2, btot[zone], mc, E);
    }
  return 0;
}
double orc_synch_F(double x) { return mcs_synch_F(x); }

// The pion-decay fold (src/photon_pion_decay.jl:40-183 -> src/pion_kafexhiu.jl:37-245), CPU twin of mcs_photon_pion: the same header
// (include/mcs_pion.h), glibc math.
int orc_photon_pion(const mcs_params* Pp, const double* dNdp_pf, const double* mom_edge_cgs, double mc, double aa, const double* target_density,
                    double scaling, int i_data, int n_photon, double emin_mev, double bins_per_dec, double* energy_erg, double* emis) {
  const int NM = Pp->num_psd_mom_bins + 2, ng = Pp->n_grid;
  if (i_data < 1 || i_data > 4 || NM > 4096) return 1;
  const double log_emin = std::log10(emin_mev * MCS_MEV_ERG);
  std::vector<double> pref(NM), T(NM), Em(NM), A(NM);
  for (int zone = 0; zone < ng; ++zone) {
    for (int i = 0; i < NM - 1; ++i) {
      const double d = dNdp_pf[(size_t)zone * NM + i], lo = mom_edge_cgs[i], hi = mom_edge_cgs[i + 1];
      const double cnt = d <= 1.0e-99 ? 1.0e-99 : d * (hi - lo);
      double t = 0, v = 0, e = 1, a = 0;
      const int ok = mcs_pion_bin(cnt, lo, hi, mc, aa, i_data, &t, &v, &e, &a);
      pref[i] = ok ? target_density[zone] * cnt * v : 0.0; T[i] = t; Em[i] = e; A[i] = a;
    }
    for (int j = 0; j < n_photon; ++j) {
      const double e = std::pow(10.0, log_emin + j * (1.0 / bins_per_dec));
      if (zone == 0 && energy_erg) energy_erg[j] = e;
      emis[(size_t)zone * n_photon + j] = mcs_pion_fold_one(pref.data(), T.data(), Em.data(), A.data(), NM - 1, i_data, e, scaling);
    }
  }
  return 0;
}
// the three functions of src/KATV2014.jl, for the tests' independent restatement
double orc_pion_sigma_pi(double Tp, int i_data) { return mcs_pion_sigma_pi(Tp, i_data, 2 * MCS_PION_MPC2 * (Tp + 2 * MCS_PION_MPC2)); }
void orc_pion_amax(double Tp, int i_data, double* Egmax, double* Amax) {
  const double s = 2 * MCS_PION_MPC2 * (Tp + 2 * MCS_PION_MPC2);
  mcs_pion_amax(Tp, i_data, s, mcs_pion_sigma_pi(Tp, i_data, s), Egmax, Amax);
}
double orc_pion_F(double Tp, double Eg, int i_data, double Egmax) { return mcs_pion_F(Tp, Eg, i_data, Egmax); }

// get_dNdp_2D (src/particle_counter.jl:343-627), CPU twin of mcs_dndp_2d: d2N/dp dcos per zone in the shock frame from the thermal
// crossings (therm_sf histogram: A9) and the psd, normalised to the zone population, rebinned by cell centres into the frame moving
// with (gam_x, beta_x) -- m = 2 of :531-598 for (gam0, beta0).  d2N: [n_grid][ntht+2][nmom+2], momentum fastest.
int orc_dndp_2d(const mcs_params* Pp, const double* T, const int64_t* I, const mcs_consumer_in* in, double gam_x, double beta_x, double* d2N) {
  const mcs_params& P = *Pp;
  mcs_layout L; mcs_tally_layout(Pp, &L);
  const int nm = P.num_psd_mom_bins, nt = P.num_psd_tht_bins, ng = P.n_grid;
  const int NM = nm + 2, NT = nt + 2;
  const double E0 = in->rest_energy;
  std::vector<double> sf(NM * (size_t)NT), dp(NM, 0.0);
  for (int k = 0; k <= nm; ++k) dp[k] = in->mom_edge_cgs[k + 1] - in->mom_edge_cgs[k];                 // :376-380 (C1)
  for (int i = 1; i <= ng; ++i) {
    const double* psd = T + L.psd + L.psd_stride_zone * (int64_t)(i - 1);
    const double* ths = T + L.therm_sf + L.psd_stride_zone * (int64_t)(i - 1);
    double* ef = d2N + (size_t)NM * NT * (i - 1);
    const int64_t ncross = I[MCS_I_NUM_CROSSINGS + (i - 1)];
    for (size_t q = 0; q < sf.size(); ++q) { sf[q] = 1.0e-99; ef[q] = 1.0e-99; }                         // :361-363
    if (ncross != 0 && in->therm_from_hist) for (size_t q = 0; q < sf.size(); ++q) sf[q] += ths[q];      // :430-451 via A9
    for (int j = 0; j <= nt; ++j) for (int k = 0; k <= nm; ++k) { const double w = psd[k + NM * j]; if (w > 1.0e-66) sf[k + NM * j] += w; }   // :466-472
    for (int j = 0; j < NT; ++j) for (int k = 0; k <= nm; ++k) if (sf[k + NM * j] > 1.0e-66) sf[k + NM * j] /= dp[k];                        // :475-480
    double dens = 0.0;                                                                                    // :487-492
    for (size_t q = 0; q < sf.size(); ++q) if (sf[q] > 1.0e-66) dens += sf[q];
    if (ncross == 0 && dens > 0) dens += in->n0;                                                          // :505-507
    const double norm = dens > 0 ? in->zone_pop[i - 1] / dens : 0.0;                                      // :510-514
    for (size_t q = 0; q < sf.size(); ++q) sf[q] = (sf[q] > 1.0e-99 && norm > 0) ? sf[q] * norm : 1.0e-99;   // :517-523
    for (int k = 0; k <= nm; ++k)                                                                          // :562-596
      for (int j = 0; j <= nt; ++j) {
        const double v = sf[k + NM * j];
        if (v <= 1.0e-66) continue;
        const double w = v * dp[k];
        const double cs = in->cos_center[j], pt = in->pt_center[k];
        const double px = pt * cs;
        const double pc = pt * MCS_C;
        const double et = MC::sqrt(pc * pc + E0 * E0);
        const double pxX = gam_x * (px - beta_x * et / MCS_C);
        const double ptX = MC::sqrt(pt * pt - px * px + pxX * pxX);
        const int kX = bin_mom(P, ptX), jX = bin_ang(P, pxX, ptX);
        ef[kX + NM * jX] += w / dp[kX];
      }
  }
  return 0;
}

// The inverse-Compton fold (src/inverse_compton.jl:36-311), CPU twin of mcs_photon_ic: the same header (include/mcs_ic.h), glibc math.
int orc_photon_ic(const mcs_params* Pp, const double* d2N /*[n_grid][ntht+2][nmom+2]*/, const double* mom_edge_cgs, double mc_e, int j_max, int n_nu,
                  const double* alpha_in, const double* n_in, int n_photon, double emin_mev, double bins_per_dec, double beam_area,
                  double* energy_erg, double* emis) {
  const int NM = Pp->num_psd_mom_bins + 2, NT = Pp->num_psd_tht_bins + 2, ng = Pp->n_grid, nm = NM - 2;
  const double log_min_rm = std::log10(emin_mev * MCS_IC_MEV_ERG / (MCS_ME * MCS_C * MCS_C));
  std::vector<double> xnum(nm + 1), gam(nm + 1);
  for (int zone = 1; zone <= ng; ++zone) {
    const double* d2 = d2N + (size_t)(zone - 1) * NM * NT;
    for (int i = 0; i <= nm; ++i) {
      const double dp = mom_edge_cgs[i + 1] - mom_edge_cgs[i];
      double mx = 0.0, sum = 0.0;
      for (int j = 0; j <= j_max; ++j) {
        const double v = d2[i + NM * j];
        const double c = v <= 1.0e-99 ? 1.0e-99 : v * dp;                  // photon_IC, :54-61
        mx = c > mx ? c : mx;
        sum += c;
      }
      xnum[i] = mx <= 1.0e-99 ? 0.0 : sum;                                 // :235-238
      gam[i] = mcs_ic_gamma(mom_edge_cgs[i], mom_edge_cgs[i + 1], mc_e);
    }
    for (int k = 0; k < n_photon; ++k) {
      const double ao = mcs_ic_alpha_out(log_min_rm, bins_per_dec, k);
      if (zone == 1 && energy_erg) energy_erg[k] = ao * (MCS_ME * MCS_C * MCS_C);
      emis[(size_t)(zone - 1) * n_photon + k] = mcs_ic_emis(mcs_ic_fold_one(xnum.data(), gam.data(), nm + 1, alpha_in, n_in, n_nu, ao), ao, beam_area);
    }
  }
  return 0;
}

}  // extern "C"
