/* mcs.h -- C ABI of the MI355X-native per-particle transport path.
 *
 * Drop-in boundary: the `for i_prt in 1:n_pts_use` loop of the reference,
 * /root/reference/src/main_loops.jl:228-292 (particle_loop + particle_finish!),
 * batched into ONE call per (iteration, species, pcut).  The reference has no
 * FFI for this path (it is pure Julia); INTEGRATION.md shows the `ccall` shim a
 * maintainer would put in place of that loop.  All quantities are fp64 cgs, as
 * in the reference after `ustrip`.
 *
 * Index conventions
 *   grid tables : n_grid+2 entries, C index == Julia OffsetVector index 0:n_grid+1
 *   zones       : Julia 1:n_grid  ->  C slot (i-1)
 *   PSD bins    : 0-based in both
 *   pcut/tcut/ion/iter numbers are passed 1-based (they enter the RNG seed formula
 *   of src/particle_loop.jl:35-40 and index pₓ_esc_feb[i_ion, i_iter]).
 *
 * Error model: every entry point returns 0 on success, non-zero otherwise;
 * mcs_last_error() returns the message (the reference's `error(...)` sites:
 * src/particle_finish.jl:104, src/all_flux.jl:73-75, src/scattering.jl:52-53,
 * src/prob_return.jl:134).  Warn-paths of the reference (@warn) are counters.
 *
 * Threading: one context per GPU; calls on one context must be serialised by
 * the caller (the reference is single-threaded and not re-entrant).
 */
#ifndef MCS_H
#define MCS_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCS_ABI_VERSION 3

/* ---- physical constants (cgs).  The reference takes them from Unitful /
 * UnitfulGaussian / PhysicalConstants.CODATA2018 (src/MonteCarloScattering.jl:10-12,
 * src/constants.jl:3); Project.toml has no lockfile, these are the CODATA-2018
 * values those packages carry. */
#define MCS_MP     1.67262192369e-24      /* proton mass [g] */
#define MCS_ME     9.1093837015e-28       /* electron mass [g] */
#define MCS_C      2.99792458e10          /* speed of light [cm/s] */
#define MCS_QCGS   4.803204712570263e-10  /* elementary charge [esu] = 1.602176634e-19 C * c/10 */
#define MCS_KB     1.380649e-16           /* Boltzmann [erg/K] */
#define MCS_SIGMA_T 6.6524587321e-25      /* Thomson cross-section [cm^2] */
#define MCS_B_CMB0 3.27e-6                /* src/constants.jl:10 [G] */
/* src/constants.jl:30: rad_loss_fac = 4/3 c sigma_T / (c^3 me^2 8 pi)  [s^2/g^2] */
#define MCS_RAD_LOSS_FAC ((4.0/3.0) * MCS_C * MCS_SIGMA_T / (MCS_C*MCS_C*MCS_C * MCS_ME*MCS_ME * 8.0 * 3.141592653589793))

/* ---- compile-time constants of the path (src/parameters.jl, src/all_flux.jl:4,
 * src/particle_finish.jl:5, src/particle_loop.jl:162, src/prob_return.jl:229) */
#define MCS_PSD_MAX      200     /* parameters.jl:18 */
#define MCS_NA_C         100     /* parameters.jl:11 */
#define MCS_E_REL_PT     0.005   /* parameters.jl:32 */
#define MCS_SPIKE_AWAY   1000.0  /* all_flux.jl:4, particle_finish.jl:5 */
#define MCS_HELIX_CAP    10000   /* particle_loop.jl:162 */
#define MCS_RETRO_XN_PER 10.0    /* prob_return.jl:229 */
/* The reference's retro_time loop is uncapped (prob_return.jl:257); a walk that never comes back to the
 * PRP would hang the GPU.  After this many inner steps of ONE walk the particle ends with i_reason 3
 * and MCS_IC_RETRO_CAP is bumped (identically in the oracle).  mcs_set_retro_cap overrides it (tests). */
#define MCS_RETRO_CAP    10000000
#define MCS_FLOOR        1.0e-99 /* particle_loop.jl:315-317, ion_init.jl:11-13 */

/* i_reason codes (src/particle_loop.jl:138, src/particle_finish.jl:81-105) */
#define MCS_REASON_SAVED      0  /* reached pcut, kept for next pcut (l_save) */
#define MCS_REASON_DOWNSTREAM 1
#define MCS_REASON_UPSTREAM   2  /* pmax or upstream FEB */
#define MCS_REASON_AGE        3
#define MCS_REASON_ZERO_E     4

/* Scalars and flags handed to particle_loop at src/main_loops.jl:236-264. */
typedef struct mcs_params {
  int32_t abi_version;          /* = MCS_ABI_VERSION */
  int32_t n_ions, n_grid, n_itrs;
  int64_t n_pts_max;            /* MonteCarloScattering.jl:488; enters the seed formula */
  int32_t i_grid_feb, i_shock;  /* MonteCarloScattering.jl:414,478 (Julia zone numbers) */
  int32_t num_psd_mom_bins, num_psd_tht_bins;
  int32_t psd_bins_per_dec_mom, psd_bins_per_dec_tht;
  double  psd_cos_fine, psd_dcos, psd_tht_min, psd_mom_min;
  double  gam0, beta0, u0, u2, bmag2;
  double  pe_crit, game_crit, eta_mfp;
  double  energy_transfer_frac;
  double  feb_upstream, feb_downstream, x_grid_stop;
  double  B_CMBz, age_max;
  double  xn_per_fine, xn_per_coarse;
  int32_t use_custom_epsB, do_rad_losses, do_retro, do_tcuts;
  int32_t dont_DSA, dont_scatter, use_custom_frg;
  int32_t track_thermal;        /* A9: bin non-injected crossings on the fly */
  int32_t state_fp32;           /* 0: fp64 particle state (the reference's precision).  1: the fp32-state variant of K1 --
                                 * state and per-step arithmetic in fp32 in normalised units (p / m_p c, x / rg0, t c / rg0),
                                 * population in HBM and all tallies fp64 (BASELINE config[4]; DESIGN.md "fp32-state variant") */
} mcs_params;

/* One particle population, struct-of-arrays, host side; element types follow the
 * Julia arrays at src/MonteCarloScattering.jl:556-585 (Float64 / Int / Bool). */
typedef struct mcs_soa {
  double  *weight, *ptot_pf, *pb_pf, *x_PT_cm, *xn_per, *prp_x_cm, *acctime_sec, *phi_rad;
  int64_t *grid, *tcut;
  uint8_t *downstream, *inj;
} mcs_soa;

/* Offsets (in doubles) of every fp64 tally inside ONE flat buffer, so that a
 * multi-GPU run merges all of them with a single sum-all-reduce.  Shapes and
 * reset cadence: SURVEY.md section 8(a) "Tally arrays". */
typedef struct mcs_layout {
  int64_t psd;            /* [nmom+2][ntht+2][n_grid], momentum fastest (MonteCarloScattering.jl:519) */
  int64_t therm_sf;       /* same shape as psd: non-injected crossings, shock frame   (A9) */
  int64_t therm_pf;       /* same shape as psd: non-injected crossings, plasma frame of zone i (A9) */
  int64_t esc_psd_up;     /* [201][201], ip fastest (MonteCarloScattering.jl:537) */
  int64_t esc_psd_down;   /* [201][201] */
  int64_t pxx_flux, pxz_flux, energy_flux;    /* [n_grid] */
  int64_t esc_flux;       /* [n_ions] */
  int64_t px_esc_feb, energy_esc_feb;         /* [n_ions][n_itrs], ion fastest */
  int64_t esc_energy_eff, esc_num_eff;        /* [201][n_ions], ip fastest */
  int64_t weight_coupled;                     /* [100][n_ions] */
  int64_t spectra_coupled;                    /* [201][100][n_ions] */
  int64_t spectra_sf, spectra_pf;             /* [201][n_grid] (2nd index = x_spec number) */
  int64_t energy_transfer_pool, energy_recv_pool; /* [n_grid] */
  int64_t scalars;        /* [4]: sumP_downstream, sumKEdensity_downstream, px_esc_upstream, energy_esc_upstream */
  int64_t total;          /* number of doubles */
  int64_t psd_stride_tht, psd_stride_zone;    /* nmom+2, (nmom+2)*(ntht+2) */
} mcs_layout;

/* int64 tallies / diagnostics, one flat buffer */
enum {
  MCS_I_NUM_CROSSINGS = 0,      /* [n_grid] src/all_flux.jl:254 */
  /* the following are offsets from n_grid */
  MCS_IC_STEPS_HELIX = 0,       /* passes of src/particle_loop.jl:154-499 */
  MCS_IC_STEPS_RETRO,           /* passes of src/prob_return.jl:257-338 */
  MCS_IC_HELIX_CAP,             /* particle_loop.jl:162 hits */
  MCS_IC_PPERP_CLAMP,           /* particle_loop.jl:640-644 hits */
  MCS_IC_PSP_CLAMP,             /* transformers.jl:562-568,592-598 hits */
  MCS_IC_MOMBIN_CLAMP,          /* get_psd_bins.jl:29-36 hits */
  MCS_IC_REASON0, MCS_IC_REASON1, MCS_IC_REASON2, MCS_IC_REASON3, MCS_IC_REASON4,
  MCS_IC_TCUT_OVERRUN,          /* tcut index past n_tcuts (reference would throw BoundsError) */
  MCS_IC_RNG_DRAWS,
  MCS_IC_ZONE_FAIL,              /* src/all_flux.jl:73-75 would throw */
  MCS_IC_RETRO_CAP,              /* retro_time walks ended by MCS_RETRO_CAP (the reference would never return) */
  MCS_IC_COUNT
};

static inline int64_t mcs_i64_total(const mcs_params* p) { return (int64_t)p->n_grid + MCS_IC_COUNT; }

static inline void mcs_tally_layout(const mcs_params* p, mcs_layout* L) {
  const int64_t nm = p->num_psd_mom_bins + 2, nt = p->num_psd_tht_bins + 2, ng = p->n_grid;
  const int64_t pm = MCS_PSD_MAX + 1;
  int64_t o = 0;
  L->psd_stride_tht = nm; L->psd_stride_zone = nm * nt;
  L->psd = o;            o += nm * nt * ng;
  L->therm_sf = o;       o += nm * nt * ng;
  L->therm_pf = o;       o += nm * nt * ng;
  L->esc_psd_up = o;     o += pm * pm;
  L->esc_psd_down = o;   o += pm * pm;
  L->pxx_flux = o;       o += ng;
  L->pxz_flux = o;       o += ng;
  L->energy_flux = o;    o += ng;
  L->esc_flux = o;       o += p->n_ions;
  L->px_esc_feb = o;     o += (int64_t)p->n_ions * p->n_itrs;
  L->energy_esc_feb = o; o += (int64_t)p->n_ions * p->n_itrs;
  L->esc_energy_eff = o; o += pm * p->n_ions;
  L->esc_num_eff = o;    o += pm * p->n_ions;
  L->weight_coupled = o; o += (int64_t)MCS_NA_C * p->n_ions;
  L->spectra_coupled = o; o += pm * MCS_NA_C * p->n_ions;
  L->spectra_sf = o;     o += pm * ng;
  L->spectra_pf = o;     o += pm * ng;
  L->energy_transfer_pool = o; o += ng;
  L->energy_recv_pool = o;     o += ng;
  L->scalars = o;        o += 4;
  L->total = o;
}

typedef struct mcs_ctx mcs_ctx;

/* ---- lifecycle ---------------------------------------------------------- */
int         mcs_abi_version(void);
/* non-inline export of mcs_tally_layout for FFI callers (Julia ccall, ctypes) */
int         mcs_get_layout(const mcs_params* p, mcs_layout* out);
const char* mcs_last_error(void);
/* device: HIP device ordinal; stream: hipStream_t (NULL = default stream). */
int mcs_create(const mcs_params* p, int device, void* stream, mcs_ctx** out);
int mcs_destroy(mcs_ctx* ctx);
int mcs_sync(mcs_ctx* ctx);

/* Optional: make the context accumulate into caller-owned DEVICE buffers (e.g.
 * torch tensors, so that torch.distributed can all-reduce them in place).
 * Without this call the context allocates its own. */
int mcs_bind_tallies(mcs_ctx* ctx, double* dev_f64, int64_t n_f64, int64_t* dev_i64, int64_t n_i64);
double*  mcs_tallies_f64_devptr(mcs_ctx* ctx);
int64_t* mcs_tallies_i64_devptr(mcs_ctx* ctx);

/* ---- per-iteration / per-species inputs (host pointers) ----------------- */
/* grid tables at src/main_loops.jl:255-260; n_entries = n_grid+2 */
int mcs_set_grid(mcs_ctx* ctx, int n_entries, const double* x_grid_cm,
                 const double* ux_sk, const double* uz_sk, const double* utot,
                 const double* gam_sf, const double* gam_ef, const double* beta_ef,
                 const double* btot, const double* theta);
/* pcuts/tcuts/x_spec/inj_fracs/eps_target at src/main_loops.jl:244,259-262 */
int mcs_set_cuts(mcs_ctx* ctx, int n_pcuts, const double* pcuts, int n_tcuts, const double* tcuts,
                 int n_xspec, const double* x_spec, const double* inj_fracs /*[n_ions]*/,
                 const double* eps_target /*[n_grid]*/);
/* resets of src/main_loops.jl:59-86 (fluxes, weight_coupled := 1e-99; pools := 0; scalars := 1e-99) */
int mcs_begin_iteration(mcs_ctx* ctx, int i_iter);
/* src/main_loops.jl:97-121,164 + clear_psd! (src/ion_init.jl:1-16): psd/esc_psd := 1e-99,
 * num_crossings/therm := 0, fluxes := 0 (quirk Q2), recv_pool := transfer_pool. */
int mcs_begin_species(mcs_ctx* ctx, int i_iter, int i_ion, double aa, double zz,
                      double pmax_cutoff, double density, double electron_weight_fac);
/* analytic fast-push fluxes of init_pop/F_update! (src/initializers.jl:1054-1068,1156) */
int mcs_set_fluxes(mcs_ctx* ctx, const double* pxx, const double* pxz, const double* energy);

/* ---- population (device resident) --------------------------------------- */
int mcs_pop_upload(mcs_ctx* ctx, int64_t n, const mcs_soa* host);
int mcs_pop_download(mcs_ctx* ctx, int64_t n, mcs_soa* host);          /* current ("new") population */
int mcs_saved_download(mcs_ctx* ctx, int64_t n, mcs_soa* host, uint8_t* l_save); /* *_saved arrays */
int64_t mcs_pop_size(mcs_ctx* ctx);
/* K3: fast-push part of init_pop (src/initializers.jl:1078-1131) +
 * assign_particle_properties_to_population! (src/ion_init.jl:29-53), on device.
 * ptot_pf_in/weight_in: host, n entries.  Population RNG: Philox key
 * (i_iter-1)*n_ions+(i_ion-1) (src/main_loops.jl:120), draw j-1 = pitch of
 * particle j, draw n_total+j-1 = phase of particle j (global j = j_offset+local). */
int mcs_init_pop(mcs_ctx* ctx, int64_t n, int64_t j_offset, int64_t n_total,
                 const double* ptot_pf_in, const double* weight_in,
                 double x_start_cm, int i_grid_start, int relativistic, int fast_push);
/* The same from the host's momentum discretisation instead of per-particle arrays (set_inj_dist,
 * src/initializers.jl:1251-1328, gives every particle of a bin the same ptot and weight): particle j
 * (global, 0-based) lies in the bin b with bin_start[b] <= j < bin_start[b+1]; bin_start has n_bins+1
 * entries, bin_start[n_bins] = n_total.  O(bins) host work and upload instead of O(N). */
int mcs_init_pop_binned(mcs_ctx* ctx, int64_t n_local, int64_t j_offset, int64_t n_total, int n_bins,
                        const double* bin_ptot_pf, const double* bin_weight, const int64_t* bin_start,
                        double x_start_cm, int i_grid_start, int relativistic, int fast_push);
/* ... for a strided shard: local particle k is global particle j_first + k * j_stride (multi-GPU: rank r of W takes
 * j_first = r, j_stride = W, so that every rank holds the same mix of the momentum-sorted injection). */
int mcs_init_pop_binned_strided(mcs_ctx* ctx, int64_t n_local, int64_t j_first, int64_t j_stride, int64_t n_total, int n_bins,
                                const double* bin_ptot_pf, const double* bin_weight, const int64_t* bin_start,
                                double x_start_cm, int i_grid_start, int relativistic, int fast_push);

/* K1: the particle loop of one pcut over the resident population.
 * i_prt_offset: global index of local particle 0 minus 1 (multi-GPU shards;
 * the RNG key uses the global i_prt).  n_saved: particles that reached pcut. */
int mcs_run_pcut(mcs_ctx* ctx, int i_pcut, int64_t i_prt_offset, int64_t* n_saved);
/* The same over a strided shard: local particle k carries the global 0-based index
 * i_prt_first + k * i_prt_stride (its RNG key uses that index + 1, as i_prt in
 * src/particle_loop.jl:35-40).  mcs_run_pcut(c, i, off, ns) == mcs_run_pcut_strided(c, i, off, 1, ns). */
int mcs_run_pcut_strided(mcs_ctx* ctx, int i_pcut, int64_t i_prt_first, int64_t i_prt_stride, int64_t* n_saved);
/* The same with an explicit index list: local particle k carries the global 0-based index dev_gidx[k] (DEVICE
 * memory, n = mcs_pop_size entries, owned by the caller and kept alive until the next mcs_run_pcut* or
 * mcs_new_pcut / mcs_split_import).  This is what a multi-GPU driver needs after a LOCAL split of an interleaved
 * shard: the children of rank r's saved particles are not an arithmetic progression of the global split. */
int mcs_run_pcut_indexed(mcs_ctx* ctx, int i_pcut, const int64_t* dev_gidx, int64_t* n_saved);
/* K2: pcut_finalize/new_pcut (src/cuts.jl:34-124) on device: stable compaction
 * of l_save and i_mult-fold replication with weight/i_mult. Returns new size. */
int mcs_new_pcut(mcs_ctx* ctx, int64_t i_mult, int64_t* n_new);
/* Multi-GPU form of new_pcut.  The reference builds the next population from ALL saved particles
 * (src/cuts.jl:34-98); with the population sharded over GPUs a local split leaves the late pcuts -- a
 * handful of saved particles, each replicated 10^5 times -- on one or two ranks.  Instead:
 *   mcs_saved_export  writes the saved particles of the last mcs_run_pcut*, compacted in index order, into
 *     caller-owned DEVICE buffers (torch tensors, all-gathered by the caller over RCCL):
 *     gidx[r] = global 0-based index of the r-th saved particle, f64[f * cap + r] = field f of it (the 8
 *     doubles of mcs_soa, in that order), meta[r] = grid | tcut << 16 | downstream << 24 | inj << 25.
 *     cap >= n_saved of that run.
 *   mcs_split_import  makes the new local population from n_parents parents in device buffers of the same
 *     layout, sorted by global index by the caller: local particle k is element o = first + k * stride of
 *     the global split population  o -> parent[o / i_mult]  with weight / i_mult  (the index arithmetic of
 *     src/cuts.jl:66-92 with a global o).  Global indices -- hence RNG keys -- are those of a one-GPU run. */
int mcs_saved_export(mcs_ctx* ctx, int64_t cap, int64_t* dev_gidx, double* dev_f64, uint32_t* dev_meta);
/* The index column of mcs_saved_export alone (8 B per saved particle): enough for the ranks to agree on every saved
 * particle's position in the global order, from which a local mcs_new_pcut's children get their global indices
 * (position * i_mult + j) without any particle leaving its GPU. */
int mcs_saved_gidx(mcs_ctx* ctx, int64_t cap, int64_t* dev_gidx);
int mcs_split_import(mcs_ctx* ctx, int64_t n_parents, int64_t cap, const double* dev_f64, const uint32_t* dev_meta,
                     int64_t i_mult, int64_t first, int64_t stride, int64_t n_local);

/* Host-buffer form of K1, the literal drop-in for the loop at main_loops.jl:228-292:
 * upload `in`, run, download saved arrays + l_save. */
int mcs_run_pcut_host(mcs_ctx* ctx, int i_pcut, int64_t n_pts_use, int64_t i_prt_offset,
                      const mcs_soa* in, mcs_soa* saved_out, uint8_t* l_save, int64_t* n_saved);

/* A species' pcuts with the long histories of every pcut finishing BESIDE the next pcut, on a second stream (the loop at
 * src/main_loops.jl:184-292 with pcut_finalize / new_pcut, src/cuts.jl:34-124, as mcs_run_pcuts_fused).  A particle is LONG in a pcut when
 * its history there took at least long_draws random draws; the next population is the children of the saved particles that are not
 * long, in index order, followed by the children of the saved long ones, in index order (the reference's order for long_draws =
 * infinity; the index keys a child's random stream, so the order is part of the result -- the oracle orders the same way:
 * orc_set_long_draws).  long_imult_max > 0: long histories are told apart only in the first pcut and in pcuts whose predecessor split by
 * at most that factor (elsewhere the pcut runs as one launch, in the reference's order): where few particles are saved and each is split a
 * hundredfold, i_mult hangs on the last long history.  n_target[k]: the target population after pcut first + k.  Outputs (host, length last - first + 1): n_use,
 * n_saved, i_mult per pcut, the main launch's kernel time; strag_out (or NULL, length 2 per pcut): particles exported, and 1 where
 * i_mult had to wait for them.  One rank with global indices 0, 1, 2, ...; fp64 state; not with sliced launches. */
int mcs_run_pcuts_pipelined(mcs_ctx* ctx, int i_pcut_first, int i_pcut_last, const int64_t* n_target, int64_t long_draws, int64_t long_imult_max,
                            int64_t* n_use_out, int64_t* n_saved_out, int64_t* i_mult_out, double* kernel_ms_out, int64_t* strag_out);

/* ---- tallies ------------------------------------------------------------ */
int mcs_read_tallies(mcs_ctx* ctx, double* host_f64 /*layout.total*/, int64_t* host_i64 /*mcs_i64_total*/);
/* A slice of the fp64 buffer, words [first, first + count) of the layout, into host_f64[0 .. count) (+ all int64 tallies
 * when host_i64 is not null).  The three histograms psd | therm_sf | therm_pf are 99 % of the buffer and have consumers on
 * the device (mcs_dndp_cr, mcs_thermo_calcs); what iter_finalize needs on the host -- fluxes, escape spectra, scalars,
 * pools (src/iter_finalize.jl:27-70) -- is the tail of the layout from esc_psd_up on: 0.8 MB instead of 61. */
int mcs_read_tallies_part(mcs_ctx* ctx, int64_t first, int64_t count, double* host_f64, int64_t* host_i64);
int mcs_write_tallies(mcs_ctx* ctx, const double* host_f64, const int64_t* host_i64);
/* The mirror of mcs_read_tallies_part: words [first, first + count) of the fp64 buffer from host_f64[0 .. count).  What the
 * host rewrites in place between iterations is small: tcut_print normalises spectra_coupled and floors weight_coupled
 * (src/io.jl:28-45, called at src/main_loops.jl:383-389). */
int mcs_write_tallies_part(mcs_ctx* ctx, int64_t first, int64_t count, const double* host_f64);

/* ---- consumers of the tallies, on the device-resident histograms ----
 * Host-made tables (O(bins), O(n_grid)); the edges are cgs momenta and true cos(theta) in the
 * intended order (consumer quirks C1, C2 in DESIGN.md). */
typedef struct mcs_consumer_in {
  const double* mom_log_cgs;    /* [nmom+2] log10 of the momentum bin edges (cgs)               */
  const double* mom_edge_cgs;   /* [nmom+2] the edges themselves                                */
  const double* cos_edge;       /* [ntht+2] true cos(theta) of the angle bin edges              */
  const double* cos_center;     /* [ntht+1] thermo_calcs.jl:57-73                               */
  const double* pt_center;      /* [nmom+1] thermo_calcs.jl:75-80 (cgs)                         */
  const double* zone_pop;       /* [n_grid] set_grid_volumes! (particle_counter.jl:1466-1524)   */
  const double* density_loc;    /* [n_grid] gam0 beta0 n0 / sqrt(gam_sf^2 - 1) (thermo_calcs.jl:258) */
  const double* cold_pressure;  /* [n_grid] density_loc^(5/3) kB T0 (thermo_calcs.jl:266)       */
  double rest_energy;           /* m c^2 of the species                                         */
  double mc;                    /* m c                                                          */
  double n0;                    /* far-upstream density of the species                          */
  double gam0;
  int therm_from_hist;          /* 1: thermal crossings from the therm_pf histogram (A9, C5)    */
} mcs_consumer_in;
/* get_dNdp_cr + the CR normalisation of get_normalized_dNdp (src/particle_counter.jl:29-306,
 * 733-790) on the resident psd.  dNdp: host [3][n_grid][nmom+2] (frame: shock, plasma, ISM).
 * diag: host [2] (cells skipped on identify_corners error paths; searches that left the table). */
int mcs_dndp_cr(mcs_ctx* ctx, const mcs_consumer_in* in, double* dNdp, int64_t* diag);
/* thermo_calcs (src/thermo_calcs.jl:30-352) on the resident psd / therm_pf / num_crossings.
 * Outputs: host [n_grid] each. */
int mcs_thermo_calcs(mcs_ctx* ctx, const mcs_consumer_in* in, double* P_par, double* P_perp, double* energy_density);

/* ---- photon post-processing: the synchrotron fold of src/synch_emission.jl:27-171 over the plasma-frame
 * dN/dp of an electron species (frame 2 of mcs_dndp_cr), for every grid zone with the zone's field of the grid tables.
 * dNdp_pf: host [n_grid][nmom+2]; mom_edge_cgs: host [nmom+2]; mc of the species; photon energies
 * E_j = emin_mev * 10^(j / bins_per_dec), j = 0 .. n_photon-1 (photon_calcs.jl:11-19,51: 1e-13 MeV, 10 per decade, 180 bins).
 * Outputs (host): energy_erg[n_photon] (may be null), emis[n_grid][n_photon] = dP/d(ln E) in erg/s per zone, floor 1e-99.
 * The photon stack is dead code in the reference and is followed as specification; F(x) is restated (include/mcs_synch.h). */
int mcs_photon_synch(mcs_ctx* ctx, const double* dNdp_pf, const double* mom_edge_cgs, double mc, int n_photon, double emin_mev,
                     double bins_per_dec, double* energy_erg, double* emis);

/* The pion-decay fold of src/photon_pion_decay.jl:40-183 -> src/pion_kafexhiu.jl:37-245 (Kafexhiu et al. 2014, src/KATV2014.jl) over the
 * plasma-frame dN/dp of a nucleus species (aa >= 1; frame 2 of mcs_dndp_cr; the thermal histogram of get_normalized_dNdp is empty, quirk C4),
 * for every grid zone.  dNdp_pf: host [n_grid][nmom+2]; mom_edge_cgs: host [nmom+2]; mc, aa of the species; target_density: host [n_grid]
 * = n0[1] gam0 beta0 / sqrt(gam_sf^2 - 1) (photon_pion_decay.jl:62-63); scaling: the heavy-nuclei factor of pion_kafexhiu.jl:60-65; i_data:
 * 1 GEANT 4 (what the reference hard-wires), 2 PYTHIA 8, 3 SIBYLL 2.1, 4 QGSJET-I; photon energies E_j = emin_mev * 10^(j / bins_per_dec)
 * (photon_calcs.jl:15-16,49: 1 MeV, 10 per decade, 120 bins).  Outputs (host): energy_erg[n_photon] (may be NULL), emis[n_grid][n_photon] =
 * dP/d(ln E) in erg/s per zone, floor 1e-99.  Dead code in the reference, followed as specification (include/mcs_pion.h: P1-P3). */
int mcs_photon_pion(mcs_ctx* ctx, const double* dNdp_pf, const double* mom_edge_cgs, double mc, double aa, const double* target_density, double scaling,
                    int i_data, int n_photon, double emin_mev, double bins_per_dec, double* energy_erg, double* emis);

/* get_dNdp_2D (src/particle_counter.jl:343-627, called at src/ion_finalize.jl:50-59) on the resident psd / therm_sf / num_crossings:
 * d2N/dp dcos of every zone, normalised to the zone population, rebinned by cell centres into the frame that moves with
 * (gam_x, beta_x) against the shock frame -- the ISM frame for (gam0, beta0), the only frame the function returns (m = 2, :538).
 * Uses mom_edge_cgs, cos_center, pt_center, zone_pop, rest_energy, n0, therm_from_hist of `in`.  The array stays on the device for
 * mcs_photon_ic; d2N: host [n_grid][ntht+2][nmom+2] (momentum fastest), floor 1e-99, may be NULL. */
int mcs_dndp_2d(mcs_ctx* ctx, const mcs_consumer_in* in, double gam_x, double beta_x, double* d2N);
/* The inverse-Compton fold of src/inverse_compton.jl:36-311 (photon_IC -> IC_emission_FCJ: Jones 1968, eq. 9) over the array the last
 * mcs_dndp_2d left on the device, per grid zone.  j_max: last angle bin inside the jet cone (inverse_compton.jl:215); alpha_in /
 * n_in [n_nu <= 60]: energies (in m_e c^2) and number densities of the incoming photon field (photon_field!, :313-383: host table);
 * photon energies E_k = emin_mev * 10^(k / bins_per_dec); beam_area = 4 pi d_lum^2 jet_sph_frac.  Outputs (host): energy_erg[n_photon]
 * (may be NULL), emis[n_grid][n_photon] = observed energy flux per d(ln E) at Earth in erg / (s cm^2), floor 1e-99 (:285-308).
 * Dead code in the reference, followed as specification (include/mcs_ic.h lists where it cannot run as written). */
int mcs_photon_ic(mcs_ctx* ctx, const double* mom_edge_cgs, double mc_e, int j_max, int n_nu, const double* alpha_in, const double* n_in, int n_photon,
                  double emin_mev, double bins_per_dec, double beam_area, double* energy_erg, double* emis);

/* ---- test / measurement hooks ------------------------------------------- */
/* evaluate device math/RNG primitives (bit-parity tests): fn ids in mcs_fn */
enum mcs_fn { MCS_FN_SIN = 0, MCS_FN_COS, MCS_FN_ASIN, MCS_FN_ACOS, MCS_FN_ATAN2, MCS_FN_LOG10,
              MCS_FN_MOD2PI, MCS_FN_SQRT, MCS_FN_DIV, MCS_FN_HYPOT1, MCS_FN_UNIFORM };
int mcs_eval_fn(mcs_ctx* ctx, int fn, int64_t n, const double* a, const double* b, double* out);
/* per-particle end state of the last mcs_run_pcut (bit-parity tests): i_reason
 * (0 = saved), helix_count, retro step count, final ptot_pf and x. Any pointer may be NULL.
 * The kernel records them only after mcs_set_debug_finals(ctx, 1) (24 B of stores per particle
 * and pcut that the product path does not need); otherwise mcs_final_download fails. */
int mcs_set_debug_finals(mcs_ctx* ctx, int on);
/* inner-step cap of one retro_time walk (0 = MCS_RETRO_CAP); tests lower it to reach the cap path */
int mcs_set_retro_cap(mcs_ctx* ctx, int64_t cap);
int mcs_final_download(mcs_ctx* ctx, int64_t n, int32_t* reason, int32_t* helix_count,
                       int32_t* retro_count, double* ptot_pf, double* x_PT_cm);
/* kernel time [ms] of the last mcs_run_pcut, from HIP events on the context stream */
double mcs_last_kernel_ms(mcs_ctx* ctx);
/* launch geometry override: blocks (0 = auto), threads per block (0 = auto) */
int mcs_set_launch(mcs_ctx* ctx, int blocks, int threads);
/* Sliced tail of mcs_run_pcut* (0 = off: one launch per pcut).  A launch cannot end before its longest history does, and
 * a history is up to 10^4 sequential passes (src/particle_loop.jl:162) while the bulk of a 10^6-particle pcut takes the
 * chip a few thousand: with budget_trips > 0 every wave that has found the work queue empty makes budget_trips more
 * trips through its loop (6 passes each), writes the complete lane state of its live particles to a device buffer and
 * ends; the library relaunches them spread over the chip's waves -- a particle that shares its wave with few others
 * advances faster, its neighbours' rare work no longer stalls it -- until none is left.  A history is the same bit for
 * bit however often it is suspended (state and RNG stream position travel with the particle).
 * mcs_last_launches: launches the last mcs_run_pcut* took.
 * mcs_last_kernel: which transport kernel they ran -- 0 the general kernel, 1 its specialisation for the common configuration,
 * 2 the one for electrons with radiative losses, 3 the fp32-state kernel, 4 its plain-loop form, 5 its specialisation for
 * electrons with radiative losses, 6 the common configuration with ion -> electron energy transfer on.  * 7 / 8: the wave-specialised kernel for the common configuration / the same with energy transfer (MCS_K1_WS=1), 9: the fp32-state
 * plain loop with the exact primitives (MCS_F32_EXACT=1), 10: the general kernel's form for sliced launches, 11 / 12 / 13: the sliced forms
 * of 1 / 2 / 6 (what mcs_run_pcuts_pipelined launches). */
int mcs_set_tail_slicing(mcs_ctx* ctx, int budget_trips);
/* A species' pcuts first .. last queued back to back: transport, pcut_finalize and new_pcut (src/cuts.jl:34-124) of every pcut with
 * nothing read back in between -- n_saved, i_mult = max(n_target / n_saved, 1) (src/cuts.jl:42) and the size of the next population
 * are decided on the device.  What the loop `for i_pcut` of src/main_loops.jl:184-292 + :293-330 does for ONE rank whose shard is
 * the whole population (global indices 0, 1, 2, ...).  n_target[k]: target population after pcut first + k (N_PTS_PCUT or
 * N_PTS_PCUT_HI).  Outputs (host arrays of last - first + 1 entries): population, saved particles and i_mult of every pcut, the
 * kernel time of each transport launch (may be NULL).  Pcuts after one that saved nobody run on an empty population.  Per-particle
 * results are those of the mcs_run_pcut / mcs_new_pcut sequence, bit for bit. */
int mcs_run_pcuts_fused(mcs_ctx* ctx, int i_pcut_first, int i_pcut_last, const int64_t* n_target, int64_t* n_use, int64_t* n_saved,
                        int64_t* i_mult, double* kernel_ms);
int mcs_last_launches(mcs_ctx* ctx);
int mcs_last_kernel(mcs_ctx* ctx);
/* compute units of the context's device (the default grid of mcs_run_pcut* is 2 workgroups per CU; a caller that keeps two
 * contexts busy on one device gives each of them one per CU: mcs_set_launch(ctx, mcs_num_cus(ctx), 256)) */
int mcs_num_cus(mcs_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MCS_H */
