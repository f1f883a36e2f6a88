// mcs_api.hip -- the C ABI of include/mcs.h over the gfx950 kernels.
//
// A context owns: device mirrors of the grid/cut tables, the resident particle
// population (two SoA buffers: current and saved) and, unless the caller binds its
// own (mcs_bind_tallies), the flat fp64/int64 tally buffers.  All work is queued
// on ONE HIP stream; mcs_run_pcut is synchronous only for the 8-byte n_saved.
// There is no CPU code path in this library.
#include "mcs_device.h"
#include "../../include/mcs_ic.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>

extern "C" {
size_t mcs_transport_smem_bytes(int n_grid, int n_tcuts);
int mcs_transport_max_entries(void);
hipError_t mcs_launch_transport(const KArgs* a_dev, int kind, int blocks, int threads, hipStream_t st);
int mcs_transport_ws_threads(void);
hipError_t mcs_launch_finalize_split_dev(const uint8_t* l_save, long long cap_n, unsigned int* block_counts, unsigned long long* block_offsets,
                                         unsigned long long* scan_total, long long* src, PcutDev* pd, PcutDev* pd_next, unsigned long long* counters,
                                         long long n_target, unsigned long long* err, DevPop sv, DevPop out, int split_blocks, hipStream_t st);
hipError_t mcs_launch_transport_f32(const KArgs* a_dev, int kind, int blocks, int threads, hipStream_t st);
hipError_t mcs_launch_compact(const uint8_t* l_save, long long n, unsigned int* block_counts, unsigned long long* block_offsets,
                              unsigned long long* total_dev, long long* src, hipStream_t st);
hipError_t mcs_launch_split(DevPop sv, DevPop out, const long long* src, long long n_new, long long i_mult, hipStream_t st);
hipError_t mcs_launch_compact_match(const uint8_t* l_save, long long n, unsigned int* block_counts, unsigned long long* block_offsets,
                                    unsigned long long* total_dev, long long* src, unsigned int match, hipStream_t st);
hipError_t mcs_launch_late_split(const uint8_t* l_save, long long n, unsigned int* block_counts, unsigned long long* block_offsets,
                                 unsigned long long* total_dev, long long* src, PcutDev* pd, long long i_mult, long long n_main_next, DevPop sv,
                                 DevPop out_at_main_end, int split_blocks, hipStream_t st);
hipError_t mcs_launch_saved_export(DevPop sv, const long long* src, long long n_saved, long long cap, long long first,
                                   long long stride, const long long* gin, long long* gidx, double* f64, uint32_t* meta,
                                   hipStream_t st);
hipError_t mcs_launch_split_import(DevPop out, const double* f64, const uint32_t* meta, long long cap, long long i_mult,
                                   long long first, long long stride, long long n_local, hipStream_t st);
hipError_t mcs_launch_init_pop(DevPop out, const double* ptot_in, const double* weight_in, long long n, long long j_offset,
                               long long j_stride, long long n_total, unsigned long long key, double m, double u, double x_start,
                               int i_grid_start, int relativistic, int fast_push, double xn_per_fine, double x_grid_stop,
                               int n_bins, const double* bin_ptot, const double* bin_weight, const long long* bin_start,
                               hipStream_t st);
hipError_t mcs_launch_fill(double* p, long long n, double v, hipStream_t st);
hipError_t mcs_launch_fold_replicas(double* dst, double* rep, long long n, int n_rep, hipStream_t st);
hipError_t mcs_launch_copy(double* dst, const double* src, long long n, hipStream_t st);
hipError_t mcs_launch_eval(int fn, long long n, const double* a, const double* b, double* out, hipStream_t st);
hipError_t mcs_launch_dndp_cr(const mcs_params* P, const double* psd, const double* gam_sf, const double* ux, const double* tabs,
                              double rest_energy, double n0, double gam0, double* out_dndp, unsigned long long* diag, hipStream_t st);
hipError_t mcs_launch_dndp_2d(const mcs_params* P, const double* psd, const double* therm_sf, const unsigned long long* num_crossings, const double* tabs,
                              double rest_energy, double n0, int therm_from_hist, double gam_x, double beta_x, double* scratch, double* ef, hipStream_t st);
hipError_t mcs_launch_photon_ic(const double* ef, const double* p_edge, const double* field, int n_grid, int NM, int NT, int j_max, int n_nu, int n_photon,
                                double log_min_rm, double bins_per_dec, double mc_e, double beam_area, double* out, hipStream_t st);
hipError_t mcs_launch_photon_pion(const double* dndp_pf, const double* p_edge, const double* target, int n_grid, int NM, int n_photon,
                                  double log_emin_erg, double bins_per_dec, double mc, double aa, double scaling, int i_data, double* out, hipStream_t st);
hipError_t mcs_launch_photon_synch(const double* dndp_pf, const double* p_edge, const double* btot, int n_grid, int NM, int n_photon,
                                   double log_emin_erg, double bins_per_dec, double mc, double* out, hipStream_t st);
hipError_t mcs_launch_thermo(const mcs_params* P, const double* psd, const double* therm_pf, const unsigned long long* num_crossings,
                             const double* gam_sf, const double* ux, const double* tabs, double rest_energy, double mc, double n0,
                             int therm_from_hist, double* scratch, double* out3, hipStream_t st);
}

#define MCS_MEV_ERG_ 1.602176634e-6
namespace {
thread_local std::string g_err;

int fail(const std::string& msg) { g_err = msg; return 1; }
#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(e_));    \
  } while (0)

struct PopBuf {
  DevPop d{};
  long long cap = 0;
};
}  // namespace

struct mcs_ctx {
  mcs_params P;
  mcs_layout L;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // tables
  double* d_tab = nullptr;     // 8 tables x (n_grid+2)
  double* d_cuts = nullptr;    // pcuts | tcuts | x_spec | inj_fracs | eps_target
  DevTables tb{};
  std::vector<double> h_inj_fracs, h_pcuts, h_ux;
  // tallies
  double* d_T = nullptr; unsigned long long* d_I = nullptr; bool own_T = false, own_I = false;
  // population
  PopBuf cur, sav, spare;      // spare: target of the next split (buffers rotate, no per-pcut allocation)
  uint8_t* d_lsave = nullptr; long long lsave_cap = 0;
  long long n = 0;             // current population size
  long long n_saved_last = 0;
  long long n_run_last = 0;    // population size of the last mcs_run_pcut (the saved arrays and src[] refer to it)
  long long idx_first = 0, idx_stride = 1;   // global index of local particle k in that run: idx_first + k * idx_stride
  const long long* idx_gidx = nullptr;       // ... or idx_gidx[k] (mcs_run_pcut_indexed; caller-owned device memory)
  bool debug_finals = false;   // mcs_set_debug_finals: record per-particle end states (tests)
  int retro_cap = MCS_RETRO_CAP;
  // sliced tail (mcs_set_tail_slicing; KArgs "sliced launches"): after the queue is exhausted a wave makes tail_budget more trips,
  // exports its live particles and ends; the host relaunches them, spread over the chip's waves, until none is left
  int tail_budget = 0;         // trips; 0 = one launch per pcut, run to the end
  int tail_rounds_last = 0;    // launches the last mcs_run_pcut* took
  int claim_max_first = 64;    // MCS_CLAIM_MAX=<n> (environment): live particles per wave in the FIRST launch of a pcut (measurements of tau(L))
  double* d_strag[2] = {nullptr, nullptr}; long long strag_cap = 0;
  int f32_blocks_per_cu = 3;   // MCS_F32_BLOCKS=<n>: resident workgroups per CU the organised fp32 kernel is launched for (its occupancy)
  bool f32_exact = false;      // MCS_F32_EXACT=1: the plain loop with the exact fp32 primitives (include/mcs_math_f32.h): the kernel the CPU restatement
                               // oracle/mcs_oracle_f32.inc reproduces bit for bit (tests)
  bool f32_loop = false;       // MCS_F32_LOOP=1: the fp32-state variant as a plain per-lane loop (the reference semantics of that variant; tests)
  bool tail_ring = true;       // MCS_TAIL_RING=0: no precomputed scatter draws in the tail (A/B measurements)
  int tail_loop = 12;          // MCS_TAIL_LOOP=<n>: live lanes at or below which an exhausted wave runs the tight tail loop (0 = off; needs the tail ring)
  int refill_min = 12;         // MCS_REFILL_MIN=<n> (environment) overrides: A/B measurements
  int defer_k = 8;             // MCS_DEFER_K=<n> (environment) overrides: A/B measurements, 1 = no deferral
  // finals
  int32_t *f_reason = nullptr, *f_helix = nullptr, *f_retro = nullptr; double *f_ptot = nullptr, *f_x = nullptr;
  long long f_cap = 0;
  // scan scratch
  unsigned int* d_bcounts = nullptr; unsigned long long* d_boffs = nullptr; long long* d_src = nullptr; long long scan_cap = 0;
  double* d_tally_rep = nullptr;               // replicas of the histograms at the head of the tally buffer (KArgs::tally_rep)
  long long rep_n = 0;                         // doubles per replica (0: no replicas)
  bool rep_dirty = false;                    // a launch may have added to the replicas since the last fold
  bool tally_replicas = true;                  // MCS_TALLY_REPLICAS_OFF=1: tally straight into T
  bool park = true;                           // MCS_PARK=0: lanes that need the full Code Blocks run them at once (A/B measurements)
  unsigned long long* d_counters = nullptr;   // [0] work counter, [1] n_saved, [2] scan total
  // staging for init_pop
  double* d_stage = nullptr; long long stage_cap = 0;
  // launch constants: host copy in PINNED memory (the upload is then a true async copy: no staging through the runtime's
  // bounce buffer, ~10 us per pcut) and device copy; read-back words of a pcut, pinned for the same reason
  KArgs* h_args_pin = nullptr; KArgs* d_args = nullptr;
  unsigned long long* h_back = nullptr;       // [0..1] n_saved | count of l_save flags, [2] exported particles of a sliced run
  // species
  int i_iter = 1, i_ion = 1;
  double aa = 1, zzq = MCS_QCGS, m = MCS_MP, mc = MCS_MP * MCS_C, pmax_cutoff = 0, density = 1, ewf = 1;
  bool have_grid = false, have_cuts = false;
  bool all_parallel = false;   // theta == 0 in every zone (mcs_set_grid)
  bool tail_merge = true;      // MCS_TAIL_MERGE=0: no consolidation of sparse waves (A/B measurements)
  int kernel_last = -1;        // mcs_last_kernel
  // fused species loop (mcs_run_pcuts_fused): launch constants of every pcut (pinned + device), the per-pcut words decided on the
  // device, one event pair per pcut
  KArgs* h_fargs = nullptr; KArgs* d_fargs = nullptr; PcutDev* d_pd = nullptr; PcutDev* h_pd = nullptr; int fused_cap = 0;
  std::vector<hipEvent_t> f_ev;
  // pipelined pcut loop (mcs_run_pcuts_pipelined): the side stream on which a pcut's long histories finish while the next pcut runs,
  // the second set of saved arrays / status bytes (pcut p's are still written while pcut p + 1 runs), the late group's scan scratch,
  // per-launch counters (device, and pinned for the one read-back per pcut), three launch-constant slots, the late group's sizes
  hipStream_t pp_s2 = nullptr; hipEvent_t pp_reset = nullptr;
  // ... and, when the runtime grants them, two streams with complementary CU masks: the side stream's waves then have their SIMDs to
  // themselves (beside a wave of the main launch on the same SIMD a long history advances at half the speed -- the kernel is issue-bound
  // -- and the side chain, not the main launch, ends the pcut); a pcut with side work runs its main launch on the masked main stream
  hipStream_t pp_s1m = nullptr, pp_s2m = nullptr; int pp_side_cus = 12;      // MCS_PIPE_SIDE_CUS=<n> (0: no masks)
  PopBuf pp_sav2; uint8_t* pp_lsave2 = nullptr; long long pp_cap = 0;
  unsigned int* pp_bcounts = nullptr; unsigned long long* pp_boffs = nullptr; long long* pp_src = nullptr;
  unsigned long long* pp_dpc = nullptr; unsigned long long* pp_hpc = nullptr;
  KArgs* pp_hargs = nullptr; KArgs* pp_dargs = nullptr; PcutDev* pp_dpdl = nullptr; PcutDev* pp_hpdl = nullptr;
  int pp_side_waves = 0;       // MCS_PIPE_SIDE_WAVES=<n>: waves the resumed long histories are spread over (0: one per SIMD of the chip)
  int pp_side_max = 64;        // MCS_PIPE_SIDE_MAX=<n>: workgroups (of 2 per CU) the main launch leaves free for them at most
  int pp_waits_last = 0;       // pcuts of the last pipelined run whose i_mult had to wait for the long histories
  bool force_general = false;  // MCS_FORCE_GENERAL=1: always the general kernel (tests compare the two)
  int k1_ws = 2;               // the wave-specialised kernels (mcs_transport_ws.inc), where they apply: MCS_K1_WS=1 always, =0 never, default (2)
                               // for populations of at least ws_auto_min particles -- measured level with transport_body at 4e6 particles, 3.8 %
                               // faster at 1e7 and 7 % slower at 2e6, where its missing tail consolidation shows (profiles/r04_ws_kernel_ab.txt)
  long long ws_auto_min = 6000000;   // MCS_WS_AUTO_MIN=<n>
  int ws_pop_max = 0;          // MCS_WS_POP=<n>: particles a block of the wave-specialised kernel holds at most (0: lanes + 160)
  int ws_serve_min = 64;       // MCS_WS_SERVE=<n>: pending particles at which a wave serves them
  // consumers (K4): table staging, outputs, thermo scratch slab
  double* d_ctab = nullptr; double* d_cout = nullptr; double* d_cscratch = nullptr; unsigned long long* d_cdiag = nullptr;
  double* d_c2d = nullptr; bool have_c2d = false;    // d2N/dp dcos of the last mcs_dndp_2d ([n_grid][ntht+2][nmom+2]), the input of mcs_photon_ic
  // launch
  int blocks = 0, threads = 256;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_ms = 0.0;
  int n_cu = 256;
};

namespace {

int pop_alloc(mcs_ctx* c, PopBuf& b, long long cap) {
  if (cap <= b.cap) return 0;
  double** f[8] = {&b.d.weight, &b.d.ptot_pf, &b.d.pb_pf, &b.d.x_PT_cm, &b.d.xn_per, &b.d.prp_x_cm, &b.d.acctime_sec, &b.d.phi_rad};
  for (auto pp : f) { if (*pp) HIPCHK(hipFree(*pp)); *pp = nullptr; HIPCHK(hipMalloc((void**)pp, (size_t)cap * sizeof(double))); }
  if (b.d.meta) HIPCHK(hipFree(b.d.meta));
  HIPCHK(hipMalloc((void**)&b.d.meta, (size_t)cap * sizeof(uint32_t)));
  b.cap = cap;
  return 0;
}
void pop_free(PopBuf& b) {
  double* f[8] = {b.d.weight, b.d.ptot_pf, b.d.pb_pf, b.d.x_PT_cm, b.d.xn_per, b.d.prp_x_cm, b.d.acctime_sec, b.d.phi_rad};
  for (auto p : f) if (p) (void)hipFree(p);
  if (b.d.meta) (void)hipFree(b.d.meta);
  b = PopBuf{};
}

int ensure_capacity(mcs_ctx* c, long long n) {
  // both buffers and every per-particle side array hold at least n entries
  if (n > c->cur.cap || n > c->sav.cap) {
    long long cap = n + n / 8 + 1024;
    // growing must preserve the current population
    if (c->n > 0 && c->cur.cap < cap) {
      PopBuf nb;
      if (pop_alloc(c, nb, cap)) return 1;
      double* src[8] = {c->cur.d.weight, c->cur.d.ptot_pf, c->cur.d.pb_pf, c->cur.d.x_PT_cm, c->cur.d.xn_per, c->cur.d.prp_x_cm, c->cur.d.acctime_sec, c->cur.d.phi_rad};
      double* dst[8] = {nb.d.weight, nb.d.ptot_pf, nb.d.pb_pf, nb.d.x_PT_cm, nb.d.xn_per, nb.d.prp_x_cm, nb.d.acctime_sec, nb.d.phi_rad};
      for (int i = 0; i < 8; ++i) HIPCHK(hipMemcpyAsync(dst[i], src[i], (size_t)c->n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(nb.d.meta, c->cur.d.meta, (size_t)c->n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      pop_free(c->cur);
      c->cur = nb;
    } else if (pop_alloc(c, c->cur, cap)) return 1;
    if (pop_alloc(c, c->sav, cap)) return 1;
  }
  if (n > c->lsave_cap) {
    long long cap = n + n / 8 + 1024;
    if (c->d_lsave) HIPCHK(hipFree(c->d_lsave));
    HIPCHK(hipMalloc((void**)&c->d_lsave, (size_t)cap));
    c->lsave_cap = cap;
  }
  if (c->debug_finals && n > c->f_cap) {
    long long cap = n + n / 8 + 1024;
    if (c->f_reason) { (void)hipFree(c->f_reason); (void)hipFree(c->f_helix); (void)hipFree(c->f_retro); (void)hipFree(c->f_ptot); (void)hipFree(c->f_x); }
    HIPCHK(hipMalloc((void**)&c->f_reason, (size_t)cap * 4)); HIPCHK(hipMalloc((void**)&c->f_helix, (size_t)cap * 4));
    HIPCHK(hipMalloc((void**)&c->f_retro, (size_t)cap * 4)); HIPCHK(hipMalloc((void**)&c->f_ptot, (size_t)cap * 8));
    HIPCHK(hipMalloc((void**)&c->f_x, (size_t)cap * 8));
    c->f_cap = cap;
  }
  if (n > c->scan_cap) {
    long long cap = n + n / 8 + 1024;
    long long nb = (cap + 1023) / 1024;
    if (c->d_bcounts) { (void)hipFree(c->d_bcounts); (void)hipFree(c->d_boffs); (void)hipFree(c->d_src); }
    HIPCHK(hipMalloc((void**)&c->d_bcounts, (size_t)nb * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void**)&c->d_boffs, (size_t)nb * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void**)&c->d_src, (size_t)cap * sizeof(long long)));
    c->scan_cap = cap;
  }
  return 0;
}

int ensure_stage(mcs_ctx* c, long long n_doubles) {
  if (n_doubles <= c->stage_cap) return 0;
  if (c->d_stage) HIPCHK(hipFree(c->d_stage));
  HIPCHK(hipMalloc((void**)&c->d_stage, (size_t)n_doubles * sizeof(double)));
  c->stage_cap = n_doubles;
  return 0;
}

int fill(mcs_ctx* c, long long off, long long n, double v) {
  HIPCHK(mcs_launch_fill(c->d_T + off, n, v, c->stream));
  return 0;
}

// host <-> packed device SoA
int upload_soa(mcs_ctx* c, PopBuf& b, long long n, const mcs_soa* h) {
  const double* src[8] = {h->weight, h->ptot_pf, h->pb_pf, h->x_PT_cm, h->xn_per, h->prp_x_cm, h->acctime_sec, h->phi_rad};
  double* dst[8] = {b.d.weight, b.d.ptot_pf, b.d.pb_pf, b.d.x_PT_cm, b.d.xn_per, b.d.prp_x_cm, b.d.acctime_sec, b.d.phi_rad};
  for (int i = 0; i < 8; ++i) HIPCHK(hipMemcpyAsync(dst[i], src[i], (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  std::vector<uint32_t> meta((size_t)n);
  for (long long k = 0; k < n; ++k) meta[k] = mcs_pack_meta((int)h->grid[k], (int)h->tcut[k], h->downstream[k] != 0, h->inj[k] != 0);
  HIPCHK(hipMemcpyAsync(b.d.meta, meta.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int download_soa(mcs_ctx* c, const PopBuf& b, long long n, mcs_soa* h) {
  double* dst[8] = {h->weight, h->ptot_pf, h->pb_pf, h->x_PT_cm, h->xn_per, h->prp_x_cm, h->acctime_sec, h->phi_rad};
  const double* src[8] = {b.d.weight, b.d.ptot_pf, b.d.pb_pf, b.d.x_PT_cm, b.d.xn_per, b.d.prp_x_cm, b.d.acctime_sec, b.d.phi_rad};
  for (int i = 0; i < 8; ++i) HIPCHK(hipMemcpyAsync(dst[i], src[i], (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  std::vector<uint32_t> meta((size_t)n);
  HIPCHK(hipMemcpyAsync(meta.data(), b.d.meta, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (long long k = 0; k < n; ++k) {
    const uint32_t mm = meta[k];
    h->grid[k] = (int64_t)(mm & 0xffffu); h->tcut[k] = (int64_t)((mm >> 16) & 0xffu);
    h->downstream[k] = (uint8_t)((mm >> 24) & 1u); h->inj[k] = (uint8_t)((mm >> 25) & 1u);
  }
  return 0;
}

}  // namespace

extern "C" {

// The thermal histograms are tallied into MCS_TALLY_REPLICAS private copies (mcs_device.h); everything that reads
// or rewrites the tally buffer folds them in first.
static int fold_replicas(mcs_ctx* c) {
  if (!c->rep_dirty || !c->d_tally_rep) return 0;
  HIPCHK(mcs_launch_fold_replicas(c->d_T, c->d_tally_rep, c->rep_n, MCS_TALLY_REPLICAS, c->stream));
  c->rep_dirty = false;
  return 0;
}

int mcs_abi_version(void) { return MCS_ABI_VERSION; }
const char* mcs_last_error(void) { return g_err.c_str(); }
int mcs_get_layout(const mcs_params* p, mcs_layout* out) { mcs_tally_layout(p, out); return 0; }

int mcs_create(const mcs_params* p, int device, void* stream, mcs_ctx** out) {
  if (!p || !out) return fail("mcs_create: null argument");
  if (p->abi_version != MCS_ABI_VERSION) return fail("mcs_create: abi_version mismatch");
  if (p->use_custom_frg) return fail("Use of custom f(r_g) not yet supported. Add functionality or use standard. (src/scattering.jl:52-53)");
  if (!p->do_retro) return fail("Code not set up for analytical PRP calculations. (src/prob_return.jl:134)");
  if (p->num_psd_mom_bins + 1 > MCS_PSD_MAX || p->num_psd_tht_bins + 1 > MCS_PSD_MAX) return fail("mcs_create: psd bins exceed psd_max (src/parameters.jl:18)");
  if (p->n_grid < 1 || p->n_grid + 2 > mcs_transport_max_entries()) return fail("mcs_create: n_grid + 2 exceeds the LDS table size (208 entries; psd_max = 200 in src/parameters.jl:18)");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail("mcs_create: no HIP device visible; the transport path has no CPU fallback");
  if (device < 0 || device >= ndev) return fail("mcs_create: device ordinal out of range");
  HIPCHK(hipSetDevice(device));
  mcs_ctx* c = new mcs_ctx();
  // from here on a failing HIP call must not leak the context and what it has allocated so far
#define CRCHK(expr)                                                                          \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      const std::string m_ = std::string(#expr) + ": " + hipGetErrorString(e_);              \
      (void)mcs_destroy(c);                                                                  \
      return fail(m_);                                                                       \
    }                                                                                        \
  } while (0)
  { const char* e = std::getenv("MCS_FORCE_GENERAL"); c->force_general = e && e[0] == '1'; }
  { const char* e = std::getenv("MCS_TAIL_MERGE"); c->tail_merge = !(e && e[0] == '0'); }
  { const char* e = std::getenv("MCS_K1_WS"); c->k1_ws = !e ? 2 : (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : 2)); }
  { const char* e = std::getenv("MCS_WS_AUTO_MIN"); if (e && std::atoll(e) >= 0) c->ws_auto_min = std::atoll(e); }
  { const char* e = std::getenv("MCS_WS_POP"); if (e && std::atoi(e) >= 64 && std::atoi(e) <= 4096) c->ws_pop_max = std::atoi(e); }
  { const char* e = std::getenv("MCS_WS_SERVE"); if (e && std::atoi(e) >= 1 && std::atoi(e) <= 64) c->ws_serve_min = std::atoi(e); }
  { const char* e = std::getenv("MCS_PARK"); c->park = !(e && e[0] == '0'); }
  { const char* e = std::getenv("MCS_TAIL_RING"); c->tail_ring = !(e && e[0] == '0'); }
  { const char* e = std::getenv("MCS_TAIL_LOOP"); if (e && std::atoi(e) >= 0 && std::atoi(e) <= 32) c->tail_loop = std::atoi(e); }
  { const char* e = std::getenv("MCS_F32_LOOP"); c->f32_loop = e && e[0] == '1'; }
  { const char* e = std::getenv("MCS_F32_EXACT"); c->f32_exact = e && e[0] == '1'; }
  { const char* e = std::getenv("MCS_F32_BLOCKS"); if (e && std::atoi(e) >= 1 && std::atoi(e) <= 8) c->f32_blocks_per_cu = std::atoi(e); }
  { const char* e = std::getenv("MCS_TAIL_BUDGET"); if (e && std::atoi(e) >= 0) c->tail_budget = std::atoi(e); }
  { const char* e = std::getenv("MCS_CLAIM_MAX"); if (e && std::atoi(e) >= 1 && std::atoi(e) <= 64) c->claim_max_first = std::atoi(e); }
  { const char* e = std::getenv("MCS_PIPE_SIDE_WAVES"); if (e && std::atoi(e) >= 1) c->pp_side_waves = std::atoi(e); }
  { const char* e = std::getenv("MCS_PIPE_SIDE_CUS"); if (e && std::atoi(e) >= 0 && std::atoi(e) <= 128) c->pp_side_cus = std::atoi(e); }
  { const char* e = std::getenv("MCS_PIPE_SIDE_MAX"); if (e && std::atoi(e) >= 0 && std::atoi(e) <= 256) c->pp_side_max = std::atoi(e); }
  { const char* e = std::getenv("MCS_REFILL_MIN"); if (e && std::atoi(e) >= 1 && std::atoi(e) <= 48) c->refill_min = std::atoi(e); }
  { const char* e = std::getenv("MCS_DEFER_K"); if (e && std::atoi(e) >= 1 && std::atoi(e) <= 40) c->defer_k = std::atoi(e); }
  c->P = *p;
  mcs_tally_layout(p, &c->L);
  c->device = device;
  if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
  else { CRCHK(hipStreamCreate(&c->stream)); c->own_stream = true; }
  hipDeviceProp_t prop;
  CRCHK(hipGetDeviceProperties(&prop, device));
  c->n_cu = prop.multiProcessorCount;
  const int ne = p->n_grid + 2;
  CRCHK(hipMalloc((void**)&c->d_tab, (size_t)8 * ne * sizeof(double)));
  CRCHK(hipMalloc((void**)&c->d_counters, 8 * sizeof(unsigned long long)));
  CRCHK(hipMalloc((void**)&c->d_args, sizeof(KArgs)));
  CRCHK(hipHostMalloc((void**)&c->h_args_pin, sizeof(KArgs)));
  CRCHK(hipHostMalloc((void**)&c->h_back, 4 * sizeof(unsigned long long)));
  { const char* e = std::getenv("MCS_TALLY_REPLICAS_OFF"); c->tally_replicas = !(e && e[0] == '1'); }
  if (c->tally_replicas) {
    c->rep_n = c->L.total;     // the whole tally buffer: the three big histograms are 99 % of it
    const size_t nrep = (size_t)MCS_TALLY_REPLICAS * (size_t)c->rep_n;
    CRCHK(hipMalloc((void**)&c->d_tally_rep, nrep * sizeof(double)));
    CRCHK(hipMemsetAsync(c->d_tally_rep, 0, nrep * sizeof(double), c->stream));
  }
  CRCHK(hipMemsetAsync(c->d_counters, 0, 8 * sizeof(unsigned long long), c->stream));
  c->own_T = c->own_I = true;      // (before the allocations: mcs_destroy on a failure below frees whichever exists)
  CRCHK(hipMalloc((void**)&c->d_T, (size_t)c->L.total * sizeof(double)));
  CRCHK(hipMalloc((void**)&c->d_I, (size_t)mcs_i64_total(p) * sizeof(unsigned long long)));
  CRCHK(hipMemsetAsync(c->d_T, 0, (size_t)c->L.total * sizeof(double), c->stream));
  CRCHK(hipMemsetAsync(c->d_I, 0, (size_t)mcs_i64_total(p) * sizeof(unsigned long long), c->stream));
  CRCHK(hipEventCreate(&c->ev0));
  CRCHK(hipEventCreate(&c->ev1));

  CRCHK(hipStreamSynchronize(c->stream));
#undef CRCHK
  *out = c;
  return 0;
}

int mcs_destroy(mcs_ctx* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  pop_free(c->cur); pop_free(c->sav); pop_free(c->spare);
  void* ptrs[] = {c->d_tab, c->d_cuts, c->d_lsave, c->f_reason, c->f_helix, c->f_retro, c->f_ptot, c->f_x,
                  c->d_bcounts, c->d_boffs, c->d_src, c->d_counters, c->d_stage, c->d_args, c->d_tally_rep,
                  c->d_ctab, c->d_cout, c->d_cscratch, c->d_cdiag, c->d_strag[0], c->d_strag[1], c->d_c2d};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (c->own_T && c->d_T) (void)hipFree(c->d_T);
  if (c->own_I && c->d_I) (void)hipFree(c->d_I);
  if (c->h_args_pin) (void)hipHostFree(c->h_args_pin);
  if (c->h_back) (void)hipHostFree(c->h_back);
  if (c->h_fargs) (void)hipHostFree(c->h_fargs);
  if (c->h_pd) (void)hipHostFree(c->h_pd);
  if (c->d_fargs) (void)hipFree(c->d_fargs);
  if (c->d_pd) (void)hipFree(c->d_pd);
  pop_free(c->pp_sav2);
  { void* pq[] = {c->pp_lsave2, c->pp_bcounts, c->pp_boffs, c->pp_src, c->pp_dpc, c->pp_dargs, c->pp_dpdl}; for (void* q : pq) if (q) (void)hipFree(q); }
  if (c->pp_hpc) (void)hipHostFree(c->pp_hpc);
  if (c->pp_hargs) (void)hipHostFree(c->pp_hargs);
  if (c->pp_hpdl) (void)hipHostFree(c->pp_hpdl);
  if (c->pp_reset) (void)hipEventDestroy(c->pp_reset);
  if (c->pp_s2) (void)hipStreamDestroy(c->pp_s2);
  if (c->pp_s1m) (void)hipStreamDestroy(c->pp_s1m);
  if (c->pp_s2m) (void)hipStreamDestroy(c->pp_s2m);
  for (auto e : c->f_ev) (void)hipEventDestroy(e);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

int mcs_sync(mcs_ctx* c) {
  HIPCHK(hipSetDevice(c->device));
  if (fold_replicas(c)) return 1;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int mcs_bind_tallies(mcs_ctx* c, double* dev_f64, int64_t n_f64, int64_t* dev_i64, int64_t n_i64) {
  HIPCHK(hipSetDevice(c->device));
  if (fold_replicas(c)) return 1;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (dev_f64) {
    if (n_f64 < c->L.total) return fail("mcs_bind_tallies: f64 buffer smaller than layout.total");
    if (c->own_T && c->d_T) (void)hipFree(c->d_T);
    c->d_T = dev_f64; c->own_T = false;
  }
  if (dev_i64) {
    if (n_i64 < mcs_i64_total(&c->P)) return fail("mcs_bind_tallies: i64 buffer too small");
    if (c->own_I && c->d_I) (void)hipFree(c->d_I);
    c->d_I = (unsigned long long*)dev_i64; c->own_I = false;
  }
  return 0;
}
double* mcs_tallies_f64_devptr(mcs_ctx* c) { return c->d_T; }
int64_t* mcs_tallies_i64_devptr(mcs_ctx* c) { return (int64_t*)c->d_I; }

int mcs_set_grid(mcs_ctx* c, int n_entries, const double* x_grid_cm, const double* ux, const double* uz, const double* utot,
                 const double* gam_sf, const double* gam_ef, const double* beta_ef, const double* btot, const double* theta) {
  HIPCHK(hipSetDevice(c->device));
  const int ne = c->P.n_grid + 2;
  if (n_entries != ne) return fail("mcs_set_grid: n_entries != n_grid+2");
  for (int i = 0; i < ne; ++i) {
    if (!(btot[i] > 0) || !(utot[i] != 0)) return fail("mcs_set_grid: btot must be > 0 and utot != 0 in every zone");
    if (!std::isfinite(x_grid_cm[i])) return fail("mcs_set_grid: x_grid_cm must be finite (use +-1e30*rg0 sentinels)");
  }
  (void)beta_ef;  // passed by the reference (main_loops.jl:256) but never read by the path
  // The kernel detects "something happened at this move" by zone changes, so two facts of the
  // reference's grid are relied upon (setup_grid, src/initializers.jl:403-476): the shock x = 0 is a
  // zone boundary, and the upstream FEB lies inside zone i_grid_feb (MonteCarloScattering.jl:414).
  {
    bool has_zero = false;
    for (int i = 0; i < ne; ++i) {
      if (x_grid_cm[i] == 0.0) has_zero = true;
      if (i > 0 && !(x_grid_cm[i] > x_grid_cm[i - 1])) return fail("mcs_set_grid: x_grid_cm must be strictly increasing");
    }
    if (!has_zero) return fail("mcs_set_grid: x_grid_cm must contain the shock position 0.0 as a zone boundary");
    const int k = c->P.i_grid_feb;
    if (k < 0 || k + 1 >= ne || !(x_grid_cm[k + 1] > c->P.feb_upstream))
      return fail("mcs_set_grid: feb_upstream must lie below x_grid_cm[i_grid_feb+1]");
  }
  const double* src[8] = {x_grid_cm, ux, uz, utot, gam_sf, gam_ef, btot, theta};
  std::vector<double> h((size_t)8 * ne);
  for (int t = 0; t < 8; ++t) std::memcpy(&h[(size_t)t * ne], src[t], (size_t)ne * sizeof(double));
  HIPCHK(hipMemcpyAsync(c->d_tab, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->tb.x_grid = c->d_tab; c->tb.ux = c->d_tab + ne; c->tb.uz = c->d_tab + 2 * ne; c->tb.utot = c->d_tab + 3 * ne;
  c->tb.gsf = c->d_tab + 4 * ne; c->tb.gef = c->d_tab + 5 * ne; c->tb.btot = c->d_tab + 6 * ne; c->tb.theta = c->d_tab + 7 * ne;
  c->h_ux.assign(ux, ux + ne);
  c->all_parallel = true;
  for (int i = 0; i < ne; ++i) if (theta[i] != 0.0) c->all_parallel = false;
  c->have_grid = true;
  return 0;
}

int mcs_set_cuts(mcs_ctx* c, int n_pcuts, const double* pcuts, int n_tcuts, const double* tcuts, int n_xspec,
                 const double* x_spec, const double* inj_fracs, const double* eps_target) {
  HIPCHK(hipSetDevice(c->device));
  if (n_pcuts < 1 || n_pcuts > MCS_NA_C) return fail("momentum-cutoffs: parameter na_c smaller than desired number of pcuts.");
  if (n_tcuts + 1 > MCS_NA_C) return fail("TCUTS: parameter na_c smaller than desired number of tcuts.");
  if (n_tcuts > 255) return fail("mcs_set_cuts: n_tcuts > 255");
  if (n_xspec > c->P.n_grid) return fail("mcs_set_cuts: n_xspec > n_grid");
  const int ng = c->P.n_grid, ni = c->P.n_ions;
  std::vector<double> h;
  h.insert(h.end(), pcuts, pcuts + n_pcuts);
  h.insert(h.end(), tcuts, tcuts + n_tcuts);
  h.insert(h.end(), x_spec, x_spec + n_xspec);
  h.insert(h.end(), inj_fracs, inj_fracs + ni);
  h.insert(h.end(), eps_target, eps_target + ng);
  if (c->d_cuts) HIPCHK(hipFree(c->d_cuts));
  HIPCHK(hipMalloc((void**)&c->d_cuts, h.size() * sizeof(double)));
  HIPCHK(hipMemcpyAsync(c->d_cuts, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  double* q = c->d_cuts;
  c->tb.pcuts = q; q += n_pcuts; c->tb.tcuts = q; q += n_tcuts; c->tb.x_spec = q; q += n_xspec;
  c->tb.inj_fracs = q; q += ni; c->tb.eps_target = q;
  c->tb.n_pcuts = n_pcuts; c->tb.n_tcuts = n_tcuts; c->tb.n_xspec = n_xspec;
  c->h_pcuts.assign(pcuts, pcuts + n_pcuts);
  c->h_inj_fracs.assign(inj_fracs, inj_fracs + ni);
  c->have_cuts = true;
  return 0;
}

int mcs_begin_iteration(mcs_ctx* c, int i_iter) {
  HIPCHK(hipSetDevice(c->device));
  if (i_iter < 1 || i_iter > c->P.n_itrs) return fail("mcs_begin_iteration: i_iter out of 1..n_itrs");
  const mcs_params& P = c->P;
  c->i_iter = i_iter;
  if (fill(c, c->L.pxx_flux, P.n_grid, MCS_FLOOR) || fill(c, c->L.pxz_flux, P.n_grid, MCS_FLOOR) ||
      fill(c, c->L.energy_flux, P.n_grid, MCS_FLOOR) || fill(c, c->L.weight_coupled, (long long)MCS_NA_C * P.n_ions, MCS_FLOOR) ||
      fill(c, c->L.scalars, 4, MCS_FLOOR) || fill(c, c->L.energy_transfer_pool, P.n_grid, 0.0) ||
      fill(c, c->L.energy_recv_pool, P.n_grid, 0.0))
    return 1;
  return 0;
}

int mcs_begin_species(mcs_ctx* c, int i_iter, int i_ion, double aa, double zz, double pmax_cutoff, double density, double ewf) {
  HIPCHK(hipSetDevice(c->device));
  const mcs_params& P = c->P;
  if (i_ion < 1 || i_ion > P.n_ions) return fail("mcs_begin_species: i_ion out of 1..n_ions");
  if (i_iter < 1 || i_iter > P.n_itrs) return fail("mcs_begin_species: i_iter out of 1..n_itrs");
  if (!(aa > 0) || zz == 0) return fail("mcs_begin_species: aa must be > 0 and zz != 0");
  c->i_iter = i_iter; c->i_ion = i_ion; c->aa = aa; c->zzq = zz * MCS_QCGS; c->m = aa * MCS_MP; c->mc = c->m * MCS_C;
  c->pmax_cutoff = pmax_cutoff; c->density = density; c->ewf = ewf;
  if (fold_replicas(c)) return 1;      // (the previous species' replicas, before its histograms are cleared)
  const long long npsd = c->L.psd_stride_zone * P.n_grid;
  const long long pm = MCS_PSD_MAX + 1;
  if (fill(c, c->L.psd, npsd, MCS_FLOOR) || fill(c, c->L.therm_sf, 2 * npsd, 0.0) ||
      fill(c, c->L.esc_psd_up, 2 * pm * pm, MCS_FLOOR) || fill(c, c->L.pxx_flux, 3LL * P.n_grid, 0.0))
    return 1;
  HIPCHK(hipMemsetAsync(c->d_I + MCS_I_NUM_CROSSINGS, 0, (size_t)P.n_grid * sizeof(unsigned long long), c->stream));
  HIPCHK(mcs_launch_copy(c->d_T + c->L.energy_recv_pool, c->d_T + c->L.energy_transfer_pool, P.n_grid, c->stream));
  return 0;
}

int mcs_set_fluxes(mcs_ctx* c, const double* pxx, const double* pxz, const double* en) {
  HIPCHK(hipSetDevice(c->device));
  const int ng = c->P.n_grid;
  HIPCHK(hipMemcpyAsync(c->d_T + c->L.pxx_flux, pxx, ng * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_T + c->L.pxz_flux, pxz, ng * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_T + c->L.energy_flux, en, ng * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int mcs_pop_upload(mcs_ctx* c, int64_t n, const mcs_soa* host) {
  HIPCHK(hipSetDevice(c->device));
  if (n < 0) return fail("mcs_pop_upload: n < 0");
  for (int64_t k = 0; k < n; ++k) {
    if (!(host->ptot_pf[k] > 0)) return fail("mcs_pop_upload: ptot_pf must be > 0 (zero-momentum particle: reference quirk G6)");
    if (host->grid[k] < 0 || host->grid[k] > c->P.n_grid) return fail("mcs_pop_upload: grid index out of 0..n_grid");
    if (host->tcut[k] < 1 || host->tcut[k] > 255) return fail("mcs_pop_upload: tcut out of range");
    // a non-finite position or momentum never satisfies an exit test (NaN compares false): the helix loop would run
    // into its cap and the retro walk into MCS_RETRO_CAP -- refuse it here (the reference would loop forever)
    if (!std::isfinite(host->ptot_pf[k]) || !std::isfinite(host->pb_pf[k]) || !std::isfinite(host->x_PT_cm[k]) ||
        !std::isfinite(host->prp_x_cm[k]) || !std::isfinite(host->acctime_sec[k]) || !std::isfinite(host->phi_rad[k]) ||
        !std::isfinite(host->weight[k]))
      return fail("mcs_pop_upload: weight, ptot_pf, pb_pf, x_PT_cm, prp_x_cm, acctime_sec and phi_rad must be finite");
    if (!(host->xn_per[k] > 0) || !std::isfinite(host->xn_per[k])) return fail("mcs_pop_upload: xn_per must be finite and > 0");
  }
  c->n = 0; c->n_run_last = -1; c->n_saved_last = 0; c->idx_gidx = nullptr;   // the saved arrays / src[] of the last run no longer describe this population
  if (ensure_capacity(c, n)) return 1;
  if (n > 0 && upload_soa(c, c->cur, n, host)) return 1;
  c->n = n;
  return 0;
}
int mcs_pop_download(mcs_ctx* c, int64_t n, mcs_soa* host) {
  HIPCHK(hipSetDevice(c->device));
  if (n > c->n) return fail("mcs_pop_download: n exceeds the population size");
  return download_soa(c, c->cur, n, host);
}
int mcs_saved_download(mcs_ctx* c, int64_t n, mcs_soa* host, uint8_t* l_save) {
  HIPCHK(hipSetDevice(c->device));
  if (n > c->n) return fail("mcs_saved_download: n exceeds the population size");
  if (host && download_soa(c, c->sav, n, host)) return 1;
  std::vector<uint8_t> own;
  if (!l_save && host) { own.resize((size_t)n); l_save = own.data(); }
  if (l_save) {
    HIPCHK(hipMemcpyAsync(l_save, c->d_lsave, (size_t)n, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream));
    for (int64_t k = 0; k < n; ++k) l_save[k] = l_save[k] == 1;      // (the device byte is a status: 1 saved, 2 ended)
  }
  if (host) {       // the *_saved arrays of the reference hold zeros where nothing was saved (main_loops.jl:184-197)
    for (int64_t k = 0; k < n; ++k) {
      if (l_save[k]) continue;
      host->weight[k] = 0; host->ptot_pf[k] = 0; host->pb_pf[k] = 0; host->x_PT_cm[k] = 0; host->xn_per[k] = 0;
      host->prp_x_cm[k] = 0; host->acctime_sec[k] = 0; host->phi_rad[k] = 0;
      host->grid[k] = 0; host->tcut[k] = 0; host->downstream[k] = 0; host->inj[k] = 0;
    }
  }
  return 0;
}
int64_t mcs_pop_size(mcs_ctx* c) { return c->n; }

int mcs_init_pop(mcs_ctx* c, int64_t n, int64_t j_offset, int64_t n_total, const double* ptot_pf_in, const double* weight_in,
                 double x_start_cm, int i_grid_start, int relativistic, int fast_push) {
  HIPCHK(hipSetDevice(c->device));
  if (!c->have_grid) return fail("mcs_init_pop: call mcs_set_grid first");
  if (n < 0 || i_grid_start < 0 || i_grid_start > c->P.n_grid) return fail("mcs_init_pop: bad arguments");
  if (!std::isfinite(x_start_cm)) return fail("mcs_init_pop: x_start_cm must be finite");
  for (int64_t k = 0; k < n; ++k)
    if (!(ptot_pf_in[k] > 0) || !std::isfinite(ptot_pf_in[k]) || !std::isfinite(weight_in[k]))
      return fail("mcs_init_pop: ptot_pf must be finite and > 0 (reference quirk G6), weight finite");
  c->n = 0; c->n_run_last = -1; c->n_saved_last = 0; c->idx_gidx = nullptr;   // the saved arrays / src[] of the last run no longer describe this population
  if (ensure_capacity(c, n)) return 1;
  if (ensure_stage(c, 2 * n + 2)) return 1;
  if (n > 0) {
    HIPCHK(hipMemcpyAsync(c->d_stage, ptot_pf_in, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_stage + n, weight_in, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const unsigned long long key = (unsigned long long)((long long)(c->i_iter - 1) * c->P.n_ions + (c->i_ion - 1));
    HIPCHK(mcs_launch_init_pop(c->cur.d, c->d_stage, c->d_stage + n, n, j_offset, 1, n_total, key, c->m, c->h_ux[i_grid_start],
                               x_start_cm, i_grid_start, relativistic, fast_push, c->P.xn_per_fine, c->P.x_grid_stop, 0, nullptr, nullptr, nullptr,
                               c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  c->n = n;
  return 0;
}

int mcs_init_pop_binned(mcs_ctx* c, int64_t n, int64_t j_offset, int64_t n_total, int n_bins, const double* bin_ptot_pf,
                        const double* bin_weight, const int64_t* bin_start, double x_start_cm, int i_grid_start,
                        int relativistic, int fast_push) {
  return mcs_init_pop_binned_strided(c, n, j_offset, 1, n_total, n_bins, bin_ptot_pf, bin_weight, bin_start, x_start_cm,
                                     i_grid_start, relativistic, fast_push);
}

int mcs_init_pop_binned_strided(mcs_ctx* c, int64_t n, int64_t j_offset, int64_t j_stride, int64_t n_total, int n_bins,
                                const double* bin_ptot_pf, const double* bin_weight, const int64_t* bin_start, double x_start_cm,
                                int i_grid_start, int relativistic, int fast_push) {
  HIPCHK(hipSetDevice(c->device));
  if (!c->have_grid) return fail("mcs_init_pop_binned: call mcs_set_grid first");
  if (n < 0 || n_bins < 1 || n_bins > 4096 || i_grid_start < 0 || i_grid_start > c->P.n_grid || !bin_ptot_pf || !bin_weight || !bin_start)
    return fail("mcs_init_pop_binned: bad arguments");
  if (bin_start[0] != 0 || bin_start[n_bins] != n_total || j_offset < 0 || j_stride < 1 ||
      (n > 0 && j_offset + (n - 1) * j_stride >= n_total))
    return fail("mcs_init_pop_binned: bin_start must run from 0 to n_total and the shard must lie inside");
  for (int b = 0; b < n_bins; ++b) {
    if (bin_start[b + 1] < bin_start[b]) return fail("mcs_init_pop_binned: bin_start must be non-decreasing");
    if (bin_start[b + 1] > bin_start[b] && (!(bin_ptot_pf[b] > 0) || !std::isfinite(bin_ptot_pf[b]) || !std::isfinite(bin_weight[b])))
      return fail("mcs_init_pop_binned: ptot_pf must be finite and > 0 (reference quirk G6), weight finite");
  }
  if (!std::isfinite(x_start_cm)) return fail("mcs_init_pop_binned: x_start_cm must be finite");
  c->n = 0; c->n_run_last = -1; c->n_saved_last = 0; c->idx_gidx = nullptr;   // the saved arrays / src[] of the last run no longer describe this population
  if (ensure_capacity(c, n)) return 1;
  const size_t nd = (size_t)3 * n_bins + 1;      // ptot | weight | start (int64 in a double slot)
  if (ensure_stage(c, (long long)nd + 2)) return 1;
  if (n > 0) {
    std::vector<double> h(nd);
    std::memcpy(h.data(), bin_ptot_pf, sizeof(double) * n_bins);
    std::memcpy(h.data() + n_bins, bin_weight, sizeof(double) * n_bins);
    std::memcpy(h.data() + 2 * n_bins, bin_start, sizeof(int64_t) * (n_bins + 1));
    HIPCHK(hipMemcpyAsync(c->d_stage, h.data(), nd * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const unsigned long long key = (unsigned long long)((long long)(c->i_iter - 1) * c->P.n_ions + (c->i_ion - 1));
    HIPCHK(mcs_launch_init_pop(c->cur.d, nullptr, nullptr, n, j_offset, j_stride, n_total, key, c->m, c->h_ux[i_grid_start], x_start_cm,
                               i_grid_start, relativistic, fast_push, c->P.xn_per_fine, c->P.x_grid_stop, n_bins, c->d_stage,
                               c->d_stage + n_bins, (const long long*)(c->d_stage + 2 * n_bins), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));     // h goes out of scope
  }
  c->n = n;
  return 0;
}

int mcs_num_cus(mcs_ctx* c) { return c->n_cu; }

int mcs_set_launch(mcs_ctx* c, int blocks, int threads) {
  if (threads != 0 && (threads % 64 != 0 || threads > 256)) return fail("mcs_set_launch: threads must be a multiple of 64, <= 256");
  c->blocks = blocks; c->threads = threads ? threads : 256;
  return 0;
}

int mcs_set_debug_finals(mcs_ctx* c, int on) { c->debug_finals = on != 0; return 0; }
int mcs_set_tail_slicing(mcs_ctx* c, int budget_trips) {
  if (budget_trips < 0 || budget_trips > (1 << 24)) return fail("mcs_set_tail_slicing: budget out of range");
  if (budget_trips > 0 && c->P.state_fp32) return fail("mcs_set_tail_slicing: the fp32-state kernels are not sliced (fp64 contexts only)");
  c->tail_budget = budget_trips;
  return 0;
}
int mcs_last_launches(mcs_ctx* c) { return c->tail_rounds_last; }
int mcs_last_kernel(mcs_ctx* c) { return c->kernel_last; }
int mcs_set_retro_cap(mcs_ctx* c, int64_t cap) {
  if (cap < 0 || cap > 2000000000LL) return fail("mcs_set_retro_cap: cap out of range");
  c->retro_cap = cap > 0 ? (int)cap : MCS_RETRO_CAP;
  return 0;
}

int mcs_run_pcut(mcs_ctx* c, int i_pcut, int64_t i_prt_offset, int64_t* n_saved) {
  return mcs_run_pcut_strided(c, i_pcut, i_prt_offset, 1, n_saved);
}

static int run_pcut_impl(mcs_ctx* c, int i_pcut, int64_t i_prt_offset, int64_t i_prt_stride, const int64_t* dev_gidx, int64_t* n_saved);

int mcs_run_pcut_strided(mcs_ctx* c, int i_pcut, int64_t i_prt_offset, int64_t i_prt_stride, int64_t* n_saved) {
  return run_pcut_impl(c, i_pcut, i_prt_offset, i_prt_stride, nullptr, n_saved);
}

int mcs_run_pcut_indexed(mcs_ctx* c, int i_pcut, const int64_t* dev_gidx, int64_t* n_saved) {
  if (!dev_gidx && c->n > 0) return fail("mcs_run_pcut_indexed: null index list");
  return run_pcut_impl(c, i_pcut, 0, 1, dev_gidx, n_saved);
}

// the launch constants of one pcut that do not depend on the kernel chosen (shared by mcs_run_pcut* and mcs_run_pcuts_fused)
static void fill_kargs(mcs_ctx* c, KArgs& a, int i_pcut, long long n, long long i_prt_offset, long long i_prt_stride, const long long* dev_gidx, int budget) {
  std::memset(&a, 0, sizeof(a));
  a.P = c->P; a.L = c->L; a.tb = c->tb; a.in = c->cur.d; a.sv = c->sav.d; a.l_save = c->d_lsave;
  a.T = c->d_T; a.I = c->d_I;
  a.aa = c->aa; a.zzq = c->zzq; a.m = c->m; a.mc = c->mc; a.pmax_cutoff = c->pmax_cutoff; a.density = c->density; a.ewf = c->ewf;
  a.inj_frac = c->h_inj_fracs[c->i_ion - 1];
  a.pcut = c->h_pcuts[i_pcut - 1];
  a.pcut_prev = i_pcut > 1 ? c->h_pcuts[i_pcut - 2] : 0.0;
  a.i_iter = c->i_iter; a.i_ion = c->i_ion; a.i_pcut = i_pcut;
  a.n = n; a.i_prt_offset = i_prt_offset; a.i_prt_stride = i_prt_stride; a.gidx = dev_gidx;
  a.retro_cap = c->retro_cap;
  a.defer_k = c->defer_k;
  a.refill_min = c->refill_min;
  // a wave whose live lanes all wait for company (fewer than defer_k of them) must be able to refill: with
  // defer_k + refill_min <= 64 either defer_k lanes are live or refill_min are idle (see the deferral in transport_body)
  if (a.defer_k > 64 - a.refill_min) a.defer_k = 64 - a.refill_min;
  a.tail_ring = c->tail_ring ? 1 : 0;
  a.tail_loop = c->tail_ring ? c->tail_loop : 0;
  // iseed_mod - i_prt, src/particle_loop.jl:35-40
  a.seed_base = (unsigned long long)((long long)(c->i_iter - 1) * c->P.n_pts_max * c->tb.n_pcuts * c->P.n_ions +
                                     (long long)(c->i_ion - 1) * c->P.n_pts_max * c->tb.n_pcuts +
                                     (long long)(i_pcut - 1) * c->P.n_pts_max);
  a.work_counter = c->d_counters; a.n_saved = c->d_counters + 1;
  a.tail_merge = c->tail_merge ? 1 : 0;
  a.wait_full = c->park ? 1 : 0;
  a.tally_rep = c->d_tally_rep; a.rep_n = c->d_tally_rep ? c->rep_n : 0;
  if (c->debug_finals) { a.f_reason = c->f_reason; a.f_helix = c->f_helix; a.f_retro = c->f_retro; a.f_ptot = c->f_ptot; a.f_x = c->f_x; }
  a.claim_max = 64; a.budget_trips = budget; a.strag_count = c->d_counters + 3;
  a.strag_out = c->d_strag[0];
}

// which transport kernel a context's current species runs (mcs_launch_transport's `kind`), without the sliced / explicit-geometry cases
static int species_kernel_kind(mcs_ctx* c, const KArgs& a, long long n, bool* ws_out) {
  const bool plain_but_etf = !c->force_general && c->all_parallel && !c->P.dont_scatter && !c->P.use_custom_epsB &&
                             !c->P.dont_DSA && !(c->P.feb_downstream > 0) && c->aa >= 1 && c->tb.n_xspec == 0 && !(a.inj_frac < 1);
  const bool plain = plain_but_etf && !(c->P.energy_transfer_frac > 0);
  const bool plain_etf = plain_but_etf && !plain;
  const bool lossy = !c->force_general && c->P.do_rad_losses && c->aa < 1 && !c->P.use_custom_epsB && !c->P.dont_scatter;
  const bool ws = (c->k1_ws == 1 || (c->k1_ws == 2 && n >= c->ws_auto_min)) && (plain || plain_etf) && !c->P.state_fp32;
  if (ws_out) *ws_out = ws;
  if (c->P.state_fp32) return c->f32_exact ? 3 : (c->f32_loop ? 1 : (lossy ? 2 : 0));
  return ws ? (plain ? 7 : 8) : (plain ? 1 : (lossy ? 2 : (plain_etf ? 6 : 0)));
}

static int run_pcut_impl(mcs_ctx* c, int i_pcut, int64_t i_prt_offset, int64_t i_prt_stride, const int64_t* dev_gidx, int64_t* n_saved) {
  HIPCHK(hipSetDevice(c->device));
  if (i_prt_offset < 0 || i_prt_stride < 1) return fail("mcs_run_pcut: i_prt_first must be >= 0 and i_prt_stride >= 1");
  if (!c->have_grid || !c->have_cuts) return fail("mcs_run_pcut: grid/cuts not set");
  if (i_pcut < 1 || i_pcut > c->tb.n_pcuts) return fail("mcs_run_pcut: i_pcut out of range");
  const long long n = c->n;
  if (ensure_capacity(c, n)) return 1;
  // main_loops.jl:184-197: l_save and the *_saved arrays start at zero.  On the device only l_save is cleared: K2 and the
  // exports read saved entries through the compacted index list, and mcs_saved_download zeroes the entries of unsaved
  // particles in the host copy it hands out (nine fills per pcut less in the timed path).
  if (n > 0) HIPCHK(hipMemsetAsync(c->d_lsave, 0, (size_t)n, c->stream));
  HIPCHK(hipMemsetAsync(c->d_counters, 0, 4 * sizeof(unsigned long long), c->stream));
  const int budget = (c->P.state_fp32 || n == 0) ? 0 : c->tail_budget;      // (the fp32 study kernel is not sliced)
  if (budget > 0) {
    // one entry per lane a launch can hold: 2 workgroups of 256 threads per CU, or the geometry of mcs_set_launch if that is larger
    long long need_cap = (long long)2 * c->n_cu * 256;
    if (c->blocks > 0 && (long long)c->blocks * c->threads > need_cap) need_cap = (long long)c->blocks * c->threads;
    if (need_cap > c->strag_cap) {
      HIPCHK(hipStreamSynchronize(c->stream));
      for (int b = 0; b < 2; ++b) { if (c->d_strag[b]) (void)hipFree(c->d_strag[b]); c->d_strag[b] = nullptr; }
      c->strag_cap = need_cap;
      for (int b = 0; b < 2; ++b) HIPCHK(hipMalloc((void**)&c->d_strag[b], (size_t)c->strag_cap * MCS_STRAG_WORDS * sizeof(double)));
    }
  }

  KArgs& a = *c->h_args_pin;     // (every launch below is followed by a stream synchronisation before this is written again)
  fill_kargs(c, a, i_pcut, n, i_prt_offset, i_prt_stride, (const long long*)dev_gidx, budget);

  const int threads = c->threads;
  int blocks = c->blocks;
  // two 256-thread blocks are resident per CU (78 KB of LDS each); the fp32-state kernel (27 KB, 119 VGPRs) fits four
  // (resident workgroups per CU: 2 for the fp64 kernel -- 78 KB of LDS each --, 3 for the organised fp32 kernel -- 168 VGPRs, 51 KB --,
  // 4 for its plain-loop form)
  const long long full = (long long)c->n_cu * (c->P.state_fp32 ? ((c->f32_loop || c->f32_exact) ? 3 : c->f32_blocks_per_cu) : 2);
  if (blocks <= 0) {
    // persistent lanes: fill the chip, never launch more lanes than particles
    const long long want = (n + threads - 1) / threads;
    blocks = (int)(want < full ? want : full);
    if (blocks < 1) blocks = 1;
  }
  if (c->claim_max_first < 64 && !c->P.state_fp32) {
    a.claim_max = c->claim_max_first; a.defer_k = 1; a.wait_full = 0; a.tail_merge = 0;
    const long long per_block = (long long)(threads / 64) * a.claim_max;
    const long long nb = (n + per_block - 1) / per_block;
    if (c->blocks <= 0) blocks = (int)(nb < full ? (nb > 0 ? nb : 1) : full);
  }
  if (budget > 0 && (long long)blocks * threads > c->strag_cap) return fail("mcs_run_pcut: launch geometry exceeds the export buffer of a sliced run");
  // the specialised kernel for the common configuration (see transport_body<PLAIN> in mcs_transport.hip)
  const bool plain_but_etf = !c->force_general && c->all_parallel && !c->P.dont_scatter && !c->P.use_custom_epsB &&
                             !c->P.dont_DSA && !(c->P.feb_downstream > 0) && c->aa >= 1 && c->tb.n_xspec == 0 && !(a.inj_frac < 1);
  const bool plain = plain_but_etf && !(c->P.energy_transfer_frac > 0);
  const bool plain_etf = plain_but_etf && !plain;      // the ions of a run with energy transfer: PLAIN with that one flag at run time
  // the specialised kernel for electrons with radiative losses (transport_body<false, LOSSY>): the loss in line in the common pass
  const bool lossy = !c->force_general && c->P.do_rad_losses && c->aa < 1 && !c->P.use_custom_epsB && !c->P.dont_scatter;
  // sliced launches (suspend / resume, fewer than 64 particles per wave) run the general kernel's SLICED form
  const bool sliced = !c->P.state_fp32 && (budget > 0 || c->claim_max_first < 64);
  // the wave-specialised form of those two (mcs_transport_ws.inc): 512-thread blocks, one per CU
  const bool ws = (c->k1_ws == 1 || (c->k1_ws == 2 && n >= c->ws_auto_min)) && (plain || plain_etf) && !c->P.state_fp32 && budget == 0 &&
                  c->claim_max_first == 64 && c->blocks <= 0;
  int k1_threads = threads;
  if (ws) {
    k1_threads = mcs_transport_ws_threads();
    const long long want = (n + k1_threads - 1) / k1_threads;
    blocks = (int)(want < c->n_cu ? (want > 0 ? want : 1) : c->n_cu);
    a.ws_pop_max = c->ws_pop_max > 0 ? c->ws_pop_max : k1_threads + 160;
    a.ws_serve_min = c->ws_serve_min;
  }
  double ms_total = 0.0;
  c->tail_rounds_last = 0;
  c->kernel_last = c->P.state_fp32 ? (c->f32_exact ? 9 : (c->f32_loop ? 4 : (lossy ? 5 : 3))) : (sliced ? 10 : (ws ? (plain ? 7 : 8) : (plain ? 1 : (lossy ? 2 : (plain_etf ? 6 : 0)))));
  for (int round = 0;; ++round) {
    HIPCHK(hipMemcpyAsync(c->d_args, c->h_args_pin, sizeof(KArgs), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->ev0, c->stream));
    if (n > 0) {
      if (c->P.state_fp32) HIPCHK(mcs_launch_transport_f32(c->d_args, c->f32_exact ? 3 : (c->f32_loop ? 1 : (lossy ? 2 : 0)), blocks, 256, c->stream));
      else HIPCHK(mcs_launch_transport(c->d_args, sliced ? 10 : (ws ? (plain ? 7 : 8) : (plain ? 1 : (lossy ? 2 : (plain_etf ? 6 : 0)))), blocks, k1_threads, c->stream));
      c->rep_dirty = true;
    }
    HIPCHK(hipEventRecord(c->ev1, c->stream));
    ++c->tail_rounds_last;
    if (budget == 0) break;
    // sliced run: how many particles did the launch export?  They are the next launch's queue, spread over the chip's waves:
    // a pass costs a wave the same with 1 live lane as with 64, but the rare work of every live lane stalls all the others,
    // so the fewer particles share a wave the faster each history advances -- and the launch waits for its longest one.
    HIPCHK(hipMemcpyAsync(c->h_back + 2, c->d_counters + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const unsigned long long n_x = c->h_back[2];
    float ms_r = 0.f;
    HIPCHK(hipEventElapsedTime(&ms_r, c->ev0, c->ev1));
    ms_total += ms_r;
    if (n_x == 0) break;
    if ((long long)n_x > c->strag_cap) return fail("mcs_run_pcut: export buffer overrun");
    HIPCHK(hipMemsetAsync(c->d_counters, 0, sizeof(unsigned long long), c->stream));        // work counter
    HIPCHK(hipMemsetAsync(c->d_counters + 3, 0, sizeof(unsigned long long), c->stream));    // export counter
    a.strag_in = c->d_strag[round & 1]; a.strag_out = c->d_strag[(round + 1) & 1];
    a.n_resume = (long long)n_x; a.fresh_lo = n;
    // measured (profiles/r03_tau_vs_lanes.txt: 2048 particles, kernel time of 14 pcuts): 64 particles per wave 45.6 ms, 32: 38.1,
    // 16: 35.6, 8: 33.4, 4: 31.6, 2 (one wave per SIMD): 31.7, 1 (two waves per SIMD): 37.9 -- so one wave per SIMD, as few
    // particles per wave as that allows, and a dense launch when that would be more than 16
    const long long waves1 = (long long)c->n_cu * (threads / 64);           // one wave per SIMD
    long long cm = ((long long)n_x + waves1 - 1) / waves1;
    if (cm > 16) cm = 64;
    a.claim_max = (int)cm;
    if (cm < 64) { a.defer_k = 1; a.wait_full = 0; a.tail_merge = 0; }
    else { a.defer_k = c->defer_k > 64 - c->refill_min ? 64 - c->refill_min : c->defer_k; a.wait_full = c->park ? 1 : 0; a.tail_merge = c->tail_merge ? 1 : 0; }
    const long long per_block = (long long)(threads / 64) * cm;
    long long nb = ((long long)n_x + per_block - 1) / per_block;
    blocks = (int)(nb < full ? nb : full);
    // few particles per wave already: nothing left to gain from another slice
    a.budget_trips = cm <= 4 ? 0 : budget;
  }
  // the compaction half of new_pcut, queued behind the kernel: src[] for mcs_new_pcut / mcs_saved_export and an
  // independent count of the l_save flags next to the kernel's own n_saved counter, read back together
  HIPCHK(mcs_launch_compact(c->d_lsave, n, c->d_bcounts, c->d_boffs, c->d_counters + 2, c->d_src, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_back, c->d_counters + 1, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const unsigned long long ns[2] = {c->h_back[0], c->h_back[1]};
  if (ws && c->h_back[2] != 0) {
    static char msg[160];
    std::snprintf(msg, sizeof msg, "mcs_run_pcut: a bounded wait of the wave-specialised kernel ran out (the launch is incomplete; code 0x%llx)", (unsigned long long)c->h_back[2]);
    return fail(msg);
  }
  if (budget == 0) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    ms_total = ms;
  }
  c->last_ms = ms_total;
  c->n_saved_last = (long long)ns[0];
  c->n_run_last = n; c->idx_first = i_prt_offset; c->idx_stride = i_prt_stride; c->idx_gidx = (const long long*)dev_gidx;
  // every entry point that hands out l_save or the saved arrays goes through here (a spilling build of the kernel
  // once miscompiled the l_save byte store, see csrc/Makefile)
  if (ns[0] != ns[1]) return fail("mcs_run_pcut: the kernel's n_saved counter and the count of l_save flags differ");
  if (n_saved) *n_saved = (int64_t)ns[0];
  return 0;
}

// ---- A species' pcuts queued back to back (the device side of it): transport, pcut_finalize and new_pcut of every
// pcut first .. last with NOTHING read back in between -- n_saved, i_mult = max(n_target / n_saved, 1) (src/cuts.jl:42) and the
// size of the next population are decided on the device (mcs_k_pcut_decide) and the next launch reads its population size there.
// One shard with global indices 0, 1, 2, ... (a single rank); not for sliced launches or an explicit launch geometry.
// n_target[k]: the target population after pcut first + k.  Outputs (host, length last - first + 1): what mcs_run_pcut / mcs_new_pcut
// would have returned per pcut, and each transport launch's kernel time.  Pcuts after the one that saved nobody run empty.
int mcs_run_pcuts_fused(mcs_ctx* c, int i_pcut_first, int i_pcut_last, const int64_t* n_target, int64_t* n_use_out, int64_t* n_saved_out,
                        int64_t* i_mult_out, double* kernel_ms_out) {
  HIPCHK(hipSetDevice(c->device));
  if (!c->have_grid || !c->have_cuts) return fail("mcs_run_pcuts_fused: grid/cuts not set");
  if (i_pcut_first < 1 || i_pcut_last > c->tb.n_pcuts || i_pcut_last < i_pcut_first) return fail("mcs_run_pcuts_fused: pcut range");
  if (!n_target || !n_use_out || !n_saved_out || !i_mult_out) return fail("mcs_run_pcuts_fused: null argument");
  if (c->tail_budget > 0 || c->claim_max_first < 64 || c->blocks > 0) return fail("mcs_run_pcuts_fused: not with sliced launches or an explicit launch geometry");
  const int npc = i_pcut_last - i_pcut_first + 1;
  long long cap_n = c->n;
  for (int k = 0; k < npc; ++k) { if (n_target[k] < 1) return fail("mcs_run_pcuts_fused: n_target < 1"); if (n_target[k] > cap_n) cap_n = n_target[k]; }
  // every population of the species fits: n_new = n_saved * (n_target / n_saved) <= max(n_target, n_saved)
  if (ensure_capacity(c, cap_n)) return 1;
  if (pop_alloc(c, c->spare, cap_n + cap_n / 8 + 1024)) return 1;
  if (c->sav.cap < cap_n && pop_alloc(c, c->sav, cap_n + cap_n / 8 + 1024)) return 1;
  if (npc > c->fused_cap) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->h_fargs) (void)hipHostFree(c->h_fargs);
    if (c->h_pd) (void)hipHostFree(c->h_pd);
    if (c->d_fargs) (void)hipFree(c->d_fargs);
    if (c->d_pd) (void)hipFree(c->d_pd);
    HIPCHK(hipHostMalloc((void**)&c->h_fargs, sizeof(KArgs) * (size_t)npc));
    HIPCHK(hipHostMalloc((void**)&c->h_pd, sizeof(PcutDev) * (size_t)(npc + 1)));
    HIPCHK(hipMalloc((void**)&c->d_fargs, sizeof(KArgs) * (size_t)npc));
    HIPCHK(hipMalloc((void**)&c->d_pd, sizeof(PcutDev) * (size_t)(npc + 1)));
    while ((int)c->f_ev.size() < 2 * npc) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); c->f_ev.push_back(e); }
    c->fused_cap = npc;
  }
  // launch constants of every pcut: the buffers rotate (cur -> saved -> spare -> cur) independently of the sizes
  PopBuf cur = c->cur, spare = c->spare;
  int kind = 0, blocks = 0, threads = c->threads;
  bool ws = false;
  for (int k = 0; k < npc; ++k) {
    KArgs& a = c->h_fargs[k];
    fill_kargs(c, a, i_pcut_first + k, 0, 0, 1, nullptr, 0);
    a.in = cur.d; a.sv = c->sav.d;
    a.n_dev = &c->d_pd[k].n_use;
    if (k == 0) {
      kind = species_kernel_kind(c, a, cap_n, &ws);
      if (ws) { threads = mcs_transport_ws_threads(); blocks = c->n_cu; }
      else { threads = 256; blocks = c->n_cu * (c->P.state_fp32 ? ((c->f32_loop || c->f32_exact) ? 3 : c->f32_blocks_per_cu) : 2); }
      const long long want = (cap_n + threads - 1) / threads;
      if (want < blocks) blocks = (int)(want > 0 ? want : 1);
    }
    if (ws) { a.ws_pop_max = c->ws_pop_max > 0 ? c->ws_pop_max : threads + 160; a.ws_serve_min = c->ws_serve_min; }
    PopBuf t = cur; cur = spare; spare = t;
  }
  std::memset(c->h_pd, 0, sizeof(PcutDev) * (size_t)(npc + 1));
  c->h_pd[0].n_use = c->n;
  HIPCHK(hipMemcpyAsync(c->d_fargs, c->h_fargs, sizeof(KArgs) * (size_t)npc, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_pd, c->h_pd, sizeof(PcutDev) * (size_t)(npc + 1), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(c->d_counters, 0, 8 * sizeof(unsigned long long), c->stream));
  const int split_blocks = (int)std::min<long long>((cap_n + 255) / 256, (long long)c->n_cu * 16);
  cur = c->cur; spare = c->spare;
  for (int k = 0; k < npc; ++k) {
    HIPCHK(hipMemsetAsync(c->d_lsave, 0, (size_t)cap_n, c->stream));
    HIPCHK(hipEventRecord(c->f_ev[2 * k], c->stream));
    if (c->P.state_fp32) HIPCHK(mcs_launch_transport_f32(c->d_fargs + k, kind, blocks, 256, c->stream));
    else HIPCHK(mcs_launch_transport(c->d_fargs + k, kind, blocks, threads, c->stream));
    HIPCHK(hipEventRecord(c->f_ev[2 * k + 1], c->stream));
    HIPCHK(mcs_launch_finalize_split_dev(c->d_lsave, cap_n, c->d_bcounts, c->d_boffs, c->d_counters + 2, c->d_src, c->d_pd + k, c->d_pd + k + 1,
                                         c->d_counters, (long long)n_target[k], c->d_counters + 4, c->sav.d, spare.d, split_blocks, c->stream));
    PopBuf t = cur; cur = spare; spare = t;
  }
  c->rep_dirty = true;
  HIPCHK(hipMemcpyAsync(c->h_pd, c->d_pd, sizeof(PcutDev) * (size_t)(npc + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_back, c->d_counters + 3, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->h_back[1] != 0) return fail("mcs_run_pcuts_fused: the kernel's n_saved counter and the count of l_save flags differ");
  if (ws && c->h_back[0] != 0) return fail("mcs_run_pcuts_fused: a bounded wait of the wave-specialised kernel ran out (a launch is incomplete)");
  double ms_sum = 0.0;
  for (int k = 0; k < npc; ++k) {
    n_use_out[k] = c->h_pd[k].n_use; n_saved_out[k] = c->h_pd[k].n_saved; i_mult_out[k] = c->h_pd[k].i_mult;
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->f_ev[2 * k], c->f_ev[2 * k + 1]));
    if (kernel_ms_out) kernel_ms_out[k] = ms;
    ms_sum += ms;
  }
  c->cur = cur; c->spare = spare;
  c->n = c->h_pd[npc].n_use;
  c->n_run_last = -1; c->n_saved_last = 0;
  c->last_ms = ms_sum; c->tail_rounds_last = npc;
  c->kernel_last = c->P.state_fp32 ? (c->f32_exact ? 9 : (c->f32_loop ? 4 : (kind == 2 ? 5 : 3))) : kind;
  return 0;
}

// ---- A species' pcuts with the long histories of pcut p finishing BESIDE pcut p + 1 (DESIGN.md "Pipelined pcuts").
// A launch waits for its longest histories -- 10^4 passes of single particles while the chip idles (40 % of an iteration at 10^6
// particles).  What stands in the way of starting the next pcut is the ORDER of its population: child o of the split is a copy of
// saved particle o / i_mult in index order (src/cuts.jl:66-92), and the index keys the child's random stream -- one unresolved
// particle leaves every index behind it open.  Here the order is made independent of the schedule: a particle is LONG in a pcut when its
// history there took at least `long_draws` random draws (a property of its stream alone), and the next population is the children of
// the saved particles that are not long, in index order, followed by the children of the saved long ones, in index order.  The oracle
// orders the same way (orc_set_long_draws), so parity stays bit for bit; long_draws is a parameter of the algorithm like the seeds.
// Per pcut: the main launch (stream) and the late launch (side stream: the children of the previous pcut's saved long particles)
// export the particles that are still running once they are long and end; both join; the main group is split and the next main
// launch starts, while on the side stream the exported particles run to their end, the late group is split and the next late launch
// runs.  i_mult = max(n_target / n_saved, 1) needs the number of long particles that will be saved: it is taken as soon as both
// bounds give the same quotient, else the pcut waits for them (counted in strag_out).  One rank, global indices 0, 1, 2, ...; fp64 state.
// Outputs as mcs_run_pcuts_fused; strag_out (or NULL): [2k] particles pcut k exported, [2k + 1] 1 if its i_mult had to wait.
int mcs_run_pcuts_pipelined(mcs_ctx* c, int i_pcut_first, int i_pcut_last, const int64_t* n_target, int64_t long_draws, int64_t long_imult_max,
                            int64_t* n_use_out, int64_t* n_saved_out, int64_t* i_mult_out, double* kernel_ms_out, int64_t* strag_out) {
  HIPCHK(hipSetDevice(c->device));
  if (!c->have_grid || !c->have_cuts) return fail("mcs_run_pcuts_pipelined: grid/cuts not set");
  if (i_pcut_first < 1 || i_pcut_last > c->tb.n_pcuts || i_pcut_last < i_pcut_first) return fail("mcs_run_pcuts_pipelined: pcut range");
  if (!n_target || !n_use_out || !n_saved_out || !i_mult_out) return fail("mcs_run_pcuts_pipelined: null argument");
  if (c->P.state_fp32) return fail("mcs_run_pcuts_pipelined: not for the fp32-state variant");
  if (c->tail_budget > 0 || c->claim_max_first < 64 || c->blocks > 0) return fail("mcs_run_pcuts_pipelined: not with sliced launches or an explicit launch geometry");
  if (long_draws < 64 || long_draws > 2000000000LL) return fail("mcs_run_pcuts_pipelined: long_draws out of range (64 .. 2e9)");
  const int npc = i_pcut_last - i_pcut_first + 1;
  long long cap_n = c->n;
  for (int k = 0; k < npc; ++k) { if (n_target[k] < 1) return fail("mcs_run_pcuts_pipelined: n_target < 1"); if (n_target[k] > cap_n) cap_n = n_target[k]; }
  const long long cap = cap_n + cap_n / 8 + 1024;
  if (ensure_capacity(c, cap_n)) return 1;
  if (pop_alloc(c, c->spare, cap)) return 1;
  if (pop_alloc(c, c->pp_sav2, cap)) return 1;
  if (c->pp_cap < cap) {
    HIPCHK(hipStreamSynchronize(c->stream));
    { void* pq[] = {c->pp_lsave2, c->pp_bcounts, c->pp_boffs, c->pp_src}; for (void* q : pq) if (q) (void)hipFree(q); }
    const long long nb = (cap + 1023) / 1024;
    HIPCHK(hipMalloc((void**)&c->pp_lsave2, (size_t)cap));
    HIPCHK(hipMalloc((void**)&c->pp_bcounts, (size_t)nb * sizeof(unsigned int)));
    HIPCHK(hipMalloc((void**)&c->pp_boffs, (size_t)nb * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void**)&c->pp_src, (size_t)cap * sizeof(long long)));
    c->pp_cap = cap;
  }
  if (!c->pp_s2) {
    HIPCHK(hipStreamCreateWithFlags(&c->pp_s2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->pp_reset, hipEventDisableTiming));
    HIPCHK(hipMalloc((void**)&c->pp_dpc, 16 * sizeof(unsigned long long)));
    HIPCHK(hipHostMalloc((void**)&c->pp_hpc, 16 * sizeof(unsigned long long)));
    HIPCHK(hipHostMalloc((void**)&c->pp_hargs, 3 * sizeof(KArgs)));
    HIPCHK(hipMalloc((void**)&c->pp_dargs, 3 * sizeof(KArgs)));
    HIPCHK(hipMalloc((void**)&c->pp_dpdl, 2 * sizeof(PcutDev)));
    HIPCHK(hipHostMalloc((void**)&c->pp_hpdl, 2 * sizeof(PcutDev)));
    if (c->pp_side_cus > 0 && c->pp_side_cus < c->n_cu) {
      const int words = (c->n_cu + 31) / 32;
      std::vector<uint32_t> m_side((size_t)words, 0u), m_main((size_t)words, 0u);
      for (int i = 0; i < c->n_cu; ++i) (i < c->pp_side_cus ? m_side : m_main)[(size_t)(i >> 5)] |= 1u << (i & 31);
      if (hipExtStreamCreateWithCUMask(&c->pp_s1m, (uint32_t)words, m_main.data()) != hipSuccess ||
          hipExtStreamCreateWithCUMask(&c->pp_s2m, (uint32_t)words, m_side.data()) != hipSuccess) {
        (void)hipGetLastError();
        if (c->pp_s1m) { (void)hipStreamDestroy(c->pp_s1m); c->pp_s1m = nullptr; }
        if (c->pp_s2m) { (void)hipStreamDestroy(c->pp_s2m); c->pp_s2m = nullptr; }
      }
    }
  }
  const int threads = 256;
  const long long full = (long long)c->n_cu * 2;            // resident workgroups of the fp64 kernels
  {   // a wave exports at most its 64 lanes, once: room for the main and the late launch of one pcut
    const long long need_cap = 2 * full * threads;
    if (need_cap > c->strag_cap) {
      HIPCHK(hipStreamSynchronize(c->stream));
      for (int b = 0; b < 2; ++b) { if (c->d_strag[b]) (void)hipFree(c->d_strag[b]); c->d_strag[b] = nullptr; }
      c->strag_cap = need_cap;
      for (int b = 0; b < 2; ++b) HIPCHK(hipMalloc((void**)&c->d_strag[b], (size_t)c->strag_cap * MCS_STRAG_WORDS * sizeof(double)));
    }
  }
  // counters: [0] work, [1] n_saved of the main launch | [2] main scan total | [3 + q] particles exported by pcut parity q |
  // [5] work, [6] n_saved of the resumed long histories | [7] work, [8] n_saved of the late launch | [9] late scan total
  unsigned long long* const pc = c->pp_dpc;
  const bool masked = c->pp_s1m && c->pp_s2m;
  hipStream_t s1 = c->stream;                              // the main stream of the CURRENT pcut: c->stream, or the masked one when the pcut has side work
  hipStream_t const s2 = masked ? c->pp_s2m : c->pp_s2;    // the side stream
  hipStream_t const s_alone = masked ? c->stream : c->pp_s2;   // long histories the pcut waits for: the whole chip
  HIPCHK(hipStreamSynchronize(c->stream));
  int kind;
  {
    KArgs t; fill_kargs(c, t, i_pcut_first, 0, 0, 1, nullptr, 0);
    int kk = species_kernel_kind(c, t, 0, nullptr);
    if (kk == 7) kk = 1;
    if (kk == 8) kk = 6;
    kind = kk == 1 ? 11 : (kk == 2 ? 12 : (kk == 6 ? 13 : 10));      // the sliced form of the species' kernel: PLAIN, LOSSY, PLAIN_ETF, general
  }
  PopBuf cur = c->cur, nxt = c->spare;
  PopBuf savb[2] = {c->sav, c->pp_sav2};
  uint8_t* lsv[2] = {c->d_lsave, c->pp_lsave2};
  long long nA = c->n, nL = 0;
  // long histories are told apart in pcut k only when the pcut before it split by at most long_imult_max (<= 0: always): where few
  // particles are saved and each is split a hundredfold, i_mult hangs on the last long history and the pcut would wait for them anyway
  // (a rule on numbers the oracle has too: driver.py applies it to set_long_draws)
  long long Bk = long_draws;
  bool side_pending = false;
  int side_blocks = 0;               // workgroups the side stream's launches of this pcut need resident beside the main launch
  long long n1_prev = 0, sofar_prev = 0;
  double ms_sum = 0.0;
  int n_done = 0;
  c->pp_waits_last = 0;
  c->tail_rounds_last = 0;
  const bool dbg_pipe = std::getenv("MCS_PIPE_DEBUG") != nullptr;
  for (int k = 0; k < npc; ++k) { n_use_out[k] = 0; n_saved_out[k] = 0; i_mult_out[k] = 1; if (kernel_ms_out) kernel_ms_out[k] = 0.0; if (strag_out) { strag_out[2 * k] = 0; strag_out[2 * k + 1] = 0; } }
  HIPCHK(hipMemsetAsync(pc, 0, 16 * sizeof(unsigned long long), s1));
  if (nA > 0) HIPCHK(hipMemsetAsync(lsv[0], 0, (size_t)nA, s1));
  auto blocks_for = [&](long long n) { const long long want = (n + threads - 1) / threads; return (int)(want < full ? (want > 0 ? want : 1) : full); };
  // the launch of the exported particles of pcut `i_pcut` (parity q), to their end, on `st`
  auto launch_resume = [&](int i_pcut, int q, long long n_pop, long long n_x, hipStream_t st, bool alone) -> int {
    KArgs& a = c->pp_hargs[1];
    fill_kargs(c, a, i_pcut, n_pop, 0, 1, nullptr, 0);
    a.in = cur.d; a.sv = savb[q].d; a.l_save = lsv[q];
    a.work_counter = pc + 5; a.n_saved = pc + 6; a.strag_count = pc + 3 + (q ^ 1); a.strag_out = c->d_strag[q ^ 1];
    a.strag_in = c->d_strag[q]; a.n_resume = n_x; a.fresh_lo = n_pop; a.long_draws = (unsigned int)Bk; a.budget_trips = 0;
    // few particles per wave (profiles/r03_tau_vs_lanes.txt), on at most one wave per SIMD of the chip
    // (alone on the chip -- the pcut waits for them -- one wave per SIMD; beside a main launch every wave they hold is taken from it:
    // 16 particles per wave cost 12 % on the longest history and 1 / 16 of the slots)
    const long long waves1 = alone ? (long long)c->n_cu * (threads / 64) : (c->pp_side_waves > 0 ? c->pp_side_waves : (n_x + 15) / 16);
    long long cm = (n_x + waves1 - 1) / waves1;
    if (cm > 16) cm = 64;
    a.claim_max = (int)cm;
    if (cm < 64) { a.defer_k = 1; a.wait_full = 0; a.tail_merge = 0; }
    const long long per_block = (long long)(threads / 64) * cm;
    const long long nb = (n_x + per_block - 1) / per_block;
    HIPCHK(hipMemcpyAsync(c->pp_dargs + 1, &a, sizeof(KArgs), hipMemcpyHostToDevice, st));
    HIPCHK(mcs_launch_transport(c->pp_dargs + 1, kind, (int)(nb < full ? nb : full), threads, st));
    if (!alone) side_blocks += (int)(nb < full ? nb : full);
    ++c->tail_rounds_last;
    return 0;
  };
  int rc = 0;                        // (a failed cross-check ends the loop: the streams are drained below before the error is returned)
  for (int k = 0; k < npc; ++k) {
    const int i_pcut = i_pcut_first + k, q = k & 1;
    // ---- the main launch: particles 0 .. nA-1 of the population
    if (nA > 0) {
      KArgs& a = c->pp_hargs[0];
      fill_kargs(c, a, i_pcut, nA, 0, 1, nullptr, Bk > 0 ? 1 : 0);
      a.in = cur.d; a.sv = savb[q].d; a.l_save = lsv[q];
      a.work_counter = pc; a.n_saved = pc + 1; a.strag_count = pc + 3 + q; a.strag_out = c->d_strag[q];
      a.long_draws = (unsigned int)Bk;
      HIPCHK(hipMemcpyAsync(c->pp_dargs, &a, sizeof(KArgs), hipMemcpyHostToDevice, s1));
      HIPCHK(hipEventRecord(c->ev0, s1));
      // (the main launch is persistent and fills every slot of the chip: it leaves room for the side stream's workgroups, which would
      // otherwise wait for its workgroups to leave -- and run after it instead of beside it)
      int blocks_a = blocks_for(nA);
      if (masked) { if (s1 == c->pp_s1m && blocks_a > 2 * (c->n_cu - c->pp_side_cus)) blocks_a = 2 * (c->n_cu - c->pp_side_cus); }
      else if (side_blocks > 0 && blocks_a > full - side_blocks) blocks_a = (int)(full - side_blocks);
      HIPCHK(mcs_launch_transport(c->pp_dargs, kind, blocks_a, threads, s1));
      HIPCHK(hipEventRecord(c->ev1, s1));
      ++c->tail_rounds_last;
    }
    c->rep_dirty = true;
    // ---- join: the side stream (the previous pcut's long histories, its late split, this pcut's late launch), then the main launch,
    // the compaction of the particles that were saved and are not long, one read-back
    const auto tj0 = std::chrono::steady_clock::now();
    if (side_pending) HIPCHK(hipStreamSynchronize(s2));
    const auto tj1 = std::chrono::steady_clock::now();
    if (side_pending) HIPCHK(hipMemcpyAsync(c->pp_hpdl + q, c->pp_dpdl + q, sizeof(PcutDev), hipMemcpyDeviceToHost, s1));
    // (the late group's size is on the device until here: the compaction below covers every index it can have)
    const long long n_hi = nA + nL;       // nL: the host's upper bound while side_pending
    HIPCHK(mcs_launch_compact_match(lsv[q], n_hi, c->d_bcounts, c->d_boffs, pc + 2, c->d_src, 1u, s1));
    HIPCHK(hipMemcpyAsync(c->pp_hpc, pc, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s1));
    HIPCHK(hipStreamSynchronize(s1));
    const auto tj2 = std::chrono::steady_clock::now();
    float ms_main = 0.f;
    if (nA > 0) { HIPCHK(hipEventElapsedTime(&ms_main, c->ev0, c->ev1)); if (kernel_ms_out) kernel_ms_out[k] = ms_main; ms_sum += ms_main; }
    const unsigned long long* h = c->pp_hpc;
    double dbg_wait_ms = 0.0;
    if (side_pending) {
      // the previous pcut is complete now: its saved long particles, the size of this pcut's late group
      const long long n5_prev = c->pp_hpdl[q].n_saved;
      nL = c->pp_hpdl[q].n_new;
      n_saved_out[k - 1] = n1_prev + n5_prev;
      if (sofar_prev + (long long)h[6] != n_saved_out[k - 1]) {
        static char msg[256];
        std::snprintf(msg, sizeof msg, "mcs_run_pcuts_pipelined: pcut %d: the kernels' n_saved counters (%lld + %llu resumed) and the count of status bytes (%lld + %lld long) differ",
                      i_pcut - 1, sofar_prev, (unsigned long long)h[6], n1_prev, n5_prev);
        rc = fail(msg); side_pending = false; break;
      }
    }
    side_pending = false;
    n_use_out[k] = nA + nL;
    n_done = k + 1;
    const long long n1 = (long long)h[2], n_T = (long long)h[3 + q];
    long long sofar = (long long)h[1] + (long long)h[8];        // saved by the main and the late launch: not long, or long and already ended
    if (strag_out) strag_out[2 * k] = n_T;
    if (n_T > c->strag_cap) { rc = fail("mcs_run_pcuts_pipelined: export buffer overrun"); break; }
    if (sofar < n1) {
      static char msg[256];
      std::snprintf(msg, sizeof msg, "mcs_run_pcuts_pipelined: pcut %d: the kernels' n_saved counters (%llu main + %llu late) are below the count of status bytes (%lld)",
                    i_pcut, (unsigned long long)h[1], (unsigned long long)h[8], n1);
      rc = fail(msg); break;
    }
    const long long target = (long long)n_target[k];
    const bool last = k == npc - 1;
    long long n_open = n_T;                 // exported particles that have not been resumed yet
    if (sofar + n_T == 0) { n_saved_out[k] = 0; break; }                       // nobody left: the species ends here
    const long long im_hi = target / (sofar + n_T) > 1 ? target / (sofar + n_T) : 1;
    const long long im_lo = sofar > 0 ? (target / sofar > 1 ? target / sofar : 1) : -1;
    // (more long histories than the side stream's CUs hold at 16 per wave, twice over: beside the main launch they would outlast it)
    const long long side_cap = masked ? (long long)c->pp_side_cus * 8 * 16 * 2 : (long long)c->pp_side_max * 4 * 16 * 2;
    if ((last || im_lo != im_hi || n_T > side_cap) && n_T > 0) {
      // i_mult depends on how many of the long histories end saved (or this is the last pcut, or they are too many): they finish first
      HIPCHK(hipMemsetAsync(pc + 5, 0, 2 * sizeof(unsigned long long), s_alone));
      if (launch_resume(i_pcut, q, nA + nL, n_T, s_alone, true)) return 1;
      HIPCHK(hipMemcpyAsync(c->pp_hpc + 6, pc + 6, sizeof(unsigned long long), hipMemcpyDeviceToHost, s_alone));
      const auto tw0 = std::chrono::steady_clock::now();
      HIPCHK(hipStreamSynchronize(s_alone));
      dbg_wait_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
      sofar += (long long)c->pp_hpc[6];
      n_open = 0;
      if (strag_out) strag_out[2 * k + 1] = 1;
      ++c->pp_waits_last;
    }
    if (dbg_pipe)
      std::fprintf(stderr, "[pipe] pcut %2d n_use %8lld (late %7lld) main %6.2f ms | side-sync %6.2f ms main-sync %6.2f ms | saved so far %8lld exported %6lld %s %6.2f ms\n",
                   i_pcut, nA + nL, nL, ms_main, std::chrono::duration<double, std::milli>(tj1 - tj0).count(),
                   std::chrono::duration<double, std::milli>(tj2 - tj1).count(), sofar, n_T, n_open == 0 && n_T > 0 ? "WAITED" : "      ", dbg_wait_ms);
    if (sofar + n_open == 0) { n_saved_out[k] = 0; break; }                  // (the long histories all ended: the species ends here)
    const long long i_mult = target / (sofar + n_open) > 1 ? target / (sofar + n_open) : 1;     // (== for both bounds when n_open > 0)
    i_mult_out[k] = i_mult;
    n_saved_out[k] = sofar;                 // (complete unless long histories are still open: then the next join adds those that end saved)
    if (last || sofar + n_open == 0) break;
    const long long B_next = (long_imult_max <= 0 || i_mult <= long_imult_max) ? long_draws : 0;
    // ---- the next pcut: main group = children of the saved particles that are not long; late group = children of the saved long ones
    const long long nA_next = n1 * i_mult;
    const long long n5_max = sofar - n1 + n_open;               // long particles that are saved, or still running
    const long long nL_max = n5_max * i_mult;
    // ((sofar + n_open) * i_mult <= max(n_target, sofar + n_open) <= cap_n: every buffer holds it)
    if (nA_next + nL_max > cap_n) { rc = fail("mcs_run_pcuts_pipelined: the next population exceeds the buffers"); break; }
    // (the long histories go first: their few waves must be resident before the next main launch fills every slot of the chip --
    // queued behind it they would start when its workgroups leave, i.e. run after it instead of beside it.  Their counters are the side
    // stream's own words; everything else is cleared on the main stream, and the late split / late launch wait for that.)
    side_blocks = 0;
    s1 = (masked && n5_max > 0) ? c->pp_s1m : c->stream;      // (both are idle: the join synchronised the host with every stream)
    if (n5_max > 0) {
      HIPCHK(hipMemsetAsync(pc + 5, 0, 2 * sizeof(unsigned long long), s2));
      if (n_open > 0 && launch_resume(i_pcut, q, nA + nL, n_open, s2, false)) return 1;
    }
    HIPCHK(hipMemsetAsync(pc, 0, 5 * sizeof(unsigned long long), s1));
    HIPCHK(hipMemsetAsync(pc + 7, 0, 3 * sizeof(unsigned long long), s1));
    if (nA_next + nL_max > 0) HIPCHK(hipMemsetAsync(lsv[q ^ 1], 0, (size_t)(nA_next + nL_max), s1));
    HIPCHK(hipEventRecord(c->pp_reset, s1));
    HIPCHK(mcs_launch_split(savb[q].d, nxt.d, c->d_src, nA_next, i_mult, s1));
    n1_prev = n1; sofar_prev = sofar;
    if (n5_max > 0) {
      HIPCHK(hipStreamWaitEvent(s2, c->pp_reset, 0));
      DevPop out = nxt.d;
      out.weight += nA_next; out.ptot_pf += nA_next; out.pb_pf += nA_next; out.x_PT_cm += nA_next; out.xn_per += nA_next;
      out.prp_x_cm += nA_next; out.acctime_sec += nA_next; out.phi_rad += nA_next; out.meta += nA_next;
      const int split_blocks = (int)std::min<long long>((nL_max + 255) / 256, (long long)c->n_cu * 4);
      HIPCHK(mcs_launch_late_split(lsv[q], nA + nL, c->pp_bcounts, c->pp_boffs, pc + 9, c->pp_src, c->pp_dpdl + (q ^ 1), i_mult, nA_next, savb[q].d,
                                   out, split_blocks < 1 ? 1 : split_blocks, s2));
      // the late launch of the next pcut: particles nA_next .. of its population, their number read on the device
      KArgs& a = c->pp_hargs[2];
      fill_kargs(c, a, i_pcut + 1, nA_next, 0, 1, nullptr, B_next > 0 ? 1 : 0);
      a.in = nxt.d; a.sv = savb[q ^ 1].d; a.l_save = lsv[q ^ 1];
      a.n_dev = &c->pp_dpdl[q ^ 1].n_use; a.fresh_lo = nA_next;
      a.work_counter = pc + 7; a.n_saved = pc + 8; a.strag_count = pc + 3 + (q ^ 1); a.strag_out = c->d_strag[q ^ 1];
      a.long_draws = (unsigned int)B_next;
      HIPCHK(hipMemcpyAsync(c->pp_dargs + 2, &a, sizeof(KArgs), hipMemcpyHostToDevice, s2));
      HIPCHK(mcs_launch_transport(c->pp_dargs + 2, kind, blocks_for(nL_max), threads, s2));
      side_blocks += blocks_for(nL_max);
      if (side_blocks > c->pp_side_max) side_blocks = c->pp_side_max;
      ++c->tail_rounds_last;
      side_pending = true;
      nL = nL_max;
    } else {
      nL = 0;
    }
    nA = nA_next;
    Bk = B_next;
    PopBuf t = cur; cur = nxt; nxt = t;
  }
  HIPCHK(hipStreamSynchronize(s2));      // (idle already on every exit but a failed cross-check)
  HIPCHK(hipStreamSynchronize(s1));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->cur = cur; c->spare = nxt;
  c->n = n_done > 0 ? n_use_out[n_done - 1] : c->n;
  c->n_run_last = -1; c->n_saved_last = 0;
  c->last_ms = ms_sum;
  c->kernel_last = kind;
  return rc;
}

int mcs_new_pcut(mcs_ctx* c, int64_t i_mult, int64_t* n_new_out) {
  HIPCHK(hipSetDevice(c->device));
  if (i_mult < 1) return fail("mcs_new_pcut: i_mult < 1");
  if (c->n != c->n_run_last) return fail("mcs_new_pcut: no mcs_run_pcut since the population changed");
  const long long n_saved = c->n_saved_last;
  const long long n_new = n_saved * i_mult;
  // the split (src[] was computed behind the transport kernel) writes into the spare buffer, then the buffers
  // rotate; nothing is read back: the new size is known on the host
  if (pop_alloc(c, c->spare, n_new + n_new / 8 + 1024)) return 1;
  HIPCHK(mcs_launch_split(c->sav.d, c->spare.d, c->d_src, n_new, i_mult, c->stream));
  PopBuf t = c->cur; c->cur = c->spare; c->spare = t;
  c->n = n_new; c->n_run_last = -1;
  if (ensure_capacity(c, n_new)) return 1;
  if (n_new_out) *n_new_out = n_new;
  return 0;
}

int mcs_saved_export(mcs_ctx* c, int64_t cap, int64_t* dev_gidx, double* dev_f64, uint32_t* dev_meta) {
  HIPCHK(hipSetDevice(c->device));
  if (c->n != c->n_run_last) return fail("mcs_saved_export: no mcs_run_pcut since the population changed");
  if (cap < c->n_saved_last) return fail("mcs_saved_export: cap < n_saved");
  if (c->n_saved_last > 0 && (!dev_gidx || !dev_f64 || !dev_meta)) return fail("mcs_saved_export: null buffer");
  HIPCHK(mcs_launch_saved_export(c->sav.d, c->d_src, c->n_saved_last, cap, c->idx_first, c->idx_stride, c->idx_gidx,
                                 (long long*)dev_gidx, dev_f64, dev_meta, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));      // the caller's library (RCCL on another stream) may read the buffers now
  return 0;
}

int mcs_saved_gidx(mcs_ctx* c, int64_t cap, int64_t* dev_gidx) {
  HIPCHK(hipSetDevice(c->device));
  if (c->n != c->n_run_last) return fail("mcs_saved_gidx: no mcs_run_pcut since the population changed");
  if (cap < c->n_saved_last) return fail("mcs_saved_gidx: cap < n_saved");
  if (c->n_saved_last > 0 && !dev_gidx) return fail("mcs_saved_gidx: null buffer");
  HIPCHK(mcs_launch_saved_export(c->sav.d, c->d_src, c->n_saved_last, cap, c->idx_first, c->idx_stride, c->idx_gidx,
                                 (long long*)dev_gidx, nullptr, nullptr, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int mcs_split_import(mcs_ctx* c, int64_t n_parents, int64_t cap, const double* dev_f64, const uint32_t* dev_meta, int64_t i_mult,
                     int64_t first, int64_t stride, int64_t n_local) {
  HIPCHK(hipSetDevice(c->device));
  if (i_mult < 1 || stride < 1 || first < 0 || n_local < 0 || n_parents < 0 || cap < n_parents)
    return fail("mcs_split_import: bad arguments");
  if (n_local > 0 && (first + (n_local - 1) * stride) / i_mult >= n_parents)
    return fail("mcs_split_import: the local slice reaches past n_parents * i_mult");
  if (n_local > 0 && (!dev_f64 || !dev_meta)) return fail("mcs_split_import: null buffer");
  if (pop_alloc(c, c->spare, n_local + n_local / 8 + 1024)) return 1;
  HIPCHK(mcs_launch_split_import(c->spare.d, dev_f64, dev_meta, cap, i_mult, first, stride, n_local, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));      // the caller may free or reuse its buffers
  PopBuf t = c->cur; c->cur = c->spare; c->spare = t;
  c->n = n_local; c->n_run_last = -1;
  if (ensure_capacity(c, n_local)) return 1;
  return 0;
}

int mcs_run_pcut_host(mcs_ctx* c, int i_pcut, int64_t n_pts_use, int64_t i_prt_offset, const mcs_soa* in, mcs_soa* saved_out,
                      uint8_t* l_save, int64_t* n_saved) {
  if (mcs_pop_upload(c, n_pts_use, in)) return 1;
  if (mcs_run_pcut(c, i_pcut, i_prt_offset, n_saved)) return 1;
  return mcs_saved_download(c, n_pts_use, saved_out, l_save);
}

int mcs_read_tallies(mcs_ctx* c, double* host_f64, int64_t* host_i64) {
  HIPCHK(hipSetDevice(c->device));
  if (fold_replicas(c)) return 1;
  if (host_f64) HIPCHK(hipMemcpyAsync(host_f64, c->d_T, (size_t)c->L.total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (host_i64) HIPCHK(hipMemcpyAsync(host_i64, c->d_I, (size_t)mcs_i64_total(&c->P) * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int mcs_read_tallies_part(mcs_ctx* c, int64_t first, int64_t count, double* host_f64, int64_t* host_i64) {
  HIPCHK(hipSetDevice(c->device));
  if (first < 0 || count < 0 || first + count > c->L.total || (count > 0 && !host_f64)) return fail("mcs_read_tallies_part: range outside the tally buffer");
  if (fold_replicas(c)) return 1;
  if (count > 0) HIPCHK(hipMemcpyAsync(host_f64, c->d_T + first, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (host_i64) HIPCHK(hipMemcpyAsync(host_i64, c->d_I, (size_t)mcs_i64_total(&c->P) * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int mcs_write_tallies_part(mcs_ctx* c, int64_t first, int64_t count, const double* host_f64) {
  HIPCHK(hipSetDevice(c->device));
  if (first < 0 || count < 0 || first + count > c->L.total || (count > 0 && !host_f64)) return fail("mcs_write_tallies_part: range outside the tally buffer");
  if (fold_replicas(c)) return 1;
  if (count > 0) HIPCHK(hipMemcpyAsync(c->d_T + first, host_f64, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int mcs_write_tallies(mcs_ctx* c, const double* host_f64, const int64_t* host_i64) {
  HIPCHK(hipSetDevice(c->device));
  if (fold_replicas(c)) return 1;
  if (host_f64) HIPCHK(hipMemcpyAsync(c->d_T, host_f64, (size_t)c->L.total * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (host_i64) HIPCHK(hipMemcpyAsync(c->d_I, host_i64, (size_t)mcs_i64_total(&c->P) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int mcs_eval_fn(mcs_ctx* c, int fn, int64_t n, const double* a, const double* b, double* out) {
  HIPCHK(hipSetDevice(c->device));
  if (ensure_stage(c, 3 * n + 3)) return 1;
  double *da = c->d_stage, *db = c->d_stage + n, *dout = c->d_stage + 2 * n;
  HIPCHK(hipMemcpyAsync(da, a, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(db, b ? b : a, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(mcs_launch_eval(fn, n, da, db, dout, c->stream));
  HIPCHK(hipMemcpyAsync(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int mcs_final_download(mcs_ctx* c, int64_t n, int32_t* reason, int32_t* helix_count, int32_t* retro_count, double* ptot_pf,
                       double* x_PT_cm) {
  HIPCHK(hipSetDevice(c->device));
  if (!c->debug_finals) return fail("mcs_final_download: end states are recorded only after mcs_set_debug_finals(ctx, 1)");
  if (n > c->f_cap) return fail("mcs_final_download: n too large");
  if (reason) HIPCHK(hipMemcpyAsync(reason, c->f_reason, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  if (helix_count) HIPCHK(hipMemcpyAsync(helix_count, c->f_helix, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  if (retro_count) HIPCHK(hipMemcpyAsync(retro_count, c->f_retro, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  if (ptot_pf) HIPCHK(hipMemcpyAsync(ptot_pf, c->f_ptot, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  if (x_PT_cm) HIPCHK(hipMemcpyAsync(x_PT_cm, c->f_x, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

double mcs_last_kernel_ms(mcs_ctx* c) { return c->last_ms; }

// ---- consumers of the tallies (K4) ------------------------------------------------------
static int consumers_ready(mcs_ctx* c, const mcs_consumer_in* in, const char* who) {
  if (!c || !in) return fail(std::string(who) + ": null argument");
  if (!c->have_grid) return fail(std::string(who) + ": grid not set");
  if (c->P.num_psd_mom_bins + 2 > 208 || c->P.num_psd_tht_bins + 2 > 208) return fail(std::string(who) + ": too many PSD bins");
  const int NM = c->P.num_psd_mom_bins + 2, NT = c->P.num_psd_tht_bins + 2, ng = c->P.n_grid;
  if (!c->d_ctab) HIPCHK(hipMalloc((void**)&c->d_ctab, sizeof(double) * (size_t)(3 * NM + 2 * NT + 4 * ng)));
  if (!c->d_cout) HIPCHK(hipMalloc((void**)&c->d_cout, sizeof(double) * (size_t)(3 * ng * NM + 3 * ng)));
  if (!c->d_cdiag) HIPCHK(hipMalloc((void**)&c->d_cdiag, sizeof(unsigned long long) * 2));
  return 0;
}

int mcs_dndp_cr(mcs_ctx* c, const mcs_consumer_in* in, double* dNdp, int64_t* diag) {
  HIPCHK(hipSetDevice(c ? c->device : 0));
  if (consumers_ready(c, in, "mcs_dndp_cr")) return 1;
  if (fold_replicas(c)) return 1;
  if (!in->mom_log_cgs || !in->mom_edge_cgs || !in->cos_edge || !in->zone_pop || !dNdp) return fail("mcs_dndp_cr: null table");
  const int NM = c->P.num_psd_mom_bins + 2, NT = c->P.num_psd_tht_bins + 2, ng = c->P.n_grid;
  std::vector<double> h((size_t)(2 * NM + NT + ng));
  memcpy(h.data(), in->mom_log_cgs, sizeof(double) * NM);
  memcpy(h.data() + NM, in->mom_edge_cgs, sizeof(double) * NM);
  memcpy(h.data() + 2 * NM, in->cos_edge, sizeof(double) * NT);
  memcpy(h.data() + 2 * NM + NT, in->zone_pop, sizeof(double) * ng);
  HIPCHK(hipMemcpyAsync(c->d_ctab, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(c->d_cdiag, 0, sizeof(unsigned long long) * 2, c->stream));
  HIPCHK(mcs_launch_dndp_cr(&c->P, c->d_T + c->L.psd, c->tb.gsf, c->tb.ux, c->d_ctab, in->rest_energy, in->n0, in->gam0,
                            c->d_cout, c->d_cdiag, c->stream));
  HIPCHK(hipMemcpyAsync(dNdp, c->d_cout, sizeof(double) * (size_t)(3 * ng * NM), hipMemcpyDeviceToHost, c->stream));
  unsigned long long hd[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(hd, c->d_cdiag, sizeof(hd), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (diag) { diag[0] = (int64_t)hd[0]; diag[1] = (int64_t)hd[1]; }
  return 0;
}

int mcs_thermo_calcs(mcs_ctx* c, const mcs_consumer_in* in, double* P_par, double* P_perp, double* energy_density) {
  HIPCHK(hipSetDevice(c ? c->device : 0));
  if (consumers_ready(c, in, "mcs_thermo_calcs")) return 1;
  if (fold_replicas(c)) return 1;
  if (!in->cos_center || !in->pt_center || !in->zone_pop || !in->density_loc || !in->cold_pressure || !P_par || !P_perp || !energy_density)
    return fail("mcs_thermo_calcs: null table");
  const int NM = c->P.num_psd_mom_bins + 2, NT = c->P.num_psd_tht_bins + 2, ng = c->P.n_grid;
  if (!c->d_cscratch) HIPCHK(hipMalloc((void**)&c->d_cscratch, sizeof(double) * (size_t)NM * NT * ng));
  std::vector<double> h((size_t)(NT + NM + 3 * ng), 0.0);
  memcpy(h.data(), in->cos_center, sizeof(double) * (NT - 1));
  memcpy(h.data() + NT, in->pt_center, sizeof(double) * (NM - 1));
  memcpy(h.data() + NT + NM, in->zone_pop, sizeof(double) * ng);
  memcpy(h.data() + NT + NM + ng, in->density_loc, sizeof(double) * ng);
  memcpy(h.data() + NT + NM + 2 * ng, in->cold_pressure, sizeof(double) * ng);
  HIPCHK(hipMemcpyAsync(c->d_ctab, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, c->stream));
  double* out3 = c->d_cout + (size_t)3 * ng * NM;
  HIPCHK(mcs_launch_thermo(&c->P, c->d_T + c->L.psd, c->d_T + c->L.therm_pf, c->d_I + MCS_I_NUM_CROSSINGS, c->tb.gsf, c->tb.ux,
                           c->d_ctab, in->rest_energy, in->mc, in->n0, in->therm_from_hist, c->d_cscratch, out3, c->stream));
  std::vector<double> o((size_t)3 * ng);
  HIPCHK(hipMemcpyAsync(o.data(), out3, sizeof(double) * o.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(P_par, o.data(), sizeof(double) * ng);
  memcpy(P_perp, o.data() + ng, sizeof(double) * ng);
  memcpy(energy_density, o.data() + 2 * ng, sizeof(double) * ng);
  return 0;
}

int mcs_photon_synch(mcs_ctx* c, const double* dNdp_pf, const double* mom_edge_cgs, double mc, int n_photon, double emin_mev,
                     double bins_per_dec, double* energy_erg, double* emis) {
  HIPCHK(hipSetDevice(c ? c->device : 0));
  if (!c || !dNdp_pf || !mom_edge_cgs || !emis) return fail("mcs_photon_synch: null argument");
  if (!c->have_grid) return fail("mcs_photon_synch: grid not set");
  if (n_photon < 1 || n_photon > 4096 || !(emin_mev > 0) || !(bins_per_dec > 0) || !(mc > 0)) return fail("mcs_photon_synch: bad arguments");
  const int NM = c->P.num_psd_mom_bins + 2, ng = c->P.n_grid;
  if (NM > 208) return fail("mcs_photon_synch: too many momentum bins");
  const size_t n_in = (size_t)ng * NM + NM, n_out = (size_t)ng * n_photon;
  if (ensure_stage(c, (long long)(n_in + n_out) + 4)) return 1;
  double* d_in = c->d_stage; double* d_out = c->d_stage + n_in;
  HIPCHK(hipMemcpyAsync(d_in, dNdp_pf, sizeof(double) * (size_t)ng * NM, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_in + (size_t)ng * NM, mom_edge_cgs, sizeof(double) * NM, hipMemcpyHostToDevice, c->stream));
  const double log_emin = std::log10(emin_mev * MCS_MEV_ERG_);
  HIPCHK(mcs_launch_photon_synch(d_in, d_in + (size_t)ng * NM, c->tb.btot, ng, NM, n_photon, log_emin, bins_per_dec, mc, d_out, c->stream));
  HIPCHK(hipMemcpyAsync(emis, d_out, sizeof(double) * n_out, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (energy_erg) for (int j = 0; j < n_photon; ++j) energy_erg[j] = std::pow(10.0, log_emin + j * (1.0 / bins_per_dec));
  return 0;
}

// The pion-decay fold (include/mcs_pion.h) over the plasma-frame dN/dp of a nucleus species.
int mcs_photon_pion(mcs_ctx* c, const double* dNdp_pf, const double* mom_edge_cgs, double mc, double aa, const double* target_density, double scaling,
                    int i_data, int n_photon, double emin_mev, double bins_per_dec, double* energy_erg, double* emis) {
  HIPCHK(hipSetDevice(c ? c->device : 0));
  if (!c || !dNdp_pf || !mom_edge_cgs || !target_density || !emis) return fail("mcs_photon_pion: null argument");
  if (!c->have_grid) return fail("mcs_photon_pion: grid not set");
  if (n_photon < 1 || n_photon > 4096 || !(emin_mev > 0) || !(bins_per_dec > 0) || !(mc > 0) || !(aa >= 1) || !(scaling >= 0))
    return fail("mcs_photon_pion: bad arguments");
  if (i_data < 1 || i_data > 4) return fail("mcs_photon_pion: i_data must be between 1 and 4");        // pion_kafexhiu.jl:81-88
  const int NM = c->P.num_psd_mom_bins + 2, ng = c->P.n_grid;
  if (NM > 208) return fail("mcs_photon_pion: too many momentum bins");
  const size_t n_in = (size_t)ng * NM + NM + ng, n_out = (size_t)ng * n_photon;
  if (ensure_stage(c, (long long)(n_in + n_out) + 4)) return 1;
  double* d_in = c->d_stage; double* d_out = c->d_stage + n_in;
  HIPCHK(hipMemcpyAsync(d_in, dNdp_pf, sizeof(double) * (size_t)ng * NM, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_in + (size_t)ng * NM, mom_edge_cgs, sizeof(double) * NM, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_in + (size_t)ng * NM + NM, target_density, sizeof(double) * ng, hipMemcpyHostToDevice, c->stream));
  const double log_emin = std::log10(emin_mev * MCS_MEV_ERG_);
  HIPCHK(mcs_launch_photon_pion(d_in, d_in + (size_t)ng * NM, d_in + (size_t)ng * NM + NM, ng, NM, n_photon, log_emin, bins_per_dec, mc, aa, scaling,
                                i_data, d_out, c->stream));
  HIPCHK(hipMemcpyAsync(emis, d_out, sizeof(double) * n_out, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (energy_erg) for (int j = 0; j < n_photon; ++j) energy_erg[j] = std::pow(10.0, log_emin + j * (1.0 / bins_per_dec));
  return 0;
}

// get_dNdp_2D (src/particle_counter.jl:343-627) on the resident psd / therm_sf / num_crossings: the d2N/dp dcos of every zone in
// the frame that moves with (gam_x, beta_x) against the shock frame (the ISM frame: gam0, beta0).  The result stays on the device
// for mcs_photon_ic; d2N (host, [n_grid][ntht+2][nmom+2], momentum fastest) may be null.
int mcs_dndp_2d(mcs_ctx* c, const mcs_consumer_in* in, double gam_x, double beta_x, double* d2N) {
  HIPCHK(hipSetDevice(c ? c->device : 0));
  if (consumers_ready(c, in, "mcs_dndp_2d")) return 1;
  if (fold_replicas(c)) return 1;
  if (!in->mom_edge_cgs || !in->cos_center || !in->pt_center || !in->zone_pop) return fail("mcs_dndp_2d: null table");
  if (!(gam_x >= 1) || !(beta_x >= 0 && beta_x < 1)) return fail("mcs_dndp_2d: bad frame");
  const int NM = c->P.num_psd_mom_bins + 2, NT = c->P.num_psd_tht_bins + 2, ng = c->P.n_grid;
  const size_t slab = (size_t)NM * NT * ng;
  if (!c->d_cscratch) HIPCHK(hipMalloc((void**)&c->d_cscratch, sizeof(double) * slab));
  if (!c->d_c2d) HIPCHK(hipMalloc((void**)&c->d_c2d, sizeof(double) * slab));
  std::vector<double> h((size_t)(2 * NM + NT + ng), 0.0);
  memcpy(h.data(), in->mom_edge_cgs, sizeof(double) * NM);
  memcpy(h.data() + NM, in->cos_center, sizeof(double) * (NT - 1));
  memcpy(h.data() + NM + NT, in->pt_center, sizeof(double) * (NM - 1));
  memcpy(h.data() + 2 * NM + NT, in->zone_pop, sizeof(double) * ng);
  HIPCHK(hipMemcpyAsync(c->d_ctab, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(mcs_launch_dndp_2d(&c->P, c->d_T + c->L.psd, c->d_T + c->L.therm_sf, c->d_I + MCS_I_NUM_CROSSINGS, c->d_ctab, in->rest_energy, in->n0,
                            in->therm_from_hist, gam_x, beta_x, c->d_cscratch, c->d_c2d, c->stream));
  if (d2N) HIPCHK(hipMemcpyAsync(d2N, c->d_c2d, sizeof(double) * slab, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->have_c2d = true;
  return 0;
}

// The inverse-Compton fold (include/mcs_ic.h) over the d2N/dp dcos the last mcs_dndp_2d left on the device.
int mcs_photon_ic(mcs_ctx* c, const double* mom_edge_cgs, double mc_e, int j_max, int n_nu, const double* alpha_in, const double* n_in, int n_photon,
                  double emin_mev, double bins_per_dec, double beam_area, double* energy_erg, double* emis) {
  HIPCHK(hipSetDevice(c ? c->device : 0));
  if (!c || !mom_edge_cgs || !alpha_in || !n_in || !emis) return fail("mcs_photon_ic: null argument");
  if (!c->have_c2d) return fail("mcs_photon_ic: no mcs_dndp_2d result on the device");
  const int NM = c->P.num_psd_mom_bins + 2, NT = c->P.num_psd_tht_bins + 2, ng = c->P.n_grid;
  if (NM > 208) return fail("mcs_photon_ic: too many momentum bins");
  if (n_photon < 1 || n_photon > 4096 || n_nu < 1 || n_nu > MCS_IC_NNU || j_max < 0 || j_max > NT - 2 || !(emin_mev > 0) || !(bins_per_dec > 0) ||
      !(mc_e > 0) || !(beam_area > 0))
    return fail("mcs_photon_ic: bad arguments");
  const size_t n_in_w = (size_t)NM + 2 * (size_t)n_nu, n_out = (size_t)ng * n_photon;
  if (ensure_stage(c, (long long)(n_in_w + n_out) + 4)) return 1;
  double* d_in = c->d_stage; double* d_out = c->d_stage + n_in_w;
  HIPCHK(hipMemcpyAsync(d_in, mom_edge_cgs, sizeof(double) * NM, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_in + NM, alpha_in, sizeof(double) * n_nu, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_in + NM + n_nu, n_in, sizeof(double) * n_nu, hipMemcpyHostToDevice, c->stream));
  const double log_min_rm = std::log10(emin_mev * MCS_MEV_ERG_ / (MCS_ME * MCS_C * MCS_C));
  HIPCHK(mcs_launch_photon_ic(c->d_c2d, d_in, d_in + NM, ng, NM, NT, j_max, n_nu, n_photon, log_min_rm, bins_per_dec, mc_e, beam_area, d_out, c->stream));
  HIPCHK(hipMemcpyAsync(emis, d_out, sizeof(double) * n_out, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (energy_erg) for (int k = 0; k < n_photon; ++k) energy_erg[k] = mcs_ic_alpha_out(log_min_rm, bins_per_dec, k) * (MCS_ME * MCS_C * MCS_C);
  return 0;
}

}  // extern "C"
