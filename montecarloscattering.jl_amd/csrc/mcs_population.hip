// mcs_population.hip -- K2 (pcut compaction + splitting), K3 (initial population),
// the elementwise test kernel and small fill/pack helpers, for gfx950.
//
// K2 replaces pcut_finalize/new_pcut (src/cuts.jl:34-124): the reference walks the
// saved flags serially and writes i_mult copies of every saved particle with
// weight/i_mult.  Here: block-wise exclusive scan of l_save (wave ballots, no
// atomics, so the order is the reference's order), then an output-centred gather
// (one lane per NEW particle, coalesced stores): new[o] = saved[src[o / i_mult]].
// K3 replaces the fast-push branch of init_pop (src/initializers.jl:1078-1131) and
// assign_particle_properties_to_population! (src/ion_init.jl:29-53).
#include "mcs_device.h"
#include "../../include/mcs_math.h"

#pragma clang fp contract(off)

namespace {
constexpr double TWOPI_ = 6.283185307179586;
constexpr double CC_ = MCS_C;

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
__device__ __forceinline__ double philox_uniform(unsigned long long key, uint32_t stream, unsigned long long j) {
  const unsigned long long blk = j >> 1;
  uint32_t o0, o1, o2, o3;
  philox_block((uint32_t)blk, (uint32_t)(blk >> 32), stream, 0u, (uint32_t)key, (uint32_t)(key >> 32), o0, o1, o2, o3);
  const uint32_t lo = (j & 1ull) ? o2 : o0, hi = (j & 1ull) ? o3 : o1;
  const unsigned long long u = ((unsigned long long)hi << 32) | lo;
  return (double)(u >> 11) * 0x1.0p-53;
}
}  // namespace

// ---------------------------------------------------------------------------------
// K2a: per-block count of saved flags.  1024 elements per 256-thread block.
extern "C" __global__ void __launch_bounds__(256)
mcs_k_count_saved(const uint8_t* __restrict__ l_save, long long n, unsigned int* __restrict__ block_counts, unsigned int match) {
  __shared__ unsigned int wsum[4];
  const long long base = (long long)blockIdx.x * 1024;
  unsigned int c = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long i = base + r * 256 + threadIdx.x;
    const bool f = i < n && l_save[i] == match;      // (the status byte: 1 saved; 5 saved and long, see KArgs::long_draws)
    c += (unsigned int)__popcll(__ballot(f));     // wave-uniform count
  }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// K2b: exclusive scan of the block counts (single 1024-thread block, serial over chunks).
extern "C" __global__ void __launch_bounds__(1024)
mcs_k_scan_blocks(const unsigned int* __restrict__ counts, long long nb, unsigned long long* __restrict__ offsets,
                  unsigned long long* __restrict__ total) {
  __shared__ unsigned long long sh[1024];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (long long c0 = 0; c0 < nb; c0 += 1024) {
    const long long i = c0 + threadIdx.x;
    const unsigned long long v = i < nb ? counts[i] : 0ull;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
      unsigned long long t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0ull;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) offsets[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// K2c: stable compaction: src[rank] = index of the rank-th saved particle.
extern "C" __global__ void __launch_bounds__(256)
mcs_k_compact_index(const uint8_t* __restrict__ l_save, long long n, const unsigned long long* __restrict__ offsets,
                    long long* __restrict__ src, unsigned int match) {
  __shared__ unsigned int wcount[4][4];   // [round][wave]
  const long long base = (long long)blockIdx.x * 1024;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  bool f[4];
  unsigned long long m[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long i = base + r * 256 + threadIdx.x;
    f[r] = i < n && l_save[i] == match;
    m[r] = __ballot(f[r]);
    if (lane == 0) wcount[r][wave] = (unsigned int)__popcll(m[r]);
  }
  __syncthreads();
  unsigned long long run = offsets[blockIdx.x];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    unsigned long long before = run;
    for (int w = 0; w < wave; ++w) before += wcount[r][w];
    if (f[r]) {
      const unsigned long long rank = before + (unsigned long long)__popcll(m[r] & ((1ull << lane) - 1ull));
      src[rank] = base + r * 256 + threadIdx.x;
    }
    run += wcount[r][0] + wcount[r][1] + wcount[r][2] + wcount[r][3];
  }
}

// ---- the same three steps and the split with the population size ON THE DEVICE (mcs_run_pcuts_fused: a species' pcuts are
// queued back to back, nothing is read back in between).  Grids are sized for the capacity; blocks beyond the population count 0.
extern "C" __global__ void __launch_bounds__(256)
mcs_k_count_saved_dev(const uint8_t* __restrict__ l_save, const PcutDev* __restrict__ pd, unsigned int* __restrict__ block_counts) {
  __shared__ unsigned int wsum[4];
  const long long n = pd->n_use;
  const long long base = (long long)blockIdx.x * 1024;
  unsigned int c = 0;
  if (base < n) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long i = base + r * 256 + threadIdx.x;
      const bool f = i < n && l_save[i] == 1;
      c += (unsigned int)__popcll(__ballot(f));
    }
  }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
extern "C" __global__ void __launch_bounds__(256)
mcs_k_compact_index_dev(const uint8_t* __restrict__ l_save, const PcutDev* __restrict__ pd, const unsigned long long* __restrict__ offsets,
                        long long* __restrict__ src) {
  __shared__ unsigned int wcount[4][4];
  const long long n = pd->n_use;
  const long long base = (long long)blockIdx.x * 1024;
  if (base >= n) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  bool f[4];
  unsigned long long m[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long i = base + r * 256 + threadIdx.x;
    f[r] = i < n && l_save[i] == 1;
    m[r] = __ballot(f[r]);
    if (lane == 0) wcount[r][wave] = (unsigned int)__popcll(m[r]);
  }
  __syncthreads();
  unsigned long long run = offsets[blockIdx.x];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    unsigned long long before = run;
    for (int w = 0; w < wave; ++w) before += wcount[r][w];
    if (f[r]) {
      const unsigned long long rank = before + (unsigned long long)__popcll(m[r] & ((1ull << lane) - 1ull));
      src[rank] = base + r * 256 + threadIdx.x;
    }
    run += wcount[r][0] + wcount[r][1] + wcount[r][2] + wcount[r][3];
  }
}
// pcut_finalize + the head of new_pcut on the device (src/cuts.jl:100-124, :42): n_saved from the flags (cross-checked against the
// transport kernel's own counter), i_mult = max(n_target / n_saved, 1), the size of the next population; the next launch's work
// counter and n_saved counter are cleared here.  One thread.
extern "C" __global__ void mcs_k_pcut_decide(PcutDev* __restrict__ pd, PcutDev* __restrict__ pd_next, const unsigned long long* __restrict__ scan_total,
                                             unsigned long long* __restrict__ counters /*[0] work, [1] n_saved (K1)*/, long long n_target,
                                             unsigned long long* __restrict__ err) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long ns = (long long)*scan_total;
  if ((unsigned long long)ns != counters[1]) atomicAdd(err, 1ull);      // (a spilling build once miscompiled the l_save store: csrc/Makefile)
  const long long im = ns > 0 ? (n_target / ns > 1 ? n_target / ns : 1) : 1;
  pd->n_saved = ns; pd->i_mult = im; pd->n_new = ns * im;
  if (pd_next) pd_next->n_use = ns * im;
  counters[0] = 0ull; counters[1] = 0ull;
}
// the late group of a pipelined pcut (mcs_run_pcuts_pipelined): the saved LONG particles are known only when the stragglers have finished,
// on the side stream -- their number, the size of their split and the population size the late launch of the next pcut reads.  One thread.
extern "C" __global__ void mcs_k_late_decide(PcutDev* __restrict__ pd, const unsigned long long* __restrict__ scan_total, long long i_mult,
                                             long long n_main_next) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long long ns = (long long)*scan_total;
  pd->n_saved = ns; pd->i_mult = i_mult; pd->n_new = ns * i_mult; pd->n_use = n_main_next + ns * i_mult;
}
extern "C" __global__ void __launch_bounds__(256)
mcs_k_split_dev(DevPop sv, DevPop out, const long long* __restrict__ src, const PcutDev* __restrict__ pd) {
  const long long n_new = pd->n_new, i_mult = pd->i_mult;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n_new; o += (long long)gridDim.x * blockDim.x) {
    const long long j = src[o / i_mult];
    out.weight[o] = sv.weight[j] / (double)i_mult;
    out.ptot_pf[o] = sv.ptot_pf[j];
    out.pb_pf[o] = sv.pb_pf[j];
    out.x_PT_cm[o] = sv.x_PT_cm[j];
    out.xn_per[o] = sv.xn_per[j];
    out.prp_x_cm[o] = sv.prp_x_cm[j];
    out.acctime_sec[o] = sv.acctime_sec[j];
    out.phi_rad[o] = sv.phi_rad[j];
    out.meta[o] = sv.meta[j];
  }
}

// K2d: new[o] = saved[src[o / i_mult]], weight / i_mult  (src/cuts.jl:66-92)
extern "C" __global__ void __launch_bounds__(256)
mcs_k_split(DevPop sv, DevPop out, const long long* __restrict__ src, long long n_new, long long i_mult) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_new) return;
  const long long j = src[o / i_mult];
  out.weight[o] = sv.weight[j] / (double)i_mult;
  out.ptot_pf[o] = sv.ptot_pf[j];
  out.pb_pf[o] = sv.pb_pf[j];
  out.x_PT_cm[o] = sv.x_PT_cm[j];
  out.xn_per[o] = sv.xn_per[j];
  out.prp_x_cm[o] = sv.prp_x_cm[j];
  out.acctime_sec[o] = sv.acctime_sec[j];
  out.phi_rad[o] = sv.phi_rad[j];
  out.meta[o] = sv.meta[j];
}

// K2e (multi-GPU new_pcut): the saved particles, compacted in index order, into caller-owned buffers that the host
// all-gathers: field f of the r-th saved particle at f64[f * cap + r]; its global index first + src[r] * stride.
extern "C" __global__ void __launch_bounds__(256)
mcs_k_saved_export(DevPop sv, const long long* __restrict__ src, long long n_saved, long long cap, long long first,
                   long long stride, const long long* __restrict__ gin, long long* __restrict__ gidx, double* __restrict__ f64,
                   uint32_t* __restrict__ meta) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_saved) return;
  const long long j = src[r];
  gidx[r] = gin ? gin[j] : first + j * stride;      // (gin: the index list of mcs_run_pcut_indexed)
  if (!f64) return;                                 // mcs_saved_gidx: the indices only
  f64[r] = sv.weight[j];
  f64[cap + r] = sv.ptot_pf[j];
  f64[2 * cap + r] = sv.pb_pf[j];
  f64[3 * cap + r] = sv.x_PT_cm[j];
  f64[4 * cap + r] = sv.xn_per[j];
  f64[5 * cap + r] = sv.prp_x_cm[j];
  f64[6 * cap + r] = sv.acctime_sec[j];
  f64[7 * cap + r] = sv.phi_rad[j];
  meta[r] = sv.meta[j];
}
// K2f: local particle k = element o = first + k * stride of the global split population o -> parent[o / i_mult]
// (src/cuts.jl:66-92 with a global o); parents in the layout of mcs_k_saved_export, sorted by global index.
extern "C" __global__ void __launch_bounds__(256)
mcs_k_split_import(DevPop out, const double* __restrict__ f64, const uint32_t* __restrict__ meta, long long cap,
                   long long i_mult, long long first, long long stride, long long n_local) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_local) return;
  const long long j = (first + k * stride) / i_mult;
  out.weight[k] = f64[j] / (double)i_mult;
  out.ptot_pf[k] = f64[cap + j];
  out.pb_pf[k] = f64[2 * cap + j];
  out.x_PT_cm[k] = f64[3 * cap + j];
  out.xn_per[k] = f64[4 * cap + j];
  out.prp_x_cm[k] = f64[5 * cap + j];
  out.acctime_sec[k] = f64[6 * cap + j];
  out.phi_rad[k] = f64[7 * cap + j];
  out.meta[k] = meta[j];
}

// K3: initial population.
extern "C" __global__ void __launch_bounds__(256)
mcs_k_init_pop(DevPop out, const double* __restrict__ ptot_in, const double* __restrict__ weight_in, long long n,
               long long j_offset, long long j_stride, long long n_total, unsigned long long key, double m, double u,
               double x_start, int i_grid_start, int relativistic, int fast_push, double xn_per_fine, double x_grid_stop,
               int n_bins, const double* __restrict__ bin_ptot, const double* __restrict__ bin_weight,
               const long long* __restrict__ bin_start) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const long long j = j_offset + k * j_stride;      // global 0-based index (j_stride > 1: a strided shard)
  double ptot, wgt;
  if (n_bins > 0) {
    // binned form (the momentum discretisation of set_inj_dist, initializers.jl:1251-1328): particle j
    // belongs to the bin b with bin_start[b] <= j < bin_start[b+1]
    int lo = 0, hi = n_bins;                 // invariant: bin_start[lo] <= j < bin_start[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bin_start[mid] <= j) lo = mid; else hi = mid; }
    ptot = bin_ptot[lo]; wgt = bin_weight[lo];
  } else {
    ptot = ptot_in[k]; wgt = weight_in[k];
  }
  const double U = philox_uniform(key, 0x504F50u, (unsigned long long)j);
  const double beta_u = u / CC_;
  double pb;
  if (!fast_push) {
    pb = ptot * 2 * (U - 0.5);                                   // initializers.jl:1006
  } else if (relativistic) {                                     // initializers.jl:1095-1110
    const double gam_pf = mcsm::hypot1(ptot / (m * CC_));
    const double beta_pf = __builtin_sqrt(1 - 1 / (gam_pf * gam_pf));
    const double bmin = __builtin_fabs((beta_u - beta_pf) / (1 - beta_u * beta_pf));
    const double bmax = __builtin_fabs((beta_u + beta_pf) / (1 + beta_u * beta_pf));
    const double bx_sf = bmin + __builtin_sqrt(U * (bmax - bmin) * (bmax - bmin));
    const double vx_pf = (bx_sf - beta_u) / (1 - bx_sf * beta_u) * CC_;
    pb = gam_pf * m * vx_pf;
  } else {                                                       // initializers.jl:1111-1126
    const double vt_pf = ptot / m;
    const double vmin = __builtin_fabs(u - vt_pf), vmax = __builtin_fabs(u + vt_pf);
    const double vx_sf = vmin + __builtin_sqrt(U * (vmax - vmin) * (vmax - vmin));
    const double vx_pf = vx_sf - u;
    pb = 1.0 * m * vx_pf;
  }
  out.weight[k] = wgt;
  out.ptot_pf[k] = ptot;
  out.pb_pf[k] = pb;
  out.x_PT_cm[k] = x_start;
  out.xn_per[k] = xn_per_fine;                                   // ion_init.jl:45-49
  out.prp_x_cm[k] = x_grid_stop;
  out.acctime_sec[k] = 0.0;
  out.phi_rad[k] = TWOPI_ * philox_uniform(key, 0x504F50u, (unsigned long long)(n_total + j));
  out.meta[k] = mcs_pack_meta(i_grid_start, 1, 0, 0);
}

// ---------------------------------------------------------------------------------
extern "C" __global__ void mcs_k_fill_f64(double* p, long long n, double v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
extern "C" __global__ void mcs_k_copy_f64(double* dst, const double* src, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}

// device evaluation of the math / RNG primitives (bit-parity tests against the oracle)
extern "C" __global__ void mcs_k_eval_fn(int fn, long long n, const double* a, const double* b, double* out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r = 0.0;
  switch (fn) {
    case MCS_FN_SIN: r = mcsm::sin(a[i]); break;
    case MCS_FN_COS: r = mcsm::cos(a[i]); break;
    case MCS_FN_ASIN: r = mcsm::asin(a[i]); break;
    case MCS_FN_ACOS: r = mcsm::acos(a[i]); break;
    case MCS_FN_ATAN2: r = mcsm::atan2(a[i], b[i]); break;
    case MCS_FN_LOG10: r = mcsm::log10(a[i]); break;
    case MCS_FN_MOD2PI: r = mcsm::mod2pi(a[i]); break;
    case MCS_FN_SQRT: r = __builtin_sqrt(a[i]); break;
    case MCS_FN_DIV: r = a[i] / b[i]; break;
    case MCS_FN_HYPOT1: r = mcsm::hypot1(a[i]); break;
    case MCS_FN_UNIFORM: r = philox_uniform((unsigned long long)a[i], 0u, (unsigned long long)b[i]); break;
    default: break;
  }
  out[i] = r;
}

// ---------------------------------------------------------------------------------
// host launchers (called from mcs_api.hip)
extern "C" {

// the compaction half of new_pcut: src[r] = local index of the r-th saved particle, *total_dev = their number.
// Queued right behind the transport kernel by mcs_run_pcut, so that ONE read-back brings both the kernel's own
// n_saved counter and this independent count of l_save (they must agree) and nothing has to be waited for later.
hipError_t mcs_launch_compact_match(const uint8_t* l_save, long long n, unsigned int* block_counts, unsigned long long* block_offsets,
                                    unsigned long long* total_dev, long long* src, unsigned int match, hipStream_t st) {
  const long long nb = (n + 1023) / 1024;
  if (nb == 0) return hipMemsetAsync(total_dev, 0, sizeof(unsigned long long), st);
  hipLaunchKernelGGL(mcs_k_count_saved, dim3((unsigned)nb), dim3(256), 0, st, l_save, n, block_counts, match);
  hipLaunchKernelGGL(mcs_k_scan_blocks, dim3(1), dim3(1024), 0, st, block_counts, nb, block_offsets, total_dev);
  hipLaunchKernelGGL(mcs_k_compact_index, dim3((unsigned)nb), dim3(256), 0, st, l_save, n, block_offsets, src, match);
  return hipGetLastError();
}
hipError_t mcs_launch_compact(const uint8_t* l_save, long long n, unsigned int* block_counts, unsigned long long* block_offsets,
                              unsigned long long* total_dev, long long* src, hipStream_t st) {
  return mcs_launch_compact_match(l_save, n, block_counts, block_offsets, total_dev, src, 1u, st);
}
// the late group of a pipelined pcut: compaction of the saved LONG particles (status 5), their split behind the main group's children
hipError_t mcs_launch_late_split(const uint8_t* l_save, long long n, unsigned int* block_counts, unsigned long long* block_offsets,
                                 unsigned long long* total_dev, long long* src, PcutDev* pd, long long i_mult, long long n_main_next, DevPop sv,
                                 DevPop out_at_main_end, int split_blocks, hipStream_t st) {
  hipError_t e = mcs_launch_compact_match(l_save, n, block_counts, block_offsets, total_dev, src, 5u, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(mcs_k_late_decide, dim3(1), dim3(64), 0, st, pd, total_dev, i_mult, n_main_next);
  hipLaunchKernelGGL(mcs_k_split_dev, dim3((unsigned)split_blocks), dim3(256), 0, st, sv, out_at_main_end, src, pd);
  return hipGetLastError();
}
// one pcut's pcut_finalize + new_pcut with the sizes on the device (see mcs_k_pcut_decide); cap_n: the largest population possible
hipError_t mcs_launch_finalize_split_dev(const uint8_t* l_save, long long cap_n, unsigned int* block_counts, unsigned long long* block_offsets,
                                         unsigned long long* scan_total, long long* src, PcutDev* pd, PcutDev* pd_next, unsigned long long* counters,
                                         long long n_target, unsigned long long* err, DevPop sv, DevPop out, int split_blocks, hipStream_t st) {
  const long long nb = (cap_n + 1023) / 1024;
  hipLaunchKernelGGL(mcs_k_count_saved_dev, dim3((unsigned)nb), dim3(256), 0, st, l_save, pd, block_counts);
  hipLaunchKernelGGL(mcs_k_scan_blocks, dim3(1), dim3(1024), 0, st, block_counts, nb, block_offsets, scan_total);
  hipLaunchKernelGGL(mcs_k_compact_index_dev, dim3((unsigned)nb), dim3(256), 0, st, l_save, pd, block_offsets, src);
  hipLaunchKernelGGL(mcs_k_pcut_decide, dim3(1), dim3(64), 0, st, pd, pd_next, scan_total, counters, n_target, err);
  hipLaunchKernelGGL(mcs_k_split_dev, dim3((unsigned)split_blocks), dim3(256), 0, st, sv, out, src, pd);
  return hipGetLastError();
}
hipError_t mcs_launch_split(DevPop sv, DevPop out, const long long* src, long long n_new, long long i_mult, hipStream_t st) {
  if (n_new > 0)
    hipLaunchKernelGGL(mcs_k_split, dim3((unsigned)((n_new + 255) / 256)), dim3(256), 0, st, sv, out, src, n_new, i_mult);
  return hipGetLastError();
}
hipError_t mcs_launch_saved_export(DevPop sv, const long long* src, long long n_saved, long long cap, long long first,
                                   long long stride, const long long* gin, long long* gidx, double* f64, uint32_t* meta,
                                   hipStream_t st) {
  if (n_saved > 0)
    hipLaunchKernelGGL(mcs_k_saved_export, dim3((unsigned)((n_saved + 255) / 256)), dim3(256), 0, st, sv, src, n_saved, cap,
                       first, stride, gin, gidx, f64, meta);
  return hipGetLastError();
}
hipError_t mcs_launch_split_import(DevPop out, const double* f64, const uint32_t* meta, long long cap, long long i_mult,
                                   long long first, long long stride, long long n_local, hipStream_t st) {
  if (n_local > 0)
    hipLaunchKernelGGL(mcs_k_split_import, dim3((unsigned)((n_local + 255) / 256)), dim3(256), 0, st, out, f64, meta, cap,
                       i_mult, first, stride, n_local);
  return hipGetLastError();
}

hipError_t mcs_launch_init_pop(DevPop out, const double* ptot_in, const double* weight_in, long long n, long long j_offset,
                               long long j_stride, long long n_total, unsigned long long key, double m, double u, double x_start,
                               int i_grid_start, int relativistic, int fast_push, double xn_per_fine, double x_grid_stop,
                               int n_bins, const double* bin_ptot, const double* bin_weight, const long long* bin_start,
                               hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(mcs_k_init_pop, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, ptot_in, weight_in, n,
                     j_offset, j_stride, n_total, key, m, u, x_start, i_grid_start, relativistic, fast_push, xn_per_fine, x_grid_stop,
                     n_bins, bin_ptot, bin_weight, bin_start);
  return hipGetLastError();
}

hipError_t mcs_launch_fill(double* p, long long n, double v, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mcs_k_fill_f64, dim3((unsigned)blocks), dim3(256), 0, st, p, n, v);
  return hipGetLastError();
}
hipError_t mcs_launch_copy(double* dst, const double* src, long long n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mcs_k_copy_f64, dim3((unsigned)blocks), dim3(256), 0, st, dst, src, n);
  return hipGetLastError();
}
hipError_t mcs_launch_eval(int fn, long long n, const double* a, const double* b, double* out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(mcs_k_eval_fn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, fn, n, a, b, out);
  return hipGetLastError();
}

}  // extern "C"

// Fold the replicas of the thermal histograms into the tally buffer and clear them (see MCS_THERM_REPLICAS).
extern "C" __global__ void mcs_k_fold_replicas(double* __restrict__ dst, double* __restrict__ rep, long long n, int n_rep) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int r = 0; r < n_rep; ++r) { s += rep[(long long)r * n + i]; rep[(long long)r * n + i] = 0.0; }
    if (s != 0.0) dst[i] += s;
  }
}
extern "C" hipError_t mcs_launch_fold_replicas(double* dst, double* rep, long long n, int n_rep, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  mcs_k_fold_replicas<<<2048, 256, 0, st>>>(dst, rep, n, n_rep);
  return hipGetLastError();
}
