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
// Execution model (CDNA4; the measurements behind it are in DESIGN.md section 4 and profiles/):
//  * one wavefront lane = one live particle; all particle state lives in VGPRs;
//  * persistent lanes: a lane whose particle finished claims the next unclaimed
//    particle (wave-aggregated atomic on one counter, 8 idle lanes at a time), so the
//    64 lanes of a wave stay busy although histories last 1 .. 10^4 passes;
//  * flag-driven loop: the common pass (scatter, clock, move, event detection) is
//    straight-line code that EVERY lane executes; everything rare -- before or after
//    the move -- sits in one region entered when the lane has an event pending or a
//    bit set in its flags register (see the comment above move_and_detect);
//  * the grid tables (n_grid+2 fp64 each) plus per-zone sin/cos(theta_B) and
//    1/(qB), and the time cuts, sit in LDS (the current zone's values in registers);
//    zone-crossing tallies are pushed as records onto per-wave LDS stacks and tallied
//    64 at a time; the three flux vectors, num_crossings and the counters are staged
//    in LDS (ds_add_f64) and flushed with one global atomic per entry per block; the
//    22 MB psd and the escape spectra take global_atomic_add_f64 (no-return);
//  * the common pass issues NO global load, NO LDS table read and NO scratch access;
//  * two kernels from this source: mcs_k_transport and its compile-time
//    specialisation for the common configuration, mcs_k_transport_plain;
//  * RNG: Philox4x32-10 keyed by the reference's iseed_mod, counter = draw
//    number; no RNG state in memory;
//  * no MFMA: scalar fp64 per-particle arithmetic.
//
// Numerics: every fp64 expression is evaluated in the same order as the CPU
// oracle (which follows the Julia source), with include/mcs_math.h for the
// transcendentals and -ffp-contract=off, so per-particle results are
// bit-identical to the oracle; only the order of the atomic tally adds differs.
// Values that are pure functions of unchanged inputs (per-zone sin/cos/1/(qB),
// cos_max of the scattering cone) are cached, which does not change a bit.
#include "mcs_device.h"
#define MCS_DEVICE_FAST_SQRT 1
#include "../../include/mcs_math.h"
#include "../../include/mcs_math_f32.h"

#pragma clang fp contract(off)

#ifndef MCS_WAVES_PER_SIMD
#define MCS_WAVES_PER_SIMD 2
#endif
#ifndef MCS_REFILL_MIN
#define MCS_REFILL_MIN 12       // idle lanes a wave collects before it claims new particles: a refill stalls the wave for a memory latency,
                                // and parked particles resume in batches of that size (final kernel: 6 -> 458 ms, 8 -> 451, 12 -> 442, 16 -> 444)
#endif
#ifndef MCS_DEFER_K
#define MCS_DEFER_K 8           // lanes with pending rare work a wave collects before it enters the rare region (see `enter` in the loop)
#endif
#ifndef MCS_PASSES_PER_ITER
#define MCS_PASSES_PER_ITER 6     // common passes per trip through the loop header (see the end of the loop)
#endif
#ifndef MCS_MERGE_POLL_MASK
#define MCS_MERGE_POLL_MASK 15u   // tail consolidation: the waves of a pair look at each other every 16 passes
#endif
#ifndef MCS_TAIL_CROSS
#define MCS_TAIL_CROSS 1          // tail loop: a plain zone crossing that ended the loop is handled right behind it (0: A/B builds)
#endif
// Rare paths (zone-crossing tallies, frame transforms, retro walk, finish): outlined
// calls with by-value arguments, or inlined (-DMCS_INLINE_COLD) -- a tuning knob.
#ifdef MCS_INLINE_COLD
#define MCS_COLD __forceinline__
#else
#define MCS_COLD __noinline__
#endif

// Launch constants live in a per-context device buffer and are read through the
// CONSTANT address space: uniform loads become s_load (scalar cache) and a field is
// fetched where it is used, so cold-path fields do not occupy SGPRs in the hot loop.
typedef const __attribute__((address_space(4))) KArgs CK;

namespace {

// Literals are materialised in scalar registers at their point of use (MCS_SC, see
// include/mcs_math.h) so that none of them is hoisted into a VGPR for the whole loop.
#define PI_ MCS_SC(3.141592653589793)
#define TWOPI_ MCS_SC(6.283185307179586)
#define HALFPI_ MCS_SC(1.5707963267948966)   /* == 3.141592653589793 / 2 exactly */
#define SIN_UL MCS_SC(0x1.fffffffffffffp-1)
#define MP_ MCS_SC(MCS_MP)
#define CC_ MCS_SC(MCS_C)

// ---- correctly rounded fp64 division without the range rescale ---------------------------
// hipcc expands a/b to div_scale x2, rcp, two Newton steps, one quotient correction, div_fmas
// and div_fixup (11 VALU); the scaling only matters for operands near the ends of the exponent
// range, which the path never divides.  rcp_refined(b) is the twice-refined reciprocal of that
// very sequence, div_r(a,b,r) its quotient step: bit-identical quotients in 8 instructions,
// and the reciprocal can be shared by divisions with the same denominator (pb/ptot and
// p_perp/ptot, scattering.jl:65-66) or cached while the denominator is unchanged.
__device__ __forceinline__ double rcp_refined(double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
__device__ __forceinline__ double div_r(double a, double b, double r) {
  const double q = a * r;
  const double rem = __builtin_fma(-b, q, a);
  return __builtin_fma(rem, r, q);
}
__device__ __forceinline__ double fdiv(double a, double b) { return div_r(a, b, rcp_refined(b)); }
#define FSQRT(x) (mcsm::sqrt_nn_(x))      /* the two square roots of a scatter: arguments in [0, 1] */

// ---- Philox4x32-10 ------------------------------------------------------------
__device__ __forceinline__ void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                             uint32_t k1, uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
#ifndef MCS_PHILOX_ROUNDS
#define MCS_PHILOX_ROUNDS 10      /* anything else is a TIMING EXPERIMENT (wrong random numbers) */
#endif
#pragma unroll
  for (int r = 0; r < MCS_PHILOX_ROUNDS; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;   // one v_mad_u64_u32 each
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    // three-input xor in ONE instruction (gfx950 v_bitop3_b32, truth table 0x96 = a ^ b ^ c)
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

__device__ __forceinline__ double u64_to_unit(uint32_t lo, uint32_t hi) {
  // (double)(u64 >> 11) * 2^-53: the 53-bit integer is hi * 2^21 + (lo >> 11), so the value is
  // hi * 2^-32 + (lo >> 11) * 2^-53 -- two exact conversions, an exact scaling and an fma whose result is
  // representable, hence exact: one shift, two v_cvt, a multiply and an fma.
  return __builtin_fma((double)hi, 0x1.0p-32, (double)(lo >> 11) * 0x1.0p-53);
}

// Per-particle stream: draw j -> words (2*(j&1), 2*(j&1)+1) of block j>>1.  The draw index is even wherever
// a call site starts (see Rng in oracle/mcs_oracle.cpp): the two draws of a scatter are one block, a single draw
// takes the first half of a block and skips the second -- no parity logic, no spare register.
// (Computing the next step's block one step ahead -- off the critical path -- was tried and
// gained nothing: hipcc does not interleave it with the fp64 chain under this register pressure.)
struct Rng {
  uint32_t k0, k1;
  uint32_t n;        // draw index (a history stays < 2^32)
  __device__ __forceinline__ void init(unsigned long long key) {
    k0 = (uint32_t)key; k1 = (uint32_t)(key >> 32); n = 0;
  }
  // a single draw: index n, then skip n+1
  __device__ __forceinline__ double rand() {
    const uint32_t j = n;
    n = j + 2u;
    uint32_t o0, o1, o2, o3;
    philox_block(j >> 1, 0u, 0u /*STREAM_PARTICLE*/, 0u, k0, k1, o0, o1, o2, o3);
    return u64_to_unit(o0, o1);
  }
  // the two draws of a scatter (src/scattering.jl:68,71): indices n, n+1 = one Philox block
  __device__ __forceinline__ void pair(double& u1, double& u2) {
    const uint32_t j = n;
    n = j + 2u;
    uint32_t o0, o1, o2, o3;
    philox_block(j >> 1, 0u, 0u, 0u, k0, k1, o0, o1, o2, o3);
    u1 = u64_to_unit(o0, o1);
    u2 = u64_to_unit(o2, o3);
  }
};

// ---- LDS (static layout: compile-time addresses, no pointer registers) -----------------
// 10 tables of n_grid+2 entries (<= MCS_MAXNE, checked by mcs_create), flux / crossing /
// counter / scalar staging, time cuts: 23 KB per workgroup.
#define MCS_MAXNE 208
__shared__ double S_x[MCS_MAXNE], S_ux[MCS_MAXNE], S_uz[MCS_MAXNE], S_ut[MCS_MAXNE], S_gsf[MCS_MAXNE], S_gef[MCS_MAXNE],
    S_bt[MCS_MAXNE], S_bsin[MCS_MAXNE], S_bcos[MCS_MAXNE], S_gd[MCS_MAXNE];
__shared__ double S_fl[3 * MCS_MAXNE];      // pxx | pxz | energy flux staging, stride MCS_MAXNE
__shared__ double S_wc[MCS_NA_C];            // weight_coupled of this ion: one address per time cut, hit by every particle that passes it
__shared__ double S_eff[2][MCS_PSD_MAX + 1]; // esc_energy_eff | esc_num_eff of this ion: one entry per momentum bin, i.e. ONE
                                            // address for a population of replicas (a global atomic there serialises at ~6 ns)
__shared__ double S_tc[MCS_NA_C];           // time cuts
__shared__ int S_nc[MCS_MAXNE];             // num_crossings staging
// Deferred zone-crossing tallies: a lane that crossed a zone boundary does NOT run the
// ~300-instruction tally code (transform, bins, atomics) on the spot -- with a handful of
// active lanes while the rest of the wave waits, that cost 43 % of the kernel.  It pushes a
// record into its wave's LDS stack; whenever 64 records are pending the whole wave pops 64
// and tallies them, one record per lane (the tallies do not feed back into the particles).
__shared__ double S_evf[4][MCS_EV_F64][MCS_EV_CAP];
__shared__ unsigned int S_evu[4][MCS_EV_CAP];
__shared__ unsigned int S_evcur[4];         // per-wave stack height

// ---- particle state (registers) ---------------------------------------------------
// The zone properties "of the current pass" (ux, uz, utot, gamma_sf, gamma_ef, sin/cos
// theta_B; particle_loop.jl:195-204) are the LDS table entries of zone `ig3`, the zone the
// particle was in when Code Block 3 last ran; they are re-read from LDS where needed.
// Rare regions: the hint lets block placement move their bodies out of line, so the common step
// falls through its branches (a taken branch refills the instruction buffer).
#define MCS_UNLIKELY(x) __builtin_expect(!!(x), 0)

// Per-particle flags (Pt::flags).  Any bit set sends the lane through the rare block before its
// next step; the bits are set where the underlying quantity changes (always in rare code), so the
// common step tests ONE register instead of re-evaluating a dozen conditions.
#define F_RS      0x001   // scatter-side derived quantities are stale (cm_val, rp_val, gyro_period, x_dt, F_NEARP, F_SAVE)
#define F_RM      0x002   // move-side derived quantities are stale (t_step, dphi, rg_val)
#define F_ZONE    0x004   // i_grid != ig3: Code Block 3 has to reload the zone (particle_loop.jl:186-246)
#define F_B1      0x008   // next pass is Code Block 1 (i_return == 1 after a PRP return, particle_loop.jl:167-177)
#define F_NEARP   0x010   // ptot_pf > pmax_cutoff: the p_max exit needs its transform (particle_loop.jl:264)
#define F_NEARFEB 0x020   // (not set since round 3: the upstream free-escape boundary is a position threshold, see `near_feb` in refresh_thr)
#define F_SAVE    0x040   // downstream && ptot_pf > pcut: saved for the next pcut at the next Code Block 3
#define F_CROSSED 0x080   // the last move changed i_grid (energy transfer test, particle_loop.jl:235)
#define F_CHECK   0x100   // a time / fine-coarse event happened: re-run the exit tests and the xn decision
#define F_CM      0x200   // only cos_max is stale (fine/coarse switch in the last pass): refresh_scatter alone
#define F_NOPARK  0x400   // just loaded, or released from waiting: run the full Code Blocks now
#define F_WAIT    0x800   // needs the full Code Blocks and waits for company (see "Waiting" in transport_body): sits out the common passes
#define F_INJCHK  0x2000  // loaded downstream-flagged, not injected, at x < 0 (a state only a caller's own population has): the `inj` update
                          // that follows EVERY move in the reference (particle_loop.jl:433-435) is due after the first move -- slow_post, event or not
#define F_LOST    0x1000  // LOSSY kernel: the in-line radiative loss of the pass left no momentum (particle_loop.jl:578-592 -> finish code 4)
struct Pt {
  double weight, ptot_pf, pb_pf, p_perp, gam_pf, x, x_old, phi, prp, acctime, xn_per;
  double dphi;                   // 2pi / xn_per (particle_loop.jl:529)
  double gyro_denom, gyro_rad, gyro_rad_tot, gyro_period, t_step;
  // Pure functions of (ptot_pf, gam_pf, gyro_denom, xn_per, prp), which change only in rare code
  // (frame transform at the shock, energy transfer, radiative loss, PRP logic, field change,
  // fine/coarse switch).  Recomputed there with the reference's own expressions: bit-identical.
  double rp_val;                 // refined 1/ptot_pf
  double cm_val;                 // cos_max of the scattering cone (scattering.jl:60)
  double rg_val;                 // refined 1/(gam_pf*m)
  double x_dt;                   // downstream_test exit threshold (refresh_dtest)
  double t_ev;                   // min(next time cut, age_max): the clock compares against one number
  // properties of zone ig3 (gamma_sf, cos theta_B, u_x, gamma_ef) and the edges of zone i_grid: read
  // from the LDS tables when the zone changes (rare code), not in every pass
  double z_gsf, z_bcos, z_ux, z_gef, z_lo, z_hi;
  // the two numbers the common pass compares the position with (refresh_thr): every position event of
  // move_and_detect -- zone edge, end of the grid, PRP, x_dt, fine/coarse switch -- implies x >= t_hi or x <= t_lo
  double t_hi, t_lo;
  // the clock of the common pass: gamma_ef of the zone while downstream, else 0 (acctime + 0 is acctime), and the time
  // of the next clock event while downstream, else +inf -- refreshed with the thresholds
  double c_gef, c_tev;
  int flags;
  int ovr_inc;                   // 1 while downstream past the last time cut (D4 counter), else 0
  unsigned n_ovr;                // passes counted by D4, flushed when the particle ends
  int i_grid, i_grid_old, ig3, helix, tcut, n_retro;
  bool downstream, inj;
  int npush;                     // tally records this lane pushed in the current pass (0..2)
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
// Event counters are bumped in LDS (one ds_add_u32) and flushed once per block: a
// per-particle global atomic on ONE address serialises at ~12 ns each (1e6 particles = 12 ms).
__shared__ unsigned int g_ctr[MCS_IC_COUNT + 1];
__shared__ double g_sc[8];
__shared__ unsigned long long S_steps[3];   // helix steps, retro steps, RNG draws of the particles this block finished   // [0..3] layout.scalars, [4] esc_flux, [5] px_esc_feb, [6] energy_esc_feb (this ion/iter)

// ---- optional phase profile (-DMCS_PROF; tools/gpu_prof.py): cycle / lane counts per loop phase
#ifdef MCS_PROF
#define MCS_NPROF 64
__device__ unsigned long long g_prof[MCS_NPROF];
__device__ unsigned long long g_wave[8192][8];   // per wave: start, counter exhausted, end (s_memrealtime, 100 MHz), live lanes at exhaustion (tools/gpu_timeline.py)
__shared__ unsigned long long S_prof[MCS_NPROF];
#define PROF_T() __builtin_amdgcn_s_memtime()
#ifdef MCS_PROF_TAIL   // count only the passes after the work counter is exhausted
#ifdef MCS_PROF_ALLPHASES   // the same timers over the whole launch
#define PROF_GATE true
#else
#define PROF_GATE exhausted
#endif
__shared__ unsigned int S_ttgate[4];            // 1 once this wave found the work counter exhausted
__shared__ unsigned long long S_tttm[4];        // time of the last mark of this wave
#define TTG_COUNT(slot) do { if (S_ttgate[threadIdx.x >> 6]) atomicAdd(&S_prof[slot], 1ull); } while (0)
#define TTG_START() do { if ((threadIdx.x & 63u) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) S_tttm[threadIdx.x >> 6] = __builtin_amdgcn_s_memtime(); } while (0)
#define TTG_MARK(slot) do { if ((threadIdx.x & 63u) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) { const unsigned long long tn__ = __builtin_amdgcn_s_memtime(); \
    if (S_ttgate[threadIdx.x >> 6]) { atomicAdd(&S_prof[slot], tn__ - S_tttm[threadIdx.x >> 6]); atomicAdd(&S_prof[(slot) + 8], 1ull); } S_tttm[threadIdx.x >> 6] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define PROF_GATE true
#endif
#define PROF_ADD(slot, v) do { if (PROF_GATE && (threadIdx.x & 63u) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicAdd(&S_prof[slot], (unsigned long long)(v)); } while (0)
#define PROF_LANES(slot, pred) do { const unsigned long long m__ = __ballot(pred); if (PROF_GATE && (threadIdx.x & 63u) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicAdd(&S_prof[slot], (unsigned long long)__popcll(m__)); } while (0)
#else
#define PROF_T() 0ull
#define PROF_ADD(slot, v) do { } while (0)
#define PROF_LANES(slot, pred) do { } while (0)
#endif
#ifndef MCS_PROF_TAIL
#define TTG_COUNT(slot) do { } while (0)
#define TTG_START() do { } while (0)
#define TTG_MARK(slot) do { } while (0)
#endif
__device__ __forceinline__ void cnt(CK* a, int which, unsigned int v = 1u) {
  (void)a;
  (void)__hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned int*)&g_ctr[which], v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void sadd(int which, double v) { ladd_f64(&g_sc[which], v); }
// a tally: into this block's replica of the histograms at the head of the buffer (see MCS_TALLY_REPLICAS), else into T
__device__ __forceinline__ void tadd(CK* a, long long off, double v) {
  double* base = a->T;
  if (off < a->rep_n) base = a->tally_rep + (long long)(blockIdx.x % MCS_TALLY_REPLICAS) * a->rep_n;
  gadd_f64(&base[off], v);
}

// src/get_psd_bins.jl:16-39
__device__ __forceinline__ int bin_momentum(CK* a, double ptot_sk) {
  int bin;
  if (ptot_sk < a->P.psd_mom_min) bin = 0;
  else bin = (int)__builtin_trunc(mcsm::log10(ptot_sk / a->P.psd_mom_min) * a->P.psd_bins_per_dec_mom) + 1;
  if (bin > a->P.num_psd_mom_bins) { cnt(a, MCS_IC_MOMBIN_CLAMP); bin = a->P.num_psd_mom_bins; }
  return bin;
}
// src/get_psd_bins.jl:73-97
__device__ __forceinline__ int bin_angle(CK* a, double px_sk, double ptot_sk) {
  if (ptot_sk == 0.0) return 0;
  const double p_cos = -px_sk / ptot_sk;
  int bin;
  if (p_cos < a->P.psd_cos_fine) {
    bin = a->P.num_psd_tht_bins - (int)__builtin_trunc((p_cos + 1) / a->P.psd_dcos);
  } else {
    const double th = mcsm::acos(p_cos);
    bin = th < a->P.psd_tht_min ? 0 : (int)__builtin_trunc(mcsm::log10(th / a->P.psd_tht_min) * a->P.psd_bins_per_dec_tht) + 1;
  }
  return bin < a->P.num_psd_tht_bins ? bin : a->P.num_psd_tht_bins;
}

// src/transformers.jl:440-476
__device__ __forceinline__ void transform_p_PS(double aa, double pb_pf, double p_perp, double gam_pf, double phi,
                                               double ux, double gsf, double bcos, double bsin, double& ptot_sk,
                                               double& px, double& py, double& pz, double& gam_sk) {
  const double m = aa * MP_;
  const double mc = m * CC_;
  const double phi_p = phi + HALFPI_;
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

// results of the rare momentum transforms, returned by value (registers)
struct Mom { double ptot, pb, pperp, gam, phi; };

// src/transformers.jl:523-607: zone io (old) -> zone in (new)
__device__ MCS_COLD Mom transform_p_PSP(CK* a, int io, int in, double r_pb, double r_pperp, double r_gam,
                                            double r_phi) {
  const double aa = a->aa;
  const double ux_o = S_ux[io], uz_o = S_uz[io], ut_o = S_ut[io], gsf_o = S_gsf[io], bcos_o = S_bcos[io], bsin_o = S_bsin[io];
  const double ux = S_ux[in], uz = S_uz[in], ut = S_ut[in], gsf = S_gsf[in], bcos = S_bcos[in], bsin = S_bsin[in];
  double phi_p = r_phi + HALFPI_;
  const double m = aa * MP_;
  const double mc = m * CC_;
  double sn, cs;
  mcsm::sincos(phi_p, &sn, &cs);
  const double p_p_cos = r_pperp * cs;
  double fx = r_pb * bcos_o - p_p_cos * bsin_o;
  double fy = r_pperp * sn;
  double fz = r_pb * bsin_o + p_p_cos * bcos_o;
  double kx, ky, kz;
  {
    const double qx = ux_o / ut_o, qz = uz_o / ut_o;
    kx = ((gsf_o - 1) * (qx * qx) + 1) * fx + (gsf_o - 1) * (ux_o * uz_o / (ut_o * ut_o)) * fz + gsf_o * r_gam * m * ux_o;
    ky = fy;
    kz = (gsf_o - 1) * (ux_o * uz_o / (ut_o * ut_o)) * fx + ((gsf_o - 1) * (qz * qz) + 1) * fz + gsf_o * r_gam * m * uz_o;
  }
  const double ptot_sk = mcsm::norm3(kx, ky, kz);
  const double pb_sk = kx * bcos + kz * bsin;
  if (ptot_sk < __builtin_fabs(pb_sk)) cnt(a, MCS_IC_PSP_CLAMP);   // the clamped shock-frame pair is never used again
  const double gam_sk = mcsm::hypot1(ptot_sk / mc);
  {
    const double qx = ux / ut, qz = uz / ut;
    fx = ((gsf - 1) * (qx * qx) + 1) * kx + (gsf - 1) * (ux * uz / (ut * ut)) * kz - gsf * gam_sk * m * ux;
    fy = ky;
    fz = (gsf - 1) * (ux * uz / (ut * ut)) * kx + ((gsf - 1) * (qz * qz) + 1) * kz - gsf * gam_sk * m * uz;
  }
  Mom r;
  r.ptot = mcsm::norm3(fx, fy, fz);
  r.pb = fx * bcos + fz * bsin;
  if (r.ptot < __builtin_fabs(r.pb)) {
    r.pperp = 1.0e-6 * r.ptot;
    r.pb = __builtin_copysign(__builtin_sqrt(r.ptot * r.ptot - r.pperp * r.pperp), r.pb);
    cnt(a, MCS_IC_PSP_CLAMP);
  } else {
    r.pperp = __builtin_sqrt(r.ptot * r.ptot - r.pb * r.pb);
  }
  r.gam = mcsm::hypot1(r.ptot / mc);
  phi_p = mcsm::atan2(fy, -fx * bsin + fz * bcos);
  r.phi = phi_p - HALFPI_;
  return r;
}

// The slowly varying part of scattering (src/scattering.jl:39-60): gyro period, r_g,tot and the
// cone cos_max -- functions of (ptot_pf, gam_pf, gyro_denom, xn_per) -- plus the refined 1/ptot_pf.
__device__ __forceinline__ void refresh_scatter(CK* a, Pt& p, double aa, double mc, double eta) {
  double grt;
  if (aa < 1 && p.ptot_pf < a->P.pe_crit) {
    grt = a->P.pe_crit * CC_ * p.gyro_denom;
    p.gyro_period = TWOPI_ * a->P.game_crit * mc * p.gyro_denom;
  } else {
    grt = p.ptot_pf * CC_ * p.gyro_denom;
    p.gyro_period = TWOPI_ * p.gam_pf * mc * p.gyro_denom;
  }
  const double vp_tg = TWOPI_ * grt;
  const double lam = eta * grt;
  p.cm_val = mcsm::cos(__builtin_sqrt(6 * vp_tg / (p.xn_per * lam)));
  p.rp_val = rcp_refined(p.ptot_pf);
}

// The same statements for the in-line loss of the LOSSY kernel: the constants handed in (they sit in registers; `a->P.x` in the
// loop is an s_load + wait), the division, the square root and the cosine in the forms of the common pass -- fdiv (the quotient
// of the compiler's own sequence without the range rescale), the square root without its class test, sincos with the
// coefficients from the table in VGPRs: bit-identical results (see rcp_refined, mcsm::sqrt_nn_, mcsm::sincos_t)
__device__ __forceinline__ void refresh_scatter_k(Pt& p, const mcsm::HotCoef& kc, double aa, double mc, double eta, double pe_crit, double game_crit) {
  double grt;
  if (aa < 1 && p.ptot_pf < pe_crit) {
    grt = pe_crit * CC_ * p.gyro_denom;
    p.gyro_period = TWOPI_ * game_crit * mc * p.gyro_denom;
  } else {
    grt = p.ptot_pf * CC_ * p.gyro_denom;
    p.gyro_period = TWOPI_ * p.gam_pf * mc * p.gyro_denom;
  }
  const double vp_tg = TWOPI_ * grt;
  const double lam = eta * grt;
  double sn, cs;
  mcsm::sincos_t(mcsm::sqrt_nn_(fdiv(6 * vp_tg, p.xn_per * lam)), &sn, &cs, kc);
  p.cm_val = cs;
  p.rp_val = rcp_refined(p.ptot_pf);
}

// The part of a scatter that depends on nothing but the particle's random stream: the two draws of block `blk` of the
// stream (k0, k1) (src/scattering.jl:68,71) and sin / cos of the azimuth phi_scat = 2 pi U2 - pi (:71).  A function of
// (key, draw index) alone, so it can be evaluated ahead of time and by any lane (see the tail ring in transport_body).
__device__ __forceinline__ void scatter_draws(uint32_t k0, uint32_t k1, uint32_t blk, const mcsm::HotCoef& kc, double& U1,
                                              double& s_ps, double& c_ps) {
  uint32_t o0, o1, o2, o3;
  philox_block(blk, 0u, 0u, 0u, k0, k1, o0, o1, o2, o3);
  U1 = u64_to_unit(o0, o1);
  const double U2 = u64_to_unit(o2, o3);
  const double phi_scat = U2 * TWOPI_ - PI_;
  mcsm::sincos_t(phi_scat, &s_ps, &c_ps, kc);
}

// src/scattering.jl:61-101: the per-step part (new pitch, phase adjustment) given the values that depend on the draws and on
// cos_max alone -- cos and sin of the deflection, cos of the azimuth, sin(azimuth) * sin(deflection): scatter_cone() --;
// straight-line code
__device__ __forceinline__ void scatter_cone(double U1, double s_ps, double omc, double& cos_d, double& sin_d, double& ssd) {
  cos_d = 1 - U1 * omc;                                              // == 1 - U1 * (1 - cos_max), scattering.jl:68
  sin_d = FSQRT(1 - cos_d * cos_d);
  ssd = s_ps * sin_d;                                                // the numerator of get_sine_adjustment (scattering.jl:94)
}
__device__ __forceinline__ void scattering_rest(Pt& p, const mcsm::HotCoef& kc, double cos_d, double sin_d, double c_ps, double ssd) {
  const double cos_old = div_r(p.pb_pf, p.ptot_pf, p.rp_val);      // == pb_pf / ptot_pf
  const double sin_old = div_r(p.p_perp, p.ptot_pf, p.rp_val);     // == p_perp / ptot_pf
  const double cos_new = cos_old * cos_d + sin_old * sin_d * c_ps;
  const double arg = __builtin_fmax(1 - cos_new * cos_new, 0.0);    // deviation D3: a rounding excess of cos_new^2 over 1 gives 0, not a DomainError
  const double sin_new = FSQRT(arg);
  p.pb_pf = p.ptot_pf * cos_new;
  p.p_perp = p.ptot_pf * sin_new;
  const double phi_p_old = p.phi + HALFPI_;
  // get_sine_adjustment (scattering.jl:93-101), evaluated unconditionally and selected: for
  // sin_new == 0 the quotient is inf/NaN and is discarded.
  double sd = fdiv(ssd, sin_new);
  // |sd| > SIN_UL ? copysign(SIN_UL, sd) : sd   as max / min (a NaN -- sin_new == 0 -- is discarded below either way)
  sd = __builtin_fmin(__builtin_fmax(sd, -SIN_UL), SIN_UL);
  const double adj = mcsm::asin_t(sd, kc);
  const double phi_p_new = sin_new != 0 ? phi_p_old + adj : phi_p_old;
  p.phi = phi_p_new - HALFPI_;
}
__device__ __forceinline__ void scattering_with(Pt& p, const mcsm::HotCoef& kc, double U1, double s_ps, double c_ps) {
  double cos_d, sin_d, ssd;
  scatter_cone(U1, s_ps, 1 - p.cm_val, cos_d, sin_d, ssd);
  scattering_rest(p, kc, cos_d, sin_d, c_ps, ssd);
}
// ---- the state-dependent half of a pass with its 64-bit literals handed in (tail loop).  The build keeps literals out of registers
// (MCS_SC, -disable-machine-licm: each is rematerialised by two s_mov where it is used), which is right for the bulk of a launch --
// scalar instructions issue beside the other wave's vector ones -- and wrong for a wave alone on its SIMD, where every instruction
// costs an issue slot: 16 s_mov per pass.  The tail loop pins the seven values in scalar registers before its first pass (sconst)
// and runs these forms: the statements of sqrt_nn_, asin_t, mod2pi (include/mcs_math.h) and scattering_rest, operand for operand.
struct TailK { double halfpi, pio2_lo, twopi, twopi_lo, inv_twopi, sin_ul, dmin; };
__device__ __forceinline__ double sqrt_nn_k(double x, const TailK& K) {
  const double y = __builtin_amdgcn_rsq(__builtin_fmax(x, K.dmin));
  double g = x * y;
  double hh = 0.5 * y;
  const double r = __builtin_fma(-hh, g, 0.5);
  g = __builtin_fma(g, r, g);
  hh = __builtin_fma(hh, r, hh);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, hh, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, hh, g);
  return g;
}
__device__ __forceinline__ double asin_tk(double x, const mcsm::HotCoef& k, const TailK& K) {
  const double ax = __builtin_fabs(x);
  const bool small = ax < 0.5;
  const double z = small ? x * x : (1.0 - ax) * 0.5;
  const double sv = small ? ax : sqrt_nn_k(z, K);
  double q = k.A12;
  q = __builtin_fma(q, z, k.A11); q = __builtin_fma(q, z, k.A10); q = __builtin_fma(q, z, k.A9); q = __builtin_fma(q, z, k.A8); q = __builtin_fma(q, z, k.A7);
  q = __builtin_fma(q, z, k.A6); q = __builtin_fma(q, z, k.A5); q = __builtin_fma(q, z, k.A4); q = __builtin_fma(q, z, k.A3); q = __builtin_fma(q, z, k.A2);
  q = __builtin_fma(q, z, k.A1); q = __builtin_fma(q, z, k.A0);
  const double t = __builtin_fma(sv * z, q, sv);
  const double big = K.halfpi - (2.0 * t - K.pio2_lo);
  return __builtin_copysign(small ? t : big, x);
}
__device__ __forceinline__ double mod2pi_k(double x, const TailK& K) {
  const double k = __builtin_floor(x * K.inv_twopi);
  double r = __builtin_fma(-k, K.twopi, x);
  r = __builtin_fma(-k, K.twopi_lo, r);
  r = r < 0.0 ? r + K.twopi : r;
  r = r >= K.twopi ? r - K.twopi : r;
  return r;
}
__device__ __forceinline__ void scattering_rest_k(Pt& p, const mcsm::HotCoef& kc, const TailK& K, double cos_d, double sin_d, double c_ps, double ssd) {
  const double cos_old = div_r(p.pb_pf, p.ptot_pf, p.rp_val);
  const double sin_old = div_r(p.p_perp, p.ptot_pf, p.rp_val);
  const double cos_new = cos_old * cos_d + sin_old * sin_d * c_ps;
  const double arg = __builtin_fmax(1 - cos_new * cos_new, 0.0);
  const double sin_new = sqrt_nn_k(arg, K);
  p.pb_pf = p.ptot_pf * cos_new;
  p.p_perp = p.ptot_pf * sin_new;
  const double phi_p_old = p.phi + K.halfpi;
  double sd = fdiv(ssd, sin_new);
  sd = __builtin_fmin(__builtin_fmax(sd, -K.sin_ul), K.sin_ul);
  const double adj = asin_tk(sd, kc, K);
  const double phi_p_new = sin_new != 0 ? phi_p_old + adj : phi_p_old;
  p.phi = phi_p_new - K.halfpi;
}
// the two draws of a scatter are indices n, n+1 = one Philox block
__device__ __forceinline__ void scattering(Rng& rng, Pt& p, const mcsm::HotCoef& kc) {
  double U1, s_ps, c_ps;
  const uint32_t j = rng.n;
  rng.n = j + 2u;
  scatter_draws(rng.k0, rng.k1, j >> 1, kc, U1, s_ps, c_ps);
  scattering_with(p, kc, U1, s_ps, c_ps);
}

// src/particle_loop.jl:639-650
__device__ __forceinline__ double perpendicular_momentum(CK* a, double ptot, double pb) {
  if (ptot < __builtin_fabs(pb)) { cnt(a, MCS_IC_PPERP_CLAMP); return 1.0e-6 * ptot; }
  return __builtin_sqrt(ptot * ptot - pb * pb);
}

// src/particle_loop.jl:578-592
__device__ __forceinline__ double radiation_loss(double B2, double pp, double dt) {
  const double dlnp = MCS_RAD_LOSS_FAC * B2 * pp * dt;
  if (dlnp > 1.0e-2) pp /= 1 + dlnp; else pp *= 1 - dlnp;
  return pp;
}

// src/cuts.jl:149-162 plus the caller's `tcut_curr += 1` (particle_loop.jl:352-358, prob_return.jl:297-304)
__device__ MCS_COLD void tcut_track(CK* a, int tcut_curr, double weight, double ptot_pf) {
  const int ion = a->i_ion - 1;
  ladd_f64(&S_wc[tcut_curr - 1], weight);
  const int i_pt = bin_momentum(a, ptot_pf);
  tadd(a, a->L.spectra_coupled + i_pt + (long long)(MCS_PSD_MAX + 1) * ((tcut_curr - 1) + (long long)MCS_NA_C * ion), weight);
}
// Tally part of all_flux! (src/all_flux.jl:84-161: transform, calculate_x_spec_spectra!,
// F_stream!, FEB tracker), entered only when the zone changed (or i_grid <= i_grid_feb,
// or x_spec detectors exist).  By-value arguments: nothing of the caller is forced to memory.
__device__ MCS_COLD void flux_tally(CK* a, double pb_pf, double p_perp, double ptot_pf, double gam_pf,
                                        double phi, double weight, double x, double x_old, int i_grid, int i_grid_old,
                                        int ig3, bool inj) {
  const auto& P = a->P;
  const double aa = a->aa;
  const double ux = S_ux[ig3], gsf = S_gsf[ig3], bcos = S_bcos[ig3], bsin = S_bsin[ig3];
  double ptot_sk, px, py, pz, gam_sk;
  transform_p_PS(aa, pb_pf, p_perp, gam_pf, phi, ux, gsf, bcos, bsin, ptot_sk, px, py, pz, gam_sk);
  const double m = aa * MP_;
  double pt_o_px_sk, abs_inv_vx;
  if (ptot_sk > __builtin_fabs(px * MCS_SPIKE_AWAY)) {
    pt_o_px_sk = MCS_SPIKE_AWAY;
    abs_inv_vx = __builtin_fabs(MCS_SPIKE_AWAY / ux);
  } else {
    pt_o_px_sk = ptot_sk / px;
    abs_inv_vx = __builtin_fabs(gam_sk * aa * MP_ / px);
  }
  double eadd;
  if ((gam_sk - 1) > MCS_E_REL_PT) eadd = (gam_sk - 1) * m * (CC_ * CC_) * weight;
  else eadd = ptot_sk * ptot_sk / (2 * m) * weight;

  const int n_xspec = a->tb.n_xspec;
  if (n_xspec > 0) {   // all_flux.jl:164-190
    double pt_o_px_pf = __builtin_fabs(ptot_pf / pb_pf);
    if (!(pt_o_px_pf < MCS_SPIKE_AWAY)) pt_o_px_pf = MCS_SPIKE_AWAY;
    const int i_pt = bin_momentum(a, ptot_sk);
    const int i_pt_pf = bin_momentum(a, ptot_pf);
    for (int i = 0; i < n_xspec; ++i) {
      const double xs = a->tb.x_spec[i];
      if ((x_old < xs && x >= xs) || (x <= xs && x_old > xs)) {
        tadd(a, a->L.spectra_sf + i_pt + (long long)(MCS_PSD_MAX + 1) * i, weight * pt_o_px_sk);
        const double Fw = __builtin_fabs(pb_pf / px) * (gam_sk / gam_pf);
        tadd(a, a->L.spectra_pf + i_pt_pf + (long long)(MCS_PSD_MAX + 1) * i, weight * pt_o_px_pf * Fw);
      }
    }
  }

  // F_stream! (all_flux.jl:197-259)
  const bool down = x > x_old;
  const int i_first = down ? i_grid_old + 1 : i_grid_old;
  const int i_last = down ? i_grid : i_grid + 1;
  const int step = down ? 1 : -1;
  const int sign_fac = down ? 1 : -1;
  const bool inj_check = !down;
  int i_pt = 0, jth = 0;
  if (inj) { i_pt = bin_momentum(a, ptot_sk); jth = bin_angle(a, px, ptot_sk); }
  const double f_pxx = sign_fac * px * weight * P.gam0 * P.u0;
  const double f_pxz = __builtin_fabs(pz) * weight * P.gam0 * P.u0;
  const double f_en = sign_fac * eadd * P.gam0 * P.u0;
  const double tw = weight * abs_inv_vx;
  int k_sf = 0, j_sf = 0;
  bool have_sf = false;
  for (int i = i_first; down ? i <= i_last : i >= i_last; i += step) {
    if (inj_check && inj && i <= P.i_grid_feb) continue;
    ladd_f64(&S_fl[i - 1], f_pxx);
    ladd_f64(&S_fl[MCS_MAXNE + i - 1], f_pxz);
    ladd_f64(&S_fl[2 * MCS_MAXNE + i - 1], f_en);
    if (inj) {
      tadd(a, a->L.psd + i_pt + a->L.psd_stride_tht * jth + a->L.psd_stride_zone * (long long)(i - 1), tw);
    } else {
      if (P.track_thermal) {   // A9: bin the thermal crossing instead of appending to a list
        if (!have_sf) { k_sf = bin_momentum(a, ptot_sk); j_sf = bin_angle(a, px, ptot_sk); have_sf = true; }
        tadd(a, a->L.therm_sf + k_sf + a->L.psd_stride_tht * j_sf + a->L.psd_stride_zone * (long long)(i - 1), tw);
        const double gam = S_gsf[i];
        const double beta = S_ux[i] / CC_;
        const double E0 = a->m * CC_ * CC_;
        const double pc = ptot_sk * CC_;
        const double etot = __builtin_sqrt(pc * pc + E0 * E0);
        double px_Xf = gam * (px - beta * etot / CC_);
        const double pt_Xf = __builtin_sqrt((ptot_sk * ptot_sk - px * px) + px_Xf * px_Xf);
        if (__builtin_fabs(px_Xf) > pt_Xf) px_Xf = __builtin_copysign(pt_Xf, px_Xf);
        const int k_pf = bin_momentum(a, pt_Xf);
        const int j_pf = bin_angle(a, px_Xf, pt_Xf);
        tadd(a, a->L.therm_pf + k_pf + a->L.psd_stride_tht * j_pf + a->L.psd_stride_zone * (long long)(i - 1), tw);
      }
      ladd_i32(&S_nc[i - 1], 1);
    }
  }
  if (inj && x < P.feb_upstream && x_old >= P.feb_upstream) {
    sadd(3, eadd * P.gam0 * P.u0);
    sadd(2, -(px * weight * P.gam0 * P.u0));
  }
}

// state that retro_time reads and writes, passed and returned by value (registers)
struct Retro {
  double ptot, pb, pperp, gam, phi, gyro_denom, acctime, tcut_next;
  int tcut, n_retro;
  uint32_t rng_n;
  bool lose_pt;
  bool capped;       // the walk was ended by MCS_RETRO_CAP (not in the reference, whose loop is uncapped)
};

// src/prob_return.jl:217-344 (with D1: the scattered pitch is kept)
__device__ MCS_COLD Retro retro_time(CK* a, Retro r, double prp, double weight, uint32_t k0, uint32_t k1) {
  const auto& P = a->P;
  const int ng = P.n_grid;
  const double aa = a->aa;
  Rng rng; rng.k0 = k0; rng.k1 = k1; rng.n = r.rng_n;
  const double xn_per = MCS_RETRO_XN_PER;
  const double phi_step = TWOPI_ / xn_per;
  const double t_step_fac = TWOPI_ * aa * MP_ * CC_ * r.gyro_denom / xn_per;
  const double ux_sk = -S_ux[ng];
  const double gsf = S_gsf[ng];
  const double gef = S_gef[ng];
  double B = S_bt[ng];
  if (P.use_custom_epsB) B *= __builtin_sqrt(P.x_grid_stop / prp);
  const double bcos = S_bcos[ng], bsin = S_bsin[ng];
  const double B_CMB_loc = P.B_CMBz * gef;
  double B2_tot = B * B + B_CMB_loc * B_CMB_loc;
  r.lose_pt = false;
  r.capped = false;
  const int n_retro_0 = r.n_retro;
  const int retro_cap = a->retro_cap;
  double x_PT = prp;
  r.phi = rng.rand() * TWOPI_;
  const int n_tcuts = a->tb.n_tcuts;
  const double mc = aa * MP_ * CC_;
  while (true) {
    ++r.n_retro;
#ifdef MCS_PROF_TAIL
    if (S_ttgate[threadIdx.x >> 6]) atomicAdd(&S_prof[39], 1ull);
#endif
    const double x_PT_old = x_PT;
    const double phi_old = r.phi;
    if (P.use_custom_epsB) {
      B = S_bt[ng] * __builtin_sqrt(P.x_grid_stop / x_PT);
      B2_tot = B * B + B_CMB_loc * B_CMB_loc;
      r.gyro_denom = 1 / (a->zzq * B);
    }
    const double gyro_rad = r.pperp * CC_ * r.gyro_denom;
    r.phi = mcsm::mod2pi(phi_old + phi_step);
    const double t_step = t_step_fac * r.gam;
    const double x_move = r.pb * t_step_fac / (aa * MP_);
    double gyr = 0.0;
    if (bsin != 0.0) gyr = gyro_rad * bsin * (mcsm::cos(r.phi) - mcsm::cos(phi_old));
    x_PT = x_PT_old + gsf * (x_move * bcos - gyr + ux_sk * t_step);
    r.acctime += t_step * gef;
    if (P.do_tcuts) {
      if (r.tcut > n_tcuts) cnt(a, MCS_IC_TCUT_OVERRUN);   // D4
      else if (r.acctime >= r.tcut_next) {
        tcut_track(a, r.tcut, weight, r.ptot);
        r.tcut += 1;
        r.tcut_next = r.tcut <= n_tcuts ? S_tc[r.tcut - 1] : __builtin_inf();
      }
    }
    r.phi = TWOPI_ * rng.rand();
    const double ptot_old = r.ptot;
    r.pb = (2 * rng.rand() - 1) * r.ptot;
    double arg = r.ptot * r.ptot - r.pb * r.pb;
    if (arg < 0) arg = 0;
    r.pperp = __builtin_sqrt(arg);
    const double cos_new = r.pb / ptot_old;
    const double sin_new = r.pperp / ptot_old;
    if (P.do_rad_losses && aa < 1) r.ptot = radiation_loss(B2_tot, r.ptot, t_step);
    if (r.ptot <= 0) {
      r.ptot = MCS_FLOOR; r.gam = 1.0; r.lose_pt = true;
      break;
    } else {
      r.pb = r.ptot * cos_new;
      r.pperp = r.ptot * sin_new;
      r.gam = mcsm::hypot1(r.ptot / mc);
    }
    if (x_PT < prp) break;
    if (r.n_retro - n_retro_0 >= retro_cap) { r.capped = true; cnt(a, MCS_IC_RETRO_CAP); break; }   // see MCS_RETRO_CAP (include/mcs.h)
  }
  r.rng_n = rng.n;
  return r;
}

__device__ __forceinline__ double vconst(double x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int vconsti(int x) { asm volatile("" : "+v"(x)); return x; }

// hot-loop constants, fetched once per wave and parked in VGPRs (see the kernel prologue)
struct Hot {
  double m, x_grid_stop, xn_coarse;     // used by every pass: parked in VGPRs (vconst)
  // Needed by rare code only: loaded once per wave into SCALAR registers behind an opaque move (sconst).
  // The allocator keeps them in SGPRs or in lanes of a spill VGPR (v_readlane at the rare use) -- a few
  // cycles either way, where re-loading from the constant buffer is an s_load + wait of ~200 cycles per
  // use (measured: it tripled the cost of the rare region).
  double u2, eta, zzq, mc, feb_down, pcut, pmax_cutoff, feb_up, age_max, inj_frac, xn_fine, aa;
  int n_grid, i_grid_feb, n_tcuts, n_xspec;
  bool custom_epsB, etf, dont_scatter, rad_losses, do_tcuts, dont_DSA;   // wave-uniform flags (scalar branches)
  bool oblique;     // some zone has b_sin != 0 (the gyro term of the move is not identically zero)
  bool odd_cfg;     // downstream FEB, no-DSA / injection probability, electrons or x_spec detectors: extra per-pass tests
  bool every_pass;  // custom eps_B, radiative losses of electrons, no-scatter runs: slow_pre has work in every pass
};
__device__ __forceinline__ double sconst(double x) { asm volatile("" : "+s"(x)); return x; }
__device__ __forceinline__ int sconsti(int x) { asm volatile("" : "+s"(x)); return x; }


// src/particle_loop.jl:652-723
__device__ MCS_COLD Mom do_energy_transfer(CK* a, int i_grid, int i_grid_old, double weight, Mom r) {
  const auto& P = a->P;
  const int i_start = i_grid_old;
  const int i_stop = i_grid < P.i_shock ? i_grid : P.i_shock;
  bool scale = false;
  const double m = a->aa * MP_;
  const double E0 = m * (CC_ * CC_);
  double gam_f = r.gam;
  double eps_max = -1e300, recv_max = 0.0;
  for (int i = i_start + 1; i <= i_stop; ++i) {
    if (i < 1 || i > P.n_grid) continue;
    const double e = a->tb.eps_target[i - 1];
    if (e > eps_max) eps_max = e;
    const double v = a->T[a->L.energy_recv_pool + (i - 1)];
    if (v > recv_max) recv_max = v;
  }
  if (a->aa >= 1 && eps_max > 0) {
    const double gam_i = mcsm::hypot1(r.ptot / a->mc);
    const double eps_stop = a->tb.eps_target[i_stop - 1];
    const double eps_start = i_start >= 1 ? a->tb.eps_target[i_start - 1] : 0.0;
    gam_f = 1 + (gam_i - 1) * (1 - eps_stop) / (1 - eps_start);
    int n_split = 0;
    for (int i = i_start + 1; i <= i_stop; ++i) if (a->tb.eps_target[i - 1] > 0) ++n_split;
    const double inc = (gam_i - gam_f) * E0 * weight / n_split;
    for (int i = i_start + 1; i <= i_stop; ++i)
      if (a->tb.eps_target[i - 1] > 0) tadd(a, a->L.energy_transfer_pool + (i - 1), inc);
    scale = true;
  } else if (recv_max > 0) {
    double sum = 0.0;
    for (int i = i_start + 1; i <= i_stop; ++i) sum += a->T[a->L.energy_recv_pool + (i - 1)];
    const double e_tr = sum * a->ewf;
    const double gam_i = mcsm::hypot1(r.ptot / a->mc);
    gam_f = gam_i + e_tr / E0;
    scale = true;
  }
  if (scale) {
    const double ptot_f = a->mc * __builtin_sqrt(gam_f * gam_f - 1);
    const double sf = ptot_f / r.ptot;
    r.pb *= sf;
    r.pperp *= sf;
    r.ptot = ptot_f;
    r.gam = gam_f;
  }
  return r;
}

// src/particle_finish.jl:46-107 (with D2).  Zone properties of zone ig3.
// `off`/`val` (optional): the escape-spectrum tally is handed back instead of being added, so that the caller can
// combine the lanes of a wave that hit the same bin (see drain_events)
__device__ MCS_COLD void particle_finish(CK* a, int i_reason, double pb_pf, double p_perp, double gam_pf,
                                             double phi, double weight, int ig3, long long* off = nullptr, double* val = nullptr) {
  const double aa = a->aa;
  const double m = aa * MP_;
  const double E0 = m * (CC_ * CC_);
  double ptot_sk, px, py, pz, gam_sk;
  transform_p_PS(aa, pb_pf, p_perp, gam_pf, phi, S_ux[ig3], S_gsf[ig3], S_bcos[ig3], S_bsin[ig3], ptot_sk, px, py, pz, gam_sk);
  const int ip = bin_momentum(a, ptot_sk);
  const int jth = bin_angle(a, px, ptot_sk);
  double wf;
  if (ptot_sk > __builtin_fabs(MCS_SPIKE_AWAY * px)) wf = gam_sk * m * MCS_SPIKE_AWAY / ptot_sk;
  else wf = gam_sk * (m / __builtin_fabs(px));
  const long long pm = MCS_PSD_MAX + 1;
  if (i_reason == 1) {
    if (off) { *off = a->L.esc_psd_down + ip + pm * jth; *val = weight * wf; }
    else tadd(a, a->L.esc_psd_down + ip + pm * jth, weight * wf);
  } else if (i_reason == 2) {
    sadd(4, weight);
    if (off) { *off = a->L.esc_psd_up + ip + pm * jth; *val = weight * wf; }
    else tadd(a, a->L.esc_psd_up + ip + pm * jth, weight * wf);
    const bool rel = (gam_sk - 1) >= MCS_E_REL_PT;
    const double E_kin = rel ? (gam_sk - 1) * E0 : ptot_sk * ptot_sk / (2 * m);
    const double eadd = E_kin * weight;
    sadd(5, __builtin_fabs(px) * weight);
    sadd(6, eadd);
    ladd_f64(&S_eff[0][ip], eadd);
    ladd_f64(&S_eff[1][ip], weight);
  }
}

// tcuts[tcut-1] or +inf past the last cut (D4)
__device__ __forceinline__ double tcut_next_of(CK* a, const Hot& h, int tcut) {
  return (h.do_tcuts && tcut <= h.n_tcuts) ? S_tc[tcut - 1] : __builtin_inf();
}
// The clock compares acctime with ONE number: the earlier of the next time cut and age_max.
__device__ __forceinline__ void refresh_time(CK* a, const Hot& h, Pt& p) {
  double t = tcut_next_of(a, h, p.tcut);
  if (h.age_max > 0 && h.age_max < t) t = h.age_max;
  p.t_ev = t;
  p.ovr_inc = (h.do_tcuts && p.downstream && p.tcut > h.n_tcuts) ? 1 : 0;
}

// src/prob_return.jl:36-173, entered only when it has something to do (the caller has
// already set i_return = 2 and filtered the no-op cases).
__device__ __forceinline__ void prob_return_events(CK* a, const Hot& h, Rng& rng, Pt& p, int& i_return, bool& lose_pt, bool& capped) {
  const auto& P = a->P;
  const double aa = h.aa, u2 = h.u2, eta = h.eta, x_grid_stop = h.x_grid_stop;
  if (p.x < x_grid_stop) {
  } else if (p.x_old < x_grid_stop && x_grid_stop <= p.x) {
    double gyro_tmp;
    if (h.custom_epsB && p.x > x_grid_stop) gyro_tmp = __builtin_sqrt(x_grid_stop / p.x); else gyro_tmp = 1.0;
    const double grt = p.ptot_pf * CC_ * gyro_tmp / (MCS_QCGS * P.bmag2);
    const double L_diff = eta / 3 * grt * p.ptot_pf / (aa * MP_ * p.gam_pf * u2);
    p.prp = p.x + 3 * L_diff;
    TTG_COUNT(56);
  } else if (p.x_old < p.prp && p.x >= p.prp) {
    const double vt = p.ptot_pf / (p.gam_pf * aa * MP_);
    const double q = (vt - u2) / (vt + u2);
    const double prob_ret = q * q;
    if (vt < u2 || rng.rand() > prob_ret) {
      i_return = 0;
      TTG_COUNT(57);
    } else {
      i_return = 1;
      TTG_COUNT(58);
      Retro r;
      r.ptot = p.ptot_pf; r.pb = p.pb_pf; r.pperp = p.p_perp; r.gam = p.gam_pf; r.phi = p.phi;
      r.gyro_denom = p.gyro_denom; r.acctime = p.acctime; r.tcut_next = tcut_next_of(a, h, p.tcut); r.tcut = p.tcut;
      r.n_retro = p.n_retro; r.rng_n = rng.n; r.lose_pt = false; r.capped = false;
      r = retro_time(a, r, p.prp, p.weight, rng.k0, rng.k1);
      p.ptot_pf = r.ptot; p.pb_pf = r.pb; p.p_perp = r.pperp; p.gam_pf = r.gam; p.phi = r.phi;
      p.gyro_denom = r.gyro_denom; p.acctime = r.acctime; p.tcut = r.tcut;
      p.n_retro = r.n_retro; rng.n = r.rng_n; lose_pt = r.lose_pt; capped = r.capped;
      p.flags |= F_RS | F_RM;          // radiative losses inside the walk change ptot_pf / gam_pf
      if (lose_pt | capped) i_return = 0;
      p.x = p.prp;
    }
  } else {
    if (aa < 1 && p.ptot_pf < a->pcut_prev && p.helix % 1000 == 0) {
      const double grt = p.ptot_pf * CC_ * p.gyro_denom;
      const double L_diff = eta / 3 * grt * p.ptot_pf / (aa * MP_ * p.gam_pf * u2);
      if (p.x > 2.0e3 * L_diff) {
        p.prp = 0.8 * p.x;
      } else {
        const double r = a->pcut_prev / p.ptot_pf;
        const double r2 = r * r;
        const double alt = x_grid_stop + L_diff * (r2 * r2 * r);
        p.prp = p.prp < alt ? p.prp : alt;
      }
    }
  }
}

// (the explicit s_waitcnt lgkmcnt(0) keeps the wait for these LDS reads here, in rare code; otherwise the
// compiler waits at the first use, which is in the common pass -- four s_waitcnt per pass for nothing)
__device__ __forceinline__ void load_zone_props(Pt& p) {
  const int z = p.ig3;
  p.z_gsf = S_gsf[z]; p.z_bcos = S_bcos[z]; p.z_ux = S_ux[z]; p.z_gef = S_gef[z];
  __builtin_amdgcn_s_waitcnt(0xC07F);
}
__device__ __forceinline__ void load_zone_edges(Pt& p) {
  p.z_lo = S_x[p.i_grid]; p.z_hi = S_x[p.i_grid + 1];
  __builtin_amdgcn_s_waitcnt(0xC07F);
}

// load a particle and run the prologue of particle_loop (src/particle_loop.jl:44-153)
__device__ __forceinline__ void load_particle(CK* a, const Hot& h, long long k, Pt& p, Rng& rng) {
  p.weight = a->in.weight[k];
  p.ptot_pf = a->in.ptot_pf[k];
  p.pb_pf = a->in.pb_pf[k];
  p.x = a->in.x_PT_cm[k];
  p.xn_per = a->in.xn_per[k];
  p.prp = a->in.prp_x_cm[k];
  p.acctime = a->in.acctime_sec[k];
  p.phi = a->in.phi_rad[k];
  const uint32_t meta = a->in.meta[k];
  p.i_grid = (int)(meta & 0xffffu);
  p.tcut = (int)((meta >> 16) & 0xffu);
  p.downstream = (meta >> 24) & 1u;
  p.inj = (meta >> 25) & 1u;
  p.i_grid_old = p.i_grid;
  p.ig3 = p.i_grid;
  load_zone_props(p); load_zone_edges(p);
  p.helix = 0; p.n_retro = 0;
  const long long gi = a->gidx ? a->gidx[k] : a->i_prt_offset + k * a->i_prt_stride;     // global 0-based index
  const unsigned long long key = a->seed_base + (unsigned long long)(gi + 1);
  rng.init(key);

  p.gam_pf = mcsm::hypot1(p.ptot_pf / h.mc);
  p.gyro_denom = S_gd[p.i_grid];
  if (h.custom_epsB && p.x > h.x_grid_stop) p.gyro_denom *= __builtin_sqrt(p.x / h.x_grid_stop);
  p.gyro_rad_tot = p.ptot_pf * CC_ * p.gyro_denom;
  p.gyro_period = TWOPI_ * p.gam_pf * h.m * CC_ * p.gyro_denom;
  p.t_step = 0.0;
  p.dphi = TWOPI_ / p.xn_per;
  p.p_perp = perpendicular_momentum(a, p.ptot_pf, p.pb_pf);
  p.gyro_rad = p.p_perp * CC_ * p.gyro_denom;
  p.x_old = 0.0;
  p.cm_val = 0.0; p.rp_val = 0.0; p.rg_val = 0.0; p.x_dt = __builtin_inf();
  p.flags = F_RS | F_RM | F_NOPARK |   // (F_NOPARK: the new particle's first Code Blocks run now)
            ((p.downstream && !p.inj && p.x < 0) ? F_INJCHK : 0);
  p.n_ovr = 0u;
  refresh_time(a, h, p);
  p.npush = 0;
}

// The zone search of all_flux! (src/all_flux.jl:68-72): forward, `findnext(>(x), x_grid, i+1) - 1` -- the first
// j >= i_grid+1 with x_grid[j] > x, minus one; backward, `findprev(<=(x), x_grid, i)` -- the first j <= i_grid,
// going down, with x_grid[j] <= x; -1 when the scan runs off the grid.  `skip` leading candidates are known to
// fail.  A linear scan costs one LDS round trip (~170 cycles of a lone wave) per zone, and one move of an
// energetic particle crosses dozens of the thin zones near the shock: three candidates are probed in one round,
// the rest is a binary search (the grid is strictly increasing, checked by mcs_set_grid) -- same result.
__device__ __forceinline__ int zone_search(double x, bool fwd, int i_grid, int ne, int skip) {
  const int st = fwd ? 1 : -1;
  const int a0 = fwd ? i_grid + 1 + skip : i_grid - skip, a1 = a0 + st, a2 = a1 + st;
  const int c0 = a0 < 0 ? 0 : (a0 > ne - 1 ? ne - 1 : a0), c1 = a1 < 0 ? 0 : (a1 > ne - 1 ? ne - 1 : a1),
            c2 = a2 < 0 ? 0 : (a2 > ne - 1 ? ne - 1 : a2);
  const double v0 = S_x[c0], v1 = S_x[c1], v2 = S_x[c2];
  const bool h0 = fwd ? v0 > x : v0 <= x, h1 = fwd ? v1 > x : v1 <= x, h2 = fwd ? v2 > x : v2 <= x;
  if (a0 != c0) return -1;
  if (h0) return fwd ? a0 - 1 : a0;
  if (a1 != c1) return -1;
  if (h1) return fwd ? a1 - 1 : a1;
  if (a2 != c2) return -1;
  if (h2) return fwd ? a2 - 1 : a2;
  // first g in [lo, hi) with x_grid[g] > x
  int lo = fwd ? a2 + 1 : 0, hi = fwd ? ne : a2;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (S_x[mid] > x) hi = mid; else lo = mid + 1;
  }
  return (fwd && lo >= ne) ? -1 : lo - 1;
}

// Zone-crossing tallies do not feed back into the particle: the lane pushes a record on its wave's
// LDS stack (see S_evf) instead of tallying on the spot.  The stack cannot overflow: it is drained
// to < 64 at the top of every pass and one pass adds at most two records per lane -- one for the move
// of the previous common pass and one for a Code Block 1 move (a second PRP return cannot follow
// directly: the retro walk leaves the particle AT the PRP, and a return needs x_old < prp).
// `known_base` >= 0: the stack height is known (first push site of the pass: the register mirror), no LDS read.
__device__ __forceinline__ void push_record(Pt& p, int ig3, int known_base = -1, uint32_t tag = 0u) {
  const unsigned long long m_ev = __builtin_amdgcn_ballot_w64(true);   // lanes that are here now
  const unsigned wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
  const unsigned base = known_base >= 0 ? (unsigned)known_base : S_evcur[wv];
  const unsigned pos = base + (unsigned)__popcll(m_ev & ((1ull << ln) - 1ull));
  S_evf[wv][0][pos] = p.pb_pf; S_evf[wv][1][pos] = p.p_perp; S_evf[wv][2][pos] = p.ptot_pf; S_evf[wv][3][pos] = p.gam_pf;
  S_evf[wv][4][pos] = p.phi; S_evf[wv][5][pos] = p.weight; S_evf[wv][6][pos] = p.x; S_evf[wv][7][pos] = p.x_old;
  S_evu[wv][pos] = (uint32_t)p.i_grid | ((uint32_t)p.i_grid_old << 8) | ((uint32_t)ig3 << 16) | ((uint32_t)(p.inj ? 1u : 0u) << 24) | tag;
  if (ln == (unsigned)(__ffsll((long long)m_ev) - 1)) S_evcur[wv] = base + (unsigned)__popcll(m_ev);
  p.npush += 1;
}

// downstream_test (particle_loop.jl:595-637) ends a particle iff  x > feb_downstream (when set)  or
// (x > 1.1 prp  and  x > 6.91 L_diff).  x_dt is that threshold, min(feb, max(1.1 prp, 6.91 L_diff)): a
// function of (prp, ptot_pf, gam_pf, gyro_rad_tot, gyro_denom), refreshed whenever one of them changed.
__device__ __forceinline__ void refresh_dtest(CK* a, const Hot& h, Pt& p) {
  const double aa = h.aa, m = aa * MP_;
  double v_fac;
  if (aa < 1 && p.ptot_pf < a->P.pe_crit) {
    const double gyro_fac = a->P.pe_crit * CC_ * p.gyro_denom;
    v_fac = gyro_fac * a->P.pe_crit / (m * a->P.game_crit * h.u2);
  } else {
    v_fac = p.gyro_rad_tot * p.ptot_pf / (m * p.gam_pf * h.u2);
  }
  const double L_diff = h.eta / 3 * v_fac;
  const double t1 = 1.1 * p.prp, t2 = 6.91 * L_diff;
  double t = t1 > t2 ? t1 : t2;
  if (h.feb_down > 0 && h.feb_down < t) t = h.feb_down;
  p.x_dt = t;
}

__device__ __forceinline__ void refresh_dtest_k(const Hot& h, Pt& p, double pe_crit, double game_crit) {
  const double aa = h.aa, m = aa * MP_;
  double v_fac;
  if (aa < 1 && p.ptot_pf < pe_crit) {
    const double gyro_fac = pe_crit * CC_ * p.gyro_denom;
    v_fac = fdiv(gyro_fac * pe_crit, m * game_crit * h.u2);
  } else {
    v_fac = fdiv(p.gyro_rad_tot * p.ptot_pf, m * p.gam_pf * h.u2);
  }
  const double L_diff = h.eta / 3 * v_fac;
  const double t1 = 1.1 * p.prp, t2 = 6.91 * L_diff;
  double t = t1 > t2 ? t1 : t2;
  if (h.feb_down > 0 && h.feb_down < t) t = h.feb_down;
  p.x_dt = t;
}

// ------------------------------------------------------------------------------------------
// The helix loop (src/particle_loop.jl:154-499), organised for the hardware.
//
// Measured on gfx950 (tools/ubench/lat.hip): a lone wave issues one instruction per 4 cycles,
// dependent or not, and every conditional region costs ~36 cycles even when no lane takes it
// (v_cmp -> SALU -> exec round trip).  So the common pass is straight-line code, executed by all
// lanes, with every rare thing -- before or after the move -- funnelled through ONE region:
//
//   loop:  [rare(lane)  if the lane has an event pending or a flag set]     <- the only exec branch
//          fast_step    scatter, clock, move, event detection                 (selects only)
//
//   rare = slow_post   everything the last move triggered (time-cut tally, no-DSA reflection, shock
//                      crossing, zone search + tally record, downstream test, PRP logic, retro walk)
//          block1_step Code Block 1 passes after a PRP return (no scatter: done here, move included)
//          slow_pre    everything before the next scatter (helix cap, zone reload, frame transform,
//                      energy transfer, the exit tests, radiative losses, refresh of the slowly
//                      varying quantities, fine/coarse switch, the save for the next pcut)
//          finish      tallies / saved-array writes of a particle that ended
//
// The arithmetic and its order are those of the reference; only the control flow differs.
// End codes: 0 = saved for the next pcut, 1..4 = i_reason.
// ------------------------------------------------------------------------------------------

// Code Block 2, the move (particle_loop.jl:392-407, 510-538) and the detection of everything that
// needs slow_post.  Straight-line; `oblique` is wave-uniform (some zone has b_sin != 0).
__device__ __forceinline__ bool move_and_detect(CK* a, const Hot& h, Pt& p, double& phi_old_out, bool& ev_cross) {
  const int ig3 = p.ig3;
  const double gsf = p.z_gsf, bcos = p.z_bcos, ux = p.z_ux;
  p.x_old = p.x;
  const double phi_old = p.phi;
  phi_old_out = phi_old;
  p.phi = mcsm::mod2pi(p.phi + p.dphi);
  const double gm = p.gam_pf * h.m;
  const double x_move = div_r(p.pb_pf * p.t_step, gm, p.rg_val);   // == pb_pf * t_step / (gam_pf * m)
  double gyr = 0.0;   // gyro_rad*b_sin*(...) is exactly +-0 for a parallel field (b_sin == 0)
  if (h.oblique) {
    const double bsin = S_bsin[ig3];
    const double g = p.gyro_rad * bsin * (mcsm::cos(p.phi) - mcsm::cos(phi_old));
    gyr = bsin != 0.0 ? g : 0.0;
  }
  const double dx = gsf * (x_move * bcos - gyr + ux * p.t_step);
  p.x = p.x_old + dx;
  // same-zone test (all_flux.jl:64-82): one boundary compare in the common case
  const bool fwd = p.x > p.x_old;
  const bool same_zone = (fwd & (p.z_hi > p.x)) | (!fwd & (p.z_lo <= p.x));   // the zone's edges are in registers
  // upward crossings that mean work: the end of the grid, the PRP (prob_return.jl:73,89) and the point
  // beyond which downstream_test ends the particle (x_dt, see refresh_dtest).  Five compares and four
  // scalar ops, no branch (a nested select compiles to exec-masked moves behind a 36-cycle skeleton).
  const bool ev_up = ((p.x >= h.x_grid_stop) & ((p.x_old < h.x_grid_stop) | ((p.x_old < p.prp) & (p.x >= p.prp)))) | (p.x > p.x_dt);
  const bool ev_xn = (p.x > p.gyro_rad_tot) != (p.xn_per == h.xn_coarse);   // fine/coarse switch due
  ev_cross = !same_zone;
  bool ev = ev_up | ev_xn;
  if (h.odd_cfg) {   // wave-uniform: configurations with more per-pass conditions
    if (h.feb_down > 0) ev |= p.x > h.feb_down;
    if (h.dont_DSA || h.inj_frac < 1) ev |= p.x <= 0 && p.x_old > 0 && !p.inj;
    if (h.aa < 1) ev |= p.x >= h.x_grid_stop;
    if (h.n_xspec != 0) ev = true;
  }
  return ev;
}

// The position events of move_and_detect, folded into two thresholds.  Between two visits to the rare region
// nothing that those events depend on changes except the position itself (zone edges, PRP, x_dt, gyro_rad_tot,
// xn_per change in rare code only), so each of them is "x has come up to T" or "x has come down to T" for a T
// known when the lane leaves the region:
//   zone edge        fwd & x >= z_hi  |  !fwd & x < z_lo          =>  x >= z_hi | x <= z_lo
//   grid end, PRP    x_old < T <= x: T above x now: up; else the lane must come back when x drops to T, so that
//                    T becomes an upward threshold again (a visit with nothing due)
//   x_dt             x > x_dt (every pass while it holds: t_hi = -inf)
//   fine / coarse    fine: x > gyro_rad_tot; coarse: x <= gyro_rad_tot
// The compares are non-strict, a SUPERSET of the events: the rare region re-derives the exact conditions from
// (x, x_old) with move_and_detect's own expressions, and a lane that stopped with nothing due just carries on --
// nothing of a particle changes while it is masked out of the common pass.  (-DMCS_CHECK_THR poisons the weight
// of a particle whose exact test fires where the thresholds do not: tests then fail loudly.)
// NO_XN (LOSSY kernel): the fine / coarse switch is decided in line in every pass (gyro_rad_tot changes with the momentum in
// every pass there), so it is no threshold
template <bool NO_XN = false>
__device__ __forceinline__ void refresh_thr(const Hot& h, Pt& p) {
  const double x = p.x, inf = __builtin_inf();
  double hi = p.z_hi, lo = p.z_lo;
  const double T1 = h.x_grid_stop, T2 = p.prp, g = p.gyro_rad_tot;
  const bool b1 = x < T1, b2 = x < T2, coarse = p.xn_per == h.xn_coarse, xg = x > g;
  hi = (b1 & (T1 < hi)) ? T1 : hi;   lo = (!b1 & (T1 > lo)) ? T1 : lo;
  hi = (b2 & (T2 < hi)) ? T2 : hi;   lo = (!b2 & (T2 > lo)) ? T2 : lo;
  {
    // The zones up to i_grid_feb (the upstream free-escape boundary lies INSIDE zone i_grid_feb).  The reference gives them two
    // things in every pass: all_flux! does not return early there (all_flux.jl:74-82) -- which, while the particle stays in its
    // zone, tallies nothing (F_stream!'s zone range is empty) except the two escape scalars when an injected particle passes
    // the boundary (all_flux.jl:150-160) --, and Code Block 3 ends an injected particle beyond it (particle_loop.jl:266-275).
    // Both need `inj` and x < feb_upstream: one more downward threshold instead of a visit to the rare region in every pass
    // (`inj` changes in rare code only, where this is refreshed).
    const double T3 = h.feb_up;
    const bool near_feb = (p.i_grid <= h.i_grid_feb) & p.inj & (x >= T3);
    lo = (near_feb & (T3 > lo)) ? T3 : lo;
  }
  hi = p.x_dt < hi ? p.x_dt : hi;
  hi = x > p.x_dt ? -inf : hi;
  const double hi_f = xg ? -inf : (g < hi ? g : hi);       // fine steps: the switch is due once x > g
  const double lo_c = xg ? (g > lo ? g : lo) : inf;        // coarse steps: due once x <= g
  p.t_hi = NO_XN ? hi : (coarse ? hi : hi_f);
  p.t_lo = NO_XN ? lo : (coarse ? lo_c : lo);
  p.c_gef = p.downstream ? p.z_gef : 0.0;
  p.c_tev = p.downstream ? p.t_ev : inf;
}

// The common pass's version of move_and_detect: same move, the position events through the two thresholds.
// Returns the events that are not position thresholds (odd configurations); ev_cross = some threshold reached.
// (K: the tail loop's literals in registers, see TailK)
__device__ __forceinline__ bool move_and_detect_thr(CK* a, const Hot& h, Pt& p, double& phi_old_out, bool& ev_cross, const bool xn_is_threshold = true,
                                                    const TailK* K = nullptr) {
  const int ig3 = p.ig3;
  const double gsf = p.z_gsf, bcos = p.z_bcos, ux = p.z_ux;
  p.x_old = p.x;
  const double phi_old = p.phi;
  phi_old_out = phi_old;
  p.phi = K ? mod2pi_k(p.phi + p.dphi, *K) : mcsm::mod2pi(p.phi + p.dphi);
  const double gm = p.gam_pf * h.m;
  const double x_move = div_r(p.pb_pf * p.t_step, gm, p.rg_val);   // == pb_pf * t_step / (gam_pf * m)
  double gyr = 0.0;
  if (h.oblique) {
    const double bsin = S_bsin[ig3];
    const double g = p.gyro_rad * bsin * (mcsm::cos(p.phi) - mcsm::cos(phi_old));
    gyr = bsin != 0.0 ? g : 0.0;
  }
  const double dx = gsf * (x_move * bcos - gyr + ux * p.t_step);
  p.x = p.x_old + dx;
  ev_cross = (p.x >= p.t_hi) | (p.x <= p.t_lo);
#ifdef MCS_CHECK_THR
  {
    const bool fwd = p.x > p.x_old;
    const bool same_zone = (fwd & (p.z_hi > p.x)) | (!fwd & (p.z_lo <= p.x));
    const bool ev_up = ((p.x >= h.x_grid_stop) & ((p.x_old < h.x_grid_stop) | ((p.x_old < p.prp) & (p.x >= p.prp)))) | (p.x > p.x_dt);
    const bool ev_xn = xn_is_threshold & ((p.x > p.gyro_rad_tot) != (p.xn_per == h.xn_coarse));      // (LOSSY: decided in line)
    const bool ev_feb = (p.i_grid <= h.i_grid_feb) & p.inj & (p.x < h.feb_up);                          // (the upstream free-escape boundary)
    if ((!same_zone | ev_up | ev_xn | ev_feb) & !ev_cross) p.weight = __builtin_nan("");
  }
#endif
  bool ev = false;
  if (h.odd_cfg) {   // wave-uniform: configurations with more per-pass conditions
    if (h.feb_down > 0) ev |= p.x > h.feb_down;
    if (h.dont_DSA || h.inj_frac < 1) ev |= p.x <= 0 && p.x_old > 0 && !p.inj;
    if (h.aa < 1) ev |= p.x >= h.x_grid_stop;
    if (h.n_xspec != 0) ev = true;
  }
  return ev;
}

// Everything the last move triggered (the tail of Code Block 2 and Code Block 3's all_flux /
// downstream_test / prob_return part, particle_loop.jl:352-358, 409-499).  Returns the end code or -1.
// DIRECT (wave-specialised kernel): the crossing is tallied on the spot instead of being pushed as a record
template <bool DIRECT = false>
__device__ __forceinline__ int slow_post(CK* a, const Hot& h, Rng& rng, Pt& p, double phi_old) {
  const double aa = h.aa;
  const int ig3 = p.ig3;
  const int i_grid_before = p.i_grid;
  p.i_grid_old = p.i_grid;
  // time cut (cuts.jl:149-162; the clock has already run, weight and ptot_pf are those of the pass)
  if (h.do_tcuts && p.downstream && !(p.tcut > h.n_tcuts) && p.acctime >= tcut_next_of(a, h, p.tcut)) {
    tcut_track(a, p.tcut, p.weight, p.ptot_pf);
    p.tcut += 1;
  }
  TTG_MARK(40);
  const bool ev_reflect = p.x <= 0 && p.x_old > 0 && !p.inj && (h.dont_DSA || h.inj_frac < 1);
  const bool ev_shock = p.x_old < 0 && p.x >= 0;
  if (!ev_shock && !ev_reflect && p.downstream && p.x < 0) p.inj = true;   // particle_loop.jl:433-435
  if (ev_reflect) {
    // the retry loop of no_DSA_loop (particle_loop.jl:555-568); its first pass was the move
    const double m = aa * MP_;
    const double gsf = S_gsf[ig3], bcos = S_bcos[ig3], bsin = S_bsin[ig3], ux = S_ux[ig3];
    while (p.x <= 0 && p.x_old > 0 && !p.inj && (h.dont_DSA || h.inj_frac < 1)) {
      if (h.dont_DSA || (rng.rand() > h.inj_frac)) {
        if (p.pb_pf < 0) p.pb_pf = -p.pb_pf; else p.phi = rng.rand() * TWOPI_;
      } else break;
      p.phi = mcsm::mod2pi(p.phi + TWOPI_ / p.xn_per);
      const double x_move = p.pb_pf * p.t_step / (p.gam_pf * m);
      double gyr = 0.0;
      if (bsin != 0.0) gyr = p.gyro_rad * bsin * (mcsm::cos(p.phi) - mcsm::cos(phi_old));
      const double dx = gsf * (x_move * bcos - gyr + ux * p.t_step);
      p.x = p.x_old + dx;
    }
  }
  if (ev_shock || ev_reflect) {
    if (p.x_old < 0 && p.x >= 0) {          // particle_loop.jl:412-429
      p.downstream = true;
      const double L_diff = h.eta / 3 * p.gyro_rad_tot * p.ptot_pf / (h.m * p.gam_pf * h.u2);
      p.prp = p.prp > L_diff ? p.prp : L_diff;
      p.flags |= F_RS;                       // F_SAVE depends on `downstream`
    }
    if (p.downstream && p.x < 0) p.inj = true;
  }
  p.flags &= ~F_INJCHK;      // (either injected now, or at x >= 0, from where x < 0 is reached through the shock: an event)
  TTG_MARK(41);
  {
    // all_flux! (all_flux.jl:45-82): zone search; a tally record only when something was crossed
    const bool fwd = p.x > p.x_old;
    const bool same_zone = fwd ? (S_x[p.i_grid + 1] > p.x) : (S_x[p.i_grid] <= p.x);
    if (!same_zone || p.i_grid <= h.i_grid_feb || h.n_xspec != 0 || ev_reflect) {
      const int found = zone_search(p.x, fwd, p.i_grid, h.n_grid + 2, 0);
      if (found < 0) { cnt(a, MCS_IC_ZONE_FAIL); return 3; }   // D6
      p.i_grid = found;
      load_zone_edges(p);
      if (!(p.i_grid == p.i_grid_old && p.i_grid > h.i_grid_feb && h.n_xspec == 0)) {
        if constexpr (DIRECT) flux_tally(a, p.pb_pf, p.p_perp, p.ptot_pf, p.gam_pf, p.phi, p.weight, p.x, p.x_old, p.i_grid, p.i_grid_old, ig3, p.inj);
        else push_record(p, ig3);
      }
    }
  }
  TTG_MARK(42);
  // downstream_test (particle_loop.jl:595-637) and prob_return, from scratch
  int i_return = 2;                              // prob_return's default (prob_return.jl:48)
  bool do_prob_ret = true;
  if (h.feb_down > 0 && p.x > h.feb_down) {
    i_return = 0; do_prob_ret = false;
  } else if (p.x > 1.1 * p.prp) {
    const double m = aa * MP_;
    double v_fac;
    if (aa < 1 && p.ptot_pf < a->P.pe_crit) {
      const double gyro_fac = a->P.pe_crit * CC_ * p.gyro_denom;
      v_fac = gyro_fac * a->P.pe_crit / (m * a->P.game_crit * h.u2);
    } else {
      v_fac = p.gyro_rad_tot * p.ptot_pf / (m * p.gam_pf * h.u2);
    }
    const double L_diff = h.eta / 3 * v_fac;
    if (p.x > 6.91 * L_diff) { i_return = 0; do_prob_ret = false; TTG_COUNT(59); }
  }
  bool lose_pt = false, capped = false;
  TTG_MARK(43);
  if (do_prob_ret) prob_return_events(a, h, rng, p, i_return, lose_pt, capped);
  TTG_MARK(44);
  if (i_return == 0) {
    if (capped) return 3;      // MCS_RETRO_CAP: ends like an aged-out particle -- nothing added to the downstream sums (as the oracle)
    double vel = p.ptot_pf / h.m;
    if ((p.gam_pf - 1) >= MCS_E_REL_PT) vel /= p.gam_pf;
    sadd(0, p.ptot_pf / 3 * vel * p.weight * a->density);
    sadd(1, (p.gam_pf - 1) * h.m * (CC_ * CC_) * p.weight * a->density);
    return lose_pt ? 4 : 1;
  }
  // the particle goes on: what the next pass has to know
  int f = p.flags | F_CHECK;
  if (i_return == 1) f |= F_B1;
  if (p.i_grid != i_grid_before) f |= F_CROSSED; else f &= ~F_CROSSED;
  if (p.i_grid != p.ig3) f |= F_ZONE;
  p.flags = f;
  refresh_time(a, h, p);
  refresh_dtest(a, h, p);      // prp may have moved (shock crossing, PRP logic)
  TTG_MARK(45);
  return -1;
}

// t_step, 2pi/xn_per and 1/(gamma m): particle_loop.jl:400,529,531
__device__ __forceinline__ void refresh_move(CK* a, const Hot& h, Pt& p) {
  p.dphi = TWOPI_ / p.xn_per;
  p.t_step = p.gyro_period / p.xn_per;
  p.rg_val = rcp_refined(p.gam_pf * (h.aa * MP_));
  p.flags &= ~F_RM;
}

// Everything before the next scatter (head of the loop body and of Code Block 3,
// particle_loop.jl:154-326, 361-385).  `t_clock` is the time step of the previous move, which the
// clock of this pass still uses (particle_loop.jl:350 precedes :400).  Returns the end code or -1.
__device__ __forceinline__ int slow_pre(CK* a, const Hot& h, const mcsm::HotCoef& kc, Rng& rng, Pt& p, double t_clock, const bool lossy = false) {
  const double aa = h.aa;
  // (LOSSY kernel: the pass that was running lost all momentum in its in-line radiative loss -- the statements below at
  // `rad_losses`, with the floors already applied and the pass already counted)
  if (lossy && (p.flags & F_LOST)) return 4;
  if (p.helix >= MCS_HELIX_CAP) {            // the pass about to start would be number cap+1: quirk Q5
    p.helix += 1;
    cnt(a, MCS_IC_HELIX_CAP);
    return 1;
  }
  int f = p.flags;
  if ((f & F_ZONE) || h.custom_epsB || (h.etf && (f & F_CROSSED))) {
    // head of Code Block 3 (particle_loop.jl:186-246)
    const int ig = p.i_grid, io = p.ig3;
    const bool etf_ev = h.etf && (f & F_CROSSED) && !p.inj && p.x_old <= 0 && p.i_grid_old != p.i_grid;
    p.ig3 = ig;
    load_zone_props(p);
    double gd;
    if (h.custom_epsB && p.x > h.x_grid_stop) {
      const double bmag = S_bt[h.n_grid] * __builtin_sqrt(h.x_grid_stop / p.x);
      gd = 1 / (h.zzq * bmag);
    } else {
      gd = S_gd[ig];                         // == 1/(zz*btot[ig]), tabulated per zone
    }
    if (gd != p.gyro_denom) { p.gyro_denom = gd; f |= F_RS | F_RM; }
    if (ig != io && S_ux[ig] != S_ux[io]) {  // same u_x => no frame transform (particle_loop.jl:214)
      const Mom r = transform_p_PSP(a, io, ig, p.pb_pf, p.p_perp, p.gam_pf, p.phi);
      p.ptot_pf = r.ptot; p.pb_pf = r.pb; p.p_perp = r.pperp; p.gam_pf = r.gam; p.phi = r.phi;
      p.gyro_rad = p.p_perp * CC_ * p.gyro_denom;
      p.gyro_rad_tot = p.ptot_pf * CC_ * p.gyro_denom;
      f |= F_RS | F_RM;
    }
    if (etf_ev) {
      Mom r; r.ptot = p.ptot_pf; r.pb = p.pb_pf; r.pperp = p.p_perp; r.gam = p.gam_pf; r.phi = p.phi;
      r = do_energy_transfer(a, p.i_grid, p.i_grid_old, p.weight, r);
      p.ptot_pf = r.ptot; p.pb_pf = r.pb; p.p_perp = r.pperp; p.gam_pf = r.gam;
      f |= F_RS | F_RM;
    }
  }
  f &= ~(F_ZONE | F_CROSSED | F_CHECK | F_NOPARK);
  p.flags = f;
  // exit tests of Code Block 3 (particle_loop.jl:251-300), after the transforms
  const int ig = p.ig3;
  if (h.dont_scatter && p.x > 10 * p.gyro_rad) { p.helix += 1; return 1; }
  if (p.ptot_pf > h.pmax_cutoff) {          // rare (only near p_max)
    double ptot_sk, px, py, pz, gam_sk;
    transform_p_PS(aa, p.pb_pf, p.p_perp, p.gam_pf, p.phi, S_ux[ig], S_gsf[ig], S_bcos[ig], S_bsin[ig], ptot_sk, px, py, pz, gam_sk);
    if (ptot_sk > h.pmax_cutoff) { p.helix += 1; return 2; }
  }
  if (p.inj && p.x < h.feb_up) { p.helix += 1; return 2; }
  if (h.age_max > 0 && p.acctime > h.age_max) { p.helix += 1; return 3; }
  if (h.rad_losses && aa < 1) {
    double bmag = S_bt[ig];
    if (h.custom_epsB && p.x > h.x_grid_stop) bmag = S_bt[h.n_grid] * __builtin_sqrt(h.x_grid_stop / p.x);
    const double ptot_old = p.ptot_pf;
    const double B_CMB_loc = a->P.B_CMBz * S_gef[ig];
    p.ptot_pf = radiation_loss(bmag * bmag + B_CMB_loc * B_CMB_loc, p.ptot_pf, t_clock);
    if (p.ptot_pf <= 0) {
      p.ptot_pf = MCS_FLOOR; p.pb_pf = MCS_FLOOR; p.p_perp = MCS_FLOOR; p.gam_pf = 1;
      p.helix += 1;
      return 4;
    }
    p.gam_pf = mcsm::hypot1(p.ptot_pf / h.mc);
    p.pb_pf *= p.ptot_pf / ptot_old;
    p.p_perp *= p.ptot_pf / ptot_old;
    p.gyro_rad_tot = p.ptot_pf * CC_ * p.gyro_denom;
    p.gyro_rad = p.p_perp * CC_ * p.gyro_denom;
    p.flags |= F_RS | F_RM;
  }
  if (p.flags & (F_RS | F_CM)) {
    if (!h.dont_scatter) refresh_scatter(a, p, aa, aa * MP_ * CC_, h.eta);   // with the OLD xn_per, as the reference
    if (p.flags & F_RS) {
      refresh_dtest(a, h, p);
      int g = p.flags & ~(F_RS | F_NEARP | F_SAVE);
      if (p.ptot_pf > h.pmax_cutoff) g |= F_NEARP;
      if (p.downstream && p.ptot_pf > h.pcut) g |= F_SAVE;
      p.flags = g;
    }
    p.flags &= ~F_CM;
  }
  if (p.flags & F_SAVE) {
    // saved for the next pcut (particle_loop.jl:361-380): scatter, clock, time cut -- no move
    p.helix += 1;
    if (!h.dont_scatter) scattering(rng, p, kc);
    p.acctime += t_clock * S_gef[ig];        // F_SAVE implies downstream
    p.n_ovr += (unsigned)p.ovr_inc;
    if (h.do_tcuts && !(p.tcut > h.n_tcuts) && p.acctime >= tcut_next_of(a, h, p.tcut)) {
      tcut_track(a, p.tcut, p.weight, p.ptot_pf);
      p.tcut += 1;
    }
    return 0;
  }
  // fine / coarse step (particle_loop.jl:382-385); decided before the scatter here, after it in the
  // reference -- the scatter touches neither x nor gyro_rad_tot.  cos_max keeps the old xn_per for
  // the coming scatter and is refreshed for the one after (F_RS), exactly as in the reference.
  const double xn = p.x > p.gyro_rad_tot ? h.xn_coarse : h.xn_fine;
  if (xn != p.xn_per) { p.xn_per = xn; p.flags |= F_CM | F_RM; }   // x_dt, F_NEARP, F_SAVE do not depend on xn_per
  if (p.flags & F_RM) refresh_move(a, h, p);
  return -1;
}

// A Code Block 1 pass (particle_loop.jl:167-177 then Code Block 2): no Code Block 3, so no scatter,
// no clock, no exit test.  Done entirely in the rare block so that the common pass has no such case.
__device__ __forceinline__ bool block1_step(CK* a, const Hot& h, Pt& p, double& phi_old, int& end) {
  if (p.helix >= MCS_HELIX_CAP) { p.helix += 1; cnt(a, MCS_IC_HELIX_CAP); end = 1; return false; }
  p.helix += 1;
  p.p_perp = perpendicular_momentum(a, p.ptot_pf, p.pb_pf);
  p.gyro_rad = p.p_perp * CC_ * p.gyro_denom;
  p.flags &= ~(F_B1 | F_CROSSED);
  if (p.flags & F_RM) refresh_move(a, h, p);
  bool cross;
  const bool other = move_and_detect(a, h, p, phi_old, cross);
  return other | cross;
}

// The most frequent event by far: the move ended in another zone and nothing else is
// going on (no flag set, no other event, not the shock, same flow speed and field on both sides, not
// the FEB zone).  What slow_post + slow_pre would do then is: inj update (particle_loop.jl:433-435),
// the all_flux record (all_flux.jl:68-82, 130-137) and the zone reload of the next Code Block 3
// (particle_loop.jl:186-204); every exit test is known to be false.  Returns false if it is not that case.
template <bool DIRECT = false>
__device__ __forceinline__ bool plain_crossing(CK* a, const Hot& h, Pt& p, unsigned stack_height) {
  const int ne = h.n_grid + 2;
  const bool fwd = p.x > p.x_old;
  // neighbour first: its edges, flow speed, field and the other zone properties in ONE round of LDS
  // reads (the same values become the cached zone properties on success); the zone search of all_flux!
  // (all_flux.jl:68-72) stops at the neighbour iff its far edge is beyond x.  All tests feed one
  // predicate: no early return, one conditional region for the commit.
  int cand = fwd ? p.i_grid + 1 : p.i_grid - 1;
  cand = cand < 0 ? 0 : (cand > ne - 2 ? ne - 2 : cand);
  double c_lo = S_x[cand], c_hi = S_x[cand + 1];
  double ux_c = S_ux[cand], gd_c = S_gd[cand], gsf_c = S_gsf[cand], bcos_c = S_bcos[cand], gef_c = S_gef[cand];
  // all seven reads are in flight before the first is waited for (otherwise the compiler sinks some of them to
  // their uses: three LDS round trips of ~130 cycles instead of one)
  asm volatile("" : "+v"(c_lo), "+v"(c_hi), "+v"(ux_c), "+v"(gd_c), "+v"(gsf_c), "+v"(bcos_c), "+v"(gef_c));
  // (selects and bitwise logic: && / ?: on comparisons compile to exec-mask regions at ~40 cycles each)
  const double edge = fwd ? c_hi : c_lo;
  const bool adjacent = (fwd ? (edge > p.x) : (edge <= p.x)) & (cand != p.i_grid);
  const bool shock = (p.x_old < 0) & (p.x >= 0);
  if (MCS_UNLIKELY(!adjacent & !shock)) {
    // one step can cross several of the thin zones near the shock: the search loop
    // (the zone's own edge and the neighbour's far edge are known to fail: two candidates skipped)
    cand = zone_search(p.x, fwd, p.i_grid, ne, 2);
    const int c = cand < 0 ? 0 : cand;
    c_lo = S_x[c]; c_hi = S_x[c + 1];
    ux_c = S_ux[c]; gd_c = S_gd[c]; gsf_c = S_gsf[c]; bcos_c = S_bcos[c]; gef_c = S_gef[c];
  }
  const bool ok = !shock & (cand >= 0) & (cand > h.i_grid_feb) & (ux_c == p.z_ux) & (gd_c == p.gyro_denom);
  if (ok) {
    p.inj = p.inj | (p.downstream & (p.x < 0));
    p.i_grid_old = p.i_grid;
    p.i_grid = cand;
    if constexpr (DIRECT) flux_tally(a, p.pb_pf, p.p_perp, p.ptot_pf, p.gam_pf, p.phi, p.weight, p.x, p.x_old, p.i_grid, p.i_grid_old, p.ig3, p.inj);
    else push_record(p, p.ig3, (int)stack_height);     // the first push site of a pass: the height is the register mirror
    p.ig3 = cand;
    p.z_lo = c_lo; p.z_hi = c_hi; p.z_gsf = gsf_c; p.z_bcos = bcos_c; p.z_ux = ux_c; p.z_gef = gef_c;
  }
  return ok;
}

}  // namespace

// Pop up to 64 records from this wave's stack and tally them, one per lane (convergent code).
__device__ __forceinline__ void drain_events(CK* a, unsigned wv, unsigned lane, bool all) {
  unsigned cnt = S_evcur[wv];
  while (cnt >= 64u || (all && cnt > 0u)) {
    const unsigned take = cnt < 64u ? cnt : 64u;
    const unsigned base = cnt - take;
    long long foff = -1; double fval = 0.0;
    if (lane < take) {
      const unsigned e = base + lane;
      const uint32_t u = S_evu[wv][e];
      if (u >> 28) {     // a finished particle (particle_finish!): reason in bits 25-27
        particle_finish(a, (int)((u >> 25) & 7u), S_evf[wv][0][e], S_evf[wv][1][e], S_evf[wv][3][e], S_evf[wv][4][e],
                        S_evf[wv][5][e], (int)((u >> 16) & 0xffu), &foff, &fval);
      } else {
        flux_tally(a, S_evf[wv][0][e], S_evf[wv][1][e], S_evf[wv][2][e], S_evf[wv][3][e], S_evf[wv][4][e], S_evf[wv][5][e],
                   S_evf[wv][6][e], S_evf[wv][7][e], (int)(u & 0xffu), (int)((u >> 8) & 0xffu), (int)((u >> 16) & 0xffu),
                   ((u >> 24) & 1u) != 0u);
      }
    }
    // the escape-spectrum tallies of the batch.  A population of replicas of ONE saved particle that leave before
    // their first scatter hits ONE bin a million times, and a global atomic on one address serialises at ~12 ns
    // (12 ms per launch, measured): when every tally of the batch goes to the same bin the wave adds them up first.
    const unsigned long long vm = __builtin_amdgcn_ballot_w64(foff >= 0);
    if (vm != 0ull) {
      const int first = __ffsll((long long)vm) - 1;
      const long long off0 = __shfl(foff, first);
      if (__builtin_amdgcn_ballot_w64(foff >= 0 && foff != off0) == 0ull) {
        double sum = foff >= 0 ? fval : 0.0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        if ((int)lane == first) tadd(a, off0, sum);
      } else if (foff >= 0) {
        tadd(a, foff, fval);
      }
    }
    cnt = base;
  }
  if (lane == 0) S_evcur[wv] = cnt;
}

// ---- tail consolidation: two sparse waves of a block become one ---------------------------------
// After the work counter is exhausted no lane is refilled and a wave decays from ~56 live lanes to the one
// longest history; a pass costs the same with 1 live lane as with 64, and the two waves that share a SIMD
// (one of each of the CU's two blocks) split its issue slots: 1.4 us per pass each instead of 0.8 us alone
// (profiles/r01_wave_timeline.txt).  So the waves of a block pair up by SIMD, (0,1) and (2,3): once the
// donor's live particles fit into the receiver's idle lanes the donor drains its tally records, writes the
// complete lane state of its particles into its (now free) record stack, hands it over and ends; the
// receiver picks the particles up at its next poll (every 16 passes).  Which wave of a pair stays is
// chosen by the wave-slot id, which differs between the two blocks resident on a CU (measured: all waves
// of a block share it), so that the surviving waves of both blocks sit on different SIMDs.  A history does
// not depend on the lane that runs it (state and RNG stream travel with the particle): results unchanged.
// Nobody ever waits: a state word per pair (open -> donated -> closed | open -> closed) is moved by CAS.
#define MCS_MB_SLOTS 32
#define MCS_MB_WORDS 36
static_assert(MCS_MB_WORDS * MCS_MB_SLOTS <= MCS_EV_F64 * MCS_EV_CAP, "the mailbox lives in the donor's record stack");
static_assert(MCS_MB_WORDS == MCS_STRAG_WORDS, "the export buffers of a sliced run hold the same lane state");
__shared__ unsigned int S_msimd[4];     // SIMD id of each wave (HW_REG_HW_ID[5:4])
__shared__ unsigned int S_mlive[4];     // live lanes an exhausted receiver last published (64 before that)
__shared__ unsigned int S_mstate[2];    // per SIMD pair: 0 open, 1 donated, 2 closed
__shared__ unsigned int S_mcount[2];    // particles in the mailbox

// STRIDE: MCS_MB_SLOTS for the LDS mailbox (word-major); 1 for the export buffers in global memory (entry-major: one
// address register per lane, every word at an immediate offset)
template <int STRIDE>
__device__ __forceinline__ void state_store(double* mb, const Pt& p, const Rng& rng, long long k, int evw, double phi_prev) {
  constexpr int S_ = STRIDE;
  const double v[30] = {p.weight, p.ptot_pf, p.pb_pf, p.p_perp, p.gam_pf, p.x, p.x_old, p.phi, p.prp, p.acctime, p.xn_per, p.dphi,
                        p.gyro_denom, p.gyro_rad, p.gyro_rad_tot, p.gyro_period, p.t_step, p.rp_val, p.cm_val, p.rg_val, p.x_dt,
                        p.t_ev, p.z_gsf, p.z_bcos, p.z_ux, p.z_gef, p.z_lo, p.z_hi, phi_prev, 0.0};
#pragma unroll
  for (int j = 0; j < 30; ++j) mb[j * S_] = v[j];
  mb[30 * S_] = __longlong_as_double(k);
  const int gridpack = p.i_grid | (p.i_grid_old << 8) | (p.ig3 << 16) | (p.tcut << 24);
  const int bits = p.ovr_inc | ((int)p.downstream << 1) | ((int)p.inj << 2) | ((evw & 7) << 3);
  mb[31 * S_] = __hiloint2double(p.flags, (int)p.n_ovr);
  mb[32 * S_] = __hiloint2double(gridpack, p.helix);
  mb[33 * S_] = __hiloint2double(p.n_retro, bits);
  mb[34 * S_] = __hiloint2double((int)rng.k0, (int)rng.k1);
  mb[35 * S_] = __hiloint2double((int)rng.n, 0);
}
template <int STRIDE, bool COHERENT>
__device__ __forceinline__ void state_load(const double* mb, Pt& p, Rng& rng, long long& k, int& evw, double& phi_prev) {
  constexpr int S_ = STRIDE;
  // COHERENT: the wave reads back what it stored to global memory earlier -- agent-scope loads, past the L1
  auto LD = [&](int j) -> double {
    if (COHERENT) return __hip_atomic_load(mb + j * S_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return mb[j * S_];
  };
  double unused__;
  double* const d[30] = {&p.weight, &p.ptot_pf, &p.pb_pf, &p.p_perp, &p.gam_pf, &p.x, &p.x_old, &p.phi, &p.prp, &p.acctime, &p.xn_per,
                         &p.dphi, &p.gyro_denom, &p.gyro_rad, &p.gyro_rad_tot, &p.gyro_period, &p.t_step, &p.rp_val, &p.cm_val,
                         &p.rg_val, &p.x_dt, &p.t_ev, &p.z_gsf, &p.z_bcos, &p.z_ux, &p.z_gef, &p.z_lo, &p.z_hi, &phi_prev, &unused__};
#pragma unroll
  for (int j = 0; j < 30; ++j) *d[j] = LD(j);
  k = __double_as_longlong(LD(30));
  const double w1 = LD(31), w2 = LD(32), w3 = LD(33), w4 = LD(34), w5 = LD(35);
  p.flags = __double2hiint(w1); p.n_ovr = (unsigned)__double2loint(w1);
  const int gridpack = __double2hiint(w2);
  p.helix = __double2loint(w2);
  p.i_grid = gridpack & 0xff; p.i_grid_old = (gridpack >> 8) & 0xff; p.ig3 = (gridpack >> 16) & 0xff; p.tcut = (gridpack >> 24) & 0xff;
  p.n_retro = __double2hiint(w3);
  const int bits = __double2loint(w3);
  p.ovr_inc = bits & 1; p.downstream = (bits >> 1) & 1; p.inj = (bits >> 2) & 1; evw = (bits >> 3) & 7;
  rng.k0 = (uint32_t)__double2hiint(w4); rng.k1 = (uint32_t)__double2loint(w4);
  rng.n = (uint32_t)__double2hiint(w5);
  p.npush = 0;
}

__device__ __forceinline__ void mb_store(unsigned box, unsigned r, const Pt& p, const Rng& rng, long long k, int evw, double phi_prev) {
  state_store<MCS_MB_SLOTS>(&S_evf[box][0][0] + r, p, rng, k, evw, phi_prev);
}
__device__ __forceinline__ void mb_load(unsigned box, unsigned r, Pt& p, Rng& rng, long long& k, int& evw, double& phi_prev) {
  state_load<MCS_MB_SLOTS, false>(&S_evf[box][0][0] + r, p, rng, k, evw, phi_prev);
}


// number of set bits of `m` below this lane (v_mbcnt: no per-lane mask register)
__device__ __forceinline__ unsigned below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Per-block set-up shared by the transport kernels: the grid tables (plus per-zone sin / cos theta_B and 1/(qB)) and the
// time cuts into LDS, the staging arrays and counters cleared.  The caller synchronises.
__device__ __forceinline__ void block_prologue(CK* a) {
  const int ne = a->P.n_grid + 2, ntc = a->tb.n_tcuts;
  for (int i = threadIdx.x; i < ne; i += blockDim.x) {
    S_x[i] = a->tb.x_grid[i]; S_ux[i] = a->tb.ux[i]; S_uz[i] = a->tb.uz[i]; S_ut[i] = a->tb.utot[i];
    S_gsf[i] = a->tb.gsf[i]; S_gef[i] = a->tb.gef[i];
    const double bt = a->tb.btot[i], th = a->tb.theta[i];
    S_bt[i] = bt;
    double sn, cs;
    mcsm::sincos(th, &sn, &cs);
    S_bsin[i] = sn; S_bcos[i] = cs;
    S_gd[i] = 1 / (a->zzq * bt);
  }
  for (int i = threadIdx.x; i < 3 * MCS_MAXNE; i += blockDim.x) S_fl[i] = 0.0;
  for (int i = threadIdx.x; i < 2 * (MCS_PSD_MAX + 1); i += blockDim.x) (&S_eff[0][0])[i] = 0.0;
  for (int i = threadIdx.x; i < MCS_NA_C; i += blockDim.x) S_wc[i] = 0.0;
  for (int i = threadIdx.x; i < ntc; i += blockDim.x) S_tc[i] = a->tb.tcuts[i];
  for (int i = threadIdx.x; i < MCS_MAXNE; i += blockDim.x) S_nc[i] = 0;
  if (threadIdx.x <= MCS_IC_COUNT) g_ctr[threadIdx.x] = 0u;
  if (threadIdx.x < 8) g_sc[threadIdx.x] = 0.0;
  if (threadIdx.x < 3) S_steps[threadIdx.x] = 0ull;
  if (threadIdx.x < 4) { S_evcur[threadIdx.x] = 0u; S_mlive[threadIdx.x] = 64u; }
#ifdef MCS_PROF_TAIL
#ifdef MCS_PROF_ALLPHASES
  if (threadIdx.x < 4) S_ttgate[threadIdx.x] = 1u;
#else
  if (threadIdx.x < 4) S_ttgate[threadIdx.x] = 0u;
#endif
#endif
  if (threadIdx.x < 2) { S_mstate[threadIdx.x] = 0u; S_mcount[threadIdx.x] = 0u; }
  if ((threadIdx.x & 63u) == 0u && threadIdx.x < 256u) S_msimd[threadIdx.x >> 6] = (unsigned)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);     // (four entries: the wave-specialised kernel runs this prologue with eight waves)
#ifdef MCS_PROF
  if (threadIdx.x < MCS_NPROF) S_prof[threadIdx.x] = 0ull;
#endif
}

// Per-block flush of the LDS staging (fluxes, crossings, counters, scalars, per-bin escape sums) into the tally buffers:
// one global atomic per entry and block.  The caller has synchronised.
__device__ __forceinline__ void block_flush(CK* a) {
  const int ng = a->P.n_grid;
  if (threadIdx.x < 3 && S_steps[threadIdx.x]) {   // one global atomic per block and counter
    const int which = threadIdx.x == 0 ? MCS_IC_STEPS_HELIX : (threadIdx.x == 1 ? MCS_IC_STEPS_RETRO : MCS_IC_RNG_DRAWS);
    gadd_u64(&a->I[ng + which], S_steps[threadIdx.x]);
  }
  if (threadIdx.x < MCS_IC_COUNT) {
    const unsigned int c = g_ctr[threadIdx.x];
    if (c) gadd_u64(&a->I[ng + threadIdx.x], (unsigned long long)c);
  } else if (threadIdx.x == MCS_IC_COUNT) {
    const unsigned int c = g_ctr[MCS_IC_COUNT];
    if (c) gadd_u64(a->n_saved, (unsigned long long)c);
  } else if (threadIdx.x < MCS_IC_COUNT + 8) {
    const int j = threadIdx.x - MCS_IC_COUNT - 1;
    const double v = g_sc[j];
    const int ion = a->i_ion - 1, iter = a->i_iter - 1;
    long long off = a->L.scalars + j;
    if (j == 4) off = a->L.esc_flux + ion;
    else if (j == 5) off = a->L.px_esc_feb + ion + (long long)a->P.n_ions * iter;
    else if (j == 6) off = a->L.energy_esc_feb + ion + (long long)a->P.n_ions * iter;
    if (v != 0.0) gadd_f64(&a->T[off], v);
  }
  for (int i = threadIdx.x; i < ng; i += blockDim.x) {
    const double v0 = S_fl[i], v1 = S_fl[MCS_MAXNE + i], v2 = S_fl[2 * MCS_MAXNE + i];
    if (v0 != 0.0) gadd_f64(&a->T[a->L.pxx_flux + i], v0);
    if (v1 != 0.0) gadd_f64(&a->T[a->L.pxz_flux + i], v1);
    if (v2 != 0.0) gadd_f64(&a->T[a->L.energy_flux + i], v2);
    const int c = S_nc[i];
    if (c) gadd_u64(&a->I[MCS_I_NUM_CROSSINGS + i], (unsigned long long)c);
  }
  for (int i = threadIdx.x; i < MCS_NA_C; i += blockDim.x) {
    const double w = S_wc[i];
    if (w != 0.0) gadd_f64(&a->T[a->L.weight_coupled + i + (long long)MCS_NA_C * (a->i_ion - 1)], w);
  }
  for (int i = threadIdx.x; i <= MCS_PSD_MAX; i += blockDim.x) {
    const long long o = i + (long long)(MCS_PSD_MAX + 1) * (a->i_ion - 1);
    const double e = S_eff[0][i], w = S_eff[1][i];
    if (e != 0.0) gadd_f64(&a->T[a->L.esc_energy_eff + o], e);
    if (w != 0.0) gadd_f64(&a->T[a->L.esc_num_eff + o], w);
  }
}

// PLAIN = the common configuration, decided by the host: scattering on, parallel field in every zone,
// no custom eps_B, no energy transfer, no electron radiative losses, no downstream FEB, DSA on with
// injection probability 1, ions, no x_spec detectors.  The flags are then compile-time constants: their
// scalar branches and the code behind them disappear from that kernel.
// LOSSY = electrons with radiative losses (and none of: custom eps_B, no-scatter), decided by the host.  The reference applies
// the loss in EVERY pass (particle_loop.jl:302-326), and everything derived from the momentum follows: in the general
// kernel such a configuration visits the rare region in every pass (h.every_pass: one common pass per trip, the whole of
// slow_pre per pass -- measured 7 000 SIMD cycles per wave-pass against ~1 100 for the common pass).  Here the loss and the
// refreshes that hang on it are straight-line code at the head of the common pass, for exactly the lanes whose other
// statements of slow_pre are no-ops (no flag set, nothing pending); a lane that has just run slow_pre in the rare region has
// had this pass's loss there and skips the block once.  Same statements in the same order on the same values.
// PLAIN_ETF: the PLAIN conditions with the ion -> electron energy transfer left as a run-time flag (the ions of a multi-species run)
// SLICED: the launch may resume lane states an earlier launch exported, hold fewer than 64 particles per wave and export its live
// particles a budget of trips after the queue ran dry (KArgs "sliced launches"; mcs_set_tail_slicing).  Measured not to pay
// (DESIGN.md "Sliced tail"), so it is a kernel of its own (mcs_k_transport_sliced, the general form): the shipping kernels carry
// neither its code nor its registers (round 3: +20 B/lane of scratch in mcs_k_transport_plain).
template <bool PLAIN, bool LOSSY = false, bool PLAIN_ETF = false, bool SLICED = false>
__device__ __forceinline__ void transport_body(const KArgs* __restrict__ ka) {
  CK* a = (CK*)ka;
  if (a->n_dev && *a->n_dev == 0) return;      // (a fused species loop launches the pcuts after the last populated one too)
  const int ne = a->P.n_grid + 2, ng = a->P.n_grid, ntc = a->tb.n_tcuts;
  block_prologue(a);
  __syncthreads();

  // Hot-loop constants are parked in VGPRs behind an opaque move: the compiler can then
  // neither re-load them from the constant buffer inside the loop (s_load + s_waitcnt) nor
  // spill them as SGPRs.
  Hot h;
  h.m = vconst(a->m); h.x_grid_stop = vconst(a->P.x_grid_stop); h.xn_coarse = vconst(a->P.xn_per_coarse);
  h.u2 = sconst(a->P.u2); h.eta = sconst(a->P.eta_mfp); h.zzq = sconst(a->zzq); h.mc = sconst(a->mc);
  h.feb_down = sconst(a->P.feb_downstream); h.pcut = sconst(a->pcut); h.pmax_cutoff = sconst(a->pmax_cutoff);
  h.feb_up = sconst(a->P.feb_upstream); h.age_max = sconst(a->P.age_max); h.inj_frac = sconst(a->inj_frac);
  h.xn_fine = sconst(a->P.xn_per_fine); h.aa = sconst(a->aa);
  h.n_grid = sconsti(ng); h.i_grid_feb = sconsti(a->P.i_grid_feb); h.n_tcuts = sconsti(ntc); h.n_xspec = sconsti(a->tb.n_xspec);
  // uniform flags stay scalar (s_cbranch): whole code regions are skipped for free
  h.custom_epsB = a->P.use_custom_epsB != 0; h.etf = a->P.energy_transfer_frac > 0;
  h.dont_scatter = a->P.dont_scatter != 0; h.rad_losses = a->P.do_rad_losses != 0;
  h.do_tcuts = a->P.do_tcuts != 0; h.dont_DSA = a->P.dont_DSA != 0;
  {
    int ob = 0;
    for (int i = threadIdx.x; i < ne; i += blockDim.x) ob |= S_bsin[i] != 0.0;
    h.oblique = __builtin_amdgcn_readfirstlane(__syncthreads_or(ob)) != 0;     // scalar: a uniform branch, not an exec mask
  }
  h.every_pass = h.custom_epsB || (h.rad_losses && h.aa < 1) || h.dont_scatter;
  h.odd_cfg = h.feb_down > 0 || h.dont_DSA || h.inj_frac < 1 || h.aa < 1 || h.n_xspec != 0;
  if (PLAIN) {
    h.custom_epsB = false; h.etf = PLAIN_ETF ? h.etf : false; h.dont_scatter = false; h.dont_DSA = false; h.oblique = false;
    h.every_pass = false; h.odd_cfg = false;
  }
  // the constants of the in-line loss (LOSSY only; dead code otherwise)
  double l_bcmb = 0, l_pe_crit = 0, l_game_crit = 0;
  if (LOSSY) {
    h.every_pass = false; h.custom_epsB = false; h.dont_scatter = false; h.rad_losses = true;
    l_bcmb = sconst(a->P.B_CMBz); l_pe_crit = sconst(a->P.pe_crit); l_game_crit = sconst(a->P.game_crit);
  }

  // the 29 constants of the per-step sincos + asin, resident in VGPRs
  mcsm::HotCoef kc;
  kc.S0 = vconst(MCS_SIN_0); kc.S1 = vconst(MCS_SIN_1); kc.S2 = vconst(MCS_SIN_2); kc.S3 = vconst(MCS_SIN_3);
  kc.S4 = vconst(MCS_SIN_4); kc.S5 = vconst(MCS_SIN_5);
  kc.C0 = vconst(MCS_COS_0); kc.C1 = vconst(MCS_COS_1); kc.C2 = vconst(MCS_COS_2); kc.C3 = vconst(MCS_COS_3);
  kc.C4 = vconst(MCS_COS_4); kc.C5 = vconst(MCS_COS_5);
  kc.A0 = vconst(MCS_ASIN_0); kc.A1 = vconst(MCS_ASIN_1); kc.A2 = vconst(MCS_ASIN_2); kc.A3 = vconst(MCS_ASIN_3);
  kc.A4 = vconst(MCS_ASIN_4); kc.A5 = vconst(MCS_ASIN_5); kc.A6 = vconst(MCS_ASIN_6); kc.A7 = vconst(MCS_ASIN_7);
  kc.A8 = vconst(MCS_ASIN_8); kc.A9 = vconst(MCS_ASIN_9); kc.A10 = vconst(MCS_ASIN_10); kc.A11 = vconst(MCS_ASIN_11);
  kc.A12 = vconst(MCS_ASIN_12);
  kc.R0 = vconst(MCS_TWO_OVER_PI); kc.R1 = vconst(MCS_PIO2_0); kc.R2 = vconst(MCS_PIO2_1); kc.R3 = vconst(MCS_PIO2_2);

  Pt p;
  Rng rng;
  // idle lanes run the common pass too (on whatever state they hold): give them valid table indices
  p.weight = 0; p.ptot_pf = 1; p.pb_pf = 0; p.p_perp = 1; p.gam_pf = 1; p.x = 0; p.x_old = 0; p.phi = 0; p.prp = 0; p.acctime = 0;
  p.xn_per = 1; p.dphi = 0; p.gyro_denom = 0; p.gyro_rad = 0; p.gyro_rad_tot = 0; p.gyro_period = 0; p.t_step = 0;
  p.rp_val = 1; p.cm_val = 1; p.rg_val = 1; p.x_dt = 0; p.t_ev = 0; p.flags = 0; p.ovr_inc = 0; p.n_ovr = 0u;
  p.z_gsf = 1; p.z_bcos = 1; p.z_ux = 0; p.z_gef = 1; p.z_lo = 0; p.z_hi = 0; p.t_hi = 0; p.t_lo = 0; p.c_gef = 0; p.c_tev = 0;
  p.i_grid = 0; p.i_grid_old = 0; p.ig3 = 0; p.helix = 0; p.tcut = 1; p.n_retro = 0; p.downstream = false; p.inj = false;
  rng.init(0ull);
  // act: -1 while the lane holds a live particle, 0 while it is idle (an all-ones / all-zeros word, so that the loop header
  // masks the pending bits with one v_and; hipcc keeps a `bool` as a byte in a VGPR and spends two VALU per ballot on it)
  int act = 0;
#define active (act != 0)
  bool exhausted = false;
  // what the last move of this lane's particle left pending, in ONE register (the loop header tests it together with
  // p.flags): bit 0 "ev" the move needs slow_post, bit 1 "ev_x" the move left its zone, bit 2 "moved" the particle has
  // made a move since it was loaded
  int evw = 0;
  double phi_prev = 0.0;  // phase before the last move (the no-DSA retry loop needs it)
  long long k = -1;
  const unsigned lane = __lane_id();
  // the launch's queue: first the resume list (lane states a previous launch exported), then the fresh particles
  // fresh_lo .. n-1 of the population (see KArgs: sliced launches)
  const unsigned long long n_resume = SLICED ? (unsigned long long)a->n_resume : 0ull;
  const long long fresh_lo = SLICED ? a->fresh_lo : 0ll;
  long long n_pop = a->n;
  if (a->n_dev) {      // (fused species loop: the size was decided on the device; one load, made wave-uniform)
    const long long v = *a->n_dev;
    n_pop = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
  }
  const unsigned long long n = n_resume + (unsigned long long)(n_pop - fresh_lo);
  const unsigned budget = SLICED ? (unsigned)__builtin_amdgcn_readfirstlane(a->budget_trips) : 0u;
  const int claim_max = SLICED ? __builtin_amdgcn_readfirstlane(a->claim_max) : 64;
  const unsigned long_draws = SLICED ? (unsigned)__builtin_amdgcn_readfirstlane((int)a->long_draws) : 0u;
  unsigned mtick_ex = 0;            // mtick when this wave found the queue exhausted

  const unsigned wv = threadIdx.x >> 6;
  // tail consolidation (see mb_store): role of this wave, decided once
  int mrole = 0;                    // 0 none / done, 1 donor, 2 receiver
  unsigned mtick = 0, mpoll_mask = ~0u;     // the poll happens when (mtick & mpoll_mask) == 0: never before exhaustion
  if (blockDim.x == 256u && a->tail_merge) {
    const unsigned s0 = S_msimd[0], s1 = S_msimd[1], s2 = S_msimd[2], s3 = S_msimd[3];
    const bool distinct = ((1u << s0) | (1u << s1) | (1u << s2) | (1u << s3)) == 15u;
    const unsigned slot = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
    if (__builtin_amdgcn_readfirstlane(distinct ? 1 : 0))
      mrole = __builtin_amdgcn_readfirstlane((int)(((S_msimd[wv] ^ slot) & 1u) == 0u ? 2 : 1));
  }
  // the pair (SIMDs 2q, 2q+1) and the partner's wave index, looked up at the (rare) uses
  auto pair_of = [&](unsigned& partner) -> unsigned {
    const unsigned mine = S_msimd[wv], want = mine ^ 1u;
    const unsigned pw = S_msimd[0] == want ? 0u : (S_msimd[1] == want ? 1u : (S_msimd[2] == want ? 2u : 3u));
    partner = (unsigned)__builtin_amdgcn_readfirstlane((int)pw);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(mine >> 1));
  };
#ifdef MCS_PROF
  const unsigned gw__ = (blockIdx.x * 4u + wv) & 8191u;
  if (lane == 0) { g_wave[gw__][0] = __builtin_amdgcn_s_memrealtime(); g_wave[gw__][1] = 0; }
  unsigned long long xp__ = 0, xr__ = 0, xl__ = 0, xf__ = 0;   // after exhaustion: passes, rare entries, live-lane sum, full-path lanes
#endif
  // Waiting for the full Code Blocks.  A lane whose particle needs them (an upward threshold, the shock, a zone with another
  // flow speed, ...) used to run them on the spot -- thousands of cycles for one or two lanes while the wave waits.
  // Instead it sets F_WAIT and sits out the common passes (nothing of its particle changes meanwhile) until the wave has
  // refill_min lanes idle or waiting; then the idle lanes claim new particles and the waiting lanes are released
  // (F_NOPARK), so that new and released particles run their Code Blocks together, in one pass.  Only young particles wait
  // (a long history is what the launch waits for at the end), and nothing waits once the work counter is exhausted.
  // (Round 1 wrote the waiting particle to a park buffer in global memory and read it back at the refill: 36 words each
  // way and a memory latency per refill for a lane that idles just the same.)
  // (readfirstlane: the compiler must see this as wave-uniform, or every scalar branch that depends on it turns into an
  // exec-mask region)
  const bool waiting_on = __builtin_amdgcn_readfirstlane((int)(a->wait_full != 0 && blockDim.x == 256u)) != 0;
  // idle + waiting lanes at which the wave has housekeeping to do: MCS_REFILL_MIN while there is unclaimed work;
  // afterwards 64 (nothing left: the wave ends)
  // (a wave that may hold only claim_max < 64 particles -- a sparse queue spread over the chip -- refills as soon as one of its
  // claim_max places is free: the host turns deferral and waiting off for such launches)
  const unsigned refill_min = claim_max < 64 ? (unsigned)(64 - claim_max + 1)
                                             : (unsigned)__builtin_amdgcn_readfirstlane(a->refill_min);   // MCS_REFILL_MIN unless overridden (A/B runs)
  unsigned refill_at = refill_min;
  unsigned ev_pending = 0;          // wave-uniform mirror of S_evcur[wv] (no LDS round trip per pass)
  // ---- tail ring.  After the work counter is exhausted a wave decays to a handful of live particles, and an instruction
  // costs the wave the same with 1 lane enabled as with 64.  Half of the common pass -- the Philox block of the two
  // draws, their conversion, sin / cos of the scattering azimuth: scatter_draws(), and cos / sin of the deflection:
  // scatter_cone() -- depends on nothing but the particle's random stream (key, draw index) and its cos_max, which
  // changes in rare code only, so the idle lanes compute it AHEAD: with L <= 32 live lanes, lane w
  // evaluates block j = w mod D of the next D = floor(64 / L) scatters of the (w / D)-th live particle, all in
  // one pass of those ~130 instructions; the results sit in the free top of the wave's record stack (entries 128..191:
  // at most 63 + 2 L <= 127 records are pending) and the live lane reads its four doubles back in each of the next D
  // passes.  An entry is addressed by draw index (entry j <-> index rb + 2j), so draws taken in rare code in between
  // (prob_return, the retro walk: whole blocks) just skip entries; a lane whose particle or cos_max changes invalidates its batch.
  // Same functions, same bits: per-particle results are unchanged.
  bool ring_on = false;             // wave-uniform: the work counter is exhausted (and KArgs::tail_ring)
  unsigned rb = 0u - 256u;          // draw index of entry 0 of this lane's batch; (rng.n - rb) / 2 >= ringD: no entry
  unsigned rrow = 0u;               // first entry of this lane's batch
  unsigned ringD = 1u;              // wave-uniform: entries per live lane in the current batch
  unsigned defer_k = h.every_pass ? 1u : (unsigned)__builtin_amdgcn_readfirstlane(a->defer_k);      // 1 once the work counter is exhausted
  p.npush = 0;
  // a new ring batch for every live particle (am: the live lanes, L of them): owners publish (key, draw index) by rank, workers evaluate
  // D = floor(64 / L) draws ahead per live particle (any D, not just powers of two: L = 10 gets 6, not 4);
  // lane -> (particle q, draw jw) by a multiply and a shift: floor(lane / D) == (lane * M) >> 16 with
  // M = floor(65536 / D) + 1 for every lane < 64 and D <= 64
  auto build_batch = [&](const unsigned long long am, const unsigned L) {
    const unsigned D = 64u / L;
    const unsigned M = 65536u / D + 1u;
    const unsigned rank = below(am);
    if (active) {
      S_evf[wv][3][128u + rank] = __hiloint2double((int)rng.k0, (int)rng.k1);
      S_evf[wv][4][128u + rank] = __hiloint2double((int)rng.n, 0);
      S_evf[wv][5][128u + rank] = 1 - p.cm_val;      // the owner's 1 - cos_max: the cone of the deflection is part of the batch
    }
    const unsigned q = (lane * M) >> 16, jw = lane - q * D;
    const bool valid = q < L;
    const unsigned qq = valid ? q : 0u;
    const double w1 = S_evf[wv][3][128u + qq], w2 = S_evf[wv][4][128u + qq], womc = S_evf[wv][5][128u + qq];
    double eU1, es, ec, ecd, esd, essd;
    scatter_draws((uint32_t)__double2hiint(w1), (uint32_t)__double2loint(w1), ((uint32_t)__double2hiint(w2) >> 1) + jw, kc, eU1, es, ec);
    scatter_cone(eU1, es, womc, ecd, esd, essd);
    if (valid) {       // entry q * D + jw == lane
      S_evf[wv][0][128u + lane] = ecd; S_evf[wv][1][128u + lane] = esd; S_evf[wv][2][128u + lane] = ec; S_evf[wv][6][128u + lane] = essd;
    }
    if (active) { rb = rng.n; rrow = rank * D; }
    ringD = D;
  };
  // (see KArgs::tail_loop)
  constexpr bool TAIL_LOOP = !LOSSY;      // (LOSSY: no tail ring -- the cone changes in every pass)
  const unsigned tail_loop_L = TAIL_LOOP ? (unsigned)__builtin_amdgcn_readfirstlane(a->tail_loop) : 0u;
  bool done = false;
  while (!done) {
    ev_pending += (unsigned)(__popcll(__builtin_amdgcn_ballot_w64(p.npush > 0)) + __popcll(__builtin_amdgcn_ballot_w64(p.npush > 1)));
    p.npush = 0;
    // ---- housekeeping behind ONE scalar branch: records to tally, idle lanes to refill, nothing left
    const unsigned long long act_mask = __builtin_amdgcn_ballot_w64(active);
    // Refills are batched: a load of new particles stalls the whole wave for a memory latency (about one
    // pass), so the wave waits until MCS_REFILL_MIN lanes are idle.  With histories of a few hundred
    // passes (late pcuts) a lane idles every 3-4 passes and refilling each at once cost ~25 % of the time.
    const int n_idle = 64 - __popcll(act_mask);
    unsigned Lh = 64u - (unsigned)n_idle;     // live lanes while this pass's rare region runs (housekeeping may add some)
    const bool waitl = (p.flags & F_WAIT) != 0;                                 // (idle lanes have flags == 0)
    const unsigned n_wait = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(waitl));
    ++mtick;
    // (bitwise | on purpose: one scalar branch, not a chain of short-circuit branches)
    if (MCS_UNLIKELY(((ev_pending >= 64u) | ((unsigned)n_idle + n_wait >= refill_at) | ((mtick & mpoll_mask) == 0u)) != 0)) {
      if (ev_pending >= 64u) { drain_events(a, wv, lane, false); ev_pending &= 63u; PROF_ADD(3, 1); }
      // the batch is complete: the waiting lanes run their Code Blocks in the coming pass, with the new particles
      if (n_wait > 0u && (exhausted || (unsigned)n_idle + n_wait >= refill_min)) {
        if (waitl) p.flags = (p.flags & ~F_WAIT) | F_NOPARK;
        PROF_ADD(36, 1); PROF_ADD(37, n_wait);
      }
      // refill idle lanes (wave-aggregated claim)
      const unsigned long long idle = ~__builtin_amdgcn_ballot_w64(active);
      if ((unsigned)n_idle + n_wait >= refill_min && !exhausted && idle != 0ull) {
        const int nfree = __popcll(idle);
        int nidle = nfree < claim_max - (64 - nfree) ? nfree : claim_max - (64 - nfree);    // places to fill: at most claim_max live
        nidle = nidle > 0 ? nidle : 0;
        const int leader = __ffsll((long long)idle) - 1;
        unsigned long long base = 0;
        if ((int)lane == leader) base = atomicAdd(a->work_counter, (unsigned long long)nidle);
        base = __shfl(base, leader);
        if (__builtin_amdgcn_readfirstlane(base >= n ? 1 : 0)) {
          exhausted = true;
          defer_k = 1u;
          ring_on = !LOSSY && __builtin_amdgcn_readfirstlane(a->tail_ring) != 0;     // (LOSSY: the cone changes in every pass)
          mtick_ex = mtick;
          if (mrole != 0 || budget != 0u) mpoll_mask = MCS_MERGE_POLL_MASK;
#ifdef MCS_PROF_TAIL
          if (lane == 0) S_ttgate[wv] = 1u;
#endif
#ifdef MCS_PROF
          if (lane == 0 && g_wave[gw__][1] == 0) { g_wave[gw__][1] = __builtin_amdgcn_s_memrealtime(); g_wave[gw__][3] = (unsigned long long)__popcll(act_mask); }
#endif
        } else if (!active) {
          const int rank = (int)below(idle);
          const unsigned long long idx = base + (unsigned long long)rank;
          if (rank < nidle && idx >= n_resume && idx < n) {
            k = fresh_lo + (long long)(idx - n_resume);
            load_particle(a, h, k, p, rng);
            act = -1; evw = 0; rb = 0u - 256u;
            // wait for the loads HERE: the common pass then carries no vmcnt wait (which would also wait for
            // every outstanding store and no-return tally atomic)
            __builtin_amdgcn_s_waitcnt(0x0F70);
          }
        }
        // the head of the queue: particles an earlier launch exported.  Their lane states come in through the wave's record
        // stack, used as a mailbox exactly as in the tail consolidation (pending records are tallied first); up to
        // MCS_MB_SLOTS at a time.  (Through LDS because 36 global loads per lane in this loop cost ~100 spilled registers.)
        if (SLICED && !exhausted && __builtin_amdgcn_readfirstlane(base < n_resume ? 1 : 0)) {
          drain_events(a, wv, lane, true); ev_pending = 0u;
          const unsigned long long left = n_resume - base;
          const unsigned cnt_r = (unsigned)(left < (unsigned long long)nidle ? left : (unsigned long long)nidle);
          const unsigned rank = below(idle);
          double* const box = &S_evf[wv][0][0];
          for (unsigned b0 = 0; b0 < cnt_r; b0 += MCS_MB_SLOTS) {
            const unsigned cb = cnt_r - b0 < MCS_MB_SLOTS ? cnt_r - b0 : MCS_MB_SLOTS;
            const double* src = a->strag_in + (base + b0) * MCS_STRAG_WORDS;
            for (unsigned i = lane; i < MCS_MB_WORDS * MCS_MB_SLOTS; i += 64u) {
              const unsigned j = i / MCS_MB_SLOTS, r = i % MCS_MB_SLOTS;
              if (r < cb) box[i] = src[r * MCS_STRAG_WORDS + j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (!active && rank >= b0 && rank < b0 + cb) {
              mb_load(wv, rank - b0, p, rng, k, evw, phi_prev);
              refresh_thr<LOSSY>(h, p);
              act = -1; rb = rng.n - 256u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
        }
        PROF_ADD(5, 1); PROF_ADD(6, nidle);
      }
      // ---- tail consolidation (see mb_store); `exhausted` holds whenever mpoll_mask is 15
      auto take_donation = [&](unsigned mpair, unsigned mpartner) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const unsigned cntm = (unsigned)__builtin_amdgcn_readfirstlane((int)S_mcount[mpair]);
        const unsigned long long idle_now = ~__builtin_amdgcn_ballot_w64(active);
        const unsigned r = below(idle_now);
        if (!active && r < cntm) {
          mb_load(mpartner, r, p, rng, k, evw, phi_prev);
          refresh_thr<LOSSY>(h, p);
          rb = rng.n - 256u;
          act = -1;
        }
        mrole = 0; mpoll_mask = budget != 0u ? MCS_MERGE_POLL_MASK : ~0u;
      };
      // (a sliced launch whose budget is spent exports below: the pair is closed first, a donation already made is taken and exported too)
      // (long_draws: the wave closes when every live particle has become long -- see KArgs -- instead of after a number of trips)
      const bool closing = SLICED && budget != 0u && exhausted &&
                           (long_draws != 0u ? __builtin_amdgcn_ballot_w64(active && rng.n < long_draws) == 0ull : mtick - mtick_ex >= budget);
      if (mrole != 0 && exhausted) {
        unsigned mpartner;
        const unsigned mpair = pair_of(mpartner);
        const int nlive = __popcll(__builtin_amdgcn_ballot_w64(active));
        if (mrole == 2) {
          // receiver: publish the room (live lanes only decrease from now on), look for a donation; when it
          // has nothing left it closes the pair -- unless the donation arrived first
          unsigned st = 0u;
          if (lane == 0) {
            S_mlive[wv] = (unsigned)nlive;
            st = (nlive == 0 || closing) ? atomicCAS(&S_mstate[mpair], 0u, 2u) : __hip_atomic_load(&S_mstate[mpair], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          st = (unsigned)__builtin_amdgcn_readfirstlane((int)st);
          if (st == 1u) {
            take_donation(mpair, mpartner);
            if (lane == 0) S_mstate[mpair] = 2u;
          } else if (st == 2u || nlive == 0 || closing) { mrole = 0; mpoll_mask = budget != 0u ? MCS_MERGE_POLL_MASK : ~0u; }
        } else {
          const int room = 64 - (int)__builtin_amdgcn_readfirstlane((int)S_mlive[mpartner]);
          if (nlive == 0 || closing) {
            if (lane == 0) (void)atomicCAS(&S_mstate[mpair], 0u, 2u);     // nothing to give: the receiver stops polling
            mrole = 0; mpoll_mask = budget != 0u ? MCS_MERGE_POLL_MASK : ~0u;
          } else if (nlive <= MCS_MB_SLOTS && nlive <= room) {
            // donor: tally the pending records (the mailbox is their stack), write the particles, hand over
            drain_events(a, wv, lane, true); ev_pending = 0u;
            if (active) mb_store(wv, below(__builtin_amdgcn_ballot_w64(active)), p, rng, k, evw, phi_prev);
            rb = rng.n - 256u;            // the mailbox has overwritten this wave's ring (it may have to carry on alone)
            unsigned st = 2u;
            if (lane == 0) S_mcount[mpair] = (unsigned)nlive;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) st = atomicCAS(&S_mstate[mpair], 0u, 1u);
            st = (unsigned)__builtin_amdgcn_readfirstlane((int)st);
            if (st == 0u) {          // handed over: this wave is done
              act = 0; p.flags = 0; p.helix = 0; evw = 0;
            }
            mrole = 0; mpoll_mask = budget != 0u ? MCS_MERGE_POLL_MASK : ~0u;     // (st == 2: the receiver had already left -- carry on alone)
          }
        }
      }
      // ---- sliced launches: the budget of trips after exhaustion is spent -- the live particles go to the export buffer
      // (complete lane state, mid-history) and the wave ends; a later launch resumes them
      // (long_draws: a donation taken just above may have brought particles that are not long yet -- everything exported must be long,
      // the order of the next population hangs on it -- so the test is made again on the lanes as they are now; the pair is closed, the
      // wave comes back at the next poll)
      if (closing && (long_draws == 0u || __builtin_amdgcn_ballot_w64(active && rng.n < long_draws) == 0ull)) {
        const unsigned long long am_x = __builtin_amdgcn_ballot_w64(active);
        if (am_x != 0ull) {
          const int nlive = __popcll(am_x);
          const int leader = __ffsll((long long)am_x) - 1;
          unsigned long long base = 0;
          if ((int)lane == leader) base = atomicAdd(a->strag_count, (unsigned long long)nlive);
          base = __shfl(base, leader);
          // (through the record stack as a mailbox, MCS_MB_SLOTS particles at a time: see the import above)
          drain_events(a, wv, lane, true); ev_pending = 0u;
          const unsigned rank = below(am_x);
          double* const box = &S_evf[wv][0][0];
          for (unsigned b0 = 0; b0 < (unsigned)nlive; b0 += MCS_MB_SLOTS) {
            const unsigned cb = (unsigned)nlive - b0 < MCS_MB_SLOTS ? (unsigned)nlive - b0 : MCS_MB_SLOTS;
            if (active && rank >= b0 && rank < b0 + cb) mb_store(wv, rank - b0, p, rng, k, evw, phi_prev);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double* dst = a->strag_out + (base + b0) * MCS_STRAG_WORDS;
            for (unsigned i = lane; i < MCS_MB_WORDS * MCS_MB_SLOTS; i += 64u) {
              const unsigned j = i / MCS_MB_SLOTS, r = i % MCS_MB_SLOTS;
              if (r < cb) dst[r * MCS_STRAG_WORDS + j] = box[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
          act = 0; p.flags = 0; p.helix = 0; evw = 0;
        }
        mrole = 0;
      }
      if (exhausted) refill_at = 64u;
      // nothing left: the loop ends after this pass (which computes on idle lanes and stores nothing) -- no jump
      // out of the middle of the loop, which costs the common pass a branch and half a dozen register copies
      const unsigned long long am_hk = __builtin_amdgcn_ballot_w64(active);
      Lh = (unsigned)__popcll(am_hk);
      done = (am_hk == 0ull) & exhausted;
    }
    {
      [[maybe_unused]] const int na__ = __popcll(__builtin_amdgcn_ballot_w64(active));
      PROF_ADD(0, 1); PROF_ADD(8, na__);
#ifdef MCS_PROF
      if (exhausted) { xp__ += 1; xl__ += (unsigned long long)na__; }
#endif
    }
    double t_clock = p.t_step;      // the clock of a pass uses the time step of the PREVIOUS move
    // ---- the one rare region (see the comment above move_and_detect)
    // (ev and ev_x are false for a particle that has not moved yet)
    // (| and &: one condition, one conditional region -- && / || compile to nested exec-mask regions)
    // (pending-move bits and flags in one test; bits 0-1 of evw are clear for a particle that has not moved yet)
    // (the helix cap is reported through the ev bit by the common pass, see below)
    const int wi = (((evw & 3) | p.flags) | (h.every_pass ? 1 : 0)) & act;
    const bool waits = (p.flags & F_WAIT) != 0;      // (after the housekeeping, which may have released the waiting lanes)
    const bool want = (wi != 0) & !waits;
    PROF_LANES(13, want);
#ifdef MCS_PROF_TAIL
    const bool rare_any__ = PROF_GATE && __builtin_amdgcn_ballot_w64(want) != 0ull;
    unsigned long long tt0__ = 0;
    if (rare_any__) tt0__ = __builtin_amdgcn_s_memtime();
#endif
    bool waits_now = false;
    bool loss_done = false;         // LOSSY: slow_pre has applied this pass's radiative loss for this lane
    // ---- deferral.  Entering the rare region costs the WAVE ~1000 cycles whatever the number of lanes in it, and in the
    // bulk of a launch some lane has a zone crossing pending in three passes out of four (2.4 lanes per entry): half of
    // all VALU issue went into the region for two or three lanes.  A lane can wait: nothing of its particle changes
    // while it is masked out of the common pass (state, RNG stream position and pending-event bits stay as they are), so
    // its history -- every bit of it -- is the same whenever the region is finally entered.  The wave therefore enters
    // only when MCS_DEFER_K lanes have work pending, or one that must not wait (a lane that has just been loaded or resumed,
    // F_NOPARK); the waiting lanes sit out the common pass.
    // After the work counter is exhausted nothing waits (defer_k = 1): the launch then waits for its longest histories;
    // configurations with work in every pass (h.every_pass) never wait either.
    const unsigned long long m_want = __builtin_amdgcn_ballot_w64(want);
    const unsigned long long m_urgent = __builtin_amdgcn_ballot_w64(want & ((wi & F_NOPARK) != 0));
    const bool enter = (m_urgent != 0ull) | ((unsigned)__popcll(m_want) >= defer_k);
    if (MCS_UNLIKELY(want & enter)) {      // (one condition, one conditional region: `enter` is wave-uniform)
      PROF_ADD(12, 1);
#ifdef MCS_PROF_TAIL
#define TT_MARK(slot) do { const unsigned long long tn__ = __builtin_amdgcn_s_memtime(); if (PROF_GATE && (threadIdx.x & 63u) == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) atomicAdd(&S_prof[slot], tn__ - tm__); tm__ = __builtin_amdgcn_s_memtime(); } while (0)
      unsigned long long tm__ = __builtin_amdgcn_s_memtime();
#else
#define TT_MARK(slot) do { } while (0)
#endif
      // (bit 1 of evw: a position threshold was reached, see refresh_thr -- what is due is derived here, with the
      // expressions of move_and_detect; state and (x, x_old) are those the move left)
      const double cm_in = p.cm_val;          // (a batch of the tail ring is built on it: compared at the end of the region)
      const bool moved = (evw & 4) != 0, thr = (evw & 2) != 0;
      bool ev = (evw & 1) != 0, ev_x = false;
      bool up_due = false, xn_due = false, feb_due = false;    // the upward thresholds / the fine-coarse switch of move_and_detect / the upstream FEB: they imply `thr`
      if (__builtin_amdgcn_ballot_w64(thr) != 0ull) {
        const bool fwd = p.x > p.x_old;
        const bool same_zone = (fwd & (p.z_hi > p.x)) | (!fwd & (p.z_lo <= p.x));
        const bool ev_up = ((p.x >= h.x_grid_stop) & ((p.x_old < h.x_grid_stop) | ((p.x_old < p.prp) & (p.x >= p.prp)))) | (p.x > p.x_dt);
        const bool ev_xn = !LOSSY & ((p.x > p.gyro_rad_tot) != (p.xn_per == h.xn_coarse));      // (LOSSY: decided in line, after the loss)
        ev_x = thr & !same_zone;
        up_due = thr & ev_up; xn_due = thr & ev_xn;
        // an injected particle beyond the upstream free-escape boundary (see refresh_thr): slow_post for the escape scalars of
        // the crossing, slow_pre for the exit
        feb_due = thr & (p.i_grid <= h.i_grid_feb) & p.inj & (p.x < h.feb_up);
        ev = ev | up_due | xn_due | feb_due;
      }
      [[maybe_unused]] const bool unusual = (p.flags != 0) | (p.helix >= MCS_HELIX_CAP) | h.every_pass;
      const bool post_pending = moved && (ev || ev_x || (p.flags & F_INJCHK) != 0);
      int end = -1;
      // What is due, from the state the move left (the expressions of move_and_detect).  A lane with nothing but
      // a plain zone crossing, a time cut, a fine/coarse switch or the cos_max refresh that follows one takes the
      // light path: those leave every other derived quantity as it is (see slow_post / slow_pre, whose remaining
      // statements are no-ops then); anything else -- flags, an upward threshold, the age limit, odd
      // configurations -- goes through the full Code Blocks.
      bool t_due = false, age_out = false;
      if (__builtin_amdgcn_ballot_w64(ev) != 0ull) {      // most entries are zone crossings only
        t_due = ev & p.downstream & (p.acctime >= p.t_ev);
        age_out = (h.age_max > 0) & (p.acctime > h.age_max);
      }
      // (energy transfer, particle_loop.jl:235-249, concerns a crossing only for a particle that has not been injected
      // yet and comes from x_old <= 0 -- slow_pre's etf_ev; `inj` as slow_post / plain_crossing leave it)
      const bool etf_due = h.etf & ev_x & !(p.inj | (p.downstream & (p.x < 0))) & (p.x_old <= 0);
      bool full = (p.flags & ~F_CM) != 0 || p.helix >= MCS_HELIX_CAP || h.every_pass || etf_due || h.custom_epsB ||
                  (ev && (h.odd_cfg || up_due || age_out || feb_due));
      TT_MARK(32);
      if (!full && ev_x) full = !plain_crossing(a, h, p, ev_pending);
      TT_MARK(33);
      // most entries are plain crossings and nothing else: one scalar branch skips what follows (a dozen
      // conditional regions at ~40 cycles each even when no lane takes them)
      // A lane that needs the full Code Blocks waits for company (see F_WAIT): decided HERE, before the branch below, so
      // that an entry whose only non-plain lanes are going to wait skips the rest of the region like a plain one.
      // (round 1 let only particles younger than 2048 passes park -- a long history is what the launch waits for --
      // when parking meant a round trip through global memory; waiting in the lane is faster without an age limit)
      if (waiting_on && !exhausted) {
        waits_now = full & moved & ((p.flags & F_NOPARK) == 0);
        p.flags = waits_now ? (p.flags | F_WAIT) : p.flags;       // nothing else is touched: the lane resumes from exactly this state
        full = full & !waits_now;
      }
      if (__builtin_amdgcn_ballot_w64(full || (!waits_now && ((p.flags & F_CM) != 0 || t_due || xn_due))) != 0ull) {
      if (!full && !waits_now) {
        if (p.flags & F_CM) {
          refresh_scatter(a, p, h.aa, h.aa * MP_ * CC_, h.eta);
          p.flags = 0;
        }
        if (t_due) {      // slow_post's time cut (cuts.jl:149-162)
          if (h.do_tcuts && !(p.tcut > h.n_tcuts) && p.acctime >= tcut_next_of(a, h, p.tcut)) {
            tcut_track(a, p.tcut, p.weight, p.ptot_pf);
            p.tcut += 1;
          }
          refresh_time(a, h, p);
        }
        if (xn_due) {     // slow_pre's fine / coarse step (particle_loop.jl:382-385)
          p.xn_per = p.x > p.gyro_rad_tot ? h.xn_coarse : h.xn_fine;
          p.flags |= F_CM;
          refresh_move(a, h, p);
        }
      }
      TT_MARK(34);
      PROF_LANES(16, full);
#ifdef MCS_PROF
      if (exhausted) { xr__ += 1; xf__ += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(full)); }
      {
        const double xup__ = p.x_old < h.x_grid_stop ? h.x_grid_stop : (p.x_old < p.prp ? p.prp : p.x_dt);
        PROF_LANES(22, moved && ((p.x > p.gyro_rad_tot) != (p.xn_per == h.xn_coarse)));
        PROF_LANES(23, moved && p.x >= xup__);
        PROF_LANES(24, moved && p.downstream && p.acctime >= p.t_ev);
        PROF_LANES(25, (p.flags & (F_RS | F_RM)) != 0);
        PROF_LANES(26, full && !unusual && !ev);
        PROF_LANES(27, p.i_grid <= h.i_grid_feb);
        PROF_LANES(28, (p.flags & F_SAVE) != 0);
        PROF_LANES(29, !moved);
      }
#endif
      if (full) {
        PROF_ADD(21, 1);
        TT_MARK(20);
        bool pend = post_pending;
        for (;;) {
          if (pend) { TTG_START(); end = slow_post(a, h, rng, p, phi_prev); if (end >= 0) break; }
          if (!(p.flags & F_B1)) break;
          pend = block1_step(a, h, p, phi_prev, end);
          if (end >= 0) break;
          pend = pend || p.i_grid <= h.i_grid_feb;      // (as the reference: all_flux! in full after every move in the FEB zones)
          t_clock = p.t_step;
        }
        TT_MARK(17);
        if (end < 0) { end = slow_pre(a, h, kc, rng, p, t_clock, LOSSY); loss_done = true; }
        TT_MARK(18);
      }
      if (end >= 0) {
        PROF_ADD(10, 1);
        const int steps = p.helix > MCS_HELIX_CAP ? MCS_HELIX_CAP : p.helix;
        // step totals: three LDS atomics per particle END instead of three 64-bit registers per lane
        atomicAdd(&S_steps[0], (unsigned long long)steps); atomicAdd(&S_steps[1], (unsigned long long)p.n_retro);
        atomicAdd(&S_steps[2], (unsigned long long)rng.n);
        if (p.n_ovr) cnt(a, MCS_IC_TCUT_OVERRUN, p.n_ovr);
        if (end == 0) {
          a->l_save[k] = SLICED ? (uint8_t)(1u | ((long_draws != 0u && rng.n >= long_draws) ? 4u : 0u)) : (uint8_t)1;
          a->sv.weight[k] = p.weight; a->sv.ptot_pf[k] = p.ptot_pf; a->sv.pb_pf[k] = p.pb_pf; a->sv.x_PT_cm[k] = p.x;
          a->sv.xn_per[k] = p.xn_per;
          a->sv.prp_x_cm[k] = p.x < p.prp ? p.prp : p.x * 1.1;   // quirk Q7
          a->sv.acctime_sec[k] = p.acctime; a->sv.phi_rad[k] = p.phi;
          a->sv.meta[k] = mcs_pack_meta(p.i_grid, p.tcut, p.downstream, p.inj);
          cnt(a, MCS_IC_COUNT);
        } else {
          // (status byte: 1 saved, 2 ended, + 4 long -- KArgs::long_draws; 0 = not resolved yet, which only a sliced run ever sees)
          a->l_save[k] = SLICED ? (uint8_t)(2u | ((long_draws != 0u && rng.n >= long_draws) ? 4u : 0u)) : (uint8_t)2;
          if (p.npush < 2) {
            // particle_finish! (transform, two bin look-ups, up to five tallies) is deferred like the zone-crossing
            // tallies: a record on the wave's stack, tallied 64 at a time (a lane pushes at most two records per pass)
            push_record(p, p.ig3, -1, (1u << 28) | ((uint32_t)end << 25));
          } else {
            particle_finish(a, end, p.pb_pf, p.p_perp, p.gam_pf, p.phi, p.weight, p.ig3);
          }
        }
        cnt(a, MCS_IC_REASON0 + end);
        if (a->f_reason) {
          a->f_reason[k] = end; a->f_helix[k] = p.helix; a->f_retro[k] = p.n_retro; a->f_ptot[k] = p.ptot_pf; a->f_x[k] = p.x;
        }
        act = 0;
        p.flags = 0; p.helix = 0;       // an idle lane must not look as if it had work
        TT_MARK(19);
      }
      }
      // the position thresholds of the common pass, from the state the lane leaves the region with (a waiting lane
      // comes back with its pending move and is refreshed then)
      if (!waits_now) refresh_thr<LOSSY>(h, p);
      if (MCS_UNLIKELY(ring_on)) rb = (p.cm_val != cm_in) ? rng.n - 256u : rb;      // the cone changed: this lane's batch is void
    }
    const bool frozen = want & !enter;  // waits for the region: sits out this pass
#ifdef MCS_PROF_TAIL
    if (rare_any__) { const unsigned long long dt__ = __builtin_amdgcn_s_memtime() - tt0__; if (lane == 0) { atomicAdd(&S_prof[30], dt__); atomicAdd(&S_prof[31], 1ull); } }
#endif
    // ---- MCS_PASSES_PER_ITER common passes per trip through the loop header.  The header (housekeeping test, the want /
    // enter logic, two branches: ~55 instructions) is a sixth of a pass; a lane that comes out of a pass with nothing
    // pending runs the next one at once, a lane with an event or a flag sits the rest of the trip out (it is one of the
    // few per cent of lanes that have an event in a given pass) and meets the rare region at the next header, as before.
    bool run = !frozen & ((p.flags & F_WAIT) == 0);
    // (who goes on after a pass is lane-mask arithmetic on the pass's own compare results -- scalar instructions, which
    // issue beside the other wave's vector ones -- instead of a round trip through the bits of evw: flags do not change
    // in a pass)
    const bool may_go_on = (p.flags == 0) & !h.every_pass;
    bool stopped = false;
    // ---- tail loop.  After the queue is exhausted the launch waits for its longest histories, a few lanes per wave at the end and
    // finally ONE particle alone on its SIMD for thousands of passes (the helix cap allows 10^4), and an instruction costs a wave
    // the same with one lane enabled as with 64: what counts is the number of instructions between two passes of that particle.
    // A wave with at most tail_loop_L live particles therefore leaves the six-pass trip (header, housekeeping test, want / enter
    // logic, per-pass ring test, lane-mask arithmetic: ~120 instructions and ~8 branches per pass of overhead beside the ~95 of
    // the state-dependent half of a pass) and runs the passes in a loop of their own: the draw-dependent half from the tail ring,
    // the state-dependent half (scattering_rest, clock, move, thresholds), ONE exit test.  It leaves the loop when a live lane
    // has an event or a flag (the rare region is due), or has used its ring entries up (the next trip builds a new batch).
    // Same statements on the same values in the same order for every particle: results are unchanged.
    bool tail_fast = false;
    unsigned long long am_t = 0ull;
    if constexpr (TAIL_LOOP) {
      if (MCS_UNLIKELY(ring_on)) {
        am_t = __builtin_amdgcn_ballot_w64(active);
        // (at least one live lane, and every live lane runs -- nothing is frozen or waits once the queue is exhausted)
        tail_fast = (Lh <= tail_loop_L) & (am_t != 0ull) & (__builtin_amdgcn_ballot_w64(active & !run) == 0ull);
      }
    }
    if (MCS_UNLIKELY(tail_fast)) {
      // (every lane computes, the idle ones on whatever state they hold -- as in the common pass of the bulk, nothing of it is stored --:
      // no exec-mask region in the loop; it runs while EVERY live lane goes on)
      unsigned jj = (rng.n - rb) >> 1;
      if (__builtin_amdgcn_ballot_w64(active && jj >= ringD) != 0ull) {
        build_batch(am_t, (unsigned)__popcll(am_t));
        jj = 0u;
      }
      unsigned e = (rrow + jj) & 63u;
      const unsigned long long m_halt = __builtin_amdgcn_ballot_w64(!may_go_on);      // (flags do not change in a pass)
      TailK K;
      K.halfpi = sconst(MCS_PIO2_DD_0); K.pio2_lo = sconst(MCS_PIO2_DD_1); K.twopi = sconst(MCS_TWOPI_DD_0); K.twopi_lo = sconst(MCS_TWOPI_DD_1);
      K.inv_twopi = sconst(MCS_INV_TWOPI); K.sin_ul = sconst(0x1.fffffffffffffp-1); K.dmin = sconst(2.2250738585072014e-308);
#pragma nounroll
      for (;;) {
        PROF_LANES(9, active); PROF_ADD(7, 1);
        p.helix += 1;
        if (!h.dont_scatter) {
          rng.n += 2u;
          const double cos_d = S_evf[wv][0][128u + e], sin_d = S_evf[wv][1][128u + e], c_ps = S_evf[wv][2][128u + e], ssd = S_evf[wv][6][128u + e];
          scattering_rest_k(p, kc, K, cos_d, sin_d, c_ps, ssd);
        }
        p.acctime = p.acctime + t_clock * p.c_gef;
        p.n_ovr += (unsigned)p.ovr_inc;
        const bool ev_time = p.acctime >= p.c_tev;
        bool x1;
        const bool e1 = move_and_detect_thr(a, h, p, phi_prev, x1, true, &K) | ev_time | (p.helix >= MCS_HELIX_CAP);
        evw = (e1 ? 5 : 4) | (x1 ? 2 : 0);
        e = (e + 1u) & 63u; jj += 1u;
        t_clock = p.t_step;
        const unsigned long long m_stop = __builtin_amdgcn_ballot_w64(e1 | x1 | (jj >= ringD)) | m_halt;
        if ((m_stop & am_t) != 0ull) break;
      }
      // ---- why the loop ended is, most of the time, a zone crossing with nothing else going on.  What the rare region does for such a
      // lane (see `full` and plain_crossing there: the same expressions on the same state) is done HERE, behind the loop: the lane comes
      // to the next loop header with nothing pending, and the trip skips the region (its skeleton is ~1 500 cycles for a wave alone on
      // its SIMD).  Anything else -- a time event, the cap, an upward threshold, the fine / coarse switch, the upstream boundary, a
      // crossing that plain_crossing refuses -- is left as the pass left it: the region derives what is due from that state, as always.
      // (The register mirror of the stack height lags this trip's pushes: brought up to date first, as the header would.)
      if (MCS_TAIL_CROSS) {
        ev_pending += (unsigned)(__popcll(__builtin_amdgcn_ballot_w64(p.npush > 0)) + __popcll(__builtin_amdgcn_ballot_w64(p.npush > 1)));
        p.npush = 0;
        if (ev_pending < 64u) {
          if (active & ((evw & 3) == 2) & (p.flags == 0) & !(h.etf | h.odd_cfg | h.custom_epsB | h.every_pass)) {
            const bool fwd = p.x > p.x_old;
            const bool same_zone = (fwd & (p.z_hi > p.x)) | (!fwd & (p.z_lo <= p.x));
            const bool ev_up = ((p.x >= h.x_grid_stop) & ((p.x_old < h.x_grid_stop) | ((p.x_old < p.prp) & (p.x >= p.prp)))) | (p.x > p.x_dt);
            const bool ev_xn = (p.x > p.gyro_rad_tot) != (p.xn_per == h.xn_coarse);
            const bool feb_due = (p.i_grid <= h.i_grid_feb) & p.inj & (p.x < h.feb_up);
            if (!(ev_up | ev_xn | feb_due)) {
              const bool handled = same_zone ? true : plain_crossing(a, h, p, ev_pending);      // (same zone: a threshold reached with nothing due)
              if (handled) { refresh_thr<false>(h, p); evw = 4; }
            }
          }
        }
      }
    } else
#pragma unroll
    for (int rep = 0; rep < MCS_PASSES_PER_ITER; ++rep) {
      if (rep > 0) {
        run = run & may_go_on & !stopped;
        t_clock = p.t_step;       // the previous move is now the one of the pass before, made with the current time step
      }
      stopped = false;
      // ---- tail ring (see above): the draw-dependent part of this pass's scatter, from the ring if the wave has one
      bool got = false;
      unsigned ridx = 0u;          // the ring entry this pass takes its four values from (read where they are used: nothing to merge on the bulk path)
      if (MCS_UNLIKELY(ring_on)) {
        // (Lh, not L: the records pushed in this pass's rare region came from up to Lh lanes -- at most 63 + 2 Lh are pending)
        const unsigned long long am = __builtin_amdgcn_ballot_w64(active);
        const unsigned L = (unsigned)__popcll(am);
        if ((L >= 1u) & (Lh <= 32u)) {
          unsigned jj = (rng.n - rb) >> 1;
          if (__builtin_amdgcn_ballot_w64(active && jj >= ringD) != 0ull) {
            build_batch(am, L);
            jj = 0u;
          }
          ridx = (rrow + jj) & 63u;
          got = true;
        } else {
          rb = rng.n - 256u;            // the top of the stack may be overwritten by records now: no batch survives
        }
      }
      // ---- the common pass, for every lane that is not waiting (idle lanes compute on stale state; nothing is stored)
      PROF_LANES(9, run & active); PROF_ADD(7, 1);
      if constexpr (LOSSY) {
        // slow_pre's radiative loss (particle_loop.jl:302-326 -> :578-592) and what hangs on the momentum, for a lane on which
        // every other statement of slow_pre is a no-op: no flag set (so not near p_max, not to be saved, zone loaded), nothing
        // pending from the last move (the clock and the position thresholds -- the upstream FEB among them -- stop a lane before this).  The
        // momentum only decreases here, so F_NEARP / F_SAVE stay clear; the cone is refreshed with the xn_per of before this
        // pass's fine / coarse decision, as slow_pre does; t_clock is the time step of the previous move.
        if (run && !(rep == 0 && loss_done)) {
          const double bmag = S_bt[p.ig3];
          const double ptot_old = p.ptot_pf;
          const double B_CMB_loc = l_bcmb * p.z_gef;
          double pn;
          {   // radiation_loss (particle_loop.jl:578-592)
            const double dlnp = MCS_RAD_LOSS_FAC * (bmag * bmag + B_CMB_loc * B_CMB_loc) * ptot_old * t_clock;
            if (dlnp > 1.0e-2) pn = fdiv(ptot_old, 1 + dlnp); else pn = ptot_old * (1 - dlnp);
          }
          if (MCS_UNLIKELY(pn <= 0)) {
            // the pass ends here with finish code 4: floors as in slow_pre, the pass counted; slow_pre returns 4 at the next header
            p.ptot_pf = MCS_FLOOR; p.pb_pf = MCS_FLOOR; p.p_perp = MCS_FLOOR; p.gam_pf = 1;
            p.helix += 1;
            p.flags |= F_LOST;
            run = false;
          } else {
            p.ptot_pf = pn;
            p.gam_pf = mcsm::hypot1(fdiv(p.ptot_pf, h.mc));
            const double ratio = fdiv(p.ptot_pf, ptot_old);
            p.pb_pf *= ratio;
            p.p_perp *= ratio;
            p.gyro_rad_tot = p.ptot_pf * CC_ * p.gyro_denom;
            p.gyro_rad = p.p_perp * CC_ * p.gyro_denom;
            refresh_scatter_k(p, kc, h.aa, h.aa * MP_ * CC_, h.eta, l_pe_crit, l_game_crit);
            refresh_dtest_k(h, p, l_pe_crit, l_game_crit);
            p.xn_per = p.x > p.gyro_rad_tot ? h.xn_coarse : h.xn_fine;
            {   // refresh_move
              const double rx = rcp_refined(p.xn_per);
              p.dphi = div_r(TWOPI_, p.xn_per, rx);
              p.t_step = div_r(p.gyro_period, p.xn_per, rx);
              p.rg_val = rcp_refined(p.gam_pf * (h.aa * MP_));
            }
            refresh_thr<true>(h, p);
          }
        }
      }
      if (run) {
        p.helix += 1;
        if (!h.dont_scatter) {
          double cos_d, sin_d, c_ps, ssd;
          const uint32_t jd = rng.n;
          rng.n = jd + 2u;
          if (got) { cos_d = S_evf[wv][0][128u + ridx]; sin_d = S_evf[wv][1][128u + ridx]; c_ps = S_evf[wv][2][128u + ridx]; ssd = S_evf[wv][6][128u + ridx]; }
          else {
            double U1, s_ps;
            scatter_draws(rng.k0, rng.k1, jd >> 1, kc, U1, s_ps, c_ps);
            scatter_cone(U1, s_ps, 1 - p.cm_val, cos_d, sin_d, ssd);
          }
          scattering_rest(p, kc, cos_d, sin_d, c_ps, ssd);
        }
        {
          // (acctime runs downstream only, particle_loop.jl:348-351: upstream c_gef is 0 and c_tev +inf, see refresh_thr)
          p.acctime = p.acctime + t_clock * p.c_gef;
          p.n_ovr += (unsigned)p.ovr_inc;
          const bool ev_time = p.acctime >= p.c_tev;
          bool x1;
          // (the helix cap rides on the ev bit: the pass about to start would be number cap + 1; slow_post finds nothing
          // due for such a lane and slow_pre ends the particle, quirk Q5)
          const bool e1 = move_and_detect_thr(a, h, p, phi_prev, x1, !LOSSY) | ev_time | (p.helix >= MCS_HELIX_CAP);
          evw = (e1 ? 5 : 4) | (x1 ? 2 : 0);
          stopped = e1 | x1;
        }
      }
    }
  }

#ifdef MCS_PROF
  if (lane == 0) { g_wave[gw__][2] = __builtin_amdgcn_s_memrealtime(); g_wave[gw__][4] = xp__; g_wave[gw__][6] = xl__;
    g_wave[gw__][5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
    g_wave[gw__][7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); }   // HW_REG_XCC_ID
#endif
  // ---- the wave's remaining tally records, then the LDS staging
  drain_events(a, wv, lane, true);
  __syncthreads();
#ifdef MCS_PROF
  if (threadIdx.x < MCS_NPROF && S_prof[threadIdx.x]) atomicAdd(&g_prof[threadIdx.x], S_prof[threadIdx.x]);
#endif
  block_flush(a);
}

#undef active

extern "C" __global__ void __launch_bounds__(256, MCS_WAVES_PER_SIMD) mcs_k_transport(const KArgs* __restrict__ ka) {
  transport_body<false>(ka);
}
extern "C" __global__ void __launch_bounds__(256, MCS_WAVES_PER_SIMD) mcs_k_transport_plain(const KArgs* __restrict__ ka) {
  transport_body<true>(ka);
}
extern "C" __global__ void __launch_bounds__(256, MCS_WAVES_PER_SIMD) mcs_k_transport_lossy(const KArgs* __restrict__ ka) {
  transport_body<false, true>(ka);
}
extern "C" __global__ void __launch_bounds__(256, MCS_WAVES_PER_SIMD) mcs_k_transport_plain_etf(const KArgs* __restrict__ ka) {
  transport_body<true, false, true>(ka);
}
extern "C" __global__ void __launch_bounds__(256, MCS_WAVES_PER_SIMD) mcs_k_transport_sliced(const KArgs* __restrict__ ka) {
  transport_body<false, false, false, true>(ka);
}
// the sliced form of the PLAIN kernel: what the pipelined pcut loop launches for the common configuration (mcs_run_pcuts_pipelined)
extern "C" __global__ void __launch_bounds__(256, MCS_WAVES_PER_SIMD) mcs_k_transport_plain_sliced(const KArgs* __restrict__ ka) {
  transport_body<true, false, false, true>(ka);
}
// ... and of the two other specialisations (electrons with radiative losses; ions with energy transfer): the species mix of BASELINE config[4]
extern "C" __global__ void __launch_bounds__(256, MCS_WAVES_PER_SIMD) mcs_k_transport_lossy_sliced(const KArgs* __restrict__ ka) {
  transport_body<false, true, false, true>(ka);
}
extern "C" __global__ void __launch_bounds__(256, MCS_WAVES_PER_SIMD) mcs_k_transport_plain_etf_sliced(const KArgs* __restrict__ ka) {
  transport_body<true, false, true, true>(ka);
}

#ifdef MCS_PROF
extern "C" int mcs_prof_waves(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave), sizeof(unsigned long long) * 8192 * 8) != hipSuccess;
}
extern "C" int mcs_prof_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * MCS_NPROF) != hipSuccess) return 1;
  if (reset) { unsigned long long z[MCS_NPROF] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif
#include "mcs_transport_ws.inc"
#include "mcs_transport_f32.inc"

extern "C" size_t mcs_transport_smem_bytes(int n_grid, int n_tcuts) { (void)n_grid; (void)n_tcuts; return 0; }   // static LDS
extern "C" int mcs_transport_max_entries(void) { return MCS_MAXNE; }

// `a_dev`: device copy of the launch constants (written by the caller on `st`).
// `kind`: 0 the general kernel; 1 / 2 / 6: the host has checked the conditions of the PLAIN / LOSSY / PLAIN_ETF specialisation (see transport_body).
extern "C" hipError_t mcs_launch_transport(const KArgs* a_dev, int kind, int blocks, int threads, hipStream_t st) {
  if (kind == 1) hipLaunchKernelGGL(mcs_k_transport_plain, dim3(blocks), dim3(threads), 0, st, a_dev);
  else if (kind == 2) hipLaunchKernelGGL(mcs_k_transport_lossy, dim3(blocks), dim3(threads), 0, st, a_dev);
  else if (kind == 6) hipLaunchKernelGGL(mcs_k_transport_plain_etf, dim3(blocks), dim3(threads), 0, st, a_dev);
  else if (kind == 10) hipLaunchKernelGGL(mcs_k_transport_sliced, dim3(blocks), dim3(threads), 0, st, a_dev);
  else if (kind == 11) hipLaunchKernelGGL(mcs_k_transport_plain_sliced, dim3(blocks), dim3(threads), 0, st, a_dev);
  else if (kind == 12) hipLaunchKernelGGL(mcs_k_transport_lossy_sliced, dim3(blocks), dim3(threads), 0, st, a_dev);
  else if (kind == 13) hipLaunchKernelGGL(mcs_k_transport_plain_etf_sliced, dim3(blocks), dim3(threads), 0, st, a_dev);
  else if (kind == 7) hipLaunchKernelGGL(mcs_k_transport_ws, dim3(blocks), dim3(threads), 0, st, a_dev);
  else if (kind == 8) hipLaunchKernelGGL(mcs_k_transport_ws_etf, dim3(blocks), dim3(threads), 0, st, a_dev);
  else hipLaunchKernelGGL(mcs_k_transport, dim3(blocks), dim3(threads), 0, st, a_dev);
  return hipGetLastError();
}
