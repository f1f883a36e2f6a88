// mcs_device.h -- device-side data structures shared by the gfx950 kernels.
#ifndef MCS_DEVICE_H
#define MCS_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcs.h"

// Device population, struct-of-arrays in HBM (coalesced: lane l touches element base+l).
// The 12 per-particle fields of src/particle_loop.jl:48-59; the two indices and two
// flags (Int64/Int64/Bool/Bool in the reference, 18 B) are packed into one 32-bit word:
//   bits 0-15 grid zone, 16-23 tcut, 24 downstream, 25 inj.        68 B per particle.
struct DevPop {
  double *weight, *ptot_pf, *pb_pf, *x_PT_cm, *xn_per, *prp_x_cm, *acctime_sec, *phi_rad;
  uint32_t* meta;
};

__host__ __device__ inline uint32_t mcs_pack_meta(int grid, int tcut, int downstream, int inj) {
  return (uint32_t)(grid & 0xffff) | ((uint32_t)(tcut & 0xff) << 16) | ((uint32_t)(downstream & 1) << 24) |
         ((uint32_t)(inj & 1) << 25);
}

struct DevTables {
  const double *x_grid, *ux, *uz, *utot, *gsf, *gef, *btot, *theta;   // n_grid+2 entries each
  const double *pcuts, *tcuts, *x_spec, *inj_fracs, *eps_target;
  int n_pcuts, n_tcuts, n_xspec;
};

// Everything one launch of the transport kernel needs (passed by value: kernarg segment).
struct KArgs {
  mcs_params P;
  mcs_layout L;
  DevTables tb;
  DevPop in, sv;
  uint8_t* l_save;
  double* T;                 // flat fp64 tallies (layout L)
  unsigned long long* I;     // int64 tallies
  // species / pcut scalars (src/main_loops.jl:97-105,205)
  double aa, zzq, m, mc, pmax_cutoff, density, ewf, inj_frac;
  double pcut, pcut_prev;
  int i_iter, i_ion, i_pcut;
  long long n;               // local population size
  const long long* n_dev;    // or, when not null: where the population size is read from on the device (mcs_run_pcuts_fused: the host
                             // does not know it -- i_mult and the size of the next population are decided on the device)
  long long i_prt_offset;    // global i_prt of local particle 0, minus 1
  long long i_prt_stride;    // global i_prt of local particle k = i_prt_offset + 1 + k * i_prt_stride (1: a contiguous shard)
  const long long* gidx;     // or, when not null: global 0-based index of local particle k (mcs_run_pcut_indexed); i_prt = gidx[k] + 1
  unsigned long long seed_base;   // iseed_mod - i_prt   (src/particle_loop.jl:35-40)
  unsigned long long* work_counter;   // next unclaimed particle
  unsigned long long* n_saved;
  int32_t *f_reason, *f_helix, *f_retro;
  double *f_ptot, *f_x;
  double* tally_rep;         // MCS_TALLY_REPLICAS private copies of T[0 .. rep_n) for the per-event tallies (the LDS-staged
  long long rep_n;           // sums are flushed into T itself); null / 0: tally into T
  int wait_full;             // 1: a lane that needs the full Code Blocks waits for company (MCS_PARK=0 turns it off: A/B runs)
  int tail_ring;             // 1: sparse waves precompute the draw-dependent part of future scatters on idle lanes (MCS_TAIL_RING=0 turns it off)
  int refill_min;            // idle lanes a wave collects before it claims new particles (MCS_REFILL_MIN = 12)
  int defer_k;               // lanes with pending rare work a wave collects before entering the rare region (MCS_DEFER_K; 1 = never wait)
  int retro_cap;             // inner steps after which one retro_time walk is ended (MCS_RETRO_CAP; tests lower it)
  int tail_merge;            // 1: sparse waves of a block consolidate after exhaustion (MCS_TAIL_MERGE=0 turns it off)
  // ---- sliced launches (DESIGN.md "Sliced tail"): a launch's queue is [resume list | fresh index range]; a wave that has found
  // the queue exhausted makes `budget_trips` more trips through the loop header, then writes the complete lane state of its
  // live particles to `strag_out` and ends; a later launch resumes them.  State and RNG stream position travel with the
  // particle (the mailbox format of the tail consolidation), so a history is the same however often it is suspended.
  const double* strag_in;    // lane states to resume first: entry e at strag_in[e * MCS_STRAG_WORDS ..]
  long long n_resume;        // entries of strag_in
  long long fresh_lo;        // then the fresh particles fresh_lo .. n-1 of `in` (queue position c >= n_resume is particle fresh_lo + c - n_resume)
  double* strag_out;         // where this launch exports to (same layout; room for one entry per lane of the launch)
  unsigned long long* strag_count;   // entries written to strag_out (device counter, zeroed by the host)
  int budget_trips;          // 0: no export, every wave runs its particles to their end
  int ws_pop_max;            // wave-specialised kernel (mcs_transport_ws.inc): particles a block holds at most (lanes + queued)
  int ws_serve_min;          // ... and the number of pending particles at which a wave serves them (64)
  int claim_max;             // live particles a wave holds at most (64; less spreads a sparse queue over the waves of the chip:
                             // a pass costs a wave the same with 1 live lane as with 64, but every live lane's rare work stalls the others)
  // ---- long histories (mcs_run_pcuts_pipelined; DESIGN.md "Pipelined pcuts").  A particle is LONG in a pcut when it has taken at least
  // `long_draws` random draws by the end of its history there -- a property of the particle alone (its stream position), whatever the
  // schedule.  A sliced launch with long_draws > 0 exports a wave's live particles once the queue is exhausted and every one of them has
  // reached long_draws (so everything exported is long; `budget_trips` only has to be non-zero), and every particle's status byte carries
  // the verdict: l_save = 1 saved, 2 ended, + 4 when long.  0: off.
  unsigned int long_draws;
  // ---- tail loop (round 4; transport_body "tail loop"): once the queue is exhausted, a wave with at most `tail_loop` live particles runs
  // its common passes in a tight loop of their own -- one pass per trip of THAT loop, the draw-dependent half from the tail ring, no
  // housekeeping, no six-pass bookkeeping -- until some live lane has an event or the ring batch is used up.  0: off (MCS_TAIL_LOOP=0).
  int tail_loop;
};

// Replicas of the tally buffer.  Particles of one population pile their tallies onto
// a few cache lines -- cold particles most of all -- and a memory-side fp64 atomic takes ~12 ns per line: the first
// pcut of a species ran at the atomic rate (21 ms instead of 6).  Each block adds to replica blockIdx % R; the host
// folds the replicas into T before anything reads the tallies (mcs_sync, mcs_read_tallies, the consumers, ...).
#ifndef MCS_TALLY_REPLICAS
#define MCS_TALLY_REPLICAS 16
#endif

// One pcut of a fused species loop (mcs_run_pcuts_fused): what the host would have read back after every pcut
struct PcutDev { long long n_use, n_saved, i_mult, n_new; };

// zone-crossing tally records staged in LDS by the transport kernel
#define MCS_EV_F64 8         // pb_pf, p_perp, ptot_pf, gam_pf, phi, weight, x, x_old
#define MCS_STRAG_WORDS 36   // doubles per exported lane state (state_store in mcs_transport.hip)
#define MCS_EV_CAP 192       // records per wave: < 64 after the drain + at most 2 per lane in one pass

#endif
