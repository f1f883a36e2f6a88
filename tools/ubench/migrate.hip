// Micro-benchmark for the design of DESIGN.md section 7, item 1 (rare work leaves the wave that runs the passes: particles
// handed between lanes and waves through LDS queues).  What does moving a particle cost?  A K1 lane state is 36 doubles.
// Each wave runs "trips" of TRIP dependent fp64 FMAs per lane on a 36-double state (a stand-in for six common passes:
// ~1700 instructions); in every trip the lanes of a changing ~1/6 of the wave (K1: 10.7 of 64 lanes have an event per trip)
//   mode 1: store their 36 doubles into a block-wide LDS pool (slots taken with one ds atomic per wave) and load 36 doubles of
//           another slot -- a particle leaves for the pending queue, one arrives from the ready queue;
//   mode 2: as 1, and every sixth trip the whole wave swaps all 64 states (the wave that serves the queue stores its running
//           particles and loads 64 pending ones);
//   mode 0: nothing (the baseline).
// 2 blocks of 256 threads per CU (LDS-limited, as K1), all CUs busy.  Prints shader cycles per trip and wave.
// Build: hipcc --offload-arch=gfx950 -O3 migrate.hip -o migrate
#include <hip/hip_runtime.h>
#include <cstdio>
#define NW 36
#define CAP 176               // pool slots per block: 176 x 36 x 8 B = 50.7 KB (the record stacks of K1 are 52 KB)
#define TRIP 44               // FMAs per state word per trip: 36 x 44 = 1584 dependent-chain FMAs (~ a K1 trip)
__global__ void __launch_bounds__(256, 2) k(double* out, unsigned long long* cyc, int mode, int trips) {
  __shared__ double pool[NW][CAP];
  __shared__ unsigned head;
  __shared__ double pad[3000];                 // 24 KB more: two blocks per CU, as K1
  if (threadIdx.x == 0) head = 0;
  for (int i = threadIdx.x; i < NW * CAP; i += 256) (&pool[0][0])[i] = 1.0 + i * 1e-9;
  for (int i = threadIdx.x; i < 3000; i += 256) pad[i] = 0.0;
  __syncthreads();
  double s[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) s[w] = 0.5 + threadIdx.x * 1e-6 + w * 1e-3 + pad[(threadIdx.x + w) % 3000];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < trips; ++it) {
#pragma unroll 1
    for (int r = 0; r < TRIP; ++r) {
#pragma unroll
      for (int w = 0; w < NW; ++w) s[w] = __builtin_fma(s[w], 0.999999, 1e-9);
    }
    if (mode) {
      const bool whole = (mode == 2) && (it % 6 == 5);
      const bool mv = whole || ((lane * 7u + (unsigned)it) % 6u == 0u);
      const unsigned long long m = __ballot(mv);
      const unsigned n = (unsigned)__popcll(m);
      unsigned base = 0;
      if (lane == (unsigned)(__ffsll((long long)m) - 1)) base = atomicAdd(&head, n);
      base = __shfl(base, __ffsll((long long)m) - 1);
      if (mv) {
        const unsigned slot = (base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))) % CAP;
        const unsigned from = (slot + 61u) % CAP;
#pragma unroll
        for (int w = 0; w < NW; ++w) pool[w][slot] = s[w];
#pragma unroll
        for (int w = 0; w < NW; ++w) s[w] = pool[w][from] * 0.5 + 0.25;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double acc = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) acc += s[w];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (lane == 0) atomicAdd(cyc, t1 - t0);
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int blocks = pr.multiProcessorCount * 2, trips = 600;
  double* out; unsigned long long* cyc;
  hipMalloc(&out, (size_t)blocks * 256 * 8); hipMalloc(&cyc, 8);
  double base = 0;
  for (int mode = 0; mode < 3; ++mode) {
    double best = 1e300;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(cyc, 0, 8);
      k<<<blocks, 256>>>(out, cyc, mode, trips);
      unsigned long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const double per = (double)h / ((double)blocks * 4 * trips);
      if (per < best) best = per;
    }
    if (mode == 0) base = best;
    printf("mode %d: %8.0f memtime ticks (100 MHz) per trip and wave", mode, best);
    if (mode) printf("  (+%.0f = +%.1f %% over the bare trip)", best - base, 100.0 * (best - base) / base);
    printf("\n");
  }
  printf("(%d CUs, %d blocks of 4 waves, 2 waves per SIMD; a trip = %d fp64 FMAs per lane)\n", pr.multiProcessorCount, blocks, NW * TRIP);
  return 0;
}
