// Micro-benchmark: issue cost of single gfx950 instructions, for ONE wave and for TWO waves on one SIMD.
// Every test is a loop of 16 x 64 copies of one instruction written in inline asm (dependent chain on one
// register unless noted), timed with s_memtime by lane 0.  With <<<1, 64>>> one wave owns its SIMD; with
// <<<1, 512>>> (<<<1, 1024>>>) the CU holds 8 (16) waves = 2 (4) per SIMD and each reports its own time: the
// per-instruction cost of a wave that shares its SIMD.  Build: hipcc --offload-arch=gfx950 -O3 issue.hip -o issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

#define BODY(ASM, ...)                                              \
  _Pragma("unroll 1") for (int it = 0; it < 16; ++it) {            \
    asm volatile(R64(ASM "\n") : __VA_ARGS__);                     \
  }

template <int MODE> __global__ void k(double* out, unsigned long long* cyc, double x0, unsigned u0) {
  double a = x0 + threadIdx.x * 1e-9, b = a + 1.25, c = 0.999999, nz = -0.0, one = 1.0;
  unsigned ua = u0 + threadIdx.x, ub = ua * 3u + 1u, uc = 0x9E3779B9u;
  unsigned long long w = ((unsigned long long)ua << 32) | ub;
  unsigned long long sa = u0, sb = 12345;
  unsigned s32a = u0, s32b = 77u;
  asm volatile("" : "+v"(nz), "+v"(one), "+v"(c));
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
  if (MODE == 0) BODY("v_fma_f64 %0, %0, %1, %2", "+v"(a) : "v"(c), "v"(b))
  if (MODE == 1) BODY("v_mul_f64 %0, %0, %1", "+v"(a) : "v"(c))
  if (MODE == 2) BODY("v_add_f64 %0, %0, %1", "+v"(a) : "v"(c))
  if (MODE == 3) BODY("v_fma_f64 %0, %0, %1, %2", "+v"(a) : "v"(c), "v"(nz))          // a*c as an fma with -0.0
  if (MODE == 4) BODY("v_fma_f64 %0, %0, %1, %2", "+v"(a) : "v"(one), "v"(c))         // a+c as an fma with 1.0
  if (MODE == 5) BODY("v_mad_u64_u32 %0, vcc, %1, %2, %0", "+v"(w) : "v"(ua), "v"(uc) : "vcc")
  if (MODE == 6) BODY("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", "+v"(ua) : "v"(ub), "v"(uc))
  if (MODE == 7) BODY("v_add_u32 %0, %0, %1", "+v"(ua) : "v"(uc))
  if (MODE == 8) BODY("v_cndmask_b32 %0, %0, %1, vcc", "+v"(ua) : "v"(ub) : )
  if (MODE == 9) BODY("v_cmp_lt_f64 vcc, %0, %1", : "v"(a), "v"(b) : "vcc")
  if (MODE == 10) BODY("v_cvt_f64_u32 %0, %1", "+v"(a) : "v"(ua))
  if (MODE == 11) BODY("v_rcp_f64 %0, %0", "+v"(a) : )
  if (MODE == 12) BODY("v_rsq_f64 %0, %0", "+v"(a) : )
  if (MODE == 13) BODY("v_trunc_f64 %0, %0", "+v"(a) : )
  if (MODE == 14) BODY("v_ldexp_f64 %0, %0, %1", "+v"(a) : "v"(ua))
  if (MODE == 15) BODY("v_mov_b32 %0, %1", "+v"(ua) : "v"(ub))
  if (MODE == 16) BODY("s_add_u32 %0, %0, %1", "+s"(s32a) : "s"(s32b) : "scc")
  if (MODE == 17) BODY("s_and_b64 %0, %0, %1", "+s"(sa) : "s"(sb) : "scc")
  if (MODE == 18) BODY("v_readlane_b32 %0, %1, 3", "+s"(s32a) : "v"(ua))
  if (MODE == 19) BODY("v_fmac_f64 %0, %1, %2", "+v"(a) : "v"(c), "v"(b))
  if (MODE == 20) BODY("v_mul_f64 %0, %1, %2", "=v"(a) : "v"(c), "v"(b))             // independent (no chain)
  if (MODE == 21) BODY("v_fma_f64 %0, %1, %2, %3", "=v"(a) : "v"(c), "v"(b), "v"(nz))  // independent
  if (MODE == 22) BODY("v_mul_lo_u32 %0, %0, %1", "+v"(ua) : "v"(uc))
  if (MODE == 23) BODY("v_cmp_lt_f64 %0, %1, %2", "=s"(sa) : "v"(a), "v"(b))          // compare into an SGPR pair
  if (MODE == 24) BODY("v_mul_f64 %0, %0, %1", "+v"(a) : "s"(sb))                     // SGPR operand
  if (MODE == 25) BODY("s_nop 0", : : )
  if (MODE == 26) BODY("v_cndmask_b32_e64 %0, %0, %1, %2", "+v"(ua) : "v"(ub), "s"(sb))      // mask in an SGPR pair (what hipcc emits)
  if (MODE == 27) BODY("v_cndmask_b32 %0, %1, %2, vcc", "=v"(ua) : "v"(ub), "v"(uc))        // vcc mask, no chain
  if (MODE == 28) BODY("v_cndmask_b32_e64 %0, %1, %2, %3", "=v"(ua) : "v"(ub), "v"(uc), "s"(sb))   // sgpr mask, no chain
  if (MODE == 29) BODY("v_cmp_lt_f64 vcc, %1, %2\nv_cndmask_b32 %0, %0, %3, vcc", "+v"(ua) : "v"(a), "v"(b), "v"(ub) : "vcc")   // cmp + select pair
  if (MODE == 30) BODY("v_cmp_lt_f64 %1, %2, %3\nv_cndmask_b32_e64 %0, %0, %4, %1", "+v"(ua), "=s"(sa) : "v"(a), "v"(b), "v"(ub))   // pair via SGPR
  if (MODE == 31) BODY("v_readfirstlane_b32 %0, %1", "=s"(s32a) : "v"(ua))
  if (MODE == 32) BODY("v_max_f64 %0, %0, %1", "+v"(a) : "v"(c))
  if (MODE == 33) BODY("v_mul_f64 %0, %0, 2.0", "+v"(a) : )                                  // inline constant operand
  if (MODE == 34) BODY("v_xor_b32 %0, %0, %1", "+v"(ua) : "v"(ub))
  if (MODE == 35) BODY("v_lshlrev_b64 %0, 3, %0", "+v"(w) : )
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = a + b + (double)(ua + ub) + (double)w + (double)sa + (double)s32a;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

static const char* names[] = {"v_fma_f64 (chain)", "v_mul_f64 (chain)", "v_add_f64 (chain)", "v_fma_f64 x*c + (-0.0)  [= mul]",
  "v_fma_f64 x*1.0 + c     [= add]", "v_mad_u64_u32 (chain)", "v_bitop3_b32", "v_add_u32", "v_cndmask_b32", "v_cmp_lt_f64 -> vcc",
  "v_cvt_f64_u32", "v_rcp_f64", "v_rsq_f64", "v_trunc_f64", "v_ldexp_f64", "v_mov_b32", "s_add_u32", "s_and_b64", "v_readlane_b32",
  "v_fmac_f64", "v_mul_f64 (independent)", "v_fma_f64 (independent)", "v_mul_lo_u32", "v_cmp_lt_f64 -> sgpr pair", "v_mul_f64 with SGPR operand", "s_nop 0",
  "v_cndmask_b32_e64 sgpr mask (chain)", "v_cndmask_b32 vcc (independent)", "v_cndmask_b32_e64 sgpr (independent)", "v_cmp_f64->vcc + v_cndmask (pair)",
  "v_cmp_f64->sgpr + v_cndmask_e64 (pair)", "v_readfirstlane_b32", "v_max_f64", "v_mul_f64 inline const", "v_xor_b32", "v_lshlrev_b64"};

template <int M> void run(double* out, unsigned long long* cyc, int threads) {
  k<M><<<1, threads>>>(out, cyc, 0.7, 3u);
}
typedef void (*runner)(double*, unsigned long long*, int);

int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 16 * 8);
  runner rs[] = {run<0>, run<1>, run<2>, run<3>, run<4>, run<5>, run<6>, run<7>, run<8>, run<9>, run<10>, run<11>, run<12>, run<13>,
                 run<14>, run<15>, run<16>, run<17>, run<18>, run<19>, run<20>, run<21>, run<22>, run<23>, run<24>, run<25>, run<26>, run<27>, run<28>, run<29>, run<30>, run<31>, run<32>, run<33>, run<34>, run<35>};
  printf("%-40s %10s %16s %16s\n", "instruction (16 x 64 per test)", "1 wave", "2 waves / SIMD", "4 waves / SIMD");
  const int thr[3] = {64, 512, 1024};
  for (int m = 0; m < 36; ++m) {
    double res[3];
    for (int cfg = 0; cfg < 3; ++cfg) {
      unsigned long long h[16] = {0};
      for (int rep = 0; rep < 3; ++rep) { rs[m](out, cyc, thr[cfg]); hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost); }
      double s = 0; int nw = thr[cfg] / 64;
      for (int w = 0; w < nw; ++w) s += (double)h[w];
      res[cfg] = s / nw / (16.0 * 64.0);
    }
    printf("%-40s %10.2f %16.2f %16.2f   memtime ticks per instruction per wave\n", names[m], res[0], res[1], res[2]);
  }
  return 0;
}
