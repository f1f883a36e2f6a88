// Micro-benchmark: issue/latency of the instructions K1 lives on, for ONE wave on a CU (gfx950).
// Each test runs a chain of N instructions, dependent (latency) or over 4 independent chains
// (issue rate), timed with s_memtime.  Build: hipcc --offload-arch=gfx950 -O3 lat.hip -o lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int MODE> __global__ void k(double* out, unsigned long long* cyc, double x0, unsigned u0) {
  double a = x0 + threadIdx.x * 1e-9, b = a + 1, c = a + 2, d = a + 3;
  const double m = 0.999999, q = 1e-9;
  unsigned ua = u0 + threadIdx.x, ub = ua + 1, uc = ua + 2, ud = ua + 3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 4; ++r) {
      if (MODE == 0) { a = __builtin_fma(a, m, q); a = __builtin_fma(a, m, q); a = __builtin_fma(a, m, q); a = __builtin_fma(a, m, q); }
      if (MODE == 1) { a = __builtin_fma(a, m, q); b = __builtin_fma(b, m, q); c = __builtin_fma(c, m, q); d = __builtin_fma(d, m, q); }
      if (MODE == 2) { a = a * m; a = a * m; a = a * m; a = a * m; }
      if (MODE == 3) { a = a + q; a = a + q; a = a + q; a = a + q; }
      if (MODE == 4) { a = __builtin_amdgcn_rcp(a); a = __builtin_amdgcn_rcp(a); a = __builtin_amdgcn_rcp(a); a = __builtin_amdgcn_rcp(a); }
      if (MODE == 5) { unsigned long long p = (unsigned long long)ua * 0xD2511F53u; ua = (unsigned)(p >> 32) ^ (unsigned)p;
                       p = (unsigned long long)ua * 0xD2511F53u; ua = (unsigned)(p >> 32) ^ (unsigned)p;
                       p = (unsigned long long)ua * 0xD2511F53u; ua = (unsigned)(p >> 32) ^ (unsigned)p;
                       p = (unsigned long long)ua * 0xD2511F53u; ua = (unsigned)(p >> 32) ^ (unsigned)p; }
      if (MODE == 6) { ua ^= ub; ua += 0x9E3779B9u; ua ^= uc; ua += 0x9E3779B9u; }
      if (MODE == 7) { a = a > 0.5 ? a * m : a + q; a = a > 0.5 ? a * m : a + q; }   // cmp + cndmask chains
      if (MODE == 9) {   // VALU -> SGPR -> SALU -> VALU round trip, 2x per r
        unsigned long long bm = __ballot(a > 0.5); a = __builtin_fma(a, m, (double)__popcll(bm) * 1e-30);
        bm = __ballot(a > 0.5); a = __builtin_fma(a, m, (double)__popcll(bm) * 1e-30); }
      if (MODE == 10) {  // divergent-if skeleton: v_cmp, s_and_saveexec, s_cbranch_execz (never taken body), s_or exec; 2x per r
        if (a > 2.0) a = __builtin_amdgcn_rcp(a);  a = __builtin_fma(a, m, q);
        if (a > 2.0) a = __builtin_amdgcn_rcp(a);  a = __builtin_fma(a, m, q); }
      if (MODE == 11) {  // 4 independent compares OR-ed, one branch (never taken)
        const bool f = (a > 2.0) | (b > 2.0) | (c > 2.0) | (d > 2.0);
        if (f) a = __builtin_amdgcn_rcp(a);
        a = __builtin_fma(a, m, q); b = __builtin_fma(b, m, q); c = __builtin_fma(c, m, q); d = __builtin_fma(d, m, q); }
      if (MODE == 8) { a = __builtin_fma(a, m, q); ua ^= ub; ua += 0x9E3779B9u; a = __builtin_fma(a, m, q); ub ^= ua; ub += 0x9E3779B9u;
                       a = __builtin_fma(a, m, q); uc ^= ub; uc += 3u; a = __builtin_fma(a, m, q); ud ^= uc; ud += 5u; }  // fp64 chain + independent int work
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = a + b + c + d + (double)(ua + ub + uc + ud);
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  const char* names[] = {"fma_f64 dependent", "fma_f64 4 chains", "mul_f64 dependent", "add_f64 dependent", "rcp_f64 dependent",
                         "mad_u64_u32+xor dependent (per pair)", "xor/add u32 dependent", "cmp+cndmask+mul/add f64 chain (per select group)",
                         "fma_f64 chain + independent int ops (per fma)", "v_cmp->s_bcnt->v_cvt->v_fma round trip (per r/4 x2)",
                         "never-taken divergent if + fma (x2 per r/4)", "4 cmps OR-ed + 1 branch + 4 fma (per r/4)"};
  for (int mode = 0; mode < 12; ++mode) {
    unsigned long long h = 0;
    for (int rep = 0; rep < 3; ++rep) {
      switch (mode) {
        case 0: k<0><<<1, 64>>>(out, cyc, 0.7, 3u); break; case 1: k<1><<<1, 64>>>(out, cyc, 0.7, 3u); break;
        case 2: k<2><<<1, 64>>>(out, cyc, 0.7, 3u); break; case 3: k<3><<<1, 64>>>(out, cyc, 0.7, 3u); break;
        case 4: k<4><<<1, 64>>>(out, cyc, 0.7, 3u); break; case 5: k<5><<<1, 64>>>(out, cyc, 0.7, 3u); break;
        case 6: k<6><<<1, 64>>>(out, cyc, 0.7, 3u); break; case 7: k<7><<<1, 64>>>(out, cyc, 0.7, 3u); break;
        case 8: k<8><<<1, 64>>>(out, cyc, 0.7, 3u); break; case 9: k<9><<<1, 64>>>(out, cyc, 0.7, 3u); break;
        case 10: k<10><<<1, 64>>>(out, cyc, 0.7, 3u); break; case 11: k<11><<<1, 64>>>(out, cyc, 0.7, 3u); break;
      }
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    }
    const double per = (double)h / (64.0 * REP) * (mode >= 9 ? 4.0 : 1.0);
    printf("%-55s %7.2f memtime-ticks per source op (total %llu)\n", names[mode], per, h);
  }
  return 0;
}
