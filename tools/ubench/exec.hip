// Micro-benchmark: does a wave64 fp64 VALU instruction get cheaper when only some quarter-waves have live lanes?
// (The tail of K1 runs waves with a handful of live lanes.)  One wave; lanes >= nlive are masked off by a branch.
// Build: hipcc --offload-arch=gfx950 -O3 exec.hip -o exec
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, unsigned long long* cyc, double x0, int nlive, int first) {
  double a = x0 + threadIdx.x * 1e-9, b = a + 1, c = a + 2, d = a + 3;
  const double m = 0.999999, q = 1e-9;
  unsigned long long t0 = 0, t1 = 0;
  if ((int)threadIdx.x >= first && (int)threadIdx.x < first + nlive) {
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int r = 0; r < 64; ++r) { a = __builtin_fma(a, m, q); b = __builtin_fma(b, m, q); c = __builtin_fma(c, m, q); d = __builtin_fma(d, m, q); }
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  out[threadIdx.x] = a + b + c + d;
  if ((int)threadIdx.x == first) cyc[0] = t1 - t0;
}
int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  const int cfg[][2] = {{64, 0}, {32, 0}, {16, 0}, {16, 16}, {16, 48}, {8, 0}, {1, 0}, {1, 63}, {2, 31}};
  for (auto& c : cfg) {
    unsigned long long h = 0;
    for (int rep = 0; rep < 3; ++rep) { k<<<1, 64>>>(out, cyc, 0.7, c[0], c[1]); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); }
    printf("live lanes [%2d, %2d): %6.2f memtime-ticks per fma_f64 (4 independent chains)\n", c[1], c[1] + c[0], (double)h / (64.0 * 256));
  }
  return 0;
}
