// What does ONE dependent fp64 operation cost a lone wave on gfx950, and what do independent ones beside it cost?  (The reference-order learners' scalar chain is a
// string of dependent fp64 operations; profiles/r06_seq_reassoc.txt.)
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o profiles/probes/bin/fp64_chain_latency profiles/probes/fp64_chain_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int N = 1 << 14;
template <int CHAINS, int OP>
__global__ void k(double* out, unsigned long long* ticks, double a, double b) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = 1.0 + 1e-9 * (threadIdx.x + c);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long c0 = wall_clock64();
#pragma unroll 1
  for (int i = 0; i < N / 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (OP == 0) x[c] = __builtin_fma(x[c], a, b);
        else if (OP == 1) x[c] = x[c] * a;
        else if (OP == 2) x[c] = x[c] + b;
        else if (OP == 3) x[c] = __builtin_amdgcn_rcp(x[c]);
        else if (OP == 4) { float f = (float)x[c]; f = __builtin_fmaf(f, (float)a, (float)b); x[c] = f; }   // (cvt + fp32 fma + cvt)
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long c1 = wall_clock64();
  double s = 0; for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
}
template <int CHAINS, int OP> void run(const char* what, int threads) {
  double* out; unsigned long long *t, h[2];
  CK(hipMalloc(&out, 4096 * 8)); CK(hipMalloc(&t, 16));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<CHAINS, OP>), dim3(1), dim3(threads), 0, 0, out, t, 0.9999999, 1e-7); CK(hipDeviceSynchronize()); }
  CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
  printf("%-44s %d waves in the workgroup, %d chain(s) per wave: %7.2f s_memtime ticks, %6.2f ns (100 MHz wall clock) per operation of a chain\n", what, threads / 64, CHAINS,
         (double)h[0] / N, (double)h[1] * 10.0 / N);
  CK(hipFree(out)); CK(hipFree(t));
}
int main() {
  run<1, 0>("v_fma_f64, dependent", 64);
  run<2, 0>("v_fma_f64, two chains interleaved", 64);
  run<4, 0>("v_fma_f64, four chains interleaved", 64);
  run<8, 0>("v_fma_f64, eight chains interleaved", 64);
  run<1, 1>("v_mul_f64, dependent", 64);
  run<1, 2>("v_add_f64, dependent", 64);
  run<1, 3>("v_rcp_f64, dependent", 64);
  run<1, 4>("fp32 fma between two conversions, dependent", 64);
  run<1, 0>("v_fma_f64, dependent", 256);
  run<1, 0>("v_fma_f64, dependent", 1024);
  run<4, 0>("v_fma_f64, four chains interleaved", 1024);
  return 0;
}
