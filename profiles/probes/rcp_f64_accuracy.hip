// How good is v_rcp_f64 on gfx950, and how many Newton steps does 1 / d need on (1, 2] (the reassociated learner's multiplier: 1 / (1 + t))?
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o profiles/probes/bin/rcp_f64_accuracy profiles/probes/rcp_f64_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k(const double* d, double* x0, double* x1, double* x2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = d[i];
  double x = __builtin_amdgcn_rcp(v);
  x0[i] = x;
  x = __builtin_fma(x, __builtin_fma(-v, x, 1.0), x);
  x1[i] = x;
  x = __builtin_fma(x, __builtin_fma(-v, x, 1.0), x);
  x2[i] = x;
}
int main() {
  const int n = 1 << 22;
  double *h = (double*)malloc(n * 8), *r0 = (double*)malloc(n * 8), *r1 = (double*)malloc(n * 8), *r2 = (double*)malloc(n * 8);
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = 1.0 + (double)(((unsigned long long)rand() << 21) ^ (unsigned long long)rand()) / 4503599627370496.0 * (i % 3 ? 1.0 : 1e-3);
  for (int i = 0; i < n; ++i) if (h[i] > 2.0) h[i] = 2.0;
  double *d, *a, *b, *c;
  CK(hipMalloc(&d, n * 8)); CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&c, n * 8));
  CK(hipMemcpy(d, h, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, a, b, c, n);
  CK(hipMemcpy(r0, a, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1, b, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2, c, n * 8, hipMemcpyDeviceToHost));
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double ex = 1.0L / (long double)h[i];
    const double u = (double)ex * 1.1102230246251565e-16 * 2;   // one ulp of the result (values in [0.5, 1))
    e0 = fmax(e0, (double)fabsl((long double)r0[i] - ex) / u); e1 = fmax(e1, (double)fabsl((long double)r1[i] - ex) / u); e2 = fmax(e2, (double)fabsl((long double)r2[i] - ex) / u);
  }
  printf("1 / d on (1, 2], %d values: v_rcp_f64 alone max error %.3g ulp (2^%.1f relative); after one Newton step %.3f ulp; after two %.3f ulp\n", n, e0, log2(e0 * 1.11e-16), e1, e2);
  return 0;
}
