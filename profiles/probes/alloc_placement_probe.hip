// Probe: the feature-major sweep on field data takes 29.6 or 35.9 ms from one fresh process to the next on the same box (profiles/r05_alloc_placement.txt).  Is it WHERE the
// 1.28 GB table of q lines lands?  Six tables of 10 M x 128 bytes allocated one after the other in one process, the same random read-modify-write of whole lines
// (20 M rows, 8 lanes per row) on each; the process run several times.
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/probes/bin/alloc_placement_probe profiles/probes/alloc_placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void rmw_k(const uint32_t* __restrict__ rows, int64_t n_idx, double* __restrict__ A) {
  const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int part = threadIdx.x & 7;
  if (g >= n_idx) return;
  double2* p = reinterpret_cast<double2*>(A + (size_t)rows[g] * 16 + 2 * part);
  double2 v = *p; v.x += 1.0; v.y -= 1.0; *p = v;
}
int main() {
  const int64_t n = 10000000, n_idx = 20000000;
  std::mt19937_64 rng(7);
  std::vector<uint32_t> h(n_idx);
  for (auto& x : h) x = (uint32_t)(rng() % n);
  uint32_t* d_rows; CK(hipMalloc(&d_rows, n_idx * 4)); CK(hipMemcpy(d_rows, h.data(), n_idx * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)((n_idx * 8 + 255) / 256);
  double* T[6];
  for (int t = 0; t < 6; ++t) {
    CK(hipMalloc(&T[t], (size_t)n * 128)); CK(hipMemset(T[t], 0, (size_t)n * 128));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(rmw_k, dim3(grid), dim3(256), 0, 0, d_rows, n_idx, T[t]);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(rmw_k, dim3(grid), dim3(256), 0, 0, d_rows, n_idx, T[t]);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("table %d at %p: %.3f ms per 20 M rows (%.2f G rows/s)\n", t, (void*)T[t], ms / 5, n_idx / (ms / 5 * 1e-3) / 1e9);
  }
  return 0;
}
