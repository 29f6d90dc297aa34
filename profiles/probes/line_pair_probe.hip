// Probe: the feature-major level kernel moves, per row of a list, the row's 128-byte q line (in and out) and 8 bytes of e that live in ANOTHER array (a whole line in,
// a partial line out).  Would keeping e NEXT TO the q line -- a 256-byte record per row, e in the second line -- make the memory system faster (two adjacent lines
// instead of two lines far apart: one DRAM page)?  Random rows of a 10 M-row table, 8 lanes per row for the line + one lane for e, read-modify-write.
//   apart   : q line from table A (128-byte rows), e from table B (16-byte pairs)       -- what als_level_allf_* does
//   adjacent: q line and e from one table of 256-byte records
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/probes/bin/line_pair_probe profiles/probes/line_pair_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool ADJ>
__global__ __launch_bounds__(256) void rmw_k(const uint32_t* __restrict__ rows, int64_t n_idx, double* __restrict__ A, double2* __restrict__ B) {
  const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;   // one group of 8 lanes per listed row
  const int part = threadIdx.x & 7;
  if (g >= n_idx) return;
  const uint32_t r = rows[g];
  const size_t stride = ADJ ? 32 : 16;
  double2 v = *reinterpret_cast<const double2*>(A + (size_t)r * stride + 2 * part);
  double e = 0.0;
  if (part == 0) e = ADJ ? A[(size_t)r * stride + 16] : B[r].y;
  v.x += 1.0; v.y -= 1.0;
  *reinterpret_cast<double2*>(A + (size_t)r * stride + 2 * part) = v;
  if (part == 0) { if (ADJ) A[(size_t)r * stride + 16] = e + 0.5; else B[r].y = e + 0.5; }
}

int main() {
  const int64_t n = 10000000, n_idx = 20000000;
  std::mt19937_64 rng(7);
  std::vector<uint32_t> h(n_idx);
  for (auto& x : h) x = (uint32_t)(rng() % n);
  uint32_t* d_rows; double *d_A, *d_A2; double2* d_B;
  CK(hipMalloc(&d_rows, n_idx * 4)); CK(hipMalloc(&d_A, (size_t)n * 128)); CK(hipMalloc(&d_A2, (size_t)n * 256)); CK(hipMalloc(&d_B, (size_t)n * 16));
  CK(hipMemcpy(d_rows, h.data(), n_idx * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_A, 0, (size_t)n * 128)); CK(hipMemset(d_A2, 0, (size_t)n * 256)); CK(hipMemset(d_B, 0, (size_t)n * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-70s %8.3f ms per 20 M rows   %6.2f G rows/s\n", name, ms / reps, n_idx / (ms / reps * 1e-3) / 1e9);
  };
  const unsigned grid = (unsigned)((n_idx * 8 + 255) / 256);
  run("apart: 128-byte line in one table, e in a table of 16-byte pairs", [&] { hipLaunchKernelGGL((rmw_k<false>), dim3(grid), dim3(256), 0, 0, d_rows, n_idx, d_A, d_B); });
  run("adjacent: 256-byte records, e in the line after the q line", [&] { hipLaunchKernelGGL((rmw_k<true>), dim3(grid), dim3(256), 0, 0, d_rows, n_idx, d_A2, (double2*)nullptr); });
  run("apart (again)", [&] { hipLaunchKernelGGL((rmw_k<false>), dim3(grid), dim3(256), 0, 0, d_rows, n_idx, d_A, d_B); });
  run("adjacent (again)", [&] { hipLaunchKernelGGL((rmw_k<true>), dim3(grid), dim3(256), 0, 0, d_rows, n_idx, d_A2, (double2*)nullptr); });
  return 0;
}
