// What would a row-tiled ALS / Gibbs V sweep get from the memory system?  
//
// als_level_k does one random 16-byte gather AND one 16-byte scatter of a row's (q, e) pair per stored nonzero, from the n-row table
// (160 MB at configs[4]).  This probe measures the two access patterns a tiled sweep would use instead:
//   gather : every workgroup of a SLICE of the table (slice_rows rows, 16 B each) gathers random rows of that slice only; the slice's
//            workgroups share one XCD (blockIdx % 8 equal) or are dealt over all eight; with or without writing the row back
//   stream : row-major pass: read idx[r] (4 B), gather 16 B from a small per-level table (level_feats rows), read (q, e)[r], write it back
// hipcc --offload-arch=gfx950 -O3 profiles/probes/gather_tiled.hip -o profiles/probes/bin/gather_tiled && profiles/probes/bin/gather_tiled
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// one thread = one "list" of `len` entries (len a multiple of 4); B workgroups per slice
template <bool SCATTER, bool XCD>
__global__ __launch_bounds__(256) void gather_k(double2* __restrict__ table, uint32_t slice_rows, int n_slices, int B, int len, uint32_t salt, double2* __restrict__ out) {
  const int b = blockIdx.x;
  int slice, chunk;
  if (XCD) { const int x = b & 7, i = b >> 3; slice = (i / B) * 8 + x; chunk = i % B; }
  else { slice = b / B; chunk = b % B; }
  if (slice >= n_slices) return;
  double2* __restrict__ T = table + (size_t)slice * slice_rows;
  const uint32_t base = ((uint32_t)(slice * B + chunk) * 256u + threadIdx.x) * (uint32_t)len + salt;
  double ax = 0.0, ay = 0.0;
  for (int t = 0; t < len; t += 4) {
    uint32_t r[4]; double2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = (uint32_t)(((uint64_t)mix32(base + (uint32_t)(t + u)) * slice_rows) >> 32);
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = T[r[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ax += v[u].x * v[u].y; ay += v[u].x * v[u].x;
      if (SCATTER) T[r[u]] = make_double2(v[u].x - 1e-9, v[u].y + 1e-9);
    }
  }
  out[((size_t)slice * B + chunk) * 256 + threadIdx.x] = make_double2(ax, ay);   // (indexed by the slice, not by the padded grid)
}

// row-major pass of a level: 16 B of (q, e) in and out per row, the row's feature index, a 16-byte gather from the level's small table
__global__ __launch_bounds__(256) void stream_k(double2* __restrict__ qe, const uint32_t* __restrict__ idx, const double2* __restrict__ lvl, int64_t n) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint32_t fi = idx[r];
  const double2 c = qe[r];
  const double2 d = lvl[fi];
  const double h = c.x - d.x;
  qe[r] = make_double2(c.x - d.y, c.y - h * d.y);
}
__global__ void fill_idx_k(uint32_t* idx, int64_t n, uint32_t feats) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r < n) idx[r] = (uint32_t)(((uint64_t)mix32((uint32_t)r * 2654435761u + 17u) * feats) >> 32);
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;   // rows of the (q, e) table
  const int reps = 20;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  double2 *table = nullptr, *out = nullptr, *lvl = nullptr;
  uint32_t* idx = nullptr;
  CK(hipMalloc(&table, (size_t)n * 16));
  CK(hipMemset(table, 0, (size_t)n * 16));
  CK(hipMalloc(&out, (size_t)(n / 4 + 65536) * 16));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("rows %lld (%.0f MB of (q, e) pairs); every launch touches each row once on average\n", (long long)n, n * 16 / 1e6);
  printf("%-10s %-7s %-5s %-8s %10s %12s\n", "slice_rows", "xcd", "len", "scatter", "ms/launch", "G entries/s");
  for (int64_t slice_rows : {(int64_t)32768, (int64_t)65536, (int64_t)131072, (int64_t)262144, (int64_t)524288, n}) {
    for (int len : {4, 8}) {
      const int n_slices = (int)(n / slice_rows);
      const int B = (int)((slice_rows + 256 * len - 1) / (256 * len));
      for (int xcd = 0; xcd < 2; ++xcd) {
        if (slice_rows == n && xcd) continue;
        for (int sc = 0; sc < 2; ++sc) {
          const int slots = xcd ? ((n_slices + 7) / 8) * 8 : n_slices;
          const dim3 g((unsigned)((int64_t)slots * B)), blk(256);
          auto launch = [&](uint32_t salt) {
            if (xcd) { if (sc) hipLaunchKernelGGL((gather_k<true, true>), g, blk, 0, s, table, (uint32_t)slice_rows, n_slices, B, len, salt, out);
                       else hipLaunchKernelGGL((gather_k<false, true>), g, blk, 0, s, table, (uint32_t)slice_rows, n_slices, B, len, salt, out); }
            else { if (sc) hipLaunchKernelGGL((gather_k<true, false>), g, blk, 0, s, table, (uint32_t)slice_rows, n_slices, B, len, salt, out);
                   else hipLaunchKernelGGL((gather_k<false, false>), g, blk, 0, s, table, (uint32_t)slice_rows, n_slices, B, len, salt, out); }
          };
          for (int i = 0; i < 3; ++i) launch(7u * i);
          CK(hipEventRecord(a, s));
          for (int i = 0; i < reps; ++i) launch(0x85ebca6bu * (uint32_t)(i + 7));
          CK(hipEventRecord(b, s));
          CK(hipEventSynchronize(b));
          CK(hipGetLastError());
          float ms = 0.f; CK(hipEventElapsedTime(&ms, a, b));
          const double entries = (double)n_slices * B * 256.0 * len;
          printf("%-10lld %-7s %-5d %-8s %10.4f %12.2f\n", (long long)slice_rows, slice_rows == n ? "-" : (xcd ? "same" : "spread"), len, sc ? "yes" : "no", ms / reps,
                 entries * reps / (ms * 1e-3) / 1e9);
        }
      }
    }
  }
  // the row-major pass
  CK(hipMalloc(&idx, (size_t)n * 4));
  for (uint32_t feats : {33334u, 1000000u}) {
    CK(hipMalloc(&lvl, (size_t)feats * 16));
    CK(hipMemset(lvl, 0, (size_t)feats * 16));
    hipLaunchKernelGGL(fill_idx_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, idx, n, feats);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stream_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, table, idx, lvl, n);
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, table, idx, lvl, n);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, a, b));
    printf("row-major pass, %u-row level table: %.4f ms/launch, %.2f G rows/s, %.2f TB/s of (4 + 16 + 16) B per row\n", feats, ms / reps, n * reps / (ms * 1e-3) / 1e9,
           36.0 * n * reps / (ms * 1e-3) / 1e12);
    CK(hipFree(lvl));
  }
  return 0;
}
