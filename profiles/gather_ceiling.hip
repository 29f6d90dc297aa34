// Measurement aid (not part of the library): the rate at which an MI355X serves uniformly random 64-byte table rows --
// the access pattern of fm_rows_forward (V rows by column id) and fm_cols_update (S rows by row id) with everything
// else stripped away.  Four lanes fetch one row (16 B each), `U` rows in flight per lane, 30 rows per output like the
// 30 nonzeros of a configs[1] example; optional 4-byte side-table gather per row (w / the multiplier).
//   hipcc --offload-arch=gfx950 -O3 -o gather_ceiling profiles/gather_ceiling.hip && ./gather_ceiling
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U, bool SIDE>
__global__ __launch_bounds__(256) void gather_k(const float4* __restrict__ table, const float* __restrict__ side, const uint32_t* __restrict__ idx,
                                                int per_out, int64_t n_out, float4* __restrict__ out) {
  const int64_t g = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2;  // one group of 4 lanes per output
  const int lig = threadIdx.x & 3;
  if (g >= n_out) return;
  const uint32_t* my = idx + g * per_out;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float s = 0.f;
  for (int t = 0; t < per_out; t += U) {
    float4 v[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t r = my[t + u < per_out ? t + u : t];
      v[u] = table[(size_t)r * 4 + lig];
      w[u] = SIDE ? side[r] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; s += w[u]; }
  }
  acc.x += s;
  out[g * 4 + lig] = acc;
}

int main() {
  const int per_out = 30;
  const int64_t n_out = 262144;  // one configs[1] tile: 7.86 M row fetches per launch
  const int64_t n_idx = n_out * per_out;
  float4* out; uint32_t* idx;
  CK(hipMalloc(&out, n_out * 4 * sizeof(float4)));
  CK(hipMalloc(&idx, n_idx * sizeof(uint32_t)));
  std::vector<uint32_t> h(n_idx);
  printf("%-34s %10s %12s %10s\n", "table (64-byte rows)", "ms/launch", "G rows/s", "TB/s (64 B)");
  for (double mb : {2.0, 16.0, 64.0, 128.0, 1024.0}) {
    const uint32_t rows = (uint32_t)(mb * 1024 * 1024 / 64);
    float4* table; float* side;
    CK(hipMalloc(&table, (size_t)rows * 64));
    CK(hipMalloc(&side, (size_t)rows * 4));
    CK(hipMemset(table, 0, (size_t)rows * 64));
    CK(hipMemset(side, 0, (size_t)rows * 4));
    uint64_t x = 88172645463325252ull;
    for (int64_t i = 0; i < n_idx; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)((x >> 11) % rows); }
    CK(hipMemcpy(idx, h.data(), n_idx * sizeof(uint32_t), hipMemcpyHostToDevice));
    for (int variant = 0; variant < 3; ++variant) {
      auto launch = [&]() {
        const dim3 grid((unsigned)((n_out * 4 + 255) / 256)), block(256);
        if (variant == 0) hipLaunchKernelGGL((gather_k<4, false>), grid, block, 0, 0, table, side, idx, per_out, n_out, out);
        else if (variant == 1) hipLaunchKernelGGL((gather_k<8, false>), grid, block, 0, 0, table, side, idx, per_out, n_out, out);
        else hipLaunchKernelGGL((gather_k<4, true>), grid, block, 0, 0, table, side, idx, per_out, n_out, out);
      };
      for (int i = 0; i < 5; ++i) launch();
      hipEvent_t a, b;
      CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      CK(hipEventRecord(a, 0));
      const int reps = 50;
      for (int i = 0; i < reps; ++i) launch();
      CK(hipEventRecord(b, 0));
      CK(hipEventSynchronize(b));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, a, b));
      ms /= reps;
      char name[96];
      snprintf(name, sizeof(name), "%6.0f MB  %s", mb, variant == 0 ? "4 in flight" : variant == 1 ? "8 in flight" : "4 in flight + 4-B side");
      printf("%-34s %10.4f %12.1f %10.2f\n", name, ms, n_idx / (ms * 1e-3) / 1e9, n_idx * 64.0 / (ms * 1e-3) / 1e12);
    }
    CK(hipFree(table)); CK(hipFree(side));
  }
  // Two gather kernels at once on two streams (a 64 MB and a 16 MB table: phase 1 of one tile beside phase 2 of another):
  // does the memory system serve more random rows in total than one kernel draws alone?
  {
    const uint32_t rows_a = 64u * 1024 * 1024 / 64, rows_b = 16u * 1024 * 1024 / 64;
    float4 *ta, *tb, *out_b; float* side; uint32_t* idx_b;
    CK(hipMalloc(&ta, (size_t)rows_a * 64)); CK(hipMalloc(&tb, (size_t)rows_b * 64)); CK(hipMalloc(&side, (size_t)rows_a * 4));
    CK(hipMalloc(&out_b, n_out * 4 * sizeof(float4))); CK(hipMalloc(&idx_b, n_idx * sizeof(uint32_t)));
    CK(hipMemset(ta, 0, (size_t)rows_a * 64)); CK(hipMemset(tb, 0, (size_t)rows_b * 64));
    uint64_t x = 1234567ull;
    for (int64_t i = 0; i < n_idx; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)((x >> 11) % rows_a); }
    CK(hipMemcpy(idx, h.data(), n_idx * sizeof(uint32_t), hipMemcpyHostToDevice));
    for (int64_t i = 0; i < n_idx; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint32_t)((x >> 11) % rows_b); }
    CK(hipMemcpy(idx_b, h.data(), n_idx * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    const dim3 grid((unsigned)((n_out * 4 + 255) / 256)), block(256);
    auto run = [&](bool concurrent, int reps) {
      CK(hipDeviceSynchronize());
      hipEvent_t a, b;
      CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      CK(hipEventRecord(a, s1));
      for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL((gather_k<4, false>), grid, block, 0, s1, ta, side, idx, per_out, n_out, out);
        hipLaunchKernelGGL((gather_k<4, false>), grid, block, 0, concurrent ? s2 : s1, tb, side, idx_b, per_out, n_out, out_b);
      }
      CK(hipStreamSynchronize(s2));
      CK(hipEventRecord(b, s1));
      CK(hipEventSynchronize(b));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, a, b));
      return ms / reps;
    };
    run(false, 5); run(true, 5);
    const float serial = run(false, 40), conc = run(true, 40);
    printf("64 MB + 16 MB tables, one launch each: back to back %.4f ms, on two streams %.4f ms (%.1f G rows/s in total)\n", serial, conc,
           2.0 * n_idx / (conc * 1e-3) / 1e9);
  }
  return 0;
}
