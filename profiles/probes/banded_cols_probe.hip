// Probe: can phase 2 of the mini-batch step (fm_cols_update_k: per feature, gather the S rows of its list and add them up) take its S rows from the XCD's L2
// instead of the fabric?  The S table of a 262 144-row step is 16.8 MB -- four times an XCD's L2 -- and every list covers its rows uniformly (ascending), so the
// walking kernel misses on almost every row and pays a 128-byte line per 64-byte row.  If every lane group owns NF features and the chip walks the ROW BANDS in
// step -- all entries with rows in band 0, then band 1, ... (the lists are sorted by row: a cursor per feature) -- an XCD only ever asks for one band (2 MB at
// eight bands) at a time and the band is read from memory once per round of resident workgroups.
//   walk   : the product's schedule in miniature -- one group of 4 lanes per list, four entries per round, fp64 sums
//   banded : NF features per lane group, NB bands; no barrier anywhere (the workgroups start together and do equal work per band)
// configs[1]'s shape: 262 144 rows x 30 entries, 1 M features (lists of 7.9 entries), k = 16 fp32 (64-byte rows).
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/probes/bin/banded_cols_probe profiles/probes/banded_cols_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void walk_k(const uint32_t* __restrict__ off, const uint32_t* __restrict__ brow, const float4* __restrict__ S, const float4* __restrict__ V,
                                              float4* __restrict__ Vout, uint32_t p) {
  const uint32_t gid = (blockIdx.x * 256 + threadIdx.x) >> 2, lig = threadIdx.x & 3;
  const uint32_t j = min(gid, p - 1);
  const uint32_t a = off[j], b = off[j + 1];
  const float4 v = V[(size_t)j * 4 + lig];
  double g0 = 0, g1 = 0, g2 = 0, g3 = 0;
  for (uint32_t t = a; t < b; t += 4) {
    uint32_t r[4]; float4 s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = brow[min(t + u, b - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = S[(size_t)r[u] * 4 + lig];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (t + u < b) { g0 += (double)s[u].x - v.x; g1 += (double)s[u].y - v.y; g2 += (double)s[u].z - v.z; g3 += (double)s[u].w - v.w; }
  }
  if (gid < p) Vout[(size_t)j * 4 + lig] = make_float4(v.x - 1e-3f * (float)g0, v.y - 1e-3f * (float)g1, v.z - 1e-3f * (float)g2, v.w - 1e-3f * (float)g3);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t BUF_SKIP = 0x80000000u;   // beyond the table: the load returns zero and no request goes out
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ float4 buf_row(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int NF>
__global__ __launch_bounds__(256) void banded_k(const uint32_t* __restrict__ off, const uint32_t* __restrict__ brow, const float4* __restrict__ S, const float4* __restrict__ V,
                                                float4* __restrict__ Vout, uint32_t p, uint32_t rows, int nb) {
  const uint32_t g = threadIdx.x >> 2, lig = threadIdx.x & 3;
  const uint32_t f0 = blockIdx.x * (64 * NF) + g;   // this group's features: f0, f0 + 64, ... (neighbouring groups read neighbouring list heads)
  uint32_t cur[NF], end[NF]; float4 v[NF]; double acc[NF][4];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const uint32_t j = min(f0 + 64 * i, p - 1);
    cur[i] = off[j]; end[i] = (f0 + 64 * i < p) ? off[j + 1] : cur[i];
    v[i] = V[(size_t)j * 4 + lig];
    acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.0;
  }
  const uint32_t band = (rows + nb - 1) / nb;
  const __amdgpu_buffer_rsrc_t s_rsrc = rsrc_of(S, rows * 64u), b_rsrc = rsrc_of(brow, off[p] * 4u);
  for (int b = 0; b < nb; ++b) {
    const uint32_t lim = (b + 1 == nb) ? 0xFFFFFFFFu : (uint32_t)(b + 1) * band;
    bool again = true;
    while (again) {   // every feature's next entry while it lies in the band: NF row ids, then NF gathers in flight per lane group
      uint32_t r[NF]; bool in[NF]; float4 s[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, (int)(cur[i] < end[i] ? cur[i] * 4u : BUF_SKIP), 0, 0);
#pragma unroll
      for (int i = 0; i < NF; ++i) { in[i] = cur[i] < end[i] && r[i] < lim; }
#pragma unroll
      for (int i = 0; i < NF; ++i) s[i] = buf_row(s_rsrc, in[i] ? r[i] * 64u + lig * 16u : BUF_SKIP);   // (masked-off lanes: no request)
      again = false;
#pragma unroll
      for (int i = 0; i < NF; ++i) if (in[i]) {
        acc[i][0] += (double)s[i].x - v[i].x; acc[i][1] += (double)s[i].y - v[i].y; acc[i][2] += (double)s[i].z - v[i].z; acc[i][3] += (double)s[i].w - v[i].w;
        ++cur[i]; again = true;
      }
      again = __any(again);
    }
  }
#pragma unroll
  for (int i = 0; i < NF; ++i) if (f0 + 64 * i < p)
    Vout[(size_t)(f0 + 64 * i) * 4 + lig] = make_float4(v[i].x - 1e-3f * (float)acc[i][0], v[i].y - 1e-3f * (float)acc[i][1], v[i].z - 1e-3f * (float)acc[i][2], v[i].w - 1e-3f * (float)acc[i][3]);
}

int main(int argc, char** argv) {
  const uint32_t rows = 262144, z = 30, p = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000000u;
  const size_t nnz = (size_t)rows * z;
  std::mt19937_64 rng(5);
  std::vector<uint32_t> col(nnz), off(p + 1, 0), brow(nnz);
  for (size_t i = 0; i < nnz; ++i) { col[i] = (uint32_t)(rng() % p); off[col[i] + 1]++; }
  for (uint32_t j = 0; j < p; ++j) off[j + 1] += off[j];
  { std::vector<uint32_t> fill(off.begin(), off.end() - 1); for (size_t i = 0; i < nnz; ++i) brow[fill[col[i]]++] = (uint32_t)(i / z); }   // rows ascending inside a list
  uint32_t *d_off, *d_brow; float4 *d_S, *d_V, *d_Vo;
  CK(hipMalloc(&d_off, (p + 1) * 4)); CK(hipMalloc(&d_brow, nnz * 4)); CK(hipMalloc(&d_S, (size_t)rows * 64)); CK(hipMalloc(&d_V, (size_t)p * 64)); CK(hipMalloc(&d_Vo, (size_t)p * 64));
  CK(hipMemcpy(d_off, off.data(), (p + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_brow, brow.data(), nnz * 4, hipMemcpyHostToDevice));
  { std::vector<float> h((size_t)rows * 16); for (auto& x : h) x = 1e-3f * (float)(rng() % 1000); CK(hipMemcpy(d_S, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
  { std::vector<float> h((size_t)p * 16); for (auto& x : h) x = 1e-3f * (float)(rng() % 1000); CK(hipMemcpy(d_V, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ref((size_t)p * 16), got((size_t)p * 16);
  auto timeit = [&](const char* name, auto launch, bool check) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const char* verdict = "";
    if (check) { CK(hipMemcpy(got.data(), d_Vo, got.size() * 4, hipMemcpyDeviceToHost)); verdict = std::equal(got.begin(), got.end(), ref.begin()) ? "  (bits = walk)" : "  (DIFFERS from walk)"; }
    printf("%-60s %8.1f us   %6.1f G rows/s%s\n", name, ms / reps * 1e3, nnz / (ms / reps * 1e-3) / 1e9, verdict);
  };
  timeit("walk: one lane group per list, 4 entries per round", [&] { hipLaunchKernelGGL(walk_k, dim3((p * 4 + 255) / 256), dim3(256), 0, 0, d_off, d_brow, d_S, d_V, d_Vo, p); }, false);
  CK(hipMemcpy(ref.data(), d_Vo, ref.size() * 4, hipMemcpyDeviceToHost));
#define BAND(NFv, NBv) timeit("banded: " #NFv " features per lane group, " #NBv " bands", [&] { hipLaunchKernelGGL((banded_k<NFv>), dim3((p + 64 * NFv - 1) / (64 * NFv)), dim3(256), 0, 0, d_off, d_brow, d_S, d_V, d_Vo, p, rows, NBv); }, true);
  BAND(4, 1) BAND(4, 8) BAND(4, 16) BAND(4, 32) BAND(8, 8) BAND(8, 16) BAND(8, 32) BAND(2, 8) BAND(2, 16) BAND(6, 16)
  return 0;
}
