// Probe for a "feature-block-major" form of the level-order V sweep (round 5, second session): before building it, what does ONE kernel per level cost on the
// configs[4] shape when a workgroup owns a BLOCK of consecutive features of the level (at most R rows) and
//   1. streams the block's region of (q, e) pairs (contiguous: the level's array is feature-block-major) into LDS at their feature-sorted place (u16 per pair),
//   2. sums every list out of LDS, takes the coordinate steps, corrects the pairs in place -- the sums never leave the workgroup and the pairs are read ONCE,
//   3. writes the pairs to the NEXT level's array, whose block B' keeps what it receives from block B as one contiguous run: the workgroup reads its pairs
//      back out of LDS in destination order (u16 per pair) and stores them at consecutive addresses run by run (u32 per pair).
// The level needs no second kernel and no chip-wide wait between sums and corrections.  Unknown before measuring: the runs are short (R^2 / n pairs: 6.7 at
// R = 8192) and unaligned -- does the L2 merge them into whole lines (neighbouring blocks run at the same time on the same XCD)?
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/probes/bin/block_level_probe profiles/probes/block_level_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ntload(const double2* p) { const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p)); return make_double2(v.x, v.y); }
template <typename T> __device__ __forceinline__ T ntl(const T* p) { return __builtin_nontemporal_load(p); }

__global__ __launch_bounds__(256) void copy_k(const double2* __restrict__ a, double2* __restrict__ b, int64_t n) {
  const int64_t i0 = (int64_t)blockIdx.x * 2048 + threadIdx.x;
  double2 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { const int64_t i = i0 + u * 256; v[u] = a[i < n ? i : n - 1]; }
#pragma unroll
  for (int u = 0; u < 8; ++u) { const int64_t i = i0 + u * 256; if (i < n) b[i] = v[u]; }
}

__device__ unsigned long long phase_ticks[8];
#define STAMP(slot) do { if (MODE & 8) { const unsigned long long now_ = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&phase_ticks[slot], now_ - last_); last_ = now_; } } while (0)
// MODE bit 0: the sums / step / correction phase runs; bit 1: the store goes to `dest` (else to the block's own region: a plain copy through LDS);
// bit 2: blocks dealt so that consecutive blocks share an XCD
template <int R, int NT_, int MODE, int LISTLEN>
__global__ __launch_bounds__(NT_) void block_level_k(const double2* __restrict__ src, double2* __restrict__ dst, const uint32_t* __restrict__ bbase, int nb,
                                                     const uint16_t* __restrict__ perm_in, const uint16_t* __restrict__ gsrc, const uint32_t* __restrict__ dest,
                                                     double* __restrict__ vout) {
  constexpr int PT = R / NT_;
  __shared__ double2 lp[R];
  int B = blockIdx.x;
  if (MODE & 4) { const int per = (nb + 7) / 8; B = (blockIdx.x & 7) * per + (blockIdx.x >> 3); if (B >= nb || (int)(blockIdx.x >> 3) >= per) return; }
  const uint32_t b0 = bbase[B], rows = bbase[B + 1] - b0;
  unsigned long long last_ = wall_clock64();
  double2 v[PT]; uint16_t pa[PT], gs[PT]; uint32_t de[PT];
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const uint32_t i = threadIdx.x + u * NT_, ic = min(i, rows - 1);
    v[u] = ntload(src + b0 + ic);
    pa[u] = ntl(perm_in + b0 + ic);
  }
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const uint32_t i = threadIdx.x + u * NT_, ic = min(i, rows - 1);
    gs[u] = ntl(gsrc + b0 + ic);
    de[u] = (MODE & 2) ? ntl(dest + b0 + ic) : b0 + ic;
  }
#pragma unroll
  STAMP(0);   // issue of all loads
  for (int u = 0; u < PT; ++u) { const uint32_t i = threadIdx.x + u * NT_; if (i < rows) lp[pa[u]] = v[u]; }
  STAMP(1);   // wait for the pairs + LDS scatter
  __syncthreads();
  STAMP(2);   // barrier
  if (MODE & 1) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int K = (LISTLEN + 63) / 64;
    for (uint32_t f = wv; f * LISTLEN < rows; f += NT_ / 64) {
      const uint32_t lo = f * LISTLEN, hi = min(lo + LISTLEN, rows);
      double2 c[K]; double h[K]; double mean = 0.0, var = 0.0;
      const double old = 0.01 * f;
#pragma unroll
      for (int k = 0; k < K; ++k) { const uint32_t i = lo + lane + 64 * k; c[k] = lp[min(i, hi - 1)]; h[k] = c[k].x - old; if (i < hi) { mean += h[k] * c[k].y; var += h[k] * h[k]; } }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { mean += __shfl_xor(mean, o); var += __shfl_xor(var, o); }
      mean -= old * var; var = 1.0 / (1.0 + var); const double nv = -var * mean; const double diff = (old - nv) * 1e-3;
      if (lane == 0) vout[(size_t)B * 64 + (f & 63)] = nv;
#pragma unroll
      for (int k = 0; k < K; ++k) { const uint32_t i = lo + lane + 64 * k; if (i < hi) lp[i] = make_double2(c[k].x - diff, c[k].y - h[k] * diff); }
    }
    STAMP(3);   // sums, steps, corrections (wave 0's share)
    __syncthreads();
    STAMP(4);   // barrier
  }
#pragma unroll
  for (int u = 0; u < PT; ++u) { const uint32_t i = threadIdx.x + u * NT_; if (i < rows) dst[de[u]] = lp[gs[u]]; }
  STAMP(5);     // LDS gather + issue of the stores
  if (MODE & 8) { __builtin_amdgcn_s_waitcnt(0); STAMP(6); }   // the stores acknowledged
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
  const int nb = argc > 2 ? atoi(argv[2]) : 1290;       // blocks per level (rows per block: n / nb on average; capacity 8192)
  const int R = argc > 3 ? atoi(argv[3]) : 8192;   // block capacity (pairs): 8192 = one workgroup per CU (128 KB of LDS), 4096 = two
  printf("n = %lld rows, %d blocks per level (%.0f rows on average, runs of %.1f pairs)\n", (long long)n, nb, (double)n / nb, (double)n / nb / nb);
  std::mt19937_64 rng(11);
  // every row: its block at this level (A) and at the next (Bn), independent and uniform
  std::vector<uint16_t> ba(n), bn(n);
  for (int64_t r = 0; r < n; ++r) { ba[r] = (uint16_t)(rng() % nb); bn[r] = (uint16_t)(rng() % nb); }
  // this level's array: rows sorted by (A, random inside) -- the order inside a block's region is whatever the previous level's runs made it
  std::vector<uint32_t> cntA(nb + 1, 0), cntB(nb + 1, 0);
  for (int64_t r = 0; r < n; ++r) { cntA[ba[r] + 1]++; cntB[bn[r] + 1]++; }
  for (int b = 0; b < nb; ++b) { cntA[b + 1] += cntA[b]; cntB[b + 1] += cntB[b]; }
  uint32_t mx = 0; for (int b = 0; b < nb; ++b) mx = std::max(mx, std::max(cntA[b + 1] - cntA[b], cntB[b + 1] - cntB[b]));
  printf("largest block: %u rows (capacity %d)\n", mx, R);
  if (mx > (uint32_t)R) { printf("a block exceeds the capacity: raise nb\n"); return 1; }
  std::vector<uint32_t> rowsA(n);          // position in this level's array -> row
  { std::vector<uint32_t> fill(cntA.begin(), cntA.end() - 1); for (int64_t r = 0; r < n; ++r) rowsA[fill[ba[r]]++] = (uint32_t)r; }
  // next level's array: block B' = runs by source block A in order; inside a run: rows in this level's order
  // C[A][B'] run lengths, runoff[B'][A]
  std::vector<uint32_t> C((size_t)nb * nb, 0);
  for (int64_t r = 0; r < n; ++r) C[(size_t)ba[r] * nb + bn[r]]++;
  std::vector<uint32_t> runstart((size_t)nb * nb);   // [A][B'] first position of the run in the next level's array
  for (int bp = 0; bp < nb; ++bp) { uint32_t at = cntB[bp]; for (int a = 0; a < nb; ++a) { runstart[(size_t)a * nb + bp] = at; at += C[(size_t)a * nb + bp]; } }
  std::vector<uint16_t> perm_in(n), gsrc(n);
  std::vector<uint32_t> dest(n);
  std::vector<uint32_t> order(R), lds_of(R);
  const bool small = R <= 4096;
  std::vector<uint8_t> seen(n, 0);
  for (int a = 0; a < nb; ++a) {
    const uint32_t b0 = cntA[a], rows = cntA[a + 1] - b0;
    // LDS place of the pair at region position i: a random permutation (the feature-sorted order is unrelated to the arrival order)
    std::iota(order.begin(), order.begin() + rows, 0u); std::shuffle(order.begin(), order.begin() + rows, rng);
    for (uint32_t i = 0; i < rows; ++i) { perm_in[b0 + i] = (uint16_t)order[i]; lds_of[i] = order[i]; }
    // destination order: region positions sorted by next block (stable)
    std::vector<uint32_t> idx(rows); std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return bn[rowsA[b0 + x]] < bn[rowsA[b0 + y]]; });
    std::vector<uint32_t> fill(nb); for (int bp = 0; bp < nb; ++bp) fill[bp] = runstart[(size_t)a * nb + bp];
    for (uint32_t k = 0; k < rows; ++k) { const uint32_t i = idx[k]; const int bp = bn[rowsA[b0 + i]]; gsrc[b0 + k] = (uint16_t)lds_of[i]; dest[b0 + k] = fill[bp]++; if (seen[dest[b0 + k]]++) { printf("dest not a bijection\n"); return 1; } }
  }
  double2 *src, *dst; uint16_t *d_pi, *d_gs; uint32_t *d_de, *d_bb; double* vout;
  CK(hipMalloc(&src, n * 16)); CK(hipMalloc(&dst, n * 16)); CK(hipMalloc(&d_pi, n * 2)); CK(hipMalloc(&d_gs, n * 2)); CK(hipMalloc(&d_de, n * 4)); CK(hipMalloc(&d_bb, (nb + 1) * 4));
  CK(hipMalloc(&vout, (size_t)nb * 64 * 8));
  { std::vector<double2> h(n); for (int64_t i = 0; i < n; ++i) h[i] = make_double2(1e-3 * (double)(rng() % 1000), 1e-3 * (double)(rng() % 1000) - 0.5); CK(hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice)); }
  CK(hipMemset(dst, 0, n * 16));
  CK(hipMemcpy(d_pi, perm_in.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_gs, gsrc.data(), n * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_de, dest.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_bb, cntA.data(), (nb + 1) * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %8.1f us   %6.2f TB/s of %.0f MB\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, bytes / 1e6);
  };
  int flip = 0;
  timeit("copy 16-B pairs", [&] { hipLaunchKernelGGL(copy_k, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, flip ? dst : src, flip ? src : dst, n); flip ^= 1; }, 32.0 * n);
  const unsigned g8 = (unsigned)(((nb + 7) / 8) * 8);
#define RUN(NTv, MODEv, name) timeit(name, [&] { hipLaunchKernelGGL((block_level_k<8192, NTv, MODEv, 300>), dim3((MODEv & 4) ? g8 : (unsigned)nb), dim3(NTv), 0, 0, flip ? dst : src, flip ? src : dst, d_bb, nb, d_pi, d_gs, d_de, vout); flip ^= 1; }, 40.0 * n);
  if (!small) {
  RUN(1024, 0, "block kernel 1024 thr: through LDS, own region (no sums)")
  RUN(1024, 1, "block kernel 1024 thr: own region + sums/steps/corrections")
  RUN(1024, 2, "block kernel 1024 thr: runs into the next level's array")
  RUN(1024, 3, "block kernel 1024 thr: runs + sums")
  RUN(1024, 6, "block kernel 1024 thr: runs, XCD-consecutive blocks")
  RUN(1024, 7, "block kernel 1024 thr: runs + sums, XCD-consecutive blocks")
  {
    unsigned long long z8[8] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(phase_ticks), z8, sizeof(z8)));
    RUN(1024, 15, "block kernel 1024 thr: runs + sums, XCD-consec, phase stamps")
    unsigned long long t8[8]; CK(hipMemcpyFromSymbol(t8, HIP_SYMBOL(phase_ticks), sizeof(t8)));
    const double per = 0.01 / (23.0 * nb);   // 3 warm-up + 20 timed launches; 100 MHz ticks -> us per workgroup
    printf("    per workgroup (us): issue loads %.2f | wait + LDS scatter %.2f | barrier %.2f | sums (wave 0) %.2f | barrier %.2f | LDS gather + store issue %.2f | stores acked %.2f\n",
           t8[0] * per, t8[1] * per, t8[2] * per, t8[3] * per, t8[4] * per, t8[5] * per, t8[6] * per);
  }
  RUN(512, 7, "block kernel 512 thr (16 pairs/thread): runs + sums, XCD-consec")
  RUN(512, 3, "block kernel 512 thr (16 pairs/thread): runs + sums")
  } else {
#define RUNS(NTv, MODEv, name) timeit(name, [&] { hipLaunchKernelGGL((block_level_k<4096, NTv, MODEv, 300>), dim3((MODEv & 4) ? g8 : (unsigned)nb), dim3(NTv), 0, 0, flip ? dst : src, flip ? src : dst, d_bb, nb, d_pi, d_gs, d_de, vout); flip ^= 1; }, 40.0 * n);
    RUNS(512, 0, "4096-pair blocks, 512 thr (2 wg/CU): through LDS, own region")
    RUNS(512, 1, "4096-pair blocks, 512 thr: own region + sums")
    RUNS(512, 6, "4096-pair blocks, 512 thr: runs, XCD-consecutive")
    RUNS(512, 7, "4096-pair blocks, 512 thr: runs + sums, XCD-consecutive")
    RUNS(512, 3, "4096-pair blocks, 512 thr: runs + sums")
    RUNS(256, 7, "4096-pair blocks, 256 thr (16 pairs/thread): runs + sums, XCD-consec")
    RUNS(1024, 7, "4096-pair blocks, 1024 thr (4 pairs/thread): runs + sums, XCD-consec")
  }
  // check of the last full launch: every destination holds a finite pair
  { std::vector<double2> h(n); CK(hipMemcpy(h.data(), flip ? src : dst, n * 16, hipMemcpyDeviceToHost)); double s = 0; for (int64_t i = 0; i < n; ++i) s += h[i].x; printf("checksum %.6f\n", s); }
  return 0;
}
