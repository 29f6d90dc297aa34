// Probe for the "pairs kept in the consuming level's list order" form of the tiled V sweep: before building it, what do its two passes
// cost on the configs[4] shape (10 M rows, 33 334 features per level, tiles of 2^ts rows)?
//   K1 sums+step : a workgroup owns FB features of the level and walks their (tile, feature) lists tile by tile -- within a tile the lists of consecutive features
//                  are contiguous, so the pairs arrive as 77 pieces of ~FB*4*16 bytes, streamed; the full sums never leave the workgroup (no per-tile partials)
//   K2 apply     : entries in level-l order: pair, feature index (u16), Delta of the feature (16-byte gather from a level-sized table), position in level l+1's
//                  order (u32): corrected pair written THERE (a permutation inside the tile's slice; the tile's workgroups share an XCD)
//   K0           : plain copy of the pairs (what the memory system gives a 160 MB -> 160 MB stream)
// build: hipcc --offload-arch=gfx950 -O3 -o profiles/probes/bin/level_order_probe profiles/probes/level_order_probe.hip
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

// K1: workgroup = FB features (FB = 256 / LG lane groups of LG lanes); lane group g owns feature f0 + g and walks its list in every tile
template <int LG>
__global__ __launch_bounds__(256) void sums_step_k(const uint32_t* __restrict__ toff, size_t nf1, uint32_t cnt, const double2* __restrict__ src, int ts, int n_tiles,
                                                   const double* __restrict__ vf, double2* __restrict__ vstep) {
  const uint32_t fi = blockIdx.x * (256 / LG) + threadIdx.x / LG;
  const int lg = threadIdx.x % LG;
  const bool live = fi < cnt;
  const uint32_t fc = live ? fi : cnt - 1;
  const double old = vf[fc];
  double mean = 0.0, var = 0.0;
  for (int t = 0; t < n_tiles; ++t) {
    const uint32_t* off = toff + (size_t)t * nf1 + fc;
    const uint32_t lb = off[0], le = off[1];
    const double2* s = src + ((size_t)t << ts);
    for (uint32_t i = lb + lg; i < le; i += LG) { const double2 c = ntload(s + i); const double h = c.x - old; mean += h * c.y; var += h * h; }
  }
#pragma unroll
  for (int o = LG / 2; o > 0; o >>= 1) { mean += __shfl_xor(mean, o); var += __shfl_xor(var, o); }
  if (lg == 0 && live) { mean -= old * var; var = 1.0 / (1.0 + var); const double nv = -var * mean; vstep[fi] = make_double2(old, (old - nv) * 1e-3); }
}
// K1b: flat form: a workgroup takes features [f0, f0 + FB) and, per tile, reads their contiguous run of pairs with all 256 threads (coalesced), each entry's
// feature from the u16 index; per-feature sums meet in LDS (one add per entry: order fixed by the static entry->thread map + a final ordered sum) -- here simply
// timed with LDS float atomics OFF: each thread keeps sums for the entries it sees and we reduce by segmented shuffles.  (cost model only)
template <int FB>
__global__ __launch_bounds__(256) void sums_flat_k(const uint32_t* __restrict__ toff, size_t nf1, uint32_t cnt, const double2* __restrict__ src, const uint16_t* __restrict__ fidx,
                                                   int ts, int n_tiles, const double* __restrict__ vf, double2* __restrict__ vstep) {
  __shared__ double2 acc[FB];
  __shared__ double sold[FB];
  const uint32_t f0 = blockIdx.x * FB;
  const uint32_t f1 = min(f0 + FB, cnt);
  for (int i = threadIdx.x; i < FB; i += 256) { acc[i] = make_double2(0.0, 0.0); sold[i] = vf[min(f0 + i, cnt - 1)]; }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int t = 0; t < n_tiles; ++t) {
    const uint32_t lb = toff[(size_t)t * nf1 + f0], le = toff[(size_t)t * nf1 + f1];
    const double2* s = src + ((size_t)t << ts);
    const uint16_t* fx = fidx + ((size_t)t << ts);
    for (uint32_t i0 = lb; i0 < le; i0 += 256) {
      const uint32_t i = i0 + threadIdx.x;
      const bool in = i < le;
      const uint32_t ic = in ? i : le - 1;
      const double2 c = ntload(s + ic);
      const uint32_t f = (uint32_t)ntl(fx + ic) - f0;
      const double h = c.x - sold[f];
      double m = in ? h * c.y : 0.0, v = in ? h * h : 0.0;
      // segmented inclusive scan over the wave by feature (entries sorted by feature): fixed association
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double m2 = __shfl_up(m, o), v2 = __shfl_up(v, o);
        const uint32_t f2 = __shfl_up(f, o);
        if (lane >= o && f2 == f) { m += m2; v += v2; }
      }
      const uint32_t fn = __shfl_down(f, 1);
      const bool tail = in && (lane == 63 || fn != f || i + 1 >= le);
      // a feature's run can straddle waves: the four waves add in wave order (barrier-separated) to stay deterministic
      for (int w = 0; w < 4; ++w) {
        if ((threadIdx.x >> 6) == w && tail) { acc[f].x += m; acc[f].y += v; }
        __syncthreads();
      }
    }
  }
  for (int i = threadIdx.x; i < FB && f0 + i < cnt; i += 256) {
    const double old = sold[i];
    double mean = acc[i].x - old * acc[i].y, var = 1.0 / (1.0 + acc[i].y);
    const double nv = -var * mean;
    vstep[f0 + i] = make_double2(old, (old - nv) * 1e-3);
  }
}

// K2: blockIdx -> (tile, chunk) with the tile's workgroups in one XCD's share of the grid
template <int R, bool NT>
__global__ __launch_bounds__(256) void apply_perm_k(const double2* __restrict__ src, double2* __restrict__ dst, const uint16_t* __restrict__ fidx, const uint32_t* __restrict__ perm,
                                                    const double2* __restrict__ vstep, int ts, int n_tiles, int B, int64_t n) {
  const int b = blockIdx.x;
  const int x = b & 7, q = b >> 3;
  const int tile = (q / B) * 8 + x, chunk = q % B;
  if (tile >= n_tiles) return;
  const int64_t base = (int64_t)tile << ts;
  const int64_t i0 = base + (int64_t)chunk * (256 * R) + threadIdx.x;
  double2 c[R]; uint32_t f[R], pm[R]; double2 s[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int64_t i = i0 + u * 256, ic = i < n ? i : n - 1;
    c[u] = NT ? ntload(src + ic) : src[ic];
    f[u] = NT ? ntl(fidx + ic) : fidx[ic];
    pm[u] = NT ? ntl(perm + ic) : perm[ic];
  }
#pragma unroll
  for (int u = 0; u < R; ++u) s[u] = vstep[f[u]];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int64_t i = i0 + u * 256;
    if (i >= n) continue;
    const double h = c[u].x - s[u].x;
    dst[base + pm[u]] = make_double2(c[u].x - s[u].y, c[u].y - h * s[u].y);
  }
}

int main(int argc, char** argv) {
  const int ts = argc > 1 ? atoi(argv[1]) : 17;
  const int n_tiles = argc > 2 ? atoi(argv[2]) : 77;
  const uint32_t cnt = argc > 3 ? (uint32_t)atoi(argv[3]) : 33334u;
  const int64_t T = 1LL << ts, n = (int64_t)n_tiles * T;
  printf("n = %lld rows, %d tiles of %lld, %u features per level\n", (long long)n, n_tiles, (long long)T, cnt);
  std::mt19937_64 rng(7);
  std::vector<uint16_t> fa(n), fb(n);
  for (int64_t r = 0; r < n; ++r) { fa[r] = (uint16_t)(rng() % cnt); fb[r] = (uint16_t)(rng() % cnt); }
  std::vector<uint32_t> ordA(n), ordB(n), posB(n), perm(n), toff((size_t)n_tiles * (cnt + 1));
  std::vector<uint16_t> fidx(n);
  for (int t = 0; t < n_tiles; ++t) {
    const int64_t b = (int64_t)t * T;
    std::iota(ordA.begin() + b, ordA.begin() + b + T, 0u); std::iota(ordB.begin() + b, ordB.begin() + b + T, 0u);
    std::stable_sort(ordA.begin() + b, ordA.begin() + b + T, [&](uint32_t x, uint32_t y) { return fa[b + x] < fa[b + y]; });
    std::stable_sort(ordB.begin() + b, ordB.begin() + b + T, [&](uint32_t x, uint32_t y) { return fb[b + x] < fb[b + y]; });
    for (int64_t i = 0; i < T; ++i) posB[b + ordB[b + i]] = (uint32_t)i;
    uint32_t* off = toff.data() + (size_t)t * (cnt + 1);
    std::fill(off, off + cnt + 1, 0u);
    for (int64_t i = 0; i < T; ++i) { perm[b + i] = posB[b + ordA[b + i]]; fidx[b + i] = fa[b + ordA[b + i]]; off[fidx[b + i] + 1]++; }
    for (uint32_t f = 0; f < cnt; ++f) off[f + 1] += off[f];
  }
  double2 *src, *dst, *vstep; uint16_t* d_fidx; uint32_t *d_perm, *d_toff; double* vf;
  CK(hipMalloc(&src, n * 16)); CK(hipMalloc(&dst, n * 16)); CK(hipMalloc(&vstep, cnt * 16)); CK(hipMalloc(&d_fidx, n * 2)); CK(hipMalloc(&d_perm, n * 4));
  CK(hipMalloc(&d_toff, toff.size() * 4)); CK(hipMalloc(&vf, cnt * 8));
  { std::vector<double2> h(n); for (int64_t i = 0; i < n; ++i) h[i] = make_double2(1e-3 * (double)(rng() % 1000), 1e-3 * (double)(rng() % 1000) - 0.5); CK(hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice)); }
  CK(hipMemset(dst, 0, n * 16)); CK(hipMemset(vf, 0, cnt * 8)); CK(hipMemset(vstep, 0, cnt * 16));
  CK(hipMemcpy(d_fidx, fidx.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_perm, perm.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_toff, toff.data(), toff.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t nf1 = cnt + 1;
  auto timeit = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f us   %6.2f TB/s of %.0f MB\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, bytes / 1e6);
  };
  timeit("copy 16-B pairs", [&] { hipLaunchKernelGGL(copy_k, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, src, dst, n); }, 32.0 * n);
  timeit("K1 sums+step LG=1 (256 features / WG)", [&] { hipLaunchKernelGGL(sums_step_k<1>, dim3((cnt + 255) / 256), dim3(256), 0, 0, d_toff, nf1, cnt, src, ts, n_tiles, vf, vstep); }, 16.0 * n);
  timeit("K1 sums+step LG=4 (64 features / WG)", [&] { hipLaunchKernelGGL(sums_step_k<4>, dim3((cnt + 63) / 64), dim3(256), 0, 0, d_toff, nf1, cnt, src, ts, n_tiles, vf, vstep); }, 16.0 * n);
  timeit("K1 sums+step LG=8 (32 features / WG)", [&] { hipLaunchKernelGGL(sums_step_k<8>, dim3((cnt + 31) / 32), dim3(256), 0, 0, d_toff, nf1, cnt, src, ts, n_tiles, vf, vstep); }, 16.0 * n);
  timeit("K1b flat sums FB=64", [&] { hipLaunchKernelGGL(sums_flat_k<64>, dim3((cnt + 63) / 64), dim3(256), 0, 0, d_toff, nf1, cnt, src, d_fidx, ts, n_tiles, vf, vstep); }, 18.0 * n);
  timeit("K1b flat sums FB=128", [&] { hipLaunchKernelGGL(sums_flat_k<128>, dim3((cnt + 127) / 128), dim3(256), 0, 0, d_toff, nf1, cnt, src, d_fidx, ts, n_tiles, vf, vstep); }, 18.0 * n);
#define K2(Rv, NTv) { const int B = (int)((T + 256 * Rv - 1) / (256 * Rv)); const unsigned g = (unsigned)(((n_tiles + 7) / 8) * 8 * B); \
    timeit("K2 apply+permute R=" #Rv " NT=" #NTv, [&] { hipLaunchKernelGGL((apply_perm_k<Rv, NTv>), dim3(g), dim3(256), 0, 0, src, dst, d_fidx, d_perm, vstep, ts, n_tiles, B, n); }, 38.0 * n); }
  K2(4, true) K2(8, true) K2(8, false) K2(2, true) K2(1, true)
  // check: dst holds a permutation of the corrected pairs (sum of e preserved up to the corrections; here just a checksum of positions written)
  { std::vector<double2> h(n); CK(hipMemcpy(h.data(), dst, n * 16, hipMemcpyDeviceToHost)); int64_t zeros = 0; for (int64_t i = 0; i < n; ++i) zeros += (h[i].x == 0.0 && h[i].y == 0.0); printf("unwritten slots: %lld\n", (long long)zeros); }
  return 0;
}
