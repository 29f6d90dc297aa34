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

// what does the ACCESS PATTERN of K1 cost by itself?  workgroup b reads PIECE pairs of every tile (piece b of the tile's slice), 8 tiles in flight per thread
// (thread = one pair of the piece); rot: workgroup b starts at tile b % n_tiles (decorrelates which slices are hot together); contig: the same bytes, contiguous
template <int PIECE, bool ROT, bool CONTIG>
__global__ __launch_bounds__(256) void pieces_k(const double2* __restrict__ src, int ts, int n_tiles, double* __restrict__ out) {
  constexpr int TPI = 256 / PIECE;   // tiles covered by one load instruction of the workgroup
  const int lane_e = threadIdx.x % PIECE, lane_t = threadIdx.x / PIECE;
  double acc = 0.0;
  const int t_rot = ROT ? blockIdx.x % n_tiles : 0;
  for (int t0 = 0; t0 < n_tiles; t0 += TPI * 8) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int t = t0 + u * TPI + lane_t;
      const bool in = t < n_tiles;
      t = in ? (t + t_rot) % n_tiles : 0;
      const size_t at = CONTIG ? ((size_t)blockIdx.x * n_tiles + t) * PIECE + lane_e : ((size_t)t << ts) + (size_t)blockIdx.x * PIECE + lane_e;
      v[u] = ntload(src + at);
      if (!in) v[u] = make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u].x * v[u].y;
  }
  if (acc == 12345.678) out[0] = acc;
}


// ---- the product's K1 (fm_als_tiled.hip: als_order_sums_k, copied by profiles/probes/sync_k1.py) built with -DFMX_K1_TIMING: clock stamps between its phases
#define FMX_K1_TIMING 1
namespace fmx { constexpr int WG_THREADS = 256; struct SweepDyn { int f, pad; double alpha, lambda, mu; const double* znorm; };
__device__ __forceinline__ bool bad_number_t(double x) { return isnan(x) || isinf(x); }
template <bool NT, typename T> __device__ __forceinline__ T stream_load(const T* p) { return __builtin_nontemporal_load(p); }
template <bool NT> __device__ __forceinline__ double2 stream_load(const double2* p) { const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p)); return make_double2(v.x, v.y); }
#ifdef FMX_K1_TIMING
__device__ unsigned long long fmx_k1_ticks[8];
#define FMX_K1_STAMP(slot) do { const unsigned long long now_ = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&fmx_k1_ticks[slot], now_ - k1_last_); k1_last_ = now_; } while (0)
#define FMX_K1_BEGIN unsigned long long k1_last_ = wall_clock64()
#else
#define FMX_K1_STAMP(slot) do {} while (0)
#define FMX_K1_BEGIN do {} while (0)
#endif
template <bool UNIT, int FBMAX, int TB, int CH, int DEPTH>
__global__ __launch_bounds__(WG_THREADS) void als_order_sums_k(const uint32_t* __restrict__ toff, size_t nf1, uint32_t lvl0, uint32_t cnt, int fb, const int64_t* __restrict__ tile_base,
                                                               const float* __restrict__ tval, const double2* __restrict__ src, int tshift, int n_tiles,
                                                               const uint32_t* __restrict__ feats, double* __restrict__ P, int kp, const SweepDyn* __restrict__ dyn,
                                                               double2* __restrict__ vstep) {
  constexpr int LG = WG_THREADS / FBMAX;       // lanes per feature: the tiles a chunk touches are dealt round-robin to the lanes of a group
  constexpr int PER = CH / WG_THREADS;
  static_assert((TB & (TB - 1)) == 0 && TB <= WG_THREADS && FBMAX == 64, "TB: a power of two, one thread per tile in the prefix step; one row of offsets per wave instruction");
  __shared__ uint32_t o[TB][FBMAX];            // list offsets of the workgroup's features in the batch's tiles (as stored: relative to the tile's first entry); fb <= FBMAX - 1
  __shared__ uint32_t vstart[TB + 1];          // the batch's runs laid end to end
  __shared__ uint32_t blk[TB];                 // position of each run's first pair inside its tile's level block
  __shared__ int64_t xbase[UNIT ? 1 : TB];     // first entry of each tile's level block in tval
  __shared__ double2 lp[CH];
  __shared__ float lx[UNIT ? 1 : CH];
  const uint32_t f0 = blockIdx.x * (uint32_t)fb;
  const int g = threadIdx.x / LG, lane = threadIdx.x % LG;
  const uint32_t fi = f0 + g;
  const bool live = g < fb && fi < cnt;
  const int gc = min(g, fb - 1);               // (lane groups beyond fb walk empty lists)
  const uint32_t feat = feats[live ? fi : cnt - 1];
  const int f = dyn->f;
  const double old = P[(size_t)feat * kp + f];
  double mean = 0.0, var = 0.0;
  FMX_K1_BEGIN;
  // position v of the virtual sequence -> its tile of the batch (the last tb with vstart[tb] <= v; empty runs are skipped by construction)
  auto tile_of = [&](uint32_t v) { int tb = 0;
#pragma unroll
    for (int st = TB / 2; st > 0; st >>= 1) tb += (vstart[tb + st] <= v) ? st : 0;
    return tb; };
  for (int t0 = 0; t0 < n_tiles; t0 += TB) {
    const int nb = min(TB, n_tiles - t0);
    __syncthreads();                            // (the walkers of the previous batch are done with o / vstart / lp)
    {
      // every thread's offset loads go out together, then land in LDS (a load-store loop would wait for one load per trip): wave w takes the tiles
      // w, w + 4, ... of the batch, lane j the offset of feature f0 + j (fb <= 63: one row of offsets is one wave instruction)
      constexpr int TPW = TB / (WG_THREADS / 64);
      const int wv = threadIdx.x >> 6, j = threadIdx.x & 63;
      const uint32_t fj = min(f0 + (uint32_t)min(j, fb), cnt);
      uint32_t ov[TPW];
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int tb = min(wv + q * (WG_THREADS / 64), nb - 1);
        ov[q] = stream_load<true>(toff + (size_t)(t0 + tb) * nf1 + lvl0 + fj);
      }
      uint32_t bv = 0; int64_t tbv = 0;
      if ((int)threadIdx.x < nb) { bv = stream_load<true>(toff + (size_t)(t0 + threadIdx.x) * nf1 + lvl0); if (!UNIT) tbv = tile_base[t0 + threadIdx.x]; }
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int tb = wv + q * (WG_THREADS / 64);
        if (tb < nb) o[tb][j] = ov[q];
      }
      if ((int)threadIdx.x < nb) { blk[threadIdx.x] = bv; if (!UNIT) xbase[threadIdx.x] = tbv + (int64_t)bv; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                     // exclusive prefix of the run lengths: one wave, TB / 64 values per lane
      uint32_t carry = 0;
      for (int b0 = 0; b0 < TB; b0 += 64) {
        const int tb = b0 + threadIdx.x;
        const uint32_t len = (tb < nb) ? o[tb][fb] - o[tb][0] : 0u;
        uint32_t inc = len;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) { const uint32_t up = __shfl_up(inc, ofs); if ((int)threadIdx.x >= ofs) inc += up; }
        if (tb < TB) vstart[tb] = carry + inc - len;
        if (tb < nb) blk[tb] = o[tb][0] - blk[tb];      // the run's first pair inside the tile's level block
        carry += __shfl(inc, 63);
      }
      if (threadIdx.x == 0) vstart[TB] = carry;
    }
    __syncthreads();
    FMX_K1_STAMP(0);                            // offsets + prefix
    const uint32_t total = vstart[TB];
    double2 pv[DEPTH][PER]; float xv[DEPTH][PER];
    auto fetch = [&](double2 (&pb)[PER], float (&xb)[PER], uint32_t c0) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const uint32_t v = min(c0 + threadIdx.x + u * WG_THREADS, total - 1);
        const int tb = tile_of(v);
        const uint32_t in_block = blk[tb] + (v - vstart[tb]);
        pb[u] = stream_load<true>(src + ((size_t)(t0 + tb) << tshift) + in_block);
        xb[u] = UNIT ? 1.0f : stream_load<true>(tval + xbase[tb] + in_block);
      }
    };
    // Every fetch is UNCONDITIONAL (positions past the end are clamped onto the last pair: one request per instruction): the number of loads outstanding at
    // every wait is then a compile-time constant and the compiler emits counted waits -- with a fetch under a condition it waited for ALL loads before every
    // LDS store (s_waitcnt vmcnt(0): the ISA of the first version), which is a prefetch depth of one whatever DEPTH says.
    if (total > 0) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) fetch(pv[d], xv[d], (uint32_t)d * CH);
      FMX_K1_STAMP(1);                          // the first fetches' address work
      for (uint32_t cbase = 0; cbase < total; cbase += DEPTH * CH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const uint32_t c0 = cbase + (uint32_t)d * CH;   // (may lie past the end in the last round: nothing is stored or walked then)
#pragma unroll
          for (int u = 0; u < PER; ++u) {
            const uint32_t v = c0 + threadIdx.x + u * WG_THREADS;
            if (v < total) { lp[v - c0] = pv[d][u]; if (!UNIT) lx[v - c0] = xv[d][u]; }
          }
          FMX_K1_STAMP(2);                      // wait for the chunk's loads + LDS stores
          __syncthreads();
          FMX_K1_STAMP(3);                      // barrier
          fetch(pv[d], xv[d], c0 + DEPTH * CH);   // refill the registers just emptied: DEPTH chunks ahead
          FMX_K1_STAMP(4);                      // the refill's address work (searches) and issue
          if (c0 < total) {
            const uint32_t c1 = min(c0 + CH, total);
            const int t_lo = tile_of(c0), t_hi = tile_of(c1 - 1);
            for (int tb = t_lo + lane; tb <= t_hi; tb += LG) {
              const uint32_t a = max(vstart[tb] + o[tb][gc] - o[tb][0], c0), b = min(vstart[tb] + o[tb][gc + 1] - o[tb][0], c1);
              // four entries' LDS reads go out together (clamped onto the list's last entry, added under a test): a loop that reads one entry per trip
              // pays the LDS latency per entry, and the longest list of the wave's 64 sets the trip count (ISA + timing: 2/3 of the kernel's time)
              for (uint32_t v = a; v < b; v += 4) {
                double2 c[4]; float x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { const uint32_t vi = min(v + i, b - 1) - c0; c[i] = lp[vi]; x[i] = UNIT ? 1.0f : lx[vi]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const float xx = x[i] * x[i];
                  const double h = (double)x[i] * c[i].x - (double)xx * old;   // :310-317
                  if (live && v + i < b) { mean += h * c[i].y; var += h * h; }
                }
              }
            }
          }
          FMX_K1_STAMP(5);                      // the walk
          __syncthreads();                      // (the walkers are done with lp before the next chunk lands in it)
          FMX_K1_STAMP(6);                      // barrier
        }
      }
    }
  }
#pragma unroll
  for (int ofs = 1; ofs < LG; ofs <<= 1) { mean += __shfl_xor(mean, ofs); var += __shfl_xor(var, ofs); }   // (a + b == b + a: every lane of the group holds the same bits)
  if (lane != 0 || !live) return;
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  const double* __restrict__ znorm = dyn->znorm;
  mean -= old * var;                               // :318
  var = 1.0 / (lambda + alpha * var);              // :319
  mean = -var * (alpha * mean - mu * lambda);      // :320
  double nv = bad_number_t(var) ? 0.0 : (znorm ? mean + sqrt(var) * znorm[feat] : mean);
  if (bad_number_t(nv)) { vstep[fi] = make_double2(old, nan("")); return; }  // CHECK_PARAM (:336): the old value stays; NaN tells the apply pass to leave the rows alone
  P[(size_t)feat * kp + f] = nv;
  vstep[fi] = make_double2(old, old - nv);
}

}  // namespace fmx

// ---- the product's K1 with knock-out switches (mode bits: 1 the LDS walk, 2 the pair loads, 4 the tile search of every load, 8 the chunk loop at all)
#undef FMX_K1_TIMING
#undef FMX_K1_STAMP
#undef FMX_K1_BEGIN
#define FMX_K1_STAMP(x) do {} while (0)
#define FMX_K1_BEGIN do {} while (0)
namespace fmx {
template <bool UNIT, int FBMAX, int TB, int CH, int DEPTH>
__global__ __launch_bounds__(WG_THREADS) void k1_ko_k(const uint32_t* __restrict__ toff, size_t nf1, uint32_t lvl0, uint32_t cnt, int fb, const int64_t* __restrict__ tile_base,
                                                               const float* __restrict__ tval, const double2* __restrict__ src, int tshift, int n_tiles,
                                                               const uint32_t* __restrict__ feats, double* __restrict__ P, int kp, const SweepDyn* __restrict__ dyn,
                                                               double2* __restrict__ vstep, int mode) {
  constexpr int LG = WG_THREADS / FBMAX;       // lanes per feature: the tiles a chunk touches are dealt round-robin to the lanes of a group
  constexpr int PER = CH / WG_THREADS;
  static_assert((TB & (TB - 1)) == 0 && TB <= WG_THREADS && FBMAX == 64, "TB: a power of two, one thread per tile in the prefix step; one row of offsets per wave instruction");
  __shared__ uint32_t o[TB][FBMAX];            // list offsets of the workgroup's features in the batch's tiles (as stored: relative to the tile's first entry); fb <= FBMAX - 1
  __shared__ uint32_t vstart[TB + 1];          // the batch's runs laid end to end
  __shared__ uint32_t blk[TB];                 // position of each run's first pair inside its tile's level block
  __shared__ int64_t xbase[UNIT ? 1 : TB];     // first entry of each tile's level block in tval
  __shared__ double2 lp[CH];
  __shared__ float lx[UNIT ? 1 : CH];
  const uint32_t f0 = blockIdx.x * (uint32_t)fb;
  const int g = threadIdx.x / LG, lane = threadIdx.x % LG;
  const uint32_t fi = f0 + g;
  const bool live = g < fb && fi < cnt;
  const int gc = min(g, fb - 1);               // (lane groups beyond fb walk empty lists)
  const uint32_t feat = feats[live ? fi : cnt - 1];
  const int f = dyn->f;
  const double old = P[(size_t)feat * kp + f];
  double mean = 0.0, var = 0.0;
  
  // position v of the virtual sequence -> its tile of the batch (the last tb with vstart[tb] <= v; empty runs are skipped by construction)
  auto tile_of = [&](uint32_t v) { int tb = 0;
#pragma unroll
    for (int st = TB / 2; st > 0; st >>= 1) tb += (vstart[tb + st] <= v) ? st : 0;
    return tb; };
  for (int t0 = 0; t0 < n_tiles; t0 += TB) {
    const int nb = min(TB, n_tiles - t0);
    __syncthreads();                            // (the walkers of the previous batch are done with o / vstart / lp)
    {
      // every thread's offset loads go out together, then land in LDS (a load-store loop would wait for one load per trip): wave w takes the tiles
      // w, w + 4, ... of the batch, lane j the offset of feature f0 + j (fb <= 63: one row of offsets is one wave instruction)
      constexpr int TPW = TB / (WG_THREADS / 64);
      const int wv = threadIdx.x >> 6, j = threadIdx.x & 63;
      const uint32_t fj = min(f0 + (uint32_t)min(j, fb), cnt);
      uint32_t ov[TPW];
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int tb = min(wv + q * (WG_THREADS / 64), nb - 1);
        ov[q] = stream_load<true>(toff + (size_t)(t0 + tb) * nf1 + lvl0 + fj);
      }
      uint32_t bv = 0; int64_t tbv = 0;
      if ((int)threadIdx.x < nb) { bv = stream_load<true>(toff + (size_t)(t0 + threadIdx.x) * nf1 + lvl0); if (!UNIT) tbv = tile_base[t0 + threadIdx.x]; }
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int tb = wv + q * (WG_THREADS / 64);
        if (tb < nb) o[tb][j] = ov[q];
      }
      if ((int)threadIdx.x < nb) { blk[threadIdx.x] = bv; if (!UNIT) xbase[threadIdx.x] = tbv + (int64_t)bv; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                     // exclusive prefix of the run lengths: one wave, TB / 64 values per lane
      uint32_t carry = 0;
      for (int b0 = 0; b0 < TB; b0 += 64) {
        const int tb = b0 + threadIdx.x;
        const uint32_t len = (tb < nb) ? o[tb][fb] - o[tb][0] : 0u;
        uint32_t inc = len;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) { const uint32_t up = __shfl_up(inc, ofs); if ((int)threadIdx.x >= ofs) inc += up; }
        if (tb < TB) vstart[tb] = carry + inc - len;
        if (tb < nb) blk[tb] = o[tb][0] - blk[tb];      // the run's first pair inside the tile's level block
        carry += __shfl(inc, 63);
      }
      if (threadIdx.x == 0) vstart[TB] = carry;
    }
    __syncthreads();
    
    const uint32_t total = vstart[TB];
    double2 pv[DEPTH][PER]; float xv[DEPTH][PER];
    auto fetch = [&](double2 (&pb)[PER], float (&xb)[PER], uint32_t c0) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const uint32_t v = min(c0 + threadIdx.x + u * WG_THREADS, total - 1);
        const int tb = (mode & 4) ? tile_of(v) : (int)(v >> 12);
        const uint32_t in_block = blk[tb] + (v - vstart[tb]);
        if (mode & 2) pb[u] = stream_load<true>(src + ((size_t)(t0 + tb) << tshift) + in_block); else pb[u] = make_double2((double)in_block, 1.0);
        xb[u] = UNIT ? 1.0f : stream_load<true>(tval + xbase[tb] + in_block);
      }
    };
    // Every fetch is UNCONDITIONAL (positions past the end are clamped onto the last pair: one request per instruction): the number of loads outstanding at
    // every wait is then a compile-time constant and the compiler emits counted waits -- with a fetch under a condition it waited for ALL loads before every
    // LDS store (s_waitcnt vmcnt(0): the ISA of the first version), which is a prefetch depth of one whatever DEPTH says.
    if ((mode & 8) && total > 0) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) fetch(pv[d], xv[d], (uint32_t)d * CH);
      
      for (uint32_t cbase = 0; cbase < total; cbase += DEPTH * CH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const uint32_t c0 = cbase + (uint32_t)d * CH;   // (may lie past the end in the last round: nothing is stored or walked then)
#pragma unroll
          for (int u = 0; u < PER; ++u) {
            const uint32_t v = c0 + threadIdx.x + u * WG_THREADS;
            if (v < total) { lp[v - c0] = pv[d][u]; if (!UNIT) lx[v - c0] = xv[d][u]; }
          }
          
          __syncthreads();
          
          fetch(pv[d], xv[d], c0 + DEPTH * CH);   // refill the registers just emptied: DEPTH chunks ahead
          
          if ((mode & 1) && c0 < total) {
            const uint32_t c1 = min(c0 + CH, total);
            const int t_lo = tile_of(c0), t_hi = tile_of(c1 - 1);
            for (int tb = t_lo + lane; tb <= t_hi; tb += LG) {
              const uint32_t a = max(vstart[tb] + o[tb][gc] - o[tb][0], c0), b = min(vstart[tb] + o[tb][gc + 1] - o[tb][0], c1);
              // four entries' LDS reads go out together (clamped onto the list's last entry, added under a test): a loop that reads one entry per trip
              // pays the LDS latency per entry, and the longest list of the wave's 64 sets the trip count (ISA + timing: 2/3 of the kernel's time)
              for (uint32_t v = a; v < b; v += 4) {
                double2 c[4]; float x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { const uint32_t vi = min(v + i, b - 1) - c0; c[i] = lp[vi]; x[i] = UNIT ? 1.0f : lx[vi]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const float xx = x[i] * x[i];
                  const double h = (double)x[i] * c[i].x - (double)xx * old;   // :310-317
                  if (live && v + i < b) { mean += h * c[i].y; var += h * h; }
                }
              }
            }
          }
          
          __syncthreads();                      // (the walkers are done with lp before the next chunk lands in it)
          
        }
      }
    }
  }
#pragma unroll
  for (int ofs = 1; ofs < LG; ofs <<= 1) { mean += __shfl_xor(mean, ofs); var += __shfl_xor(var, ofs); }   // (a + b == b + a: every lane of the group holds the same bits)
  if (lane != 0 || !live) return;
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  const double* __restrict__ znorm = dyn->znorm;
  mean -= old * var;                               // :318
  var = 1.0 / (lambda + alpha * var);              // :319
  mean = -var * (alpha * mean - mu * lambda);      // :320
  double nv = bad_number_t(var) ? 0.0 : (znorm ? mean + sqrt(var) * znorm[feat] : mean);
  if (bad_number_t(nv)) { vstep[fi] = make_double2(old, nan("")); return; }  // CHECK_PARAM (:336): the old value stays; NaN tells the apply pass to leave the rows alone
  P[(size_t)feat * kp + f] = nv;
  vstep[fi] = make_double2(old, old - nv);
}

}  // namespace fmx

// ---- K1, lean: NIT tiles per trip, one LDS slot of S pairs per tile (every run of the workgroup's FB features in a tile is at most S pairs: checked by the
// host), thread e of the workgroup loads pair e of each of the trip's runs (no search, no prefix: the mapping is static), the lanes of a feature's group walk
// the trip's tiles; run bounds and list bounds come straight from global memory TWO trips ahead, the pairs one trip ahead: no load waits on another.
namespace fmx {
template <bool UNIT, int FB, int NIT, int S>
__global__ __launch_bounds__(WG_THREADS) void k1_lean_k(const uint32_t* __restrict__ toff, size_t nf1, uint32_t lvl0, uint32_t cnt, const int64_t* __restrict__ tile_base,
                                                        const float* __restrict__ tval, const double2* __restrict__ src, int tshift, int n_tiles,
                                                        const uint32_t* __restrict__ feats, double* __restrict__ P, int kp, const SweepDyn* __restrict__ dyn,
                                                        double2* __restrict__ vstep) {
  constexpr int LG = WG_THREADS / FB;          // lanes per feature
  constexpr int TPL = (NIT + LG - 1) / LG;     // tiles per lane and trip
  static_assert(S <= WG_THREADS, "");
  __shared__ double2 lp[NIT][S];
  __shared__ float lx[UNIT ? 1 : NIT * S];
  const uint32_t f0 = blockIdx.x * FB;
  const int g = threadIdx.x / LG, lane = threadIdx.x % LG;
  const uint32_t fi = f0 + g;
  const bool live = fi < cnt;
  const uint32_t feat = feats[live ? fi : cnt - 1];
  const int f = dyn->f;
  const double old = P[(size_t)feat * kp + f];
  const uint32_t fa = min(f0 + g, cnt), fb = min(f0 + g + 1, cnt), fend = min(f0 + FB, cnt);
  double mean = 0.0, var = 0.0;
  // run bounds of the trip's tiles: lane u of every wave loads tile u's three offsets with VECTOR loads (uniform scalar loads share the LDS counter: the
  // walk's first LDS read would wait for them) and the values are read out of that lane when the pairs are fetched
  struct Offs { uint32_t ra, rb, rbase, la[TPL], lb[TPL]; int64_t xb; };
  const int wl = threadIdx.x & 63;
  auto offsets = [&](int t0, Offs& o) {
    {
      const int t = min(t0 + min(wl, NIT - 1), n_tiles - 1);
      const uint32_t* off = toff + (size_t)t * nf1 + lvl0;
      o.ra = stream_load<true>(off + f0); o.rb = stream_load<true>(off + fend); o.rbase = stream_load<true>(off);
      o.xb = UNIT ? 0 : stream_load<true>(tile_base + t);
    }
#pragma unroll
    for (int q = 0; q < TPL; ++q) {
      const int u = lane + q * LG;
      const int t = min(t0 + u, n_tiles - 1);
      const uint32_t* off = toff + (size_t)t * nf1 + lvl0;
      const bool in = u < NIT && t0 + u < n_tiles;
      o.la[q] = stream_load<true>(off + fa); o.lb[q] = in ? stream_load<true>(off + fb) : 0u;
      if (!in) o.lb[q] = o.la[q];
    }
  };
  double2 pv[NIT]; float xv[NIT];
  uint32_t rl_cur[NIT];
  auto fetch = [&](int t0, const Offs& o) {
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int t = min(t0 + u, n_tiles - 1);
      const uint32_t a = __builtin_amdgcn_readlane(o.ra, u), b = __builtin_amdgcn_readlane(o.rb, u), base = __builtin_amdgcn_readlane(o.rbase, u);
      const uint32_t rl = (t0 + u < n_tiles) ? b - a : 0u;
      rl_cur[u] = rl;
      const uint32_t e = min((uint32_t)threadIdx.x, rl > 0 ? rl - 1 : 0u);
      pv[u] = stream_load<true>(src + ((size_t)t << tshift) + (a - base) + e);
      if (!UNIT) { const int64_t xb = ((int64_t)__builtin_amdgcn_readlane((int)(o.xb >> 32), u) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)o.xb, u);
                   xv[u] = stream_load<true>(tval + xb + a + e); } else xv[u] = 1.0f;
    }
  };
  Offs cur, nxt;
  offsets(0, cur);
  offsets(NIT, nxt);
  fetch(0, cur);
  for (int t0 = 0; t0 < n_tiles; t0 += NIT) {
    uint32_t wa[TPL], wb[TPL];
#pragma unroll
    for (int q = 0; q < TPL; ++q) {   // this lane's lists, relative to their runs (the run start of tile u sits in lane u of cur.ra)
      const int u = min(lane + q * LG, NIT - 1);
      const uint32_t a0 = __shfl(cur.ra, u);
      wa[q] = cur.la[q] - a0; wb[q] = cur.lb[q] - a0;
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u)
      if (threadIdx.x < rl_cur[u]) { lp[u][threadIdx.x] = pv[u]; if (!UNIT) lx[u * S + threadIdx.x] = xv[u]; }
    __syncthreads();
    Offs after;
    if (t0 + NIT < n_tiles) fetch(t0 + NIT, nxt);          // the next trip's pairs ...
    if (t0 + 2 * NIT < n_tiles) offsets(t0 + 2 * NIT, after);   // ... and the bounds of the one after it, in flight while this trip's runs are walked
    else { after.ra = after.rb = after.rbase = 0; after.xb = 0; for (int q = 0; q < TPL; ++q) { after.la[q] = 0; after.lb[q] = 0; } }
#pragma unroll
    for (int q = 0; q < TPL; ++q) {
      const int u = lane + q * LG;
      if (u < NIT)
        for (uint32_t v = wa[q]; v < wb[q]; ++v) {
          const double2 c = lp[u][v];
          const float x = UNIT ? 1.0f : lx[u * S + v];
          const float xx = x * x;
          const double h = (double)x * c.x - (double)xx * old;
          mean += h * c.y; var += h * h;
        }
    }
    __syncthreads();
    cur = nxt; nxt = after;
  }
#pragma unroll
  for (int ofs = 1; ofs < LG; ofs <<= 1) { mean += __shfl_xor(mean, ofs); var += __shfl_xor(var, ofs); }
  if (lane != 0 || !live) return;
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  const double* __restrict__ znorm = dyn->znorm;
  mean -= old * var;
  var = 1.0 / (lambda + alpha * var);
  mean = -var * (alpha * mean - mu * lambda);
  double nv = bad_number_t(var) ? 0.0 : (znorm ? mean + sqrt(var) * znorm[feat] : mean);
  if (bad_number_t(nv)) { vstep[fi] = make_double2(old, nan("")); return; }
  P[(size_t)feat * kp + f] = nv;
  vstep[fi] = make_double2(old, old - nv);
}
}  // namespace fmx


// ---- K1, one WAVE per 16 features, no LDS arrays, no barrier: per tile the wave's run (<= 64 pairs in the regular case) is ONE load instruction, one pair per
// lane; the four lanes of a feature fetch "their" entries of its list from the lanes that hold them (cross-lane reads).  Software pipeline over batches of NA
// tiles: offsets two batches ahead, pairs one batch ahead -- no load of the loop waits on another.  Runs longer than 64 pairs: the rest straight from memory.
namespace fmx {
template <bool UNIT, int NA, bool TR>
__global__ __launch_bounds__(256) void k1_wave_k(const uint32_t* __restrict__ toff, size_t nf1, uint32_t lvl0, uint32_t cnt, const int64_t* __restrict__ tile_base,
                                                 const float* __restrict__ tval, const double2* __restrict__ src, int tshift, int n_tiles,
                                                 const uint32_t* __restrict__ feats, double* __restrict__ P, int kp, const SweepDyn* __restrict__ dyn,
                                                 double2* __restrict__ vstep) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t f0 = wave * 16;
  if (f0 >= cnt) return;
  const int g = lane >> 2, j = lane & 3;
  const uint32_t fi = f0 + g;
  const bool live = fi < cnt;
  const uint32_t feat = feats[live ? fi : cnt - 1];
  const int f = dyn->f;
  const double old = P[(size_t)feat * kp + f];
  const uint32_t fo = min(f0 + (uint32_t)min(lane, 16), cnt);   // lanes 0..16 hold the offsets of features f0 .. f0 + 16
  double mean = 0.0, var = 0.0;
  auto sh = [&](double v, int from) { return __hiloint2double(__shfl(__double2hiint(v), from), __shfl(__double2loint(v), from)); };
  struct Offs { uint32_t off[NA], base[NA]; };
  struct Pairs { double2 pv[NA]; uint32_t r0[NA], rl[NA], la[NA], lb[NA]; };
  auto load_offs = [&](int t0, Offs& o) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int t = min(t0 + u, n_tiles - 1);
      const uint32_t* row = toff + (size_t)t * nf1 + lvl0;
      o.off[u] = stream_load<true>(row + fo);
      o.base[u] = stream_load<true>(row);
    }
  };
  auto load_pairs = [&](int t0, const Offs& o, Pairs& p) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int t = min(t0 + u, n_tiles - 1);
      const bool in = t0 + u < n_tiles;
      p.r0[u] = __shfl(o.off[u], 0);
      p.rl[u] = in ? __shfl(o.off[u], 16) - p.r0[u] : 0u;
      p.la[u] = in ? __shfl(o.off[u], g) - p.r0[u] : 0u;
      p.lb[u] = in ? __shfl(o.off[u], g + 1) - p.r0[u] : 0u;
      const uint32_t e = min((uint32_t)lane, p.rl[u] > 0 ? p.rl[u] - 1 : 0u);
      p.r0[u] -= o.base[u];   // the run's first pair inside the tile's level block
      p.pv[u] = stream_load<true>(src + ((size_t)t << tshift) + p.r0[u] + e);
    }
  };
  Offs o_next, o_after; Pairs cur, nxt;
  load_offs(0, o_next);
  load_offs(NA, o_after);
  load_pairs(0, o_next, cur);
  for (int t0 = 0; t0 < n_tiles; t0 += NA) {
    load_pairs(t0 + NA, o_after, nxt);          // (past the end: clamped addresses, empty lists)
    load_offs(t0 + 2 * NA, o_next);
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const uint32_t la = cur.la[u], lb = cur.lb[u], hi = min(lb, 64u);
      for (uint32_t r = 0; __any(la + r < hi); r += 4) {   // (a UNIFORM trip count: a lane that left the loop could no longer lend its pair to the others)
        const uint32_t idx = la + r + j;
        const double qx = sh(cur.pv[u].x, (int)min(idx, 63u)), ey = sh(cur.pv[u].y, (int)min(idx, 63u));
        const double h = qx - old;
        if (live && idx < hi) { mean += h * ey; var += h * h; }
      }
      if (cur.rl[u] > 64) {   // a long run: the rest straight from memory
        const int t = min(t0 + u, n_tiles - 1);
        for (uint32_t idx = max(la, 64u) + j; idx < lb; idx += 4) {
          const double2 c = src[((size_t)t << tshift) + cur.r0[u] + idx];
          const double h = c.x - old;
          if (live) { mean += h * c.y; var += h * h; }
        }
      }
    }
    cur = nxt; { Offs tmp = o_after; o_after = o_next; o_next = tmp; }
  }
  mean += __shfl_xor(mean, 1); var += __shfl_xor(var, 1);
  mean += __shfl_xor(mean, 2); var += __shfl_xor(var, 2);
  if (j != 0 || !live) return;
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  mean -= old * var;
  var = 1.0 / (lambda + alpha * var);
  mean = -var * (alpha * mean - mu * lambda);
  const double nv = mean;
  P[(size_t)feat * kp + f] = nv;
  vstep[fi] = make_double2(old, old - nv);
}
}  // namespace fmx


// ---- K1, one wave per FW features, E = FW / 16 pairs per lane per tile, the sums kept in LDS accumulators OWNED BY THE WAVE and fed by ds_add_f64 (no other
// wave ever touches them, a wave's LDS operations execute in order, and lanes of ONE instruction that hit the same accumulator are serialised by the LDS in a
// fixed order: reproducible, checked by running twice).  No list walk, no search, no barrier: per tile the run bounds, E pair loads, E feature-index loads.
namespace fmx {
template <bool UNIT, int NA, int FW, int MODE = 3>   // MODE bit 0: the LDS atomics; bit 1: the feature-index loads
__global__ __launch_bounds__(256) void k1_atom_k(const uint32_t* __restrict__ toff, size_t nf1, uint32_t lvl0, uint32_t cnt, const uint16_t* __restrict__ fidx,
                                                 const double2* __restrict__ src, int tshift, int n_tiles, const uint32_t* __restrict__ feats, double* __restrict__ P,
                                                 int kp, const SweepDyn* __restrict__ dyn, double2* __restrict__ vstep) {
  constexpr int E = FW / 16;
  __shared__ double acc[4][FW][2];
  __shared__ double oldv[4][FW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t f0 = (blockIdx.x * 4 + wv) * FW;
  if (f0 >= cnt) return;
  const uint32_t fend = min(f0 + (uint32_t)FW, cnt);
  const int f = dyn->f;
  for (int i = lane; i < FW; i += 64) {
    const uint32_t fi = min(f0 + (uint32_t)i, cnt - 1);
    oldv[wv][i] = P[(size_t)feats[fi] * kp + f];
    acc[wv][i][0] = 0.0; acc[wv][i][1] = 0.0;
  }
  const uint32_t which = lane == 0 ? f0 : (lane == 1 ? fend : 0u);   // lanes 0, 1, 2: the run's first offset, its end, the level block's first offset
  struct Offs { uint32_t o[NA]; };
  struct Pairs { double2 pv[NA][E]; uint32_t fx[NA][E], rl[NA], r0[NA]; };
  auto load_offs = [&](int t0, Offs& o) {
#pragma unroll
    for (int u = 0; u < NA; ++u) { const int t = min(t0 + u, n_tiles - 1); o.o[u] = stream_load<true>(toff + (size_t)t * nf1 + lvl0 + which); }
  };
  auto load_pairs = [&](int t0, const Offs& o, Pairs& p) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int t = min(t0 + u, n_tiles - 1);
      const uint32_t a = __builtin_amdgcn_readlane(o.o[u], 0), b = __builtin_amdgcn_readlane(o.o[u], 1), base = __builtin_amdgcn_readlane(o.o[u], 2);
      p.rl[u] = t0 + u < n_tiles ? b - a : 0u;
      p.r0[u] = a - base;
#pragma unroll
      for (int q = 0; q < E; ++q) {
        const size_t at = ((size_t)t << tshift) + p.r0[u] + min((uint32_t)(lane + 64 * q), p.rl[u] > 0 ? p.rl[u] - 1 : 0u);
        p.pv[u][q] = stream_load<true>(src + at);
        p.fx[u][q] = (MODE & 2) ? (uint32_t)stream_load<true>(fidx + at) : f0 + (uint32_t)(lane & 15);
      }
    }
  };
  double dm = 0.0, dv = 0.0;
  auto add = [&](double2 c, uint32_t fx) {
    const int g = (int)(fx - f0);
    if (MODE & 1) {
      const double h = c.x - oldv[wv][g];
      unsafeAtomicAdd(&acc[wv][g][0], h * c.y);
      unsafeAtomicAdd(&acc[wv][g][1], h * h);
    } else { const double h = c.x - (double)g; dm += h * c.y; dv += h * h; }
  };
  Offs o_next, o_after; Pairs cur, nxt;
  load_offs(0, o_next);
  load_offs(NA, o_after);
  load_pairs(0, o_next, cur);
  for (int t0 = 0; t0 < n_tiles; t0 += NA) {
    load_pairs(t0 + NA, o_after, nxt);
    load_offs(t0 + 2 * NA, o_next);
#pragma unroll
    for (int u = 0; u < NA; ++u) {
#pragma unroll
      for (int q = 0; q < E; ++q)
        if ((uint32_t)(lane + 64 * q) < cur.rl[u]) add(cur.pv[u][q], cur.fx[u][q]);
      if (cur.rl[u] > 64u * E) {   // a long run: the rest in pieces of 64, straight from memory
        const int t = min(t0 + u, n_tiles - 1);
        for (uint32_t i = 64 * E + lane; __any(i < cur.rl[u]); i += 64)
          if (i < cur.rl[u]) { const size_t at = ((size_t)t << tshift) + cur.r0[u] + i; add(src[at], fidx[at]); }
      }
    }
    cur = nxt; { Offs tmp = o_after; o_after = o_next; o_next = tmp; }
  }
  if (!(MODE & 1)) { acc[wv][lane & (FW - 1)][0] = dm; acc[wv][lane & (FW - 1)][1] = dv; }
  for (int i = lane; i < FW && f0 + i < cnt; i += 64) {
    const uint32_t fi = f0 + i;
    const uint32_t feat = feats[fi];
    double mean = acc[wv][i][0], var = acc[wv][i][1];
    const double old = oldv[wv][i];
    const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
    mean -= old * var;
    var = 1.0 / (lambda + alpha * var);
    mean = -var * (alpha * mean - mu * lambda);
    P[(size_t)feat * kp + f] = mean;
    vstep[fi] = make_double2(old, old - mean);
  }
}
}  // namespace fmx


// K2 with its source read in short pieces (what a feature-block-major array would give the apply kernel: a (tile, 16-feature block) piece is ~32-64 pairs):
// thread i of a workgroup reads pair (piece(i / PL), i % PL) where the workgroup's 256 / PL pieces lie PSTRIDE pairs apart; the writes are K2's own scatter.
template <int PL>
__global__ __launch_bounds__(256) void apply_pieces_k(const double2* __restrict__ src, double2* __restrict__ dst, const uint16_t* __restrict__ fidx, const uint32_t* __restrict__ perm,
                                                      const double2* __restrict__ vstep, int ts, int n_tiles, int B, int64_t n, int64_t pstride) {
  const int b = blockIdx.x;
  const int x = b & 7, q = b >> 3;
  const int tile = (q / B) * 8 + x, chunk = q % B;
  if (tile >= n_tiles) return;
  const int64_t base = (int64_t)tile << ts;
  const int64_t i = base + (int64_t)chunk * 256 + threadIdx.x;          // the entry whose perm / destination this thread handles (K2's own mapping)
  // where its PAIR and feature index are read from: piece p of the workgroup, far from the others
  const int p_ = threadIdx.x / PL, e_ = threadIdx.x % PL;
  const int64_t at = (((int64_t)b * (256 / PL) + p_) * pstride + e_) % n;
  const double2 c = ntload(src + at);
  const uint32_t f = ntl(fidx + at);
  const uint32_t pm = ntl(perm + (i < n ? i : n - 1));
  const double2 s = vstep[f];
  if (i >= n) return;
  const double h = c.x - s.x;
  dst[base + pm] = make_double2(c.x - s.y, c.y - h * s.y);
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
#define PIECES(Pv, ROTv, CONTIGv) timeit("pieces of " #Pv " pairs/tile, rot=" #ROTv " contig=" #CONTIGv, [&] { hipLaunchKernelGGL((pieces_k<Pv, ROTv, CONTIGv>), dim3((unsigned)(T / Pv)), dim3(256), 0, 0, src, ts, n_tiles, vf); }, 16.0 * n);
  PIECES(128, false, false) PIECES(128, true, false) PIECES(128, false, true) PIECES(256, false, false) PIECES(256, true, false) PIECES(256, false, true) PIECES(64, false, false) PIECES(64, true, false)
  uint32_t* d_feats; int64_t* d_tb; fmx::SweepDyn* d_dyn; double* d_P;
  { std::vector<uint32_t> hf(cnt); std::iota(hf.begin(), hf.end(), 0u); CK(hipMalloc(&d_feats, cnt * 4)); CK(hipMemcpy(d_feats, hf.data(), cnt * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_tb, (n_tiles + 1) * 8)); CK(hipMemset(d_tb, 0, (n_tiles + 1) * 8)); fmx::SweepDyn hd{0, 0, 1.0, 1.0, 0.0, nullptr}; CK(hipMalloc(&d_dyn, sizeof(hd)));
    CK(hipMemcpy(d_dyn, &hd, sizeof(hd), hipMemcpyHostToDevice)); CK(hipMalloc(&d_P, cnt * 8)); CK(hipMemset(d_P, 0, cnt * 8)); }
  auto k1_timed = [&](const char* name, auto kern, int fbv) {
    unsigned long long z8[8] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(fmx::fmx_k1_ticks), z8, sizeof(z8)));
    const unsigned grid = (cnt + fbv - 1) / fbv;
    timeit(name, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d_toff, nf1, 0u, cnt, fbv, d_tb, (const float*)nullptr, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vstep); }, 16.0 * n);
    unsigned long long t8[8]; CK(hipMemcpyFromSymbol(t8, HIP_SYMBOL(fmx::fmx_k1_ticks), sizeof(t8)));
    const double per = 1.0 / (23.0 * grid);   // 3 warm-up + 20 timed launches; ticks of the 100 MHz wall clock -> us per workgroup: x 0.01
    printf("    per workgroup (us): offsets+prefix %.2f | first fetch %.2f | wait+store %.2f | barrier %.2f | refill %.2f | walk %.2f | barrier %.2f\n",
           t8[0] * per * 0.01, t8[1] * per * 0.01, t8[2] * per * 0.01, t8[3] * per * 0.01, t8[4] * per * 0.01, t8[5] * per * 0.01, t8[6] * per * 0.01);
  };
#define K1KO(MODEv) timeit("product K1 (fb 44, depth 1) knock-out mode " #MODEv, [&] { hipLaunchKernelGGL((fmx::k1_ko_k<true, 64, 128, 1024, 1>), dim3((cnt + 43) / 44), dim3(256), 0, 0, \
    d_toff, nf1, 0u, cnt, 44, d_tb, (const float*)nullptr, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vstep, MODEv); }, 16.0 * n);
  K1KO(15) K1KO(14) K1KO(13) K1KO(12) K1KO(10) K1KO(8) K1KO(0)
  k1_timed("product K1 fb=44 depth 1", fmx::als_order_sums_k<true, 64, 128, 1024, 1>, 44);
  k1_timed("product K1 fb=44 depth 3", fmx::als_order_sums_k<true, 64, 128, 1024, 3>, 44);
  k1_timed("product K1 fb=33 depth 1", fmx::als_order_sums_k<true, 64, 128, 1024, 1>, 33);
  {  // the lean K1 against the generic one: same vstep?
    double2* vs2; CK(hipMalloc(&vs2, cnt * 16)); CK(hipMemset(vs2, 0, cnt * 16)); CK(hipMemset(vstep, 0, cnt * 16));
    uint32_t maxrun = 0; for (int t = 0; t < n_tiles; ++t) for (uint32_t b0 = 0; b0 < cnt; b0 += 32) { const uint32_t* off = toff.data() + (size_t)t * (cnt + 1); maxrun = std::max(maxrun, off[std::min(b0 + 32, cnt)] - off[b0]); }
    printf("longest run of 32 features in a tile: %u pairs\n", maxrun);
    CK(hipMemset(d_P, 0, cnt * 8));
    hipLaunchKernelGGL((fmx::als_order_sums_k<true, 64, 128, 1024, 1>), dim3((cnt + 31) / 32), dim3(256), 0, 0, d_toff, nf1, 0u, cnt, 32, d_tb, (const float*)nullptr, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vstep);
    CK(hipMemset(d_P, 0, cnt * 8));
    hipLaunchKernelGGL((fmx::k1_lean_k<true, 32, 8, 256>), dim3((cnt + 31) / 32), dim3(256), 0, 0, d_toff, nf1, 0u, cnt, d_tb, (const float*)nullptr, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vs2);
    CK(hipDeviceSynchronize());
    std::vector<double2> a(cnt), b(cnt); CK(hipMemcpy(a.data(), vstep, cnt * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), vs2, cnt * 16, hipMemcpyDeviceToHost));
    double worst = 0.0; for (uint32_t i = 0; i < cnt; ++i) worst = std::max(worst, std::abs(a[i].y - b[i].y) / std::max(1e-300, std::abs(a[i].y)));
    printf("lean vs generic K1: largest relative difference of a step %.3e\n", worst);
    CK(hipMemset(d_P, 0, cnt * 8));
  }
  {  // the wave kernel against the product's: same steps?
    double2* vs2; CK(hipMalloc(&vs2, cnt * 16)); CK(hipMemset(vs2, 0, cnt * 16)); CK(hipMemset(vstep, 0, cnt * 16)); CK(hipMemset(d_P, 0, cnt * 8));
    hipLaunchKernelGGL((fmx::als_order_sums_k<true, 64, 128, 1024, 1>), dim3((cnt + 31) / 32), dim3(256), 0, 0, d_toff, nf1, 0u, cnt, 32, d_tb, (const float*)nullptr, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vstep);
    CK(hipMemset(d_P, 0, cnt * 8));
    hipLaunchKernelGGL((fmx::k1_wave_k<true, 8, false>), dim3((cnt + 63) / 64), dim3(256), 0, 0, d_toff, nf1, 0u, cnt, d_tb, (const float*)nullptr, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vs2);
    CK(hipDeviceSynchronize());
    std::vector<double2> a(cnt), b(cnt); CK(hipMemcpy(a.data(), vstep, cnt * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), vs2, cnt * 16, hipMemcpyDeviceToHost));
    double worst = 0.0; for (uint32_t i = 0; i < cnt; ++i) worst = std::max(worst, std::abs(a[i].y - b[i].y) / std::max(1e-300, std::abs(a[i].y)));
    printf("wave K1 vs product K1: largest relative difference of a step %.3e\n", worst);
    CK(hipMemset(d_P, 0, cnt * 8));
  }
  {  // the LDS-atomic kernel against the product's: same steps?  twice: the same bits?
    double2 *vs2, *vs3; CK(hipMalloc(&vs2, cnt * 16)); CK(hipMalloc(&vs3, cnt * 16)); CK(hipMemset(vs2, 0, cnt * 16)); CK(hipMemset(vs3, 0, cnt * 16)); CK(hipMemset(vstep, 0, cnt * 16)); CK(hipMemset(d_P, 0, cnt * 8));
    hipLaunchKernelGGL((fmx::als_order_sums_k<true, 64, 128, 1024, 1>), dim3((cnt + 31) / 32), dim3(256), 0, 0, d_toff, nf1, 0u, cnt, 32, d_tb, (const float*)nullptr, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vstep);
    CK(hipMemset(d_P, 0, cnt * 8));
    hipLaunchKernelGGL((fmx::k1_atom_k<true, 8, 32>), dim3((cnt + 127) / 128), dim3(256), 0, 0, d_toff, nf1, 0u, cnt, d_fidx, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vs2);
    CK(hipMemset(d_P, 0, cnt * 8));
    hipLaunchKernelGGL((fmx::k1_atom_k<true, 8, 32>), dim3((cnt + 127) / 128), dim3(256), 0, 0, d_toff, nf1, 0u, cnt, d_fidx, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vs3);
    CK(hipDeviceSynchronize());
    std::vector<double2> a(cnt), b(cnt), c(cnt); CK(hipMemcpy(a.data(), vstep, cnt * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), vs2, cnt * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), vs3, cnt * 16, hipMemcpyDeviceToHost));
    double worst = 0.0; size_t diff = 0; for (uint32_t i = 0; i < cnt; ++i) { worst = std::max(worst, std::abs(a[i].y - b[i].y) / std::max(1e-300, std::abs(a[i].y))); diff += (b[i].y != c[i].y) || (b[i].x != c[i].x); }
    printf("LDS-atomic K1 vs product K1: largest relative difference of a step %.3e; two runs differ in %zu of %u steps\n", worst, diff, cnt);
    CK(hipMemset(d_P, 0, cnt * 8));
  }
#define K1A(NAv, FWv) timeit("LDS-atomic K1, features per wave " #FWv ", tiles in flight " #NAv, [&] { hipLaunchKernelGGL((fmx::k1_atom_k<true, NAv, FWv>), dim3((cnt + 4 * FWv - 1) / (4 * FWv)), dim3(256), 0, 0, \
    d_toff, nf1, 0u, cnt, d_fidx, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vstep); }, 18.0 * n);
  K1A(4, 16) K1A(8, 16) K1A(4, 32) K1A(8, 32)
#define K1AM(NAv, FWv, MODEv) timeit("LDS-atomic K1, features per wave " #FWv ", tiles in flight " #NAv ", knock-out mode " #MODEv, [&] { hipLaunchKernelGGL((fmx::k1_atom_k<true, NAv, FWv, MODEv>), dim3((cnt + 4 * FWv - 1) / (4 * FWv)), dim3(256), 0, 0, \
    d_toff, nf1, 0u, cnt, d_fidx, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vstep); }, 18.0 * n);
  K1AM(4, 16, 2) K1AM(4, 16, 0) K1AM(8, 16, 0) K1AM(4, 32, 0) K1AM(8, 32, 0)
#define K1W(NAv) timeit("wave K1 (16 features per wave, no LDS) tiles in flight " #NAv, [&] { hipLaunchKernelGGL((fmx::k1_wave_k<true, NAv, false>), dim3((cnt + 63) / 64), dim3(256), 0, 0, \
    d_toff, nf1, 0u, cnt, d_tb, (const float*)nullptr, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vstep); }, 16.0 * n);
  K1W(4) K1W(8) K1W(16)
  uint32_t* d_toffT;
  { std::vector<uint32_t> tt((size_t)(cnt + 1) * n_tiles); for (int t = 0; t < n_tiles; ++t) for (uint32_t f = 0; f <= cnt; ++f) tt[(size_t)f * n_tiles + t] = toff[(size_t)t * (cnt + 1) + f];
    CK(hipMalloc(&d_toffT, tt.size() * 4)); CK(hipMemcpy(d_toffT, tt.data(), tt.size() * 4, hipMemcpyHostToDevice)); }
#define K1WT(NAv) timeit("wave K1, offsets [feature][tile], tiles in flight " #NAv, [&] { hipLaunchKernelGGL((fmx::k1_wave_k<true, NAv, true>), dim3((cnt + 63) / 64), dim3(256), 0, 0, \
    d_toffT, nf1, 0u, cnt, d_tb, (const float*)nullptr, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vstep); }, 16.0 * n);

#define K1L(FBv, NITv, Sv) timeit("lean K1 FB=" #FBv " NIT=" #NITv " S=" #Sv, [&] { hipLaunchKernelGGL((fmx::k1_lean_k<true, FBv, NITv, Sv>), dim3((cnt + FBv - 1) / FBv), dim3(256), 0, 0, \
    d_toff, nf1, 0u, cnt, d_tb, (const float*)nullptr, src, ts, n_tiles, d_feats, d_P, 1, d_dyn, vstep); }, 16.0 * n);
  K1L(32, 8, 256) K1L(32, 8, 192) K1L(32, 12, 192) K1L(32, 16, 192) K1L(16, 16, 128) K1L(16, 8, 128) K1L(64, 8, 256) K1L(64, 4, 256)
  timeit("K1 sums+step LG=1 (256 features / WG)", [&] { hipLaunchKernelGGL(sums_step_k<1>, dim3((cnt + 255) / 256), dim3(256), 0, 0, d_toff, nf1, cnt, src, ts, n_tiles, vf, vstep); }, 16.0 * n);
  timeit("K1 sums+step LG=4 (64 features / WG)", [&] { hipLaunchKernelGGL(sums_step_k<4>, dim3((cnt + 63) / 64), dim3(256), 0, 0, d_toff, nf1, cnt, src, ts, n_tiles, vf, vstep); }, 16.0 * n);
  timeit("K1 sums+step LG=8 (32 features / WG)", [&] { hipLaunchKernelGGL(sums_step_k<8>, dim3((cnt + 31) / 32), dim3(256), 0, 0, d_toff, nf1, cnt, src, ts, n_tiles, vf, vstep); }, 16.0 * n);
  timeit("K1b flat sums FB=64", [&] { hipLaunchKernelGGL(sums_flat_k<64>, dim3((cnt + 63) / 64), dim3(256), 0, 0, d_toff, nf1, cnt, src, d_fidx, ts, n_tiles, vf, vstep); }, 18.0 * n);
  timeit("K1b flat sums FB=128", [&] { hipLaunchKernelGGL(sums_flat_k<128>, dim3((cnt + 127) / 128), dim3(256), 0, 0, d_toff, nf1, cnt, src, d_fidx, ts, n_tiles, vf, vstep); }, 18.0 * n);
#define K2(Rv, NTv) { const int B = (int)((T + 256 * Rv - 1) / (256 * Rv)); const unsigned g = (unsigned)(((n_tiles + 7) / 8) * 8 * B); \
    timeit("K2 apply+permute R=" #Rv " NT=" #NTv, [&] { hipLaunchKernelGGL((apply_perm_k<Rv, NTv>), dim3(g), dim3(256), 0, 0, src, dst, d_fidx, d_perm, vstep, ts, n_tiles, B, n); }, 38.0 * n); }
  K2(4, true) K2(8, true) K2(8, false) K2(2, true) K2(1, true)

  // ---- can the sums kernel of level l+1 hide under the apply kernel of level l?  K1 on a tile range needs only K2 of THOSE tiles done.  G groups of tiles:
  // serial = K2(all) ; K1(all).  pipelined = K2(g0) ; { K2(g1) || K1(g0) } ; ... ; K1(g_last), on two streams with events.  (K1 then leaves per-group partial
  // sums; the probe does not combine them: timing only.)
  {
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t ev[64]; for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    auto k2 = [&](int ta, int tb_, hipStream_t st) { const int nt = tb_ - ta; const int B = (int)(T / 256); const unsigned g = (unsigned)(((nt + 7) / 8) * 8 * B);
      hipLaunchKernelGGL((apply_perm_k<1, true>), dim3(g), dim3(256), 0, st, src + ((size_t)ta << ts), dst + ((size_t)ta << ts), d_fidx + ((size_t)ta << ts), d_perm + ((size_t)ta << ts), vstep, ts, nt, B, (int64_t)nt << ts); };
    auto k1 = [&](int ta, int tb_, hipStream_t st) { const int nt = tb_ - ta;
      hipLaunchKernelGGL((fmx::k1_ko_k<true, 64, 128, 1024, 1>), dim3((cnt + 43) / 44), dim3(256), 0, st, d_toff + (size_t)ta * nf1, nf1, 0u, cnt, 44, d_tb, (const float*)nullptr, dst + ((size_t)ta << ts), ts, nt, d_feats, d_P, 1, d_dyn, vstep, 15); };
    for (int G : {1, 2, 4, 8}) {
      const int reps = 10;
      float ms = 0;
      for (int rep = -2; rep < reps; ++rep) {
        if (rep == 0) { CK(hipDeviceSynchronize()); CK(hipEventRecord(t0, sa)); }
        int e = 0;
        for (int lvl = 0; lvl < 4; ++lvl) {   // four "levels" back to back: K2(l) on sa, K1(l+1) on sb, next level's K2 waits for the last K1
          for (int g = 0; g < G; ++g) {
            const int ta = n_tiles * g / G, tb_ = n_tiles * (g + 1) / G;
            k2(ta, tb_, sa); CK(hipEventRecord(ev[e], sa)); CK(hipStreamWaitEvent(sb, ev[e], 0)); e = (e + 1) % 64;
            k1(ta, tb_, sb);
          }
          CK(hipEventRecord(ev[e], sb)); CK(hipStreamWaitEvent(sa, ev[e], 0)); e = (e + 1) % 64;
        }
      }
      CK(hipEventRecord(t1, sa)); CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
      printf("K2(l) + K1(l+1) in %d tile groups on two streams: %.1f us per level\n", G, ms / (reps * 4) * 1e3);
    }
  }
  { const int B = (int)(T / 256); const unsigned g = (unsigned)(((n_tiles + 7) / 8) * 8 * B);
#define K2P(PLv) timeit("K2 apply+permute, source read in pieces of " #PLv " pairs", [&] { hipLaunchKernelGGL((apply_pieces_k<PLv>), dim3(g), dim3(256), 0, 0, src, dst, d_fidx, d_perm, vstep, ts, n_tiles, B, n, (int64_t)4999); }, 38.0 * n);
    K2P(256) K2P(64) K2P(32) K2P(16) }
  // check: dst holds a permutation of the corrected pairs (sum of e preserved up to the corrections; here just a checksum of positions written)
  { std::vector<double2> h(n); CK(hipMemcpy(h.data(), dst, n * 16, hipMemcpyDeviceToHost)); int64_t zeros = 0; for (int64_t i = 0; i < n; ++i) zeros += (h[i].x == 0.0 && h[i].y == 0.0); printf("unwritten slots: %lld\n", (long long)zeros); }
  return 0;
}
