// Row-tiled form of one level of the ALS / Gibbs sweeps (MCMC_ALS_Learner::update_v, solver/MCMC_ALS_Learner.h:283-351; update_w, :208-256).
//
// als_level_k (fm_als_kernels.hip) walks a feature's CSC column: per stored nonzero one random 16-byte gather AND one 16-byte scatter of the
// row's (q, e) pair, from a table of n rows -- 160 MB at configs[4], i.e. every access a miss of the XCD's 4 MB L2, a whole line moved for
// 16 bytes used (profiles/r04_pmc_summary_mcmc_untiled.json).  On matrices whose levels are few and wide (one column per field: the levels
// ARE the fields) the same level is done here as three passes, none of which touches the big table at random:
//
//   sums   als_tile_sums_k   the rows are cut into tiles whose (q, e) slice fits an XCD's L2 (131 072 rows = 2 MB); a tile's workgroups share
//                            one XCD (equal blockIdx % 8: placement for speed only, the result does not depend on it).  Per (tile, feature of
//                            the level) the tile's entries of that feature -- kept sorted by feature, rows ascending: `trow`, `tval`, `toff` --
//                            are walked by a lane group: sum h e, sum h^2 against the L2-resident slice.  One (mean, var) pair per (tile, feature).
//   step   als_tile_step_k   per feature: the tiles' pairs added in tile order (fixed: reproducible run to run), the coordinate step of
//                            :318-336 exactly as als_level_k takes it, V written, (v_old, v_old - v_new) left in a level-sized table.
//   apply  als_rows_apply_k  ROW-major: row r reads the level's entry it holds (`lfi`, `lval`: level-major copies of the CSR), the 16-byte
//                            (v_old, diff) pair of its feature (a table of one level's features: L2-resident) and streams (q, e)[r] through:
//                            q -= x diff, e -= h diff (:341-350).  Coalesced in and out.
//
// Same arithmetic per entry as als_level_k; only the association of the two sums differs (lane-strided butterfly there, list order inside a
// tile then tile order here): oracle parity 1e-10 (tests/test_gpu_configs4.py), bitwise run to run.
#include <cstring>  // rocprim's texture_cache_iterator.hpp uses memset without including it
#include <memory>

#include <rocprim/rocprim.hpp>

#include "fmx_internal.h"

namespace fmx {

constexpr uint32_t TILED_NONE = 0xFFFFFFFFu;

struct AlsTiled {
  int64_t n = 0;
  int tshift = 17;               // tile_rows = 1 << tshift
  int n_tiles = 0;
  int lg = 1;                    // lanes per (tile, feature) list in als_tile_sums_k
  uint32_t n_feats = 0;          // features of the tiled levels, ordered by (level, index)
  uint32_t max_cnt = 0;          // most features in one tiled level
  int n_slots = 0;               // tiled levels
  int unit = 0;
  std::vector<int> slot_of_level;          // [L] index of the level among the tiled ones, -1: the level keeps the column-walking kernels
  std::vector<uint32_t> lvl0, cnt;         // per slot: first feature (position in `feats`) and number of features
  uint32_t* feats = nullptr;     // [n_feats] feature ids
  void* lfi = nullptr;           // [n_slots][n] index (inside its level) of the feature row r holds at that level (u16 when every level has fewer than 65 535
                                 // features -- lfi16 -- else u32), all ones: none
  int lfi16 = 0;
  float* lval = nullptr;         // [n_slots][n] its value (null: every value is 1.0f)
  uint32_t* toff = nullptr;      // [n_tiles][n_feats + 1] entry offsets of the (tile, feature) lists, relative to the tile's first entry
  int64_t* tile_base = nullptr;  // [n_tiles + 1] (device) first entry of each tile in trow / tval
  uint32_t* trow = nullptr;      // [entries + 1] row inside the tile
  float* tval = nullptr;         // [entries + 1] (null: unit values)
  ~AlsTiled() {
    (void)hipFree(feats); (void)hipFree(lfi); (void)hipFree(lval); (void)hipFree(toff); (void)hipFree(tile_base); (void)hipFree(trow); (void)hipFree(tval);
  }
};

void als_tiled_free(fmx_matrix* m) {
  delete reinterpret_cast<AlsTiled*>(m->als_tiled);
  m->als_tiled = nullptr;
}

// ---- plan ------------------------------------------------------------------------------------------------------------------------------------
// level-major copies of the CSR: for every tiled level the feature (as its index inside the level) and value each row holds there
template <typename IT>
__global__ void tiled_rows_k(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const float* __restrict__ val, int64_t n,
                             const int* __restrict__ slot_of_feat, const uint32_t* __restrict__ idx_in_level, IT* __restrict__ lfi, float* __restrict__ lval) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  for (int64_t t = row_ptr[r]; t < row_ptr[r + 1]; ++t) {
    const uint32_t j = col[t];
    const int s = slot_of_feat[j];
    if (s < 0) continue;
    lfi[(size_t)s * n + r] = (IT)idx_in_level[j];
    if (lval) lval[(size_t)s * n + r] = val[t];
  }
}

// entries of every (tile, feature) list: one wave per CSC column (rows ascending inside a column, so a tile's entries of it are one run)
__global__ __launch_bounds__(WG_THREADS) void tiled_count_k(const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow, const uint32_t* __restrict__ rank_of,
                                                            uint32_t p, int tshift, size_t nf1, uint32_t* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int64_t j = ((int64_t)blockIdx.x * WG_THREADS + threadIdx.x) >> 6;
  if (j >= (int64_t)p) return;
  const uint32_t k = rank_of[j];
  if (k == TILED_NONE) return;
  const int64_t b = col_ptr[j], e = col_ptr[j + 1];
  for (int64_t t = b + lane; t < e; t += 64) atomicAdd(&counts[(size_t)(crow[t] >> tshift) * nf1 + k], 1u);  // (integer counts: the order of the adds does not matter)
}

__global__ __launch_bounds__(WG_THREADS) void tiled_scatter_k(const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow, const float* __restrict__ cval,
                                                              const uint32_t* __restrict__ rank_of, uint32_t p, int tshift, size_t nf1,
                                                              const uint32_t* __restrict__ toff, const int64_t* __restrict__ tile_base,
                                                              uint32_t* __restrict__ trow, float* __restrict__ tval) {
  const int lane = threadIdx.x & 63;
  const int64_t j = ((int64_t)blockIdx.x * WG_THREADS + threadIdx.x) >> 6;
  if (j >= (int64_t)p) return;
  const uint32_t k = rank_of[j];
  if (k == TILED_NONE) return;
  const int64_t b = col_ptr[j], e = col_ptr[j + 1];
  for (int64_t t = b + lane; t < e; t += 64) {
    const uint32_t r = crow[t];
    const uint32_t tile = r >> tshift, lo = tile << tshift;
    int64_t a = b, z = t;  // first entry of the column whose row lies in this tile: in [b, t]
    while (a < z) {
      const int64_t mid = (a + z) >> 1;
      if (crow[mid] < lo) a = mid + 1; else z = mid;
    }
    const int64_t dst = tile_base[tile] + (int64_t)toff[(size_t)tile * nf1 + k] + (t - a);
    trow[dst] = r - lo;
    if (tval) tval[dst] = cval[t];
  }
}

__global__ void tiled_tile_base_k(const int64_t* __restrict__ row_ptr, int64_t n, int tshift, int n_tiles, int64_t* __restrict__ tile_base) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > n_tiles) return;
  const int64_t r = (int64_t)t << tshift;
  tile_base[t] = row_ptr[r < n ? r : n];
}

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s && *s ? atoi(s) : dflt;
}

// Which levels of the exact plan go through the tiled form, and their plan.  Called at the end of build_plan (the CSC and the levels exist).
// A failure to allocate leaves the matrix without a tiled plan (the column-walking kernels do every level then): never an error.
int als_tiled_build(fmx_matrix* m, hipStream_t stream) {
  als_tiled_free(m);
  if (m->als_approx || m->n == 0 || m->nnz == 0) return FMX_OK;
  // FMX_ALS_TILED: 0 never, 1 wherever a level qualifies (tests: small matrices), unset: where the (q, e) table no longer fits the L2s together
  const int mode = env_int("FMX_ALS_TILED", -1);
  if (mode == 0) return FMX_OK;
  if (mode < 0 && m->n < (1 << 21)) return FMX_OK;
  const int L = (int)m->als_level_ptr.size() - 1;
  const uint32_t p = m->p;
  if (L <= 0 || m->n >= (1LL << 32)) return FMX_OK;
  // a level qualifies when every feature of it is a "light" one (at most ALS_HEAVY entries: als_heavy / als_vh hold none of the level) and it is wide
  // enough to fill the chip; the level-major copies cost 4 (8) bytes per row and tiled level, so deep plans (thousands of narrow levels) never qualify
  const int64_t min_feats = mode > 0 ? 1 : 2048;
  std::vector<int> slot((size_t)L, -1);
  int n_slots = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t c = m->als_level_ptr[(size_t)l + 1] - m->als_level_ptr[(size_t)l];
    const int64_t h = m->als_heavy_ptr[(size_t)l + 1] - m->als_heavy_ptr[(size_t)l];
    const int64_t v = m->als_vh_ptr.empty() ? 0 : m->als_vh_ptr[(size_t)l + 1] - m->als_vh_ptr[(size_t)l];
    if (h == 0 && v == 0 && c >= min_feats) slot[(size_t)l] = n_slots++;
  }
  if (n_slots == 0 || n_slots > (mode > 0 ? 4096 : 256)) return FMX_OK;
  if (mode < 0 && (double)n_slots * (double)m->n > 8.0 * (double)m->nnz + 1e6) return FMX_OK;   // the level-major copies would dwarf the matrix
  std::unique_ptr<AlsTiled> T(new AlsTiled());
  T->n = m->n;
  int rows_want = env_int("FMX_ALS_TILE_ROWS", 131072);
  int ts = 4;
  while ((1 << (ts + 1)) <= rows_want && ts < 24) ++ts;
  T->tshift = ts;
  T->n_tiles = (int)((m->n + (1LL << ts) - 1) >> ts);
  T->lg = env_int("FMX_ALS_TILE_LG", 1);
  if (T->lg != 1 && T->lg != 2 && T->lg != 4 && T->lg != 8) T->lg = 1;
  T->unit = m->unit_values;
  T->n_slots = n_slots;
  T->slot_of_level = slot;
  // features of the tiled levels by (level, index): als_feats already holds the light features in that order
  std::vector<uint32_t> all_light((size_t)m->als_level_ptr[(size_t)L]);
  if (!all_light.empty()) FMX_HIP(hipMemcpy(all_light.data(), m->als_feats, all_light.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  std::vector<uint32_t> feats, rank_of((size_t)p, TILED_NONE), idx_in((size_t)p, 0u);
  std::vector<int> slot_of_feat((size_t)p, -1);
  T->lvl0.assign((size_t)n_slots, 0u); T->cnt.assign((size_t)n_slots, 0u);
  for (int l = 0; l < L; ++l) {
    const int s = slot[(size_t)l];
    if (s < 0) continue;
    T->lvl0[(size_t)s] = (uint32_t)feats.size();
    for (int64_t q = m->als_level_ptr[(size_t)l]; q < m->als_level_ptr[(size_t)l + 1]; ++q) {
      const uint32_t j = all_light[(size_t)q];
      idx_in[j] = (uint32_t)(feats.size() - T->lvl0[(size_t)s]);
      rank_of[j] = (uint32_t)feats.size();
      slot_of_feat[j] = s;
      feats.push_back(j);
    }
    T->cnt[(size_t)s] = (uint32_t)(feats.size() - T->lvl0[(size_t)s]);
    if (T->cnt[(size_t)s] > T->max_cnt) T->max_cnt = T->cnt[(size_t)s];
  }
  T->n_feats = (uint32_t)feats.size();
  const size_t nf1 = (size_t)T->n_feats + 1;
  if ((double)nf1 * T->n_tiles * 4.0 > 4e9) return FMX_OK;   // directories of more than 4 GB: smaller tiles than this matrix wants
  struct Tmp {
    uint32_t *rank_of = nullptr, *idx_in = nullptr, *counts = nullptr;
    int* slot_of_feat = nullptr;
    void* scan = nullptr;
    ~Tmp() { (void)hipFree(rank_of); (void)hipFree(idx_in); (void)hipFree(counts); (void)hipFree(slot_of_feat); (void)hipFree(scan); }
  } w;
  auto ok = [](hipError_t e) { if (e != hipSuccess) (void)hipGetLastError(); return e == hipSuccess; };
  const size_t sn = (size_t)n_slots * (size_t)m->n;
  T->lfi16 = T->max_cnt < 0xFFFFu ? 1 : 0;
  const size_t isz = T->lfi16 ? 2 : 4;
  if (!ok(hipMalloc(&w.rank_of, (size_t)p * 4)) || !ok(hipMalloc(&w.idx_in, (size_t)p * 4)) || !ok(hipMalloc(&w.slot_of_feat, (size_t)p * 4)) ||
      !ok(hipMalloc(&w.counts, nf1 * T->n_tiles * 4)) || !ok(hipMalloc(&T->feats, (size_t)(T->n_feats ? T->n_feats : 1) * 4)) ||
      !ok(hipMalloc(&T->lfi, sn * isz)) || (!T->unit && !ok(hipMalloc(&T->lval, sn * 4))) || !ok(hipMalloc(&T->toff, nf1 * T->n_tiles * 4)) ||
      !ok(hipMalloc(&T->tile_base, ((size_t)T->n_tiles + 1) * 8)) || !ok(hipMalloc(&T->trow, ((size_t)m->nnz + 1) * 4)) ||
      (!T->unit && !ok(hipMalloc(&T->tval, ((size_t)m->nnz + 1) * 4))))
    return FMX_OK;
  FMX_HIP(hipMemcpyAsync(w.rank_of, rank_of.data(), (size_t)p * 4, hipMemcpyHostToDevice, stream));
  FMX_HIP(hipMemcpyAsync(w.idx_in, idx_in.data(), (size_t)p * 4, hipMemcpyHostToDevice, stream));
  FMX_HIP(hipMemcpyAsync(w.slot_of_feat, slot_of_feat.data(), (size_t)p * 4, hipMemcpyHostToDevice, stream));
  FMX_HIP(hipMemcpyAsync(T->feats, feats.data(), (size_t)T->n_feats * 4, hipMemcpyHostToDevice, stream));
  FMX_HIP(hipMemsetAsync(T->lfi, 0xFF, sn * isz, stream));
  FMX_HIP(hipMemsetAsync(w.counts, 0, nf1 * T->n_tiles * 4, stream));
  FMX_HIP(hipMemsetAsync(T->trow, 0, ((size_t)m->nnz + 1) * 4, stream));
  if (T->tval) FMX_HIP(hipMemsetAsync(T->tval, 0, ((size_t)m->nnz + 1) * 4, stream));
  if (T->lfi16) hipLaunchKernelGGL((tiled_rows_k<uint16_t>), dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, stream, m->row_ptr, m->col, m->val, m->n, w.slot_of_feat, w.idx_in,
                                   reinterpret_cast<uint16_t*>(T->lfi), T->lval);
  else hipLaunchKernelGGL((tiled_rows_k<uint32_t>), dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, stream, m->row_ptr, m->col, m->val, m->n, w.slot_of_feat, w.idx_in,
                          reinterpret_cast<uint32_t*>(T->lfi), T->lval);
  hipLaunchKernelGGL(tiled_tile_base_k, dim3((unsigned)(T->n_tiles / 256 + 1)), dim3(256), 0, stream, m->row_ptr, m->n, ts, T->n_tiles, T->tile_base);
  const unsigned col_grid = (unsigned)(((int64_t)p * 64 + WG_THREADS - 1) / WG_THREADS);
  hipLaunchKernelGGL(tiled_count_k, dim3(col_grid), dim3(WG_THREADS), 0, stream, m->col_ptr, m->crow, w.rank_of, p, ts, nf1, w.counts);
  size_t scan_bytes = 0;
  FMX_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, w.counts, T->toff, 0u, nf1, rocprim::plus<uint32_t>(), stream));
  if (!ok(hipMalloc(&w.scan, scan_bytes ? scan_bytes : 16))) return FMX_OK;
  for (int t = 0; t < T->n_tiles; ++t)
    FMX_HIP(rocprim::exclusive_scan(w.scan, scan_bytes, w.counts + (size_t)t * nf1, T->toff + (size_t)t * nf1, 0u, nf1, rocprim::plus<uint32_t>(), stream));
  hipLaunchKernelGGL(tiled_scatter_k, dim3(col_grid), dim3(WG_THREADS), 0, stream, m->col_ptr, m->crow, m->cval, w.rank_of, p, ts, nf1, T->toff, T->tile_base, T->trow, T->tval);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipStreamSynchronize(stream));
  m->als_tiled = T.release();
  return FMX_OK;
}

int als_tiled_info(const fmx_matrix* m, int32_t* levels_tiled, int64_t* tile_rows, int32_t* n_tiles) {
  const AlsTiled* T = reinterpret_cast<const AlsTiled*>(m->als_tiled);
  if (levels_tiled) *levels_tiled = T ? T->n_slots : 0;
  if (tile_rows) *tile_rows = T ? (1LL << T->tshift) : 0;
  if (n_tiles) *n_tiles = T ? T->n_tiles : 0;
  return FMX_OK;
}

// ---- the sweep -------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool bad_number_t(double x) { return isnan(x) || isinf(x); }

// The (q, e) pairs are the one table every pass of every level re-reads (160 MB at configs[4]: it fits the 256 MB Infinity Cache, but not next
// to the 40 MB per level of lists, of level-major indices and of per-tile sums that are read ONCE).  Those streams are loaded non-temporally
// so that they do not push the pairs out (NT = false: the default cache policy, for A/B: FMX_ALS_NT=0).
template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
typedef double fmx_v2d __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 stream_load(const double2* p) {   // (the builtin takes native vectors, not HIP's struct)
  if constexpr (NT) { const fmx_v2d v = __builtin_nontemporal_load(reinterpret_cast<const fmx_v2d*>(p)); return make_double2(v.x, v.y); }
  else return *p;
}

// the level's current coordinates, gathered once into a level-sized vector (read coalesced by every tile's lists)
template <bool W>
__global__ void als_tile_prep_k(const uint32_t* __restrict__ feats, uint32_t cnt, const double* __restrict__ P, int kp, const SweepDyn* __restrict__ dyn,
                                double* __restrict__ vf) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  vf[i] = P[W ? (size_t)feats[i] : (size_t)feats[i] * kp + dyn->f];
}

// blockIdx -> (tile, chunk of the level's features): the B workgroups of a tile are consecutive in ONE XCD's share of the grid (blocks are dealt
// round-robin over the eight XCDs, so blocks b and b + 8 share one), and an XCD works through its tiles one after the other -- the tile's
// (q, e) slice is fetched into that L2 once and gathered from there.  Placement is for speed only.
template <bool W, int LG, bool UNIT, bool NT, int U>
__global__ __launch_bounds__(WG_THREADS) void als_tile_sums_k(const uint32_t* __restrict__ toff, size_t nf1, uint32_t lvl0, uint32_t cnt,
                                                              const int64_t* __restrict__ tile_base, const uint32_t* __restrict__ trow, const float* __restrict__ tval,
                                                              const double* __restrict__ vf, const double2* __restrict__ qe, int tshift, int n_tiles, int B,
                                                              double2* __restrict__ partial, uint32_t max_cnt) {
  const int b = blockIdx.x;
  const int x = b & 7, i = b >> 3;
  const int tile = (i / B) * 8 + x, chunk = i % B;
  if (tile >= n_tiles) return;
  constexpr int LISTS = WG_THREADS / LG;
  const uint32_t fi_raw = (uint32_t)chunk * LISTS + threadIdx.x / LG;
  const bool live = fi_raw < cnt;             // (a whole lane group is live or not)
  const uint32_t fi = live ? fi_raw : cnt - 1;
  const int lg = threadIdx.x % LG;
  const uint32_t* off = toff + (size_t)tile * nf1 + lvl0 + fi;
  const uint32_t lb = stream_load<NT>(off), le = live ? stream_load<NT>(off + 1) : lb;
  const int64_t tb = tile_base[tile];
  const double2* __restrict__ slice = qe + ((size_t)tile << tshift);
  const double old = vf[fi];
  double mean = 0.0, var = 0.0;
  // loads are unconditional on a clamped index, the values selected afterwards (a load under a condition is a branch whose join waits: DESIGN 6.2)
  for (uint32_t t0 = lb + lg; t0 < le; t0 += LG * U) {
    uint32_t rr[U]; float xs[U]; double2 c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t t = t0 + u * LG, tc = t < le ? t : t0;
      rr[u] = stream_load<NT>(trow + tb + tc);
      xs[u] = UNIT ? 1.0f : stream_load<NT>(tval + tb + tc);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) c[u] = slice[rr[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (t0 + u * LG >= le) continue;
      if (W) { const double xd = (double)xs[u]; mean += c[u].y * xd - old * xd * xd; var += xd * xd; }                                       // :216-219
      else { const float xx = xs[u] * xs[u]; const double h = (double)xs[u] * c[u].x - (double)xx * old; mean += h * c[u].y; var += h * h; }  // :310-317
    }
  }
#pragma unroll
  for (int o = LG / 2; o > 0; o >>= 1) { mean += __shfl_xor(mean, o); var += __shfl_xor(var, o); }
  if (lg == 0 && live) partial[(size_t)tile * max_cnt + fi] = make_double2(mean, var);
}

// 16 features per workgroup, 16 threads per feature: thread (tl, fl) adds the pairs of tiles tl, tl + 16, ... of feature fl in that order, the sixteen
// part sums meet in LDS and are added in tl order -- a fixed association (reproducible), and sixteen loads in flight per feature where one thread per
// feature walked the tiles one load at a time (24 us of a 245 us level).
// Workgroups beyond the level's own (step_blocks) gather the coordinates of the NEXT tiled level of the sweep into the other half of the
// coordinate buffer (features of different levels are different features: nothing this level writes is read there) -- the prep launch of that level is saved.
template <bool W, bool NT>
__global__ __launch_bounds__(WG_THREADS) void als_tile_step_k(const uint32_t* __restrict__ feats, uint32_t cnt, const double2* __restrict__ partial, uint32_t max_cnt, int n_tiles,
                                                              double* __restrict__ P, int kp, const SweepDyn* __restrict__ dyn, const double* __restrict__ vf,
                                                              double2* __restrict__ vstep, unsigned step_blocks, const uint32_t* __restrict__ next_feats, uint32_t next_cnt,
                                                              double* __restrict__ next_vf) {
  __shared__ double2 red[16][16];
  if (blockIdx.x >= step_blocks) {
    const uint32_t i = (blockIdx.x - step_blocks) * WG_THREADS + threadIdx.x;
    if (i < next_cnt) next_vf[i] = P[W ? (size_t)next_feats[i] : (size_t)next_feats[i] * kp + dyn->f];
    return;
  }
  const int fl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const uint32_t fi = blockIdx.x * 16 + fl;
  const uint32_t fc = fi < cnt ? fi : cnt - 1;
  double mean = 0.0, var = 0.0;
  for (int t = tl; t < n_tiles; t += 16) { const double2 s = stream_load<NT>(partial + (size_t)t * max_cnt + fc); mean += s.x; var += s.y; }
  red[tl][fl] = make_double2(mean, var);
  __syncthreads();
  if (tl != 0 || fi >= cnt) return;
  mean = 0.0; var = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) { mean += red[q][fl].x; var += red[q][fl].y; }
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  const double* __restrict__ znorm = dyn->znorm;
  const uint32_t i = feats[fi];
  const double old = vf[fi];
  double nv;
  if (W) {
    var = 1.0 / (lambda + alpha * var);
    mean = -var * (alpha * mean - mu * lambda);
    nv = bad_number_t(var) ? 0.0 : (znorm ? mean + var * znorm[i] : mean);      // (the variance as Rf_rnorm's sd: :239, kept)
  } else {
    mean -= old * var;                               // :318
    var = 1.0 / (lambda + alpha * var);              // :319
    mean = -var * (alpha * mean - mu * lambda);      // :320
    nv = bad_number_t(var) ? 0.0 : (znorm ? mean + sqrt(var) * znorm[i] : mean);
  }
  if (bad_number_t(nv)) { vstep[fi] = make_double2(old, nan("")); return; }  // CHECK_PARAM (:336): keep the old value; NaN tells the apply pass to skip the feature's rows
  P[W ? (size_t)i : (size_t)i * kp + dyn->f] = nv;
  vstep[fi] = make_double2(old, old - nv);
}

// R rows per thread, a workgroup's rows contiguous per r (coalesced): all of a thread's loads go out before the first is used
// QNEXT (the LAST level of a factor's sweep): this factor's q is dead once its last correction is applied, so the pass stores the NEXT factor's q
// (qnext[r], one coalesced double per row, from the factor-major table of all factors' q) in its place -- for EVERY row, also those the level does
// not touch -- and the per-factor pick pass over the pairs is saved.
template <bool W, bool UNIT, bool NT, int R, bool QNEXT, typename IT>
__global__ __launch_bounds__(WG_THREADS) void als_rows_apply_k(const IT* __restrict__ lfi, const float* __restrict__ lval, int64_t n,
                                                               const double2* __restrict__ vstep, double2* __restrict__ qe, const double* __restrict__ qnext) {
  const int64_t r0 = (int64_t)blockIdx.x * (WG_THREADS * R) + threadIdx.x;
  uint32_t fi[R]; float x[R]; double2 c[R], s[R]; double qn[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int64_t r = r0 + (int64_t)u * WG_THREADS, rc = r < n ? r : n - 1;
    { const IT raw = stream_load<NT>(lfi + rc); fi[u] = raw == (IT)~(IT)0 ? TILED_NONE : (uint32_t)raw; }
    x[u] = UNIT ? 1.0f : stream_load<NT>(lval + rc);
    c[u] = qe[rc];
    qn[u] = QNEXT ? stream_load<NT>(qnext + rc) : 0.0;
  }
#pragma unroll
  for (int u = 0; u < R; ++u) s[u] = vstep[fi[u] == TILED_NONE ? 0u : fi[u]];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int64_t r = r0 + (int64_t)u * WG_THREADS;
    if (r >= n) continue;
    const bool skip = fi[u] == TILED_NONE || s[u].y != s[u].y;
    if (skip) { if (QNEXT) qe[r] = make_double2(qn[u], c[u].y); continue; }
    if (W) {
      qe[r] = make_double2(c[u].x, c[u].y - (double)x[u] * s[u].y);                           // :246-252
    } else {
      const float xx = x[u] * x[u];
      const double h = (double)x[u] * c[u].x - (double)xx * s[u].x;
      qe[r] = make_double2(QNEXT ? qn[u] : c[u].x - (double)x[u] * s[u].y, c[u].y - h * s[u].y);   // :341-350
    }
  }
}

static int tile_ws(fmx_engine* e, const AlsTiled* T, double2** partial, double** vf, double2** vstep) {
  const size_t need = ((size_t)T->n_tiles * T->max_cnt + T->max_cnt) * sizeof(double2) + 2 * (size_t)T->max_cnt * sizeof(double);   // (vf: two halves)
  if (e->als_tile_ws_bytes < need) {
    FMX_HIP(hipStreamSynchronize(e->stream));
    (void)hipFree(e->als_tile_ws); e->als_tile_ws = nullptr; e->als_tile_ws_bytes = 0;
    FMX_HIP(hipMalloc(&e->als_tile_ws, need));
    e->als_tile_ws_bytes = need;
  }
  *partial = reinterpret_cast<double2*>(e->als_tile_ws);
  *vstep = *partial + (size_t)T->n_tiles * T->max_cnt;
  *vf = reinterpret_cast<double*>(*vstep + T->max_cnt);
  return FMX_OK;
}

// one level of the w sweep (W) or of one factor of the V sweep in the tiled form; *done = false: the level is not a tiled one
template <bool W>
int als_tiled_level(fmx_engine* e, fmx_matrix* m, int level, bool last, double2* d_qe, const SweepDyn* dyn, bool* done) {
  *done = false;
  const AlsTiled* T = reinterpret_cast<const AlsTiled*>(m->als_tiled);
  if (!T || level >= (int)T->slot_of_level.size() || T->slot_of_level[(size_t)level] < 0) return FMX_OK;
  const int s = T->slot_of_level[(size_t)level];
  const uint32_t lvl0 = T->lvl0[(size_t)s], cnt = T->cnt[(size_t)s];
  double2 *partial = nullptr, *vstep = nullptr;
  double* vf2 = nullptr;
  FMX_TRY(tile_ws(e, T, &partial, &vf2, &vstep));
  double* P = W ? e->dw : e->dV;
  const size_t nf1 = (size_t)T->n_feats + 1;
  const dim3 blk(WG_THREADS);
  // FMX_ALS_NT: bit 0 the lists of the sums pass, bit 1 the streams of the correction pass, bit 2 the per-tile sums read by the step kernel
  static const int nt_mask = env_int("FMX_ALS_NT", 6);
  static const int rows_per_thread = env_int("FMX_ALS_APPLY_ROWS", 8);
  static const bool fold_prep = env_int("FMX_ALS_FOLD_PREP", 1) != 0;
  static const int sums_u = env_int("FMX_ALS_SUMS_U", 4);
  // this level's coordinates: gathered by the previous tiled level's step kernel (its spare workgroups), or here
  int buf = 0;
  if (fold_prep && e->als_vf_slot == s) buf = e->als_vf_buf;
  else hipLaunchKernelGGL((als_tile_prep_k<W>), dim3((cnt + 255) / 256), dim3(256), 0, e->stream, T->feats + lvl0, cnt, (const double*)P, e->kp64, dyn, vf2);
  double* vf = vf2 + (size_t)buf * T->max_cnt;
  const int lists = WG_THREADS / T->lg;
  const int B = (int)((cnt + lists - 1) / lists);
  const dim3 g((unsigned)(((T->n_tiles + 7) / 8) * 8 * B));
#define FMX_SUMS3(LGv, UNITv, NTv, Uv)                                                                                                                    \
  hipLaunchKernelGGL((als_tile_sums_k<W, LGv, UNITv, NTv, Uv>), g, blk, 0, e->stream, T->toff, nf1, lvl0, cnt, T->tile_base, T->trow, T->tval, (const double*)vf, \
                     (const double2*)d_qe, T->tshift, T->n_tiles, B, partial, T->max_cnt)
#define FMX_SUMS2(LGv, UNITv, NTv) do { if (sums_u == 8) FMX_SUMS3(LGv, UNITv, NTv, 8); else FMX_SUMS3(LGv, UNITv, NTv, 4); } while (0)
#define FMX_SUMS(LGv)                                                                                                                                     \
  do {                                                                                                                                                    \
    if (T->unit) { if (nt_mask & 1) FMX_SUMS2(LGv, true, true); else FMX_SUMS2(LGv, true, false); }                                                         \
    else { if (nt_mask & 1) FMX_SUMS2(LGv, false, true); else FMX_SUMS2(LGv, false, false); }                                                               \
  } while (0)
  switch (T->lg) {
    case 2: FMX_SUMS(2); break;
    case 4: FMX_SUMS(4); break;
    case 8: FMX_SUMS(8); break;
    default: FMX_SUMS(1); break;
  }
#undef FMX_SUMS
#undef FMX_SUMS2
#undef FMX_SUMS3
  // the next tiled level of this sweep, if the very next level is one (anything in between may not be skipped: it would run after the gather,
  // which is harmless -- other features -- but keep the rule simple)
  const uint32_t* next_feats = nullptr; uint32_t next_cnt = 0;
  int next_slot = -1;
  if (fold_prep && !last && level + 1 < (int)T->slot_of_level.size() && T->slot_of_level[(size_t)level + 1] >= 0) {
    next_slot = T->slot_of_level[(size_t)level + 1];
    next_feats = T->feats + T->lvl0[(size_t)next_slot]; next_cnt = T->cnt[(size_t)next_slot];
  }
  double* next_vf = vf2 + (size_t)(1 - buf) * T->max_cnt;
  const unsigned sgrid = (cnt + 15) / 16, pgrid = (next_cnt + WG_THREADS - 1) / WG_THREADS;
  if (nt_mask & 4) hipLaunchKernelGGL((als_tile_step_k<W, true>), dim3(sgrid + pgrid), blk, 0, e->stream, T->feats + lvl0, cnt, (const double2*)partial, T->max_cnt, T->n_tiles, P, e->kp64, dyn,
                                      (const double*)vf, vstep, sgrid, next_feats, next_cnt, next_vf);
  else hipLaunchKernelGGL((als_tile_step_k<W, false>), dim3(sgrid + pgrid), blk, 0, e->stream, T->feats + lvl0, cnt, (const double2*)partial, T->max_cnt, T->n_tiles, P, e->kp64, dyn,
                          (const double*)vf, vstep, sgrid, next_feats, next_cnt, next_vf);
  e->als_vf_slot = next_slot; e->als_vf_buf = 1 - buf;
  const uint32_t* lfi32 = T->lfi16 ? nullptr : reinterpret_cast<const uint32_t*>(T->lfi) + (size_t)s * T->n;
  const uint16_t* lfi16 = T->lfi16 ? reinterpret_cast<const uint16_t*>(T->lfi) + (size_t)s * T->n : nullptr;
  const float* lval = T->lval ? T->lval + (size_t)s * T->n : nullptr;
  const double* qnext = (!W && last) ? e->als_qnext : nullptr;
#define FMX_APPLY(UNITv, NTv, Rv, QNv)                                                                                                                    \
  do {                                                                                                                                                    \
    const dim3 ag((unsigned)((T->n + WG_THREADS * Rv - 1) / (WG_THREADS * Rv)));                                                                            \
    if (lfi16) hipLaunchKernelGGL((als_rows_apply_k<W, UNITv, NTv, Rv, QNv, uint16_t>), ag, blk, 0, e->stream, lfi16, lval, T->n, (const double2*)vstep, d_qe, qnext); \
    else hipLaunchKernelGGL((als_rows_apply_k<W, UNITv, NTv, Rv, QNv, uint32_t>), ag, blk, 0, e->stream, lfi32, lval, T->n, (const double2*)vstep, d_qe, qnext);       \
  } while (0)
#define FMX_APPLY_Q(UNITv, NTv, Rv) do { if (qnext) FMX_APPLY(UNITv, NTv, Rv, true); else FMX_APPLY(UNITv, NTv, Rv, false); } while (0)
#define FMX_APPLY_R(Rv)                                                                                                                                   \
  do {                                                                                                                                                    \
    if (T->unit) { if (nt_mask & 2) FMX_APPLY_Q(true, true, Rv); else FMX_APPLY_Q(true, false, Rv); }                                                       \
    else { if (nt_mask & 2) FMX_APPLY_Q(false, true, Rv); else FMX_APPLY_Q(false, false, Rv); }                                                             \
  } while (0)
  switch (rows_per_thread) {
    case 1: FMX_APPLY_R(1); break;
    case 2: FMX_APPLY_R(2); break;
    case 4: FMX_APPLY_R(4); break;
    default: FMX_APPLY_R(8); break;
  }
#undef FMX_APPLY_R
#undef FMX_APPLY_Q
#undef FMX_APPLY
  if (qnext) e->als_qnext = nullptr;   // folded: v_sweep_enqueue skips the next factor's pick
  *done = true;
  return FMX_OK;
}
template int als_tiled_level<true>(fmx_engine*, fmx_matrix*, int, bool, double2*, const SweepDyn*, bool*);
template int als_tiled_level<false>(fmx_engine*, fmx_matrix*, int, bool, double2*, const SweepDyn*, bool*);

}  // namespace fmx
