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
  // The LEVEL-ORDER form of the V sweep (below): available when the plan is COMPLETE -- every level of the plan is a tiled one and every row holds exactly
  // one feature of every level (one-column-per-field data) -- so that position i of tile t's level-s block, entry tile_base[t] + toff[t][lvl0_s] + i of
  // trow / tval, is also a position of the tile's slice of a (q, e) array kept in level s's list order.
  int complete = 0;
  uint32_t max_list = 0;         // longest (tile, feature) list of the plan (decides whether the sums kernel may keep its offsets in 16 bits)
  uint32_t* perm = nullptr;      // [n_slots][n] position, inside the same tile, that the row at position i of level s's order has in the order of level s + 1 (cyclic)
  void* fidx = nullptr;          // [n_slots][n] index (inside its level) of the feature whose list position i belongs to (u16 / u32 like lfi)
  void* blocks = nullptr;        // the BLOCK form of the level-order sweep (fm_als_blocks.hip), where it applies: perm / fidx are not built then
  ~AlsTiled() {
    als_blocks_free(blocks);
    (void)hipFree(feats); (void)hipFree(lfi); (void)hipFree(lval); (void)hipFree(toff); (void)hipFree(tile_base); (void)hipFree(trow); (void)hipFree(tval);
    (void)hipFree(perm); (void)hipFree(fidx);
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

struct AlsTiled;
static int order_build(fmx_matrix* m, AlsTiled* T, const std::vector<uint32_t>& h_feats, hipStream_t stream);

// Which levels of the exact plan go through the tiled form, and their plan.  Called at the end of build_plan (the CSC and the levels exist).
// A failure to allocate -- here, or of the sweep's workspace later (als_tiled_level drops the plan then) -- leaves the matrix without a tiled plan (the
// column-walking kernels do every level): never an error.
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
  // tile size: 131 072 rows (a 2 MB slice of pairs) for the three-pass form; 65 536 where the plan looks complete (every level tiled, one entry per row and
  // level) and the V sweep will take the level-order form: its permuting scatter merges better in the L2 on 1 MB slices (apply kernel 92 against 107 us per level
  // at configs[4], the sums kernel 65 against 57: 116.8 against 112.0 M examples/s, profiles/r05_order_ab11.txt).  FMX_ALS_TILE_ROWS pins it.
  const bool order_candidate = env_int("FMX_ALS_ORDER", 1) != 0 && n_slots == L && m->nnz == (int64_t)n_slots * m->n;
  int rows_want = env_int("FMX_ALS_TILE_ROWS", order_candidate ? 65536 : 131072);
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
  FMX_TRY(order_build(m, T.get(), feats, stream));
  m->als_tiled = T.release();
  return FMX_OK;
}

// ---- the level-order plan ----------------------------------------------------------------------------------------------------------------
// complete: the level-s block of every tile holds exactly the tile's rows
__global__ void order_complete_k(const uint32_t* __restrict__ toff, size_t nf1, const uint32_t* __restrict__ lvl0, const uint32_t* __restrict__ cnt, int n_slots, int n_tiles,
                                 int64_t n, int tshift, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots * n_tiles) return;
  const int s = i / n_tiles, t = i % n_tiles;
  const int64_t r0 = (int64_t)t << tshift, r1 = min(n, r0 + ((int64_t)1 << tshift));
  const uint32_t* off = toff + (size_t)t * nf1;
  if ((int64_t)off[lvl0[s]] != (int64_t)s * (r1 - r0) || (int64_t)off[lvl0[s] + cnt[s]] != (int64_t)(s + 1) * (r1 - r0)) *bad = 1;
}
__global__ void order_max_list_k(const uint32_t* __restrict__ toff, size_t nf1, int n_tiles, uint32_t n_feats, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t len = 0;
  if (i < (size_t)n_tiles * n_feats) { const size_t t = i / n_feats, f = i % n_feats; len = toff[t * nf1 + f + 1] - toff[t * nf1 + f]; }
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) len = max(len, (uint32_t)__shfl_xor((int)len, ofs));
  if ((threadIdx.x & 63) == 0 && len > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, len);   // (an atomic only where it can still raise the value: 2.4 M waves on one word took 27 ms)
}
// inv[s][row] = position of the row in its tile's level-s order
__global__ void order_inverse_k(const uint32_t* __restrict__ trow, const int64_t* __restrict__ tile_base, int64_t n, int tshift, int n_slots, uint32_t* __restrict__ inv) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (slot, position) flattened: slot-major
  if (g >= (int64_t)n_slots * n) return;
  const int s = (int)(g / n);
  const int64_t at = g % n, t = at >> tshift, r0 = t << tshift, rows = min(n, r0 + ((int64_t)1 << tshift)) - r0;
  const uint32_t row = trow[tile_base[t] + (int64_t)s * rows + (at - r0)];
  inv[(size_t)s * n + r0 + row] = (uint32_t)(at - r0);
}
template <typename IT>
__global__ void order_perm_k(const uint32_t* __restrict__ trow, const int64_t* __restrict__ tile_base, int64_t n, int tshift, int n_slots, const uint32_t* __restrict__ inv,
                             const IT* __restrict__ lfi, uint32_t* __restrict__ perm, IT* __restrict__ fidx) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)n_slots * n) return;
  const int s = (int)(g / n), s2 = s + 1 < n_slots ? s + 1 : 0;
  const int64_t at = g % n, t = at >> tshift, r0 = t << tshift, rows = min(n, r0 + ((int64_t)1 << tshift)) - r0;
  const int64_t row = r0 + trow[tile_base[t] + (int64_t)s * rows + (at - r0)];
  perm[g] = inv[(size_t)s2 * n + row];
  fidx[g] = lfi[(size_t)s * n + row];
}

// Called at the end of als_tiled_build: the extra arrays of the level-order form where the plan is complete.  Any failure (incomplete plan, no memory) just
// leaves the form unavailable.  FMX_ALS_ORDER=0 switches it off (A/B runs, and the tests that compare the forms).
static int order_build(fmx_matrix* m, AlsTiled* T, const std::vector<uint32_t>& h_feats, hipStream_t stream) {
  const int order_env = env_int("FMX_ALS_ORDER", 2);   // 0: never, 1: the tile form, 2 (default): the block form where it applies, else the tile form
  if (order_env == 0) return FMX_OK;
  const int L = (int)m->als_level_ptr.size() - 1;
  int nonempty = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t c = m->als_level_ptr[(size_t)l + 1] - m->als_level_ptr[(size_t)l], h = m->als_heavy_ptr[(size_t)l + 1] - m->als_heavy_ptr[(size_t)l];
    const int64_t v = m->als_vh_ptr.empty() ? 0 : m->als_vh_ptr[(size_t)l + 1] - m->als_vh_ptr[(size_t)l];
    if (c + h + v > 0) { ++nonempty; if (T->slot_of_level[(size_t)l] < 0) return FMX_OK; }   // a level that keeps the column-walking kernels
  }
  if (nonempty != T->n_slots || T->n_slots < 1 || m->nnz != (int64_t)T->n_slots * m->n) return FMX_OK;
  auto ok = [](hipError_t e) { if (e != hipSuccess) (void)hipGetLastError(); return e == hipSuccess; };
  struct Tmp { uint32_t *lvl0 = nullptr, *cnt = nullptr, *inv = nullptr; int* bad = nullptr; ~Tmp() { (void)hipFree(lvl0); (void)hipFree(cnt); (void)hipFree(inv); (void)hipFree(bad); } } w;
  const size_t sn = (size_t)T->n_slots * (size_t)m->n, isz = T->lfi16 ? 2 : 4;
  if (!ok(hipMalloc(&w.lvl0, (size_t)T->n_slots * 4)) || !ok(hipMalloc(&w.cnt, (size_t)T->n_slots * 4)) || !ok(hipMalloc(&w.bad, sizeof(int)))) return FMX_OK;
  FMX_HIP(hipMemcpyAsync(w.lvl0, T->lvl0.data(), (size_t)T->n_slots * 4, hipMemcpyHostToDevice, stream));
  FMX_HIP(hipMemcpyAsync(w.cnt, T->cnt.data(), (size_t)T->n_slots * 4, hipMemcpyHostToDevice, stream));
  FMX_HIP(hipMemsetAsync(w.bad, 0, sizeof(int), stream));
  const int pairs = T->n_slots * T->n_tiles;
  hipLaunchKernelGGL(order_complete_k, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, T->toff, (size_t)T->n_feats + 1, w.lvl0, w.cnt, T->n_slots, T->n_tiles, m->n, T->tshift, w.bad);
  int bad = 0;
  FMX_HIP(hipMemcpyAsync(&bad, w.bad, sizeof(int), hipMemcpyDeviceToHost, stream));
  FMX_HIP(hipStreamSynchronize(stream));
  if (bad) return FMX_OK;   // some row lacks a level (or holds two features of one): the level blocks are not the tiles' rows
  if (order_env != 1) {
    const AlsBlocksIn in{m->n, T->n_slots, T->lvl0.data(), T->cnt.data(), h_feats.data(), T->feats, T->unit};
    FMX_TRY(als_blocks_build(m, in, &T->blocks, stream));
    if (T->blocks) { T->complete = 1; return FMX_OK; }
  }
  if (!ok(hipMalloc(&w.inv, sn * 4)) || !ok(hipMalloc(&T->perm, sn * 4)) || !ok(hipMalloc(&T->fidx, sn * isz))) {
    (void)hipFree(T->perm); (void)hipFree(T->fidx); T->perm = nullptr; T->fidx = nullptr;
    return FMX_OK;
  }
  {
    uint32_t* d_max = reinterpret_cast<uint32_t*>(w.bad);   // (done with: reused for the maximum)
    FMX_HIP(hipMemsetAsync(d_max, 0, sizeof(uint32_t), stream));
    const size_t pairs_tf = (size_t)T->n_tiles * T->n_feats;
    hipLaunchKernelGGL(order_max_list_k, dim3((unsigned)((pairs_tf + 255) / 256)), dim3(256), 0, stream, (const uint32_t*)T->toff, (size_t)T->n_feats + 1, T->n_tiles, T->n_feats, d_max);
    FMX_HIP(hipMemcpyAsync(&T->max_list, d_max, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  }
  const unsigned grid = (unsigned)((sn + 255) / 256);
  hipLaunchKernelGGL(order_inverse_k, dim3(grid), dim3(256), 0, stream, T->trow, T->tile_base, m->n, T->tshift, T->n_slots, w.inv);
  if (T->lfi16) hipLaunchKernelGGL((order_perm_k<uint16_t>), dim3(grid), dim3(256), 0, stream, T->trow, T->tile_base, m->n, T->tshift, T->n_slots, (const uint32_t*)w.inv,
                                   reinterpret_cast<const uint16_t*>(T->lfi), T->perm, reinterpret_cast<uint16_t*>(T->fidx));
  else hipLaunchKernelGGL((order_perm_k<uint32_t>), dim3(grid), dim3(256), 0, stream, T->trow, T->tile_base, m->n, T->tshift, T->n_slots, (const uint32_t*)w.inv,
                          reinterpret_cast<const uint32_t*>(T->lfi), T->perm, reinterpret_cast<uint32_t*>(T->fidx));
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipStreamSynchronize(stream));
  T->complete = 1;
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
  if (tile_ws(e, T, &partial, &vf2, &vstep) != FMX_OK) {
    // no room for the per-tile sums: this matrix keeps the column-walking kernels from here on (one failed allocation, not one per level of every factor)
    (void)hipGetLastError();
    als_tiled_free(m);
    return FMX_OK;   // *done == false: the caller runs als_level_k on this level
  }
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


// ---- the level-order form of the V sweep -------------------------------------------------------------------------------------------------
// What bounds the three passes above is the one random 16-byte access per stored nonzero (the sums pass's gather: 13 M L2 requests and 92 us of a 181 us
// level at configs[4], profiles/r04_pmc_summary_mcmc.json) PLUS two streaming passes over the pairs around it.  Here the (q, e) pairs are kept physically in
// the list order of the level that consumes them next -- tile t's slice holds its rows sorted by (feature of level s, row) -- so that
//   sums + step  als_order_sums_k   is a STREAM: a wave owns 16 consecutive features of the level; inside a tile their lists are one contiguous run of
//                                   pairs (one load instruction), which the lanes add into the wave's own LDS accumulators by feature index.  The sums of
//                                   a feature never leave the wave: no per-tile partial sums, no separate step kernel -- the coordinate step of :318-336 is
//                                   taken right there.  Reproducible bit for bit (wave-owned accumulators, in-order LDS).  als_order_walk_k is the earlier
//                                   form (a workgroup's runs staged through LDS, lists walked by lane groups): FMX_ALS_ORDER_SUMS=walk.
//   apply        als_order_apply_k  reads the pairs in level s's order (stream), the feature's (v_old, diff) by the entry's feature index (neighbouring
//                                   entries share it), corrects (:341-350) and writes each pair to its position in level s + 1's order: a permutation inside
//                                   the tile's slice -- the one random 16-byte access per nonzero that is left.  A tile's workgroups share an XCD and there are
//                                   enough of them per tile that an XCD works on ONE slice at a time: the scattered writes merge in its L2.
// The last level's apply of a factor writes into level 0's order and stores the NEXT factor's q (gathered by row from the factor-major table); entry and exit
// of the sweep convert between row order and level 0's order.  Measured before building: profiles/r05_level_order_probe.txt.
// TB tiles' offsets are taken in at once (all of configs[4]'s 77): their runs of pairs, laid end to end, are one virtual sequence that the workgroup
// streams in chunks of CH entries.  What bounds such a kernel is BYTES IN FLIGHT: a load comes back after ~5 us under load, so 6 TB/s want ~30 MB outstanding
// chip-wide (profiles/r05_level_order_probe*.txt: the bare access pattern streams at 6.3 TB/s with 33 MB in flight; a version with one chunk in flight per
// workgroup sat at 3 TB/s whatever its tiling).  The loads therefore run DEPTH chunks ahead in registers -- the LDS only ever holds the chunk being walked --
// and the grid is sized to ONE round of resident workgroups: `fb` features per workgroup (at most FBMAX), chosen by the host from the level's feature count.
// -DFMX_K1_TIMING (profiles/probes/level_order_probe.hip builds this kernel with it): clock stamps between the phases, summed per phase over wave 0 of every
// workgroup into fmx_k1_ticks[]; nothing in the product build
#ifdef FMX_K1_TIMING
__device__ unsigned long long fmx_k1_ticks[8];
#define FMX_K1_STAMP(slot) do { const unsigned long long now_ = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&fmx_k1_ticks[slot], now_ - k1_last_); k1_last_ = now_; } while (0)
#define FMX_K1_BEGIN unsigned long long k1_last_ = wall_clock64()
#else
#define FMX_K1_STAMP(slot) do {} while (0)
#define FMX_K1_BEGIN do {} while (0)
#endif
// OT: the type the list offsets are kept in (relative to their run's first pair): uint16_t where no run of a workgroup's features inside a tile can pass 65 535
// pairs (the host knows the longest (tile, feature) list), which halves the offsets' share of the LDS: four workgroups per CU instead of three, and TB = 256
// tiles in one batch where the tiles are 65 536 rows.  MINW: workgroups per CU the register allocation is held to.
template <bool UNIT, int FBMAX, int TB, int CH, int DEPTH, typename OT, int MINW>
__global__ __launch_bounds__(WG_THREADS, MINW) void als_order_walk_k(const uint32_t* __restrict__ toff, size_t nf1, uint32_t lvl0, uint32_t cnt, int fb, const int64_t* __restrict__ tile_base,
                                                               const float* __restrict__ tval, const double2* __restrict__ src, int tshift, int n_tiles,
                                                               const uint32_t* __restrict__ feats, double* __restrict__ P, int kp, const SweepDyn* __restrict__ dyn,
                                                               double2* __restrict__ vstep) {
  constexpr int LG = WG_THREADS / FBMAX;       // lanes per feature: the tiles a chunk touches are dealt round-robin to the lanes of a group
  constexpr int PER = CH / WG_THREADS;
  static_assert((TB & (TB - 1)) == 0 && TB <= WG_THREADS && FBMAX == 64, "TB: a power of two, one thread per tile in the prefix step; one row of offsets per wave instruction");
  __shared__ OT o[TB][FBMAX];                  // list offsets of the workgroup's features in the batch's tiles, relative to the run's first pair (o[tb][0] = 0); fb <= FBMAX - 1
  __shared__ uint32_t vstart[TB + 1];          // the batch's runs laid end to end
  __shared__ uint32_t blk[TB];                 // position of each run's first pair inside its tile's level block
  __shared__ int64_t xbase[UNIT ? 1 : TB];     // first entry of each tile's level block in tval
  __shared__ double2 lp[CH];
  __shared__ float lx[UNIT ? 1 : CH];
  constexpr uint32_t MAPPED = 1u << 16;        // virtual positions the tile map covers (a batch of regular data: TB x fb x a few entries)
  __shared__ uint8_t tmap[MAPPED / 64 + 1];    // tile of virtual position 64 k: tile_of() then needs one or two LDS reads instead of log2(TB) dependent ones
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
  // position v of the virtual sequence -> its tile of the batch (the last tb with vstart[tb] <= v; empty runs are skipped by construction).  The binary
  // search is log2(TB) DEPENDENT LDS reads: 16 of the kernel's 56 us when every load and every chunk's walk began with one (knock-outs,
  // profiles/r05_k1_knockouts.txt).  tile_of() starts from a map of every 64th position instead and steps over at most a few short runs.
  auto tile_search = [&](uint32_t v) { int tb = 0;
#pragma unroll
    for (int st = TB / 2; st > 0; st >>= 1) tb += (vstart[tb + st] <= v) ? st : 0;
    return tb; };
  bool mapped = false;
  auto tile_of = [&](uint32_t v) {
    if (!mapped) return tile_search(v);
    int tb = tmap[v >> 6];
    while (vstart[tb + 1] <= v) ++tb;           // (v < total = vstart[TB]: ends at the latest at TB - 1)
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
        const uint32_t first = __shfl(ov[q], 0);   // (lane 0 holds the offset of feature f0: the run's first pair)
        if (tb < nb) { o[tb][j] = (OT)(ov[q] - first); if (j == 0) blk[tb] = first; }
      }
      __syncthreads();
      if ((int)threadIdx.x < nb) { blk[threadIdx.x] -= bv; if (!UNIT) xbase[threadIdx.x] = tbv + (int64_t)bv; }   // the run's first pair inside the tile's level block
    }
    __syncthreads();
    if (threadIdx.x < 64) {                     // exclusive prefix of the run lengths: one wave, TB / 64 values per lane
      uint32_t carry = 0;
      for (int b0 = 0; b0 < TB; b0 += 64) {
        const int tb = b0 + threadIdx.x;
        const uint32_t len = (tb < nb) ? (uint32_t)o[tb][fb] : 0u;
        uint32_t inc = len;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) { const uint32_t up = __shfl_up(inc, ofs); if ((int)threadIdx.x >= ofs) inc += up; }
        if (tb < TB) vstart[tb] = carry + inc - len;
        carry += __shfl(inc, 63);
      }
      if (threadIdx.x == 0) vstart[TB] = carry;
    }
    __syncthreads();
    const uint32_t total = vstart[TB];
    mapped = total <= MAPPED;                    // (uniform)
    if (mapped) {
      for (uint32_t q = threadIdx.x; q * 64 < total; q += WG_THREADS) tmap[q] = (uint8_t)tile_search(q * 64);
      __syncthreads();
    }
    FMX_K1_STAMP(0);                            // offsets + prefix + map
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
            // The walk is a chain of LDS latencies (list bounds, then entries): 25 of the kernel's 56 us when every lane took one list at a time
            // (knock-outs, profiles/r05_k1_knockouts.txt).  A lane therefore takes TWO of its lists per trip -- their bounds are read together, then the
            // first four entries of both (clamped, added under a test), then what is left of either.  Any fixed association is as good as any other.
            auto entry = [&](uint32_t v, bool on) {
              const double2 c = lp[v - c0];
              const float x = UNIT ? 1.0f : lx[v - c0];
              const float xx = x * x;
              const double h = (double)x * c.x - (double)xx * old;   // :310-317
              if (live && on) { mean += h * c.y; var += h * h; }
            };
            for (int tb = t_lo + lane; tb <= t_hi; tb += 2 * LG) {
              const int tb2 = tb + LG <= t_hi ? tb + LG : tb;
              const uint32_t s1 = vstart[tb], oa1 = o[tb][gc], ob1 = o[tb][gc + 1];
              const uint32_t s2 = vstart[tb2], oa2 = o[tb2][gc], ob2 = o[tb2][gc + 1];
              const uint32_t a1 = max(s1 + oa1, c0), b1 = min(s1 + ob1, c1);
              uint32_t a2 = max(s2 + oa2, c0), b2 = min(s2 + ob2, c1);
              if (tb2 == tb) { a2 = 0; b2 = 0; }
              double2 e1[4], e2[4]; float x1[4], x2[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const uint32_t v1 = min(a1 + i, b1 > a1 ? b1 - 1 : a1) - c0, v2 = (b2 > a2 ? min(a2 + i, b2 - 1) : c0) - c0;
                e1[i] = lp[v1 < CH ? v1 : 0]; e2[i] = lp[v2 < CH ? v2 : 0];
                x1[i] = UNIT ? 1.0f : lx[v1 < CH ? v1 : 0]; x2[i] = UNIT ? 1.0f : lx[v2 < CH ? v2 : 0];
              }
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float xx = x1[i] * x1[i];
                const double h = (double)x1[i] * e1[i].x - (double)xx * old;   // :310-317
                if (live && a1 + i < b1) { mean += h * e1[i].y; var += h * h; }
              }
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float xx = x2[i] * x2[i];
                const double h = (double)x2[i] * e2[i].x - (double)xx * old;
                if (live && a2 + i < b2) { mean += h * e2[i].y; var += h * h; }
              }
              for (uint32_t v = a1 + 4; v < b1; ++v) entry(v, true);
              for (uint32_t v = a2 + 4; v < b2; ++v) entry(v, true);
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

// The sums + step kernel the product takes [r5, last form]: ONE WAVE per 16 consecutive features of the level, one pair per lane per tile (the wave's run of a
// tile is at most 64 pairs on regular data: one load instruction; longer runs: the rest in pieces of 64), the two sums of every feature kept in LDS
// accumulators OWNED BY THE WAVE and fed by ds_add_f64.  No other wave ever touches them, a wave's LDS operations execute in order, and the lanes of ONE
// instruction that hit the same accumulator are serialised by the LDS in a fixed order: the sums are reproducible bit for bit (tests: two sweeps, two engines),
// in an association of their own like every other form (1e-10 against the oracle).  No list walk, no search, no barrier: per tile the run's bounds (three
// offsets, two batches of tiles ahead), its pairs and feature indices (one batch ahead), ~30 instructions.  55 us per level at configs[4] against the
// walking kernel's 65 (als_order_walk_k below, FMX_ALS_ORDER_SUMS=walk); what is left is the memory pattern itself (44-47 us with the adds knocked out:
// profiles/r05_k1_atom_probe2.txt).
template <bool UNIT, int NA, typename IT>
__global__ __launch_bounds__(WG_THREADS) void als_order_sums_k(const uint32_t* __restrict__ toff, size_t nf1, uint32_t lvl0, uint32_t cnt, const int64_t* __restrict__ tile_base,
                                                               const float* __restrict__ tval, const IT* __restrict__ fidx, const double2* __restrict__ src, int tshift,
                                                               int n_tiles, const uint32_t* __restrict__ feats, double* __restrict__ P, int kp,
                                                               const SweepDyn* __restrict__ dyn, double2* __restrict__ vstep) {
  constexpr int WAVES = WG_THREADS / 64;
  __shared__ double acc[WAVES][16][2];
  __shared__ double oldv[WAVES][16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t f0 = (blockIdx.x * WAVES + wv) * 16u;
  if (f0 >= cnt) return;                          // (whole waves leave: nothing below synchronises across waves)
  const uint32_t fend = min(f0 + 16u, cnt);
  const int f = dyn->f;
  if (lane < 16) {
    const uint32_t fi = min(f0 + (uint32_t)lane, cnt - 1);
    oldv[wv][lane] = P[(size_t)feats[fi] * kp + f];
    acc[wv][lane][0] = 0.0; acc[wv][lane][1] = 0.0;
  }
  const uint32_t which = lane == 0 ? f0 : (lane == 1 ? fend : 0u);   // lanes 0, 1, 2 hold: the run's first offset, its end, the level block's first offset
  struct Offs { uint32_t o[NA]; };
  struct Run { double2 pv[NA]; float xv[NA]; uint32_t fx[NA], rl[NA]; size_t at0[NA]; int64_t x0[NA]; };
  auto load_offs = [&](int t0, Offs& o) {
#pragma unroll
    for (int u = 0; u < NA; ++u) { const int t = min(t0 + u, n_tiles - 1); o.o[u] = stream_load<true>(toff + (size_t)t * nf1 + lvl0 + which); }
  };
  auto load_run = [&](int t0, const Offs& o, Run& p) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int t = min(t0 + u, n_tiles - 1);
      const uint32_t a = __builtin_amdgcn_readlane(o.o[u], 0), b = __builtin_amdgcn_readlane(o.o[u], 1), base = __builtin_amdgcn_readlane(o.o[u], 2);
      p.rl[u] = t0 + u < n_tiles ? b - a : 0u;
      p.at0[u] = ((size_t)t << tshift) + (a - base);              // the run's first pair (and feature index) in the level-ordered arrays
      const uint32_t e = min((uint32_t)lane, p.rl[u] > 0 ? p.rl[u] - 1 : 0u);
      p.pv[u] = stream_load<true>(src + p.at0[u] + e);
      p.fx[u] = (uint32_t)stream_load<true>(fidx + p.at0[u] + e);
      if (!UNIT) { p.x0[u] = tile_base[t] + (int64_t)a; p.xv[u] = stream_load<true>(tval + p.x0[u] + e); } else { p.x0[u] = 0; p.xv[u] = 1.0f; }
    }
  };
  auto add = [&](double2 c, float x, uint32_t fx) {
    const int g = (int)(fx - f0) & 15;
    const float xx = x * x;
    const double h = (double)x * c.x - (double)xx * oldv[wv][g];   // :310-317
    unsafeAtomicAdd(&acc[wv][g][0], h * c.y);
    unsafeAtomicAdd(&acc[wv][g][1], h * h);
  };
  Offs o_next, o_after; Run cur, nxt;
  load_offs(0, o_next);
  load_offs(NA, o_after);
  load_run(0, o_next, cur);
  for (int t0 = 0; t0 < n_tiles; t0 += NA) {
    load_run(t0 + NA, o_after, nxt);              // (past the end: clamped addresses, empty runs)
    load_offs(t0 + 2 * NA, o_next);
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      if ((uint32_t)lane < cur.rl[u]) add(cur.pv[u], cur.xv[u], cur.fx[u]);
      if (cur.rl[u] > 64u)                        // a long run: the rest in pieces of 64, straight from memory
        for (uint32_t i = 64u + lane; __any(i < cur.rl[u]); i += 64u)
          if (i < cur.rl[u]) add(src[cur.at0[u] + i], UNIT ? 1.0f : tval[cur.x0[u] + i], (uint32_t)fidx[cur.at0[u] + i]);
    }
    cur = nxt; { const Offs tmp = o_after; o_after = o_next; o_next = tmp; }
  }
  if (lane >= 16 || f0 + (uint32_t)lane >= cnt) return;
  const uint32_t fi = f0 + (uint32_t)lane;
  const uint32_t feat = feats[fi];
  double mean = acc[wv][lane][0], var = acc[wv][lane][1];
  const double old = oldv[wv][lane];
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  const double* __restrict__ znorm = dyn->znorm;
  mean -= old * var;                               // :318
  var = 1.0 / (lambda + alpha * var);              // :319
  mean = -var * (alpha * mean - mu * lambda);      // :320
  const double nv = bad_number_t(var) ? 0.0 : (znorm ? mean + sqrt(var) * znorm[feat] : mean);
  if (bad_number_t(nv)) { vstep[fi] = make_double2(old, nan("")); return; }  // CHECK_PARAM (:336): the old value stays; NaN tells the apply pass to leave the rows alone
  P[(size_t)feat * kp + f] = nv;
  vstep[fi] = make_double2(old, old - nv);
}

// blockIdx -> (tile, chunk) with a tile's workgroups consecutive in ONE XCD's share of the grid (as als_tile_sums_k)
template <bool UNIT, int R, bool QNEXT, typename IT>
__global__ __launch_bounds__(WG_THREADS) void als_order_apply_k(const double2* __restrict__ src, double2* __restrict__ dst, const IT* __restrict__ fidx, const uint32_t* __restrict__ perm,
                                                                const uint32_t* __restrict__ toff, size_t nf1, uint32_t lvl0, const int64_t* __restrict__ tile_base,
                                                                const float* __restrict__ tval, const uint32_t* __restrict__ trow, const double2* __restrict__ vstep,
                                                                const double* __restrict__ qnext, int tshift, int n_tiles, int B, int64_t n) {
  const int b = blockIdx.x;
  const int x8 = b & 7, qd = b >> 3;
  const int tile = (qd / B) * 8 + x8, chunk = qd % B;
  if (tile >= n_tiles) return;
  const int64_t base = (int64_t)tile << tshift;
  const int64_t rows = min(n, base + ((int64_t)1 << tshift)) - base;
  const int64_t ebase = (UNIT && !QNEXT) ? 0 : tile_base[tile] + (int64_t)toff[(size_t)tile * nf1 + lvl0];   // the tile's level block in trow / tval
  const int64_t i0 = (int64_t)chunk * (WG_THREADS * R) + threadIdx.x;
  double2 c[R], s[R]; uint32_t pm[R]; float xs[R]; double qn[R]; IT fx[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int64_t i = i0 + u * WG_THREADS, ic = i < rows ? i : rows - 1;
    c[u] = stream_load<true>(src + base + ic);
    fx[u] = stream_load<true>(fidx + base + ic);
    pm[u] = stream_load<true>(perm + base + ic);
    xs[u] = UNIT ? 1.0f : stream_load<true>(tval + ebase + ic);
    qn[u] = QNEXT ? qnext[base + stream_load<true>(trow + ebase + ic)] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < R; ++u) s[u] = vstep[fx[u]];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const int64_t i = i0 + u * WG_THREADS;
    if (i >= rows) continue;
    const bool skip = s[u].y != s[u].y;
    const float xx = xs[u] * xs[u];
    const double h = (double)xs[u] * c[u].x - (double)xx * s[u].x;
    const double q2 = QNEXT ? qn[u] : (skip ? c[u].x : c[u].x - (double)xs[u] * s[u].y);   // :341-350
    dst[base + pm[u]] = make_double2(q2, skip ? c[u].y : c[u].y - h * s[u].y);
  }
}

// row order -> level 0's order (q of the first factor from the factor-major table, e from the pairs) and back (e only: q of the last factor is dead)
__global__ void als_order_enter_k(const double2* __restrict__ qe, const double* __restrict__ Q0, const uint32_t* __restrict__ trow, const int64_t* __restrict__ tile_base,
                                  int64_t n, int tshift, double2* __restrict__ dst) {
  const int64_t at = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (at >= n) return;
  const int64_t t = at >> tshift, r0 = t << tshift;
  const int64_t row = r0 + trow[tile_base[t] + (at - r0)];   // (level 0's block is the first of the tile)
  dst[at] = make_double2(Q0[row], qe[row].y);
}
__global__ void als_order_exit_k(const double2* __restrict__ src, const uint32_t* __restrict__ trow, const int64_t* __restrict__ tile_base, int64_t n, int tshift,
                                 double2* __restrict__ qe) {
  const int64_t at = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (at >= n) return;
  const int64_t t = at >> tshift, r0 = t << tshift;
  qe[r0 + trow[tile_base[t] + (at - r0)]] = src[at];
}

bool als_order_ready(const fmx_matrix* m) {
  const AlsTiled* T = reinterpret_cast<const AlsTiled*>(m->als_tiled);
  return T && T->complete && !m->als_approx;
}

static int order_buffers(fmx_engine* e, int64_t n) {
  if (e->als_lo_rows >= n && e->als_lo[0] && e->als_lo[1]) return FMX_OK;
  FMX_HIP(hipStreamSynchronize(e->stream));
  (void)hipFree(e->als_lo[0]); (void)hipFree(e->als_lo[1]); e->als_lo[0] = e->als_lo[1] = nullptr; e->als_lo_rows = 0;
  if (hipMalloc(&e->als_lo[0], ((size_t)n + 1) * sizeof(double2)) != hipSuccess || hipMalloc(&e->als_lo[1], ((size_t)n + 1) * sizeof(double2)) != hipSuccess) {   // (+ 1: the spare pair the block form's masked-off lanes store to)
    (void)hipGetLastError();
    (void)hipFree(e->als_lo[0]); e->als_lo[0] = nullptr;
    return FMX_ERR_HIP;   // (the caller falls back to the three-pass form)
  }
  e->als_lo_rows = n;
  return FMX_OK;
}

int als_order_prepare(fmx_engine* e, fmx_matrix* m, const uint32_t** colP, const float** valP, const uint32_t** row0) {
  *colP = nullptr; *valP = nullptr;
  if (row0) *row0 = nullptr;
  e->als_q_level0 = 0;
  const AlsTiled* T = reinterpret_cast<const AlsTiled*>(m->als_tiled);
  if (!T || !T->complete || !T->blocks || m->als_approx) return FMX_OK;
  if (order_buffers(e, m->n) != FMX_OK) { (void)hipGetLastError(); return FMX_OK; }
  als_blocks_csr(T->blocks, colP, valP, row0);
  e->als_q_level0 = 1;
  return FMX_OK;
}

// enter: d_qe (row order; e current) + Q0 (q of the first factor, row order) -> buffer 0 in level 0's order.  *ok = false: no memory, use the other form.
int als_order_enter(fmx_engine* e, fmx_matrix* m, const double2* d_qe, const double* d_Q0, bool* ok) {
  *ok = false;
  const AlsTiled* T = reinterpret_cast<const AlsTiled*>(m->als_tiled);
  if (!T || !T->complete) return FMX_OK;
  if (T->blocks) {
    if (!e->als_q_level0) return FMX_OK;   // (the q table is in row order: als_order_prepare found no room for the pair buffers; the other forms take the sweep)
    FMX_TRY(als_blocks_enter(e, T->blocks, d_qe, d_Q0, reinterpret_cast<double2*>(e->als_lo[0])));
    e->als_lo_cur = 0;
    *ok = true;
    return FMX_OK;
  }
  double2* ws_partial; double* ws_vf; double2* vstep;
  if (order_buffers(e, m->n) != FMX_OK || tile_ws(e, T, &ws_partial, &ws_vf, &vstep) != FMX_OK) { (void)hipGetLastError(); return FMX_OK; }
  hipLaunchKernelGGL(als_order_enter_k, dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, e->stream, d_qe, d_Q0, (const uint32_t*)T->trow, (const int64_t*)T->tile_base, m->n, T->tshift,
                     reinterpret_cast<double2*>(e->als_lo[0]));
  e->als_lo_cur = 0;
  *ok = true;
  return FMX_OK;
}

// one level (slot s = the s-th non-empty level) of one factor: sums + step, then apply into the next level's order; d_qnext (row order) on the last level
// of a factor that has a successor
int als_order_level(fmx_engine* e, fmx_matrix* m, int s, const SweepDyn* dyn, const double* d_qnext, double* d_qprev_out) {
  const AlsTiled* T = reinterpret_cast<const AlsTiled*>(m->als_tiled);
  if (T->blocks) {
    FMX_TRY(als_blocks_level(e, T->blocks, s, reinterpret_cast<const double2*>(e->als_lo[e->als_lo_cur]), reinterpret_cast<double2*>(e->als_lo[1 - e->als_lo_cur]),
                             T->feats + T->lvl0[(size_t)s], dyn, d_qnext, d_qprev_out));
    e->als_lo_cur = 1 - e->als_lo_cur;
    return FMX_OK;
  }
  double2 *partial = nullptr, *vstep = nullptr; double* vf = nullptr;
  FMX_TRY(tile_ws(e, T, &partial, &vf, &vstep));
  const uint32_t lvl0 = T->lvl0[(size_t)s], cnt = T->cnt[(size_t)s];
  const size_t nf1 = (size_t)T->n_feats + 1;
  const double2* src = reinterpret_cast<const double2*>(e->als_lo[e->als_lo_cur]);
  double2* dst = reinterpret_cast<double2*>(e->als_lo[1 - e->als_lo_cur]);
  const dim3 blk(WG_THREADS);
  // features per workgroup: one round of resident workgroups (the kernel's occupancy x the device's CUs), every workgroup the same share of the level.
  // FMX_ALS_ORDER_FB pins it (tuning).  Which instantiation: 128 or 256 tiles per batch by the plan's tile count; 16-bit list offsets where no run of 63
  // lists inside a tile can pass 65 535 pairs.
  static const int fb_env = env_int("FMX_ALS_ORDER_FB", 0);
  static const int rr = env_int("FMX_ALS_ORDER_R", 1);
  // (per DEVICE: engines may live on several -- ADVICE r5; 64 devices is more than a node holds)
  constexpr int MAX_DEV = 64;
  const int dev_slot = (e->cfg.device >= 0 && e->cfg.device < MAX_DEV) ? e->cfg.device : 0;
  static int n_cus_dev[MAX_DEV] = {};
  int& n_cus = n_cus_dev[dev_slot];
  if (n_cus == 0) { hipDeviceProp_t pr{}; n_cus = (hipGetDeviceProperties(&pr, e->cfg.device) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
  const bool narrow = (uint64_t)T->max_list * 63u <= 65535u;
#define FMX_OSUMS(UNITv, TBv, OTv, MINWv)                                                                                                                   \
  do {                                                                                                                                                      \
    auto kern = als_order_walk_k<UNITv, 64, TBv, 1024, 1, OTv, MINWv>;                                                                                        \
    static int per_cu_dev[MAX_DEV] = {};                                                                                                                      \
    int& per_cu = per_cu_dev[dev_slot];                                                                                                                       \
    if (per_cu == 0) { int nbk = 0; per_cu = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, kern, WG_THREADS, 0) == hipSuccess && nbk > 0) ? nbk : 2; }  \
    const uint32_t slots = (uint32_t)(per_cu * n_cus);                                                                                                        \
    int fbv = fb_env > 0 ? fb_env : (int)((cnt + slots - 1) / slots);                                                                                         \
    fbv = fbv < 8 ? 8 : (fbv > 63 ? 63 : fbv);                                                                                                                \
    hipLaunchKernelGGL(kern, dim3((cnt + fbv - 1) / fbv), blk, 0, e->stream, (const uint32_t*)T->toff, nf1, lvl0, cnt, fbv, (const int64_t*)T->tile_base,    \
                       (const float*)T->tval, src, T->tshift, T->n_tiles, (const uint32_t*)(T->feats + lvl0), e->dV, e->kp64, dyn, vstep);                   \
  } while (0)
#define FMX_OSUMS_U(TBv, OTv, MINWv) do { if (T->unit) FMX_OSUMS(true, TBv, OTv, MINWv); else FMX_OSUMS(false, TBv, OTv, MINWv); } while (0)
  // (four workgroups per CU for the 128-tile / 16-bit form fit the LDS but cost register spills and bought nothing: 57.0 against 53.7 us, profiles/r05_order_ab11.txt)
  static const bool walk = [] { const char* v = getenv("FMX_ALS_ORDER_SUMS"); return v && v[0] == 'w'; }();
  if (!walk) {
    // the product form: one wave per 16 features, wave-owned LDS accumulators (als_order_sums_k)
    const dim3 sg((cnt + 63) / 64);
    const void* fxs = (const char*)T->fidx + (size_t)s * T->n * (T->lfi16 ? 2 : 4);
#define FMX_OATOM(UNITv, ITv) hipLaunchKernelGGL((als_order_sums_k<UNITv, 4, ITv>), sg, blk, 0, e->stream, (const uint32_t*)T->toff, nf1, lvl0, cnt, (const int64_t*)T->tile_base, \
                                                 (const float*)T->tval, (const ITv*)fxs, src, T->tshift, T->n_tiles, (const uint32_t*)(T->feats + lvl0), e->dV, e->kp64, dyn, vstep)
    if (T->unit) { if (T->lfi16) FMX_OATOM(true, uint16_t); else FMX_OATOM(true, uint32_t); }
    else { if (T->lfi16) FMX_OATOM(false, uint16_t); else FMX_OATOM(false, uint32_t); }
#undef FMX_OATOM
  } else if (T->n_tiles <= 128) { if (narrow) FMX_OSUMS_U(128, uint16_t, 3); else FMX_OSUMS_U(128, uint32_t, 3); }
  else { if (narrow) FMX_OSUMS_U(256, uint16_t, 3); else FMX_OSUMS_U(256, uint32_t, 1); }
#undef FMX_OSUMS_U
#undef FMX_OSUMS
  const int64_t tile_rows = (int64_t)1 << T->tshift;
  const void* fx = (const char*)T->fidx + (size_t)s * T->n * (T->lfi16 ? 2 : 4);
  const uint32_t* pm = T->perm + (size_t)s * T->n;
#define FMX_OAPPLY(UNITv, Rv, QNv)                                                                                                                            \
  do {                                                                                                                                                        \
    const int B = (int)((tile_rows + WG_THREADS * Rv - 1) / (WG_THREADS * Rv));                                                                                 \
    const dim3 ag((unsigned)(((T->n_tiles + 7) / 8) * 8 * B));                                                                                                  \
    if (T->lfi16) hipLaunchKernelGGL((als_order_apply_k<UNITv, Rv, QNv, uint16_t>), ag, blk, 0, e->stream, src, dst, (const uint16_t*)fx, pm, (const uint32_t*)T->toff, nf1, lvl0, \
                                     (const int64_t*)T->tile_base, (const float*)T->tval, (const uint32_t*)T->trow, (const double2*)vstep, d_qnext, T->tshift, T->n_tiles, B, T->n); \
    else hipLaunchKernelGGL((als_order_apply_k<UNITv, Rv, QNv, uint32_t>), ag, blk, 0, e->stream, src, dst, (const uint32_t*)fx, pm, (const uint32_t*)T->toff, nf1, lvl0,          \
                            (const int64_t*)T->tile_base, (const float*)T->tval, (const uint32_t*)T->trow, (const double2*)vstep, d_qnext, T->tshift, T->n_tiles, B, T->n);          \
  } while (0)
#define FMX_OAPPLY_Q(UNITv, Rv) do { if (d_qnext) FMX_OAPPLY(UNITv, Rv, true); else FMX_OAPPLY(UNITv, Rv, false); } while (0)
#define FMX_OAPPLY_R(Rv) do { if (T->unit) FMX_OAPPLY_Q(true, Rv); else FMX_OAPPLY_Q(false, Rv); } while (0)
  switch (rr) {
    case 2: FMX_OAPPLY_R(2); break;
    case 4: FMX_OAPPLY_R(4); break;
    default: FMX_OAPPLY_R(1); break;
  }
#undef FMX_OAPPLY_R
#undef FMX_OAPPLY_Q
#undef FMX_OAPPLY
  e->als_lo_cur = 1 - e->als_lo_cur;
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int als_order_form(const fmx_matrix* m) {
  const AlsTiled* T = reinterpret_cast<const AlsTiled*>(m->als_tiled);
  return (T && T->complete && !m->als_approx) ? (T->blocks ? 2 : 1) : 0;
}

int als_order_w_sweep(fmx_engine* e, fmx_matrix* m, double2* d_qe, const SweepDyn* dyn, bool* done) {
  *done = false;
  const AlsTiled* T = reinterpret_cast<const AlsTiled*>(m->als_tiled);
  if (!T || !T->complete || !T->blocks || m->als_approx) return FMX_OK;
  static const bool on = env_int("FMX_ALS_BLOCK_W", 1) != 0;   // (A/B: the three-pass form for the w sweep)
  if (!on) return FMX_OK;
  if (order_buffers(e, m->n) != FMX_OK) { (void)hipGetLastError(); return FMX_OK; }
  FMX_TRY(als_blocks_enter(e, T->blocks, d_qe, nullptr, reinterpret_cast<double2*>(e->als_lo[0])));
  e->als_lo_cur = 0;
  for (int s = 0; s < T->n_slots; ++s) {
    FMX_TRY(als_blocks_level(e, T->blocks, s, reinterpret_cast<const double2*>(e->als_lo[e->als_lo_cur]), reinterpret_cast<double2*>(e->als_lo[1 - e->als_lo_cur]),
                             T->feats + T->lvl0[(size_t)s], dyn, nullptr, nullptr, true));
    e->als_lo_cur = 1 - e->als_lo_cur;
  }
  FMX_TRY(als_blocks_exit(e, T->blocks, reinterpret_cast<const double2*>(e->als_lo[e->als_lo_cur]), d_qe, nullptr, true));
  *done = true;
  return FMX_OK;
}

int als_order_levels(const fmx_matrix* m) {
  const AlsTiled* T = reinterpret_cast<const AlsTiled*>(m->als_tiled);
  return T ? T->n_slots : 0;
}

// exit: the current buffer (level 0's order: the last apply of the last factor wrote there) back to d_qe in row order
uint64_t als_order_plan_uid(const fmx_matrix* m) {
  const AlsTiled* T = reinterpret_cast<const AlsTiled*>(m->als_tiled);
  return T ? als_blocks_uid(T->blocks) : 0;
}

int als_order_exit(fmx_engine* e, fmx_matrix* m, double2* d_qe, double* d_qlast_out) {
  const AlsTiled* T = reinterpret_cast<const AlsTiled*>(m->als_tiled);
  if (T->blocks) return als_blocks_exit(e, T->blocks, reinterpret_cast<const double2*>(e->als_lo[e->als_lo_cur]), d_qe, d_qlast_out);
  hipLaunchKernelGGL(als_order_exit_k, dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, e->stream, reinterpret_cast<const double2*>(e->als_lo[e->als_lo_cur]), (const uint32_t*)T->trow,
                     (const int64_t*)T->tile_base, m->n, T->tshift, d_qe);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

}  // namespace fmx
