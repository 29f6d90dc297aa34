// The BLOCK form of the level-order V sweep (MCMC_ALS_Learner::update_v, solver/MCMC_ALS_Learner.h:283-351): ONE kernel per level of one factor.
//
// The tile form (fm_als_tiled.hip, als_order_*) keeps the (q, e) pairs tile-major and needs two kernels per level with a chip-wide wait between them: the sums of a
// feature come from every tile, so the pairs are read twice (sums, then corrections), and the permuting scatter of the corrections is 16 bytes at a time.
// Here the level's array is FEATURE-BLOCK-MAJOR: the features of a level are cut into blocks of consecutive features holding at most R rows between them
// (R = 8 192 pairs = 128 KB: one workgroup's LDS), and block B's region of the array holds exactly the rows of its features.  Everything a coordinate step of a
// feature needs is then inside ONE workgroup:
//   1. the block's region streams in (contiguous), every pair lands in LDS at its feature-sorted slot (u16 per pair: `perm_in`);
//   2. lane groups sum their lists out of LDS (Sum h e, Sum h^2, :310-317), take the coordinate steps (:318-336) and correct their lists in place (:341-350): the
//      sums never leave the workgroup, the pairs are read from memory ONCE per level, and nothing waits for another workgroup;
//   3. the corrected pairs leave for the NEXT level's array, in which block B' keeps what it receives from block B as one contiguous RUN: the workgroup reads
//      its pairs back out of LDS in destination order (u16 per pair: `gsrc`) and stores them at consecutive addresses run by run (u32 per pair: `dest`).
// The runs are short (R^2 / n pairs: 6.7 at configs[4]) and unaligned; consecutive blocks run at the same time on the same XCD (blockIdx -> block below), so
// the neighbouring runs of a line meet in that L2.  Measured before building (profiles/probes/block_level_probe.hip, profiles/r05_block_probe.txt): 89 us per
// level at configs[4]'s shape against 59 + 96 us of the tile form's two kernels; a plain copy of the pairs takes 53 us.
//
// Arithmetic per entry as in every other form; the sums of a list are associated in (lane of the group, then butterfly) order over the list's rows ascending:
// 1e-10 against the oracle and the other forms, bitwise run to run (tests/test_gpu_configs4.py).
#include <cstring>  // rocprim's texture_cache_iterator.hpp uses memset without including it
#include <memory>

#include <rocprim/rocprim.hpp>

#include "fmx_internal.h"

namespace fmx {

namespace {

constexpr int BLK_THREADS = 1024;
constexpr int BLK_MAXF = 1024;       // most features in one block (their old values, ids and list bounds live in LDS)
constexpr int BLK_R_UNIT = 8192;     // pairs per block: 128 KB of LDS
constexpr int BLK_R_VAL = 6144;      // with a value per entry next to the pair: 96 + 24 KB

struct AlsBlocks {
  int64_t n = 0;
  int L = 0, R = 0, unit = 0;
  uint64_t uid = 0;               // one per plan built (a q table carried from sweep to sweep belongs to ONE plan)
  std::vector<uint32_t> nblk;     // per level
  std::vector<size_t> boff;       // per level: first entry of the level in bbase / bfeat (nblk + 1 entries each)
  std::vector<size_t> foff;       // per level: first entry of the level in loff (cnt + 1 entries each)
  std::vector<int> lg;            // per level: lanes per list
  uint32_t last_cnt = 0;          // features of the last level
  uint32_t* bbase = nullptr;      // first position of every block in its level's array (= rows of the features before it)
  uint32_t* bfeat = nullptr;      // first feature (index inside the level) of every block
  uint32_t* loff = nullptr;       // per level [cnt + 1]: first position of every feature's list in slot order (feature-major, rows ascending)
  uint16_t* perm_in = nullptr;    // [L][n] LDS slot of the pair at array position i (inside its block)
  uint16_t* gsrc = nullptr;       // [L][n] per block, in destination order: the LDS slot ...
  uint32_t* dest = nullptr;       // [L][n] ... and the position in the next level's array
  float* xs = nullptr;            // [L][n] entry values in slot order (null: every value is 1.0f)
  uint32_t* row0 = nullptr;       // [n] row at position i of level 0's array
  uint32_t* colP = nullptr;       // [nnz] the CSR's columns with the rows in level 0's array order (every row holds L entries): the forward pass that builds q for all
  float* valP = nullptr;          // [nnz] factors runs on this copy and leaves q in level 0's order -- a factor's first level takes its q in as a stream (null: unit values)
  ~AlsBlocks() {
    (void)hipFree(bbase); (void)hipFree(bfeat); (void)hipFree(loff); (void)hipFree(perm_in); (void)hipFree(gsrc); (void)hipFree(dest); (void)hipFree(xs);
    (void)hipFree(row0); (void)hipFree(colP); (void)hipFree(valP);
  }
};

int env_int_b(const char* name, int dflt) {
  const char* s = getenv(name);
  return s && *s ? atoi(s) : dflt;
}

// ---- plan kernels ----------------------------------------------------------------------------------------------------------------------------
// one wave per feature of the level: block and slot of every row of its list, the values in slot order
__global__ __launch_bounds__(256) void blocks_rows_k(const uint32_t* __restrict__ feats, uint32_t cnt, const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow,
                                                     const float* __restrict__ cval, const uint32_t* __restrict__ loff, const uint16_t* __restrict__ blk_of_feat,
                                                     const uint32_t* __restrict__ bbase, uint16_t* __restrict__ blkrow, uint16_t* __restrict__ slotrow, float* __restrict__ xs) {
  const int lane = threadIdx.x & 63;
  const uint32_t fi = (blockIdx.x * 256u + threadIdx.x) >> 6;
  if (fi >= cnt) return;
  const uint32_t j = feats[fi];
  const int64_t b = col_ptr[j], e = col_ptr[j + 1];
  const uint16_t blk = blk_of_feat[fi];
  const uint32_t l0 = loff[fi], bb = bbase[blk];
  for (int64_t t = b + lane; t < e; t += 64) {
    const uint32_t r = crow[t];
    blkrow[r] = blk;
    slotrow[r] = (uint16_t)(l0 + (uint32_t)(t - b) - bb);
    if (xs) xs[l0 + (uint32_t)(t - b)] = cval[t];
  }
}
__global__ void blocks_keys_k(const uint16_t* __restrict__ blk_s, const uint16_t* __restrict__ blk_prev, int64_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  keys[r] = ((uint32_t)blk_s[r] << 16) | (uint32_t)blk_prev[r];
  vals[r] = (uint32_t)r;
}
// perm_in[i] = slot of the row at position i
__global__ void blocks_place_k(const uint32_t* __restrict__ rowat, const uint16_t* __restrict__ slotrow, int64_t n, uint16_t* __restrict__ perm_in) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  perm_in[i] = slotrow[rowat[i]];
}
// keys of the destination order: the block (this level) of the row at position i of the NEXT level's array
__global__ void blocks_keys2_k(const uint32_t* __restrict__ rowat_next, const uint16_t* __restrict__ blk_s, int64_t n, uint16_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = blk_s[rowat_next[i]];
  vals[i] = (uint32_t)i;
}
__global__ void blocks_dest_k(const uint32_t* __restrict__ sorted_pos, const uint32_t* __restrict__ rowat_next, const uint16_t* __restrict__ slotrow, int64_t n,
                              uint32_t* __restrict__ dest, uint16_t* __restrict__ gsrc) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t d = sorted_pos[k], r = rowat_next[d];
  dest[k] = d;
  gsrc[k] = slotrow[r];
}
// the CSR with its rows in level 0's array order: L lanes-worth of entries per row (every row holds exactly L)
__global__ void blocks_permute_csr_k(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const float* __restrict__ val, const uint32_t* __restrict__ row0, int64_t n,
                                     int L, uint32_t* __restrict__ colP, float* __restrict__ valP) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n * L) return;
  const int64_t i = g / L; const int t = (int)(g % L);
  const int64_t src = row_ptr[row0[i]] + t;
  colP[g] = col[src];
  if (valP) valP[g] = val[src];
}

// ---- the sweep -------------------------------------------------------------------------------------------------------------------------------
typedef double blk_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_pair(const double2* p) { const blk_v2d v = __builtin_nontemporal_load(reinterpret_cast<const blk_v2d*>(p)); return make_double2(v.x, v.y); }
template <typename T> __device__ __forceinline__ T nt_ld(const T* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ bool bad_number_b(double x) { return isnan(x) || isinf(x); }

// one entry of a list, the coordinate step and the correction: the V sweep's (:310-350) and, with W, the w sweep's (update_w, :208-256) -- the arithmetic of
// als_level_k / als_tile_*_k, operation for operation
template <bool W>
__device__ __forceinline__ void blk_acc(const double2 c, const float x, const double old, double& mean, double& var) {
  if (W) { const double xd = (double)x; mean += c.y * xd - old * xd * xd; var += xd * xd; }                                            // :216-219
  else { const float xx = x * x; const double h = (double)x * c.x - (double)xx * old; mean += h * c.y; var += h * h; }                 // :310-317
}
template <bool W>
__device__ __forceinline__ double blk_step(double mean, double var, const double old, const double alpha, const double lambda, const double mu, const bool gibbs, const double z) {
  if (W) {
    var = 1.0 / (lambda + alpha * var);
    mean = -var * (alpha * mean - mu * lambda);
    return bad_number_b(var) ? 0.0 : (gibbs ? mean + var * z : mean);             // (the variance as Rf_rnorm's sd: :239, kept)
  }
  mean -= old * var;                               // :318
  var = 1.0 / (lambda + alpha * var);              // :319
  mean = -var * (alpha * mean - mu * lambda);      // :320
  return bad_number_b(var) ? 0.0 : (gibbs ? mean + sqrt(var) * z : mean);
}
template <bool W>
__device__ __forceinline__ double2 blk_fix(const double2 c, const float x, const double old, const double diff) {
  if (W) return make_double2(c.x, c.y - (double)x * diff);                                                                             // :246-252
  const float xx = x * x;
  const double h = (double)x * c.x - (double)xx * old;
  return make_double2(c.x - (double)x * diff, c.y - h * diff);                                                                         // :341-350
}

// blockIdx -> block: consecutive blocks share an XCD (workgroups are dealt round-robin over the eight XCDs) and run there at about the same time, so the runs
// that neighbouring blocks write into one line of the next level's array meet in that XCD's L2 (73 against 89 us per level in the probe).  Placement is for
// speed only.
template <bool W, bool UNIT, int R, int LG, bool QIN>
__global__ __launch_bounds__(BLK_THREADS) void als_block_level_k(const double2* __restrict__ src, double2* __restrict__ dst, const uint32_t* __restrict__ bbase,
                                                                 const uint32_t* __restrict__ bfeat, int nb, const uint32_t* __restrict__ loff, const uint32_t* __restrict__ feats,
                                                                 const uint16_t* __restrict__ perm_in, const uint16_t* __restrict__ gsrc, const uint32_t* __restrict__ dest,
                                                                 const float* __restrict__ xs, double* __restrict__ P, int kp, const SweepDyn* __restrict__ dyn,
                                                                 const double* __restrict__ qin, double* __restrict__ qprev_out, uint32_t n) {
  constexpr int NT = BLK_THREADS, PT = R / NT, NG = NT / LG;
  static_assert(R % NT == 0, "whole pairs per thread");
  __shared__ double2 lp[R];
  __shared__ float lx[UNIT ? 1 : R];
  __shared__ double oldv[BLK_MAXF];
  __shared__ double zn[BLK_MAXF];   // the features' standard normals (Gibbs): fetched with the old values, so that no load inside the list loop waits for the streams
  __shared__ uint32_t lfeat[BLK_MAXF];
  __shared__ uint16_t lo[BLK_MAXF + 2];
  const int per = (nb + 7) >> 3;
  const int B = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (B >= nb) return;
  const uint32_t b0 = bbase[B], rows = bbase[B + 1] - b0;
  const uint32_t f0 = bfeat[B], nf = bfeat[B + 1] - f0;
  const int f = dyn->f;
  double2 v[PT]; uint16_t pa[PT], gs[PT]; uint32_t de[PT]; float xv[PT]; double qn[PT];
  // every load goes out before anything is used; positions past the block's end are clamped onto its last pair (nothing of them is stored)
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const uint32_t i = threadIdx.x + u * NT, ic = min(b0 + min(i, rows ? rows - 1 : 0u), n - 1);   // (an empty block: features without rows still take their step)
    v[u] = nt_pair(src + ic);
    pa[u] = nt_ld(perm_in + ic);
    xv[u] = UNIT ? 1.0f : nt_ld(xs + ic);
    qn[u] = QIN ? nt_ld(qin + ic) : 0.0;   // a factor's FIRST level: this factor's q, in the array's own order (the pair still carries the previous factor's)
  }
  if (threadIdx.x < nf) {
    const uint32_t ft = feats[f0 + threadIdx.x];
    lfeat[threadIdx.x] = ft;
    oldv[threadIdx.x] = P[W ? (size_t)ft : (size_t)ft * kp + f];
    zn[threadIdx.x] = dyn->znorm ? dyn->znorm[ft] : 0.0;
  }
  if (threadIdx.x <= nf) lo[threadIdx.x] = (uint16_t)(loff[f0 + threadIdx.x] - b0);
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const uint32_t i = threadIdx.x + u * NT, ic = min(b0 + min(i, rows ? rows - 1 : 0u), n - 1);
    gs[u] = nt_ld(gsrc + ic);
    de[u] = nt_ld(dest + ic);
  }
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const uint32_t i = threadIdx.x + u * NT;
    if (i < rows) {
      lp[pa[u]] = QIN ? make_double2(qn[u], v[u].y) : v[u]; if (!UNIT) lx[i] = xv[u];
      if (QIN && qprev_out) qprev_out[b0 + i] = v[u].x;   // (q carried from sweep to sweep: the pair still holds the PREVIOUS factor's final q, in level 0's order)
    }
  }
  __syncthreads();
  {
    const int g = threadIdx.x / LG, l = threadIdx.x % LG;
    const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
    const bool gibbs = dyn->znorm != nullptr;
    for (uint32_t fi = g; fi < nf; fi += NG) {
      const uint32_t a = lo[fi], b = lo[fi + 1];
      const double old = oldv[fi];
      double mean = 0.0, var = 0.0;
      for (uint32_t t0 = a + l; t0 < b; t0 += 4 * LG) {
        double2 c[4]; float x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t t = min(t0 + u * LG, b - 1); c[u] = lp[t]; x[u] = UNIT ? 1.0f : lx[t]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (t0 + u * LG < b) blk_acc<W>(c[u], x[u], old, mean, var);
      }
#pragma unroll
      for (int o = 1; o < LG; o <<= 1) { mean += __shfl_xor(mean, o); var += __shfl_xor(var, o); }   // (a + b == b + a: every lane of the group holds the same bits)
      const uint32_t feat = lfeat[fi];
      const double nv = blk_step<W>(mean, var, old, alpha, lambda, mu, gibbs, zn[fi]);
      if (bad_number_b(nv)) continue;                  // CHECK_PARAM (:336): the old value stays and the rows keep their pairs
      if (l == 0) P[W ? (size_t)feat : (size_t)feat * kp + f] = nv;
      const double diff = old - nv;
      for (uint32_t t0 = a + l; t0 < b; t0 += 4 * LG) {
        double2 c[4]; float x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t t = min(t0 + u * LG, b - 1); c[u] = lp[t]; x[u] = UNIT ? 1.0f : lx[t]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (t0 + u * LG < b) lp[t0 + u * LG] = blk_fix<W>(c[u], x[u], old, diff);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PT; ++u) {
    const uint32_t i = threadIdx.x + u * NT;
    if (i < rows) dst[de[u]] = lp[gs[u]];
  }
}

// The same level as a PIPELINE over the blocks of one CU: one workgroup per CU stays resident and takes the blocks B, B + slots, B + 2 slots, ... of its XCD's
// share (at any moment an XCD still works on consecutive blocks).  While block k is summed, stepped and corrected in LDS, the pairs of block k + 1 are on their
// way into registers; while they are written to LDS, the stores of block k are still draining.  Three things make that hold on gfx950 (each read off the ISA):
//   * loads and stores share ONE counter (vmcnt) and return out of order with respect to each other, so while both kinds are outstanding the compiler can only
//     wait for everything: the prefetch is therefore waited for BEFORE the block's stores are issued (it had the whole LDS phase to land), and nothing loaded
//     is first used after them -- the feature ids of a block arrive TWO blocks ahead so that its old values (loads that depend on them) go out with the prefetch;
//   * __syncthreads() fences global memory too (s_waitcnt vmcnt(0) before s_barrier: the stores would have to be acknowledged before the next block may enter
//     the LDS); the barriers here order LDS only (lds_barrier);
//   * no memory operation sits under a condition -- lanes past the block's end store to a spare slot (lp[R], dst[n]) -- and no store sits inside the list loop.
// Same bits as als_block_level_k: the same slots, the same lane groups, the same order.
// -DFMX_BLK_KO=bits (profiles/probes/block_knockouts.sh, never the product build): knock-outs of the pipelined kernel, fixed at compile time -- 1: no list
// loop; 2: the pairs leave for the block's own region (a linear store); 4: the LDS is written and read linearly (no slots); 8: no index loads (16, 32: A/B
// builds with correct results: non-temporal pair stores; default-policy pair loads).  Wrong results,
// the time of what is left (profiles/r05_block_knockouts.txt).
#ifndef FMX_BLK_KO
#define FMX_BLK_KO 0
#endif
#define BLK_KO(bit) (((FMX_BLK_KO) & (bit)) != 0)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool W, bool UNIT, int R, int LG>
__global__ __launch_bounds__(BLK_THREADS) void als_block_level_pipe_k(const double2* __restrict__ src, double2* __restrict__ dst, const uint32_t* __restrict__ bbase,
                                                                      const uint32_t* __restrict__ bfeat, int nb, const uint32_t* __restrict__ loff, const uint32_t* __restrict__ feats,
                                                                      uint32_t cnt, const uint16_t* __restrict__ perm_in, const uint16_t* __restrict__ gsrc,
                                                                      const uint32_t* __restrict__ dest, const float* __restrict__ xs, double* __restrict__ P, int kp,
                                                                      const SweepDyn* __restrict__ dyn, uint32_t n) {
  constexpr int NT = BLK_THREADS, PT = R / NT, NG = NT / LG;
  __shared__ double2 lp[R + 1];
  __shared__ float lx[UNIT ? 1 : R + 1];
  __shared__ double oldv[BLK_MAXF];
  __shared__ double newv[BLK_MAXF];   // the stepped coordinates: stored to P by one thread per feature after the list loop
  __shared__ double zn[BLK_MAXF];
  __shared__ uint32_t lfeat[BLK_MAXF];
  __shared__ uint16_t lo[BLK_MAXF + 2];
  const int per = (nb + 7) >> 3, slots = (int)(gridDim.x >> 3);
  const int xcd = (int)(blockIdx.x & 7);
  int B = xcd * per + (int)(blockIdx.x >> 3);
  const int Bend = min((xcd + 1) * per, nb);
  if (B >= Bend) return;
  const int f = dyn->f;
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  const bool gibbs = dyn->znorm != nullptr;
  const size_t pmul = W ? 1 : (size_t)kp, pofs = W ? 0 : (size_t)f;   // coordinate of feature j: P[j * pmul + pofs] (w[j], or V[j][f])
  const double* __restrict__ zsrc = gibbs ? dyn->znorm : P;   // (every load unconditional: the ALS form reads a word it does not use)
  const size_t zmul = gibbs ? 1 : pmul;
  const uint32_t tid = threadIdx.x;
  struct Geo { uint32_t b0, rows, f0, nf; };
  auto geo = [&](int Bx) {   // a block past the workgroup's last: no rows, no features (its loads re-read one word of the last block)
    const bool ok = Bx < Bend;
    const int Bc = ok ? Bx : Bend - 1;
    Geo g; g.b0 = bbase[Bc]; g.rows = ok ? bbase[Bc + 1] - g.b0 : 0u; g.f0 = bfeat[Bc]; g.nf = ok ? bfeat[Bc + 1] - g.f0 : 0u;
    return g;
  };
  auto load_feat = [&](const Geo& g) { return feats[min(g.f0 + min(tid, g.nf ? g.nf - 1 : 0u), cnt - 1)]; };
  auto load_pairs = [&](const Geo& g, double2 (&v_)[PT], uint16_t (&pa_)[PT], float (&xv_)[PT], uint32_t& lov_) {
    lov_ = loff[min(g.f0 + min(tid, g.nf), cnt)] - g.b0;
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      const uint32_t i = tid + u * NT, ic = min(g.b0 + min(i, g.rows ? g.rows - 1 : 0u), n - 1);
      v_[u] = BLK_KO(32) ? src[ic] : nt_pair(src + ic);
      pa_[u] = BLK_KO(8) ? (uint16_t)i : nt_ld(perm_in + ic);
      xv_[u] = UNIT ? 1.0f : nt_ld(xs + ic);
    }
  };
  Geo cur = geo(B), nxt = geo(B + slots);
  double2 v[PT]; uint16_t pa[PT]; float xv[PT]; uint32_t ft, lov, ftn; double old, zv;
  ft = load_feat(cur);
  ftn = load_feat(nxt);
  load_pairs(cur, v, pa, xv, lov);
  old = P[(size_t)ft * pmul + pofs];
  zv = zsrc[(size_t)ft * zmul];
  for (;;) {
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      const uint32_t i = tid + u * NT;
      lp[i < cur.rows ? (BLK_KO(4) ? i : (uint32_t)pa[u]) : (uint32_t)R] = v[u];
      if (!UNIT) lx[i < cur.rows ? i : (uint32_t)R] = xv[u];
    }
    { const uint32_t ts = tid < cur.nf ? tid : (uint32_t)(BLK_MAXF - 1); lfeat[ts] = ft; oldv[ts] = old; newv[ts] = old; zn[ts] = zv; }   // (nf < BLK_MAXF: the last slot is spare)
    lo[tid <= cur.nf ? tid : (uint32_t)(BLK_MAXF + 1)] = (uint16_t)lov;
    lds_barrier();
    // this block's way out, the next block's way in, the block after's feature ids: all in flight behind the LDS phase below
    uint16_t gs[PT]; uint32_t de[PT];
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      const uint32_t i = tid + u * NT, ic = min(cur.b0 + min(i, cur.rows ? cur.rows - 1 : 0u), n - 1);
      gs[u] = BLK_KO(8) ? (uint16_t)min(i, cur.rows ? cur.rows - 1 : 0u) : nt_ld(gsrc + ic);
      de[u] = BLK_KO(8) ? ic : nt_ld(dest + ic);
    }
    const bool more = B + slots < Bend;   // (uniform)
    const Geo aft = geo(B + 2 * slots);
    double2 vn[PT]; uint16_t pan[PT]; float xvn[PT]; uint32_t lovn;
    load_pairs(nxt, vn, pan, xvn, lovn);
    const double oldn = P[(size_t)ftn * pmul + pofs];   // (the next block's features: none of them is stepped by this block)
    const double zvn = zsrc[(size_t)ftn * zmul];
    const uint32_t ftnn = load_feat(aft);
    if (!BLK_KO(1)) {
      const int g = tid / LG, l = tid % LG;
      for (uint32_t fi = g; fi < cur.nf; fi += NG) {
        const uint32_t a = lo[fi], b = lo[fi + 1];
        const double oldf = oldv[fi];
        double mean = 0.0, var = 0.0;
        for (uint32_t t0 = a + l; t0 < b; t0 += 4 * LG) {
          double2 c[4]; float x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const uint32_t t = min(t0 + u * LG, b - 1); c[u] = lp[t]; x[u] = UNIT ? 1.0f : lx[t]; }
#pragma unroll
          for (int u = 0; u < 4; ++u) if (t0 + u * LG < b) blk_acc<W>(c[u], x[u], oldf, mean, var);
        }
#pragma unroll
        for (int o = 1; o < LG; o <<= 1) { mean += __shfl_xor(mean, o); var += __shfl_xor(var, o); }
        const double nv = blk_step<W>(mean, var, oldf, alpha, lambda, mu, gibbs, zn[fi]);
        if (bad_number_b(nv)) continue;                  // CHECK_PARAM (:336): newv keeps the old value
        if (l == 0) newv[fi] = nv;
        const double diff = oldf - nv;
        for (uint32_t t0 = a + l; t0 < b; t0 += 4 * LG) {
          double2 c[4]; float x[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const uint32_t t = min(t0 + u * LG, b - 1); c[u] = lp[t]; x[u] = UNIT ? 1.0f : lx[t]; }
#pragma unroll
          for (int u = 0; u < 4; ++u) if (t0 + u * LG < b) lp[t0 + u * LG] = blk_fix<W>(c[u], x[u], oldf, diff);
        }
      }
    }
    lds_barrier();
    { const uint32_t ts = min(tid, cur.nf - 1); P[(size_t)lfeat[ts] * pmul + pofs] = newv[ts]; }   // (threads past the last feature repeat its store: same address, same value)
#pragma unroll
    for (int u = 0; u < PT; ++u) {
      const uint32_t i = tid + u * NT;
      const double2 c = lp[BLK_KO(4) ? min(i, (uint32_t)R) : (uint32_t)gs[u]];
      double2* const at = dst + (i < cur.rows ? (BLK_KO(2) ? cur.b0 + i : de[u]) : n);
      if (BLK_KO(16)) { blk_v2d cv; cv.x = c.x; cv.y = c.y; __builtin_nontemporal_store(cv, reinterpret_cast<blk_v2d*>(at)); }   // (16: non-temporal stores -- an A/B, not a knock-out)
      else *at = c;
    }
    if (!more) break;
    lds_barrier();   // (the pairs have left the LDS -- not yet the CU -- before the next block lands in it)
    B += slots; cur = nxt; nxt = aft; ft = ftn; ftn = ftnn; lov = lovn; old = oldn; zv = zvn;
#pragma unroll
    for (int u = 0; u < PT; ++u) { v[u] = vn[u]; pa[u] = pan[u]; xv[u] = xvn[u]; }
  }
}

// row order -> level 0's array (q of the first factor from the factor-major table, e from the pairs) and back
__global__ void als_block_enter_k(const double2* __restrict__ qe, const double* __restrict__ Q0, const uint32_t* __restrict__ row0, int64_t n, double2* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = row0[i];
  dst[i] = make_double2(Q0 ? Q0[i] : 0.0, qe[r].y);   // (Q: built on the permuted CSR, already in this order; the w sweep has no q)
}
__global__ void als_block_exit_k(const double2* __restrict__ src, const uint32_t* __restrict__ row0, int64_t n, double2* __restrict__ qe, double* __restrict__ qlast_out, int e_only) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double2 c = src[i];
  if (e_only) qe[row0[i]].y = c.y;   // (the w sweep: the rows' q is none of its business)
  else qe[row0[i]] = c;
  if (qlast_out) qlast_out[i] = c.x;   // (q carried from sweep to sweep: the last factor's final q)
}
// a 64-bit fingerprint of the fp64 V table: sum over i of mix(bits of element i XOR i x golden ratio) modulo 2^64, mix = splitmix64's finaliser -- integer adds commute, so
// any reduction order gives the same word; every element is scrambled before it is summed (the round-5 form summed bits x (2 i + 1): linear, so sign flips of an even number
// of entries, V -> -V included, cancelled -- ADVICE r5).  A backstop only: every host-side writer of V drops the carried table explicitly (als_q_invalidate).
__device__ __forceinline__ unsigned long long vhash_mix(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void als_vhash_k(const double* __restrict__ V, size_t count, unsigned long long* __restrict__ out) {
  unsigned long long h = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
    h += vhash_mix((unsigned long long)__double_as_longlong(V[i]) ^ ((unsigned long long)i * 0x9E3779B97F4A7C15ull));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) h += ((unsigned long long)(unsigned)__shfl_xor((int)(h >> 32), o) << 32) + (unsigned)__shfl_xor((int)(h & 0xFFFFFFFFull), o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

}  // namespace

void als_blocks_free(void* b) { delete reinterpret_cast<AlsBlocks*>(b); }

// *out stays null where the form does not apply (a list longer than a block, too many blocks, no memory): never an error
int als_blocks_build(fmx_matrix* m, const AlsBlocksIn& in, void** out, hipStream_t stream) {
  *out = nullptr;
  const int64_t n = in.n;
  const int L = in.n_slots;
  if (L < 1 || n < 1 || n >= (1LL << 32)) return FMX_OK;
  std::unique_ptr<AlsBlocks> Bk(new AlsBlocks());
  Bk->n = n; Bk->L = L; Bk->unit = in.unit;
  int R = in.unit ? BLK_R_UNIT : BLK_R_VAL;
  const int r_env = env_int_b("FMX_ALS_BLOCK_ROWS", 0);   // (tests: several blocks on a small matrix; at most the kernel's capacity)
  if (r_env > 0 && r_env < R) R = r_env;
  Bk->R = R;
  const int maxf = BLK_MAXF - 1;   // (thread nf of the workgroup loads the end of the last list: nf < BLK_THREADS)
  // list lengths of every feature (the CSC's column pointers), blocks by greedy filling
  std::vector<int64_t> cp((size_t)m->p + 1);
  FMX_HIP(hipMemcpy(cp.data(), m->col_ptr, cp.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
  size_t n_feats = 0;
  for (int s = 0; s < L; ++s) n_feats += in.cnt[s];
  std::vector<uint32_t> h_loff(n_feats + (size_t)L), h_bbase, h_bfeat;
  std::vector<uint16_t> h_blk_of_feat(n_feats);
  Bk->nblk.assign((size_t)L, 0u); Bk->boff.assign((size_t)L, 0); Bk->foff.assign((size_t)L, 0); Bk->lg.assign((size_t)L, 64);
  size_t fo = 0;
  for (int s = 0; s < L; ++s) {
    const uint32_t cnt = in.cnt[s];
    const uint32_t* fs = in.h_feats + in.lvl0[s];
    Bk->foff[(size_t)s] = fo;
    Bk->boff[(size_t)s] = h_bbase.size();
    uint64_t at = 0; uint32_t in_block = 0, feats_in_block = 0, nb = 0;
    h_bbase.push_back(0u); h_bfeat.push_back(0u);
    for (uint32_t i = 0; i < cnt; ++i) {
      const int64_t len = cp[(size_t)fs[i] + 1] - cp[(size_t)fs[i]];
      if (len > R) return FMX_OK;                       // a list that no block holds: the tile form keeps this matrix
      if (in_block + len > (uint32_t)R || feats_in_block == (uint32_t)maxf) {
        h_bbase.push_back((uint32_t)at); h_bfeat.push_back(i); ++nb; in_block = 0; feats_in_block = 0;
      }
      h_loff[fo + i] = (uint32_t)at;
      if (nb > 65534u) return FMX_OK;
      h_blk_of_feat[fo - (size_t)s + i] = (uint16_t)nb;
      at += (uint64_t)len; in_block += (uint32_t)len; ++feats_in_block;
    }
    if (at != (uint64_t)n) return FMX_OK;               // (a complete plan: every row once per level)
    h_loff[fo + cnt] = (uint32_t)at;
    h_bbase.push_back((uint32_t)at); h_bfeat.push_back(cnt); ++nb;
    Bk->nblk[(size_t)s] = nb;
    const double avg = cnt ? (double)n / cnt : 0.0;
    Bk->lg[(size_t)s] = avg >= 48.0 ? 64 : (avg >= 12.0 ? 16 : (avg >= 3.0 ? 4 : 1));
    fo += (size_t)cnt + 1;
    if (s == L - 1) Bk->last_cnt = cnt;
  }
  // (h_blk_of_feat is indexed by feature position over all levels: fo - s = features before the level)
  auto ok = [](hipError_t e) { if (e != hipSuccess) (void)hipGetLastError(); return e == hipSuccess; };
  struct Tmp {
    uint16_t *blk_of_feat = nullptr, *blkrow = nullptr, *slotrow = nullptr, *keys16 = nullptr, *keys16o = nullptr;
    uint32_t *rowat = nullptr, *keys = nullptr, *keys_o = nullptr, *vals = nullptr, *vals_o = nullptr;
    void* sort_ws = nullptr;
    ~Tmp() { (void)hipFree(blk_of_feat); (void)hipFree(blkrow); (void)hipFree(slotrow); (void)hipFree(keys16); (void)hipFree(keys16o); (void)hipFree(rowat); (void)hipFree(keys);
             (void)hipFree(keys_o); (void)hipFree(vals); (void)hipFree(vals_o); (void)hipFree(sort_ws); }
  } w;
  const size_t sn = (size_t)L * (size_t)n;
  size_t ws1 = 0, ws2 = 0;
  FMX_HIP(rocprim::radix_sort_pairs(nullptr, ws1, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 32, stream));
  FMX_HIP(rocprim::radix_sort_pairs(nullptr, ws2, (uint16_t*)nullptr, (uint16_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 16, stream));
  const size_t ws = ws1 > ws2 ? ws1 : ws2;
  if (!ok(hipMalloc(&Bk->bbase, h_bbase.size() * 4)) || !ok(hipMalloc(&Bk->bfeat, h_bfeat.size() * 4)) || !ok(hipMalloc(&Bk->loff, h_loff.size() * 4)) ||
      !ok(hipMalloc(&Bk->perm_in, sn * 2)) || !ok(hipMalloc(&Bk->gsrc, sn * 2)) || !ok(hipMalloc(&Bk->dest, sn * 4)) || (!in.unit && !ok(hipMalloc(&Bk->xs, sn * 4))) ||
      !ok(hipMalloc(&Bk->row0, (size_t)n * 4)) || !ok(hipMalloc(&Bk->colP, (size_t)n * L * 4)) || (!in.unit && !ok(hipMalloc(&Bk->valP, (size_t)n * L * 4))) ||
      !ok(hipMalloc(&w.blk_of_feat, (n_feats ? n_feats : 1) * 2)) || !ok(hipMalloc(&w.blkrow, sn * 2)) || !ok(hipMalloc(&w.slotrow, sn * 2)) || !ok(hipMalloc(&w.rowat, sn * 4)) ||
      !ok(hipMalloc(&w.keys, (size_t)n * 4)) || !ok(hipMalloc(&w.keys_o, (size_t)n * 4)) || !ok(hipMalloc(&w.vals, (size_t)n * 4)) || !ok(hipMalloc(&w.vals_o, (size_t)n * 4)) ||
      !ok(hipMalloc(&w.keys16, (size_t)n * 2)) || !ok(hipMalloc(&w.keys16o, (size_t)n * 2)) || !ok(hipMalloc(&w.sort_ws, ws ? ws : 16)))
    return FMX_OK;
  FMX_HIP(hipMemcpyAsync(Bk->bbase, h_bbase.data(), h_bbase.size() * 4, hipMemcpyHostToDevice, stream));
  FMX_HIP(hipMemcpyAsync(Bk->bfeat, h_bfeat.data(), h_bfeat.size() * 4, hipMemcpyHostToDevice, stream));
  FMX_HIP(hipMemcpyAsync(Bk->loff, h_loff.data(), h_loff.size() * 4, hipMemcpyHostToDevice, stream));
  FMX_HIP(hipMemcpyAsync(w.blk_of_feat, h_blk_of_feat.data(), n_feats * 2, hipMemcpyHostToDevice, stream));
  const unsigned row_grid = (unsigned)((n + 255) / 256);
  size_t feats_before = 0;
  for (int s = 0; s < L; ++s) {
    const uint32_t cnt = in.cnt[s];
    hipLaunchKernelGGL(blocks_rows_k, dim3((unsigned)(((size_t)cnt * 64 + 255) / 256)), dim3(256), 0, stream, in.d_feats + in.lvl0[s], cnt, (const int64_t*)m->col_ptr,
                       (const uint32_t*)m->crow, (const float*)m->cval, (const uint32_t*)(Bk->loff + Bk->foff[(size_t)s]), (const uint16_t*)(w.blk_of_feat + feats_before),
                       (const uint32_t*)(Bk->bbase + Bk->boff[(size_t)s]), w.blkrow + (size_t)s * n, w.slotrow + (size_t)s * n, Bk->xs ? Bk->xs + (size_t)s * n : nullptr);
    feats_before += cnt;
  }
  // every level's array order: rows by (block at the level, block at the level before (cyclic: level 0 follows the last level), row)
  for (int s = 0; s < L; ++s) {
    const int sp = s > 0 ? s - 1 : L - 1;
    hipLaunchKernelGGL(blocks_keys_k, dim3(row_grid), dim3(256), 0, stream, (const uint16_t*)(w.blkrow + (size_t)s * n), (const uint16_t*)(w.blkrow + (size_t)sp * n), n, w.keys, w.vals);
    size_t wsz = ws;
    FMX_HIP(rocprim::radix_sort_pairs(w.sort_ws, wsz, w.keys, w.keys_o, w.vals, w.rowat + (size_t)s * n, (size_t)n, 0, 32, stream));
    hipLaunchKernelGGL(blocks_place_k, dim3(row_grid), dim3(256), 0, stream, (const uint32_t*)(w.rowat + (size_t)s * n), (const uint16_t*)(w.slotrow + (size_t)s * n), n,
                       Bk->perm_in + (size_t)s * n);
  }
  // every block's destination order: its rows by their position in the next level's array
  for (int s = 0; s < L; ++s) {
    const int sn_ = s + 1 < L ? s + 1 : 0;
    const uint32_t* rowat_next = w.rowat + (size_t)sn_ * n;
    hipLaunchKernelGGL(blocks_keys2_k, dim3(row_grid), dim3(256), 0, stream, rowat_next, (const uint16_t*)(w.blkrow + (size_t)s * n), n, w.keys16, w.vals);
    size_t wsz = ws;
    FMX_HIP(rocprim::radix_sort_pairs(w.sort_ws, wsz, w.keys16, w.keys16o, w.vals, w.vals_o, (size_t)n, 0, 16, stream));
    hipLaunchKernelGGL(blocks_dest_k, dim3(row_grid), dim3(256), 0, stream, (const uint32_t*)w.vals_o, rowat_next, (const uint16_t*)(w.slotrow + (size_t)s * n), n,
                       Bk->dest + (size_t)s * n, Bk->gsrc + (size_t)s * n);
  }
  FMX_HIP(hipMemcpyAsync(Bk->row0, w.rowat, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
  hipLaunchKernelGGL(blocks_permute_csr_k, dim3((unsigned)(((size_t)n * L + 255) / 256)), dim3(256), 0, stream, (const int64_t*)m->row_ptr, (const uint32_t*)m->col, (const float*)m->val,
                     (const uint32_t*)Bk->row0, n, L, Bk->colP, Bk->valP);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipStreamSynchronize(stream));
  { static uint64_t next_uid = 0; Bk->uid = ++next_uid; }
  *out = Bk.release();
  return FMX_OK;
}

void als_blocks_csr(const void* b, const uint32_t** colP, const float** valP, const uint32_t** row0) {
  const AlsBlocks* Bk = reinterpret_cast<const AlsBlocks*>(b);
  *colP = Bk->colP; *valP = Bk->valP;
  if (row0) *row0 = Bk->row0;
}

uint64_t als_blocks_uid(const void* b) { return b ? reinterpret_cast<const AlsBlocks*>(b)->uid : 0; }

// fingerprint of the engine's fp64 V table (waits for the stream)
int als_vhash(fmx_engine* e, uint64_t* out) {
  if (!e->als_hash_word) FMX_HIP(hipMalloc(&e->als_hash_word, sizeof(unsigned long long)));
  FMX_HIP(hipMemsetAsync(e->als_hash_word, 0, sizeof(unsigned long long), e->stream));
  const size_t count = (size_t)e->p * (size_t)e->kp64;
  hipLaunchKernelGGL(als_vhash_k, dim3(2048), dim3(256), 0, e->stream, (const double*)e->dV, count, reinterpret_cast<unsigned long long*>(e->als_hash_word));
  unsigned long long h = 0;
  FMX_HIP(hipMemcpyAsync(&h, e->als_hash_word, sizeof(h), hipMemcpyDeviceToHost, e->stream));
  FMX_HIP(hipStreamSynchronize(e->stream));
  *out = (uint64_t)h;
  return FMX_OK;
}

int als_blocks_info(const void* b, int32_t* block_rows, int32_t* blocks_level0) {
  const AlsBlocks* Bk = reinterpret_cast<const AlsBlocks*>(b);
  if (block_rows) *block_rows = Bk ? Bk->R : 0;
  if (blocks_level0) *blocks_level0 = Bk && !Bk->nblk.empty() ? (int32_t)Bk->nblk[0] : 0;
  return FMX_OK;
}

int als_blocks_enter(fmx_engine* e, const void* b, const double2* d_qe, const double* d_Q0, double2* dst) {
  const AlsBlocks* Bk = reinterpret_cast<const AlsBlocks*>(b);
  hipLaunchKernelGGL(als_block_enter_k, dim3((unsigned)((Bk->n + 255) / 256)), dim3(256), 0, e->stream, d_qe, d_Q0, (const uint32_t*)Bk->row0, Bk->n, dst);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int als_blocks_exit(fmx_engine* e, const void* b, const double2* src, double2* d_qe, double* d_qlast_out, bool e_only) {
  const AlsBlocks* Bk = reinterpret_cast<const AlsBlocks*>(b);
  hipLaunchKernelGGL(als_block_exit_k, dim3((unsigned)((Bk->n + 255) / 256)), dim3(256), 0, e->stream, src, (const uint32_t*)Bk->row0, Bk->n, d_qe, d_qlast_out, e_only ? 1 : 0);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// one level (slot s) of one factor; d_feats: the level's feature ids; d_qin (level 0's array order): this factor's q, taken in by its first level
int als_blocks_level(fmx_engine* e, const void* b, int s, const double2* src, double2* dst, const uint32_t* d_feats, const SweepDyn* dyn, const double* d_qin, double* d_qprev_out, bool w_sweep) {
  const AlsBlocks* Bk = reinterpret_cast<const AlsBlocks*>(b);
  const int nb = (int)Bk->nblk[(size_t)s];
  const dim3 grid((unsigned)(((nb + 7) / 8) * 8)), blk(BLK_THREADS);
  const uint32_t* bbase = Bk->bbase + Bk->boff[(size_t)s];
  const uint32_t* bfeat = Bk->bfeat + Bk->boff[(size_t)s];
  const uint32_t* loff = Bk->loff + Bk->foff[(size_t)s];
  const uint16_t* pin = Bk->perm_in + (size_t)s * Bk->n;
  const uint16_t* gs = Bk->gsrc + (size_t)s * Bk->n;
  const uint32_t* de = Bk->dest + (size_t)s * Bk->n;
  const float* xs = Bk->xs ? Bk->xs + (size_t)s * Bk->n : nullptr;
  FMX_CHECK(!d_qin || (s == 0 && !w_sweep), FMX_ERR_STATE, "a factor's q enters at its FIRST level");
  double* const Pw = w_sweep ? e->dw : e->dV;
#define FMX_BLK(UNITv, Rv, LGv, QNv)                                                                                                                              \
  do {                                                                                                                                                            \
    if (w_sweep) hipLaunchKernelGGL((als_block_level_k<true, UNITv, Rv, LGv, false>), grid, blk, 0, e->stream, src, dst, bbase, bfeat, nb, loff, d_feats, pin, gs, de, xs, Pw, e->kp64, dyn, \
                                    d_qin, d_qprev_out, (uint32_t)Bk->n);                                                                                          \
    else hipLaunchKernelGGL((als_block_level_k<false, UNITv, Rv, LGv, QNv>), grid, blk, 0, e->stream, src, dst, bbase, bfeat, nb, loff, d_feats, pin, gs, de, xs, Pw, e->kp64, dyn, d_qin,  \
                            d_qprev_out, (uint32_t)Bk->n);                                                                                                         \
  } while (0)
#define FMX_BLK_Q(UNITv, Rv, LGv) do { if (d_qin) FMX_BLK(UNITv, Rv, LGv, true); else FMX_BLK(UNITv, Rv, LGv, false); } while (0)
#define FMX_BLK_L(UNITv, Rv)                                                                                                                                      \
  do {                                                                                                                                                            \
    switch (Bk->lg[(size_t)s]) {                                                                                                                                  \
      case 64: FMX_BLK_Q(UNITv, Rv, 64); break;                                                                                                                   \
      case 16: FMX_BLK_Q(UNITv, Rv, 16); break;                                                                                                                   \
      case 4: FMX_BLK_Q(UNITv, Rv, 4); break;                                                                                                                     \
      default: FMX_BLK_Q(UNITv, Rv, 1); break;                                                                                                                    \
    }                                                                                                                                                             \
  } while (0)
  const bool pipe = env_int_b("FMX_ALS_BLOCK_PIPE", 1) != 0;   // (read per launch: the tests compare the two kernels in one process)
  if (pipe && !d_qin) {
    // one resident workgroup per CU (the LDS admits one): `slots` of them per XCD, each walking its XCD's share of the blocks
    static int n_cus = 0;
    if (n_cus == 0) { hipDeviceProp_t pr{}; n_cus = (hipGetDeviceProperties(&pr, e->cfg.device) == hipSuccess && pr.multiProcessorCount >= 8) ? pr.multiProcessorCount : 256; }
    const int per = (nb + 7) / 8;
    const int slots = per < n_cus / 8 ? per : n_cus / 8;
    const dim3 pgrid((unsigned)(slots * 8));
    const uint32_t cnt = (uint32_t)(Bk->foff.size() > (size_t)s + 1 ? Bk->foff[(size_t)s + 1] - Bk->foff[(size_t)s] - 1 : Bk->last_cnt);
#define FMX_BLKP(UNITv, Rv, LGv)                                                                                                                                  \
  do {                                                                                                                                                            \
    if (w_sweep) hipLaunchKernelGGL((als_block_level_pipe_k<true, UNITv, Rv, LGv>), pgrid, blk, 0, e->stream, src, dst, bbase, bfeat, nb, loff, d_feats, cnt, pin, gs, de, xs, Pw, e->kp64, \
                                    dyn, (uint32_t)Bk->n);                                                                                                         \
    else hipLaunchKernelGGL((als_block_level_pipe_k<false, UNITv, Rv, LGv>), pgrid, blk, 0, e->stream, src, dst, bbase, bfeat, nb, loff, d_feats, cnt, pin, gs, de, xs, Pw, e->kp64, dyn,   \
                            (uint32_t)Bk->n);                                                                                                                      \
  } while (0)
#define FMX_BLKP_L(UNITv, Rv)                                                                                                                                     \
  do {                                                                                                                                                            \
    switch (Bk->lg[(size_t)s]) {                                                                                                                                  \
      case 64: FMX_BLKP(UNITv, Rv, 64); break;                                                                                                                    \
      case 16: FMX_BLKP(UNITv, Rv, 16); break;                                                                                                                    \
      case 4: FMX_BLKP(UNITv, Rv, 4); break;                                                                                                                      \
      default: FMX_BLKP(UNITv, Rv, 1); break;                                                                                                                     \
    }                                                                                                                                                             \
  } while (0)
    if (Bk->unit) FMX_BLKP_L(true, BLK_R_UNIT); else FMX_BLKP_L(false, BLK_R_VAL);
#undef FMX_BLKP_L
#undef FMX_BLKP
    FMX_HIP(hipGetLastError());
    return FMX_OK;
  }
  if (Bk->unit) FMX_BLK_L(true, BLK_R_UNIT); else FMX_BLK_L(false, BLK_R_VAL);
#undef FMX_BLK_L
#undef FMX_BLK_Q
#undef FMX_BLK
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

}  // namespace fmx
