// ALS V-column sweep: MCMC_ALS_Learner::update_v, ALS branch (solver/MCMC_ALS_Learner.h:272-354), with the
// cached residual e and the per-factor cache q = X v_f it maintains (:291-299, :341-350).
//
// The reference walks the features of one factor strictly in index order (Gauss-Seidel): feature i reads and then
// corrects q[r], e[r] of every row r that holds it.  Two features interact only if they share a row, so the exact
// sequential result is kept by "level scheduling": level(i) = 1 + max level of the smaller-index features sharing a
// row with i; all features of one level are independent and are processed by one launch, levels in ascending order.
// Field-structured data (one active feature per field and row: one-hot encodings, the synthetic workload's strata,
// MovieLens user/item) has as many levels as fields.  One wavefront owns one feature: lanes stride over the
// feature's CSC column, reduce sum(h*e), sum(h*h), form the new v, then apply the rank-1 corrections with plain
// stores (no two features of a level touch the same row, so there are no atomics and the result is reproducible).
// fp64 throughout, like the reference; x*x is a float product there (:314,:345) and here.
// q and e live interleaved as one {q[r], e[r]} pair per row, so a stored nonzero costs one 16-byte gather (one line)
// per pass instead of two.
#include <atomic>
#include <cmath>
#include <ctime>

#include "fmx_internal.h"
#include "fm_probit.h"

namespace fmx {

// ---- levels ------------------------------------------------------------------------------------------------
// one relaxation sweep over the rows (columns ascending inside a row): level[c_j] >= level[c_{j-1}] + 1
__global__ void level_relax_k(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, int64_t n,
                              int* __restrict__ level, int* __restrict__ changed) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int run = -1;
  bool ch = false;
  for (int64_t t = row_ptr[r]; t < row_ptr[r + 1]; ++t) {
    const uint32_t c = col[t];
    const int want = run + 1;
    int cur = level[c];
    if (cur < want) {
      const int old = atomicMax(&level[c], want);
      cur = old > want ? old : want;
      ch = true;
    }
    run = cur;
  }
  if (ch) *changed = 1;
}

// The same levels by a frontier walk (Kahn's layering; level = longest chain of row predecessors, exactly the relaxation's fixed
// point): a feature is READY once every row holding it has it at its head -- `ready[j]` counts those rows against the column's
// length -- and the round in which it becomes ready is its level.  Each round touches only the frontier's columns, so the whole
// walk reads every entry once (the relaxation reads all of them once per LEVEL: 8 155 sweeps of 120 M entries for i.i.d. columns).
// Rows must not hold a column twice (they never converge under the relaxation either).
__global__ void level_heads_k(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, int64_t n, const int64_t* __restrict__ col_ptr,
                              int* __restrict__ ready, uint32_t* __restrict__ frontier, int* __restrict__ n_front) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || row_ptr[r] == row_ptr[r + 1]) return;
  const uint32_t h = col[row_ptr[r]];
  const int c = atomicAdd(&ready[h], 1) + 1;
  if ((int64_t)c == col_ptr[h + 1] - col_ptr[h]) frontier[atomicAdd(n_front, 1)] = h;
}

// what happens to one entry (row r) of a frontier feature: the row moves on to its next entry, which may complete a feature
__device__ __forceinline__ void level_advance(uint32_t r, const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                              const int64_t* __restrict__ col_ptr, int* __restrict__ pos, int* __restrict__ ready,
                                              uint32_t* __restrict__ f_out, int* __restrict__ n_out) {
  const int64_t at = row_ptr[r] + (int64_t)(++pos[r]);  // this row's head was the frontier feature: nobody else touches it this round
  if (at < row_ptr[r + 1]) {
    const uint32_t h = col[at];
    const int c = atomicAdd(&ready[h], 1) + 1;
    if ((int64_t)c == col_ptr[h + 1] - col_ptr[h]) f_out[atomicAdd(n_out, 1)] = h;
  }
}

// one round: wave w takes frontier feature w, w + waves, ...; counters rotate over three slots (in, out, the one to clear).
// A feature with a LONG column (a Zipf head holds a million entries: one wave walking it is milliseconds, and the rounds are a
// dependent chain) is only recorded here; level_heavy_k, launched behind every round, walks such columns with the whole grid.
constexpr int64_t LEVEL_HEAVY = 16384;
__global__ __launch_bounds__(WG_THREADS) void level_round_k(const uint32_t* __restrict__ f_in, uint32_t* __restrict__ f_out, int* __restrict__ cnt, int round,
                                                            const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col,
                                                            const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow,
                                                            int* __restrict__ pos, int* __restrict__ ready, int* __restrict__ level,
                                                            unsigned long long* __restrict__ assigned, uint32_t* __restrict__ heavy, int* __restrict__ n_heavy) {
  const int n_in = cnt[round % 3];
  int* n_out = cnt + (round + 1) % 3;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cnt[(round + 2) % 3] = 0; *assigned += (unsigned long long)n_in; }
  const int lane = threadIdx.x & 63;
  const int waves = (int)(gridDim.x * (WG_THREADS / 64));
  for (int w = (int)(((int64_t)blockIdx.x * WG_THREADS + threadIdx.x) >> 6); w < n_in; w += waves) {
    const uint32_t j = f_in[w];
    if (lane == 0) level[j] = round;
    const int64_t b = col_ptr[j], e = col_ptr[j + 1];
    if (e - b > LEVEL_HEAVY) {
      if (lane == 0) heavy[atomicAdd(n_heavy, 1)] = j;
      continue;
    }
    for (int64_t t = b + lane; t < e; t += 64) level_advance(crow[t], row_ptr, col, col_ptr, pos, ready, f_out, n_out);
  }
}

// the long columns recorded by this round (usually none: the kernel then ends at once), every thread of the grid striding over each
__global__ __launch_bounds__(WG_THREADS) void level_heavy_k(const uint32_t* __restrict__ heavy, int* __restrict__ n_heavy, uint32_t* __restrict__ f_out,
                                                            int* __restrict__ cnt, int round, const int64_t* __restrict__ row_ptr,
                                                            const uint32_t* __restrict__ col, const int64_t* __restrict__ col_ptr,
                                                            const uint32_t* __restrict__ crow, int* __restrict__ pos, int* __restrict__ ready,
                                                            int* __restrict__ done_flag) {
  const int nh = *n_heavy;
  if (nh == 0) return;
  int* n_out = cnt + (round + 1) % 3;
  const int64_t stride = (int64_t)gridDim.x * WG_THREADS, me = (int64_t)blockIdx.x * WG_THREADS + threadIdx.x;
  for (int q = 0; q < nh; ++q) {
    const uint32_t j = heavy[q];
    for (int64_t t = col_ptr[j] + me; t < col_ptr[j + 1]; t += stride) level_advance(crow[t], row_ptr, col, col_ptr, pos, ready, f_out, n_out);
  }
  // the last workgroup to finish clears the list for the next round
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(done_flag, 1) == (int)gridDim.x - 1) { *n_heavy = 0; *done_flag = 0; }
}

// ---- the FEATURE-MAJOR coloured order (cfg.als_max_levels = -2) ----------------------------------------------------------------------------
// als_level_k steps ONE factor of a level's features: per entry a random 16-byte gather and scatter of the row's (q_f, e) -- a whole line moved each way for 16 bytes
// used, k times per sweep.  A sweep that chooses its own order may as well step ALL k factors of a feature while its rows are on the chip: the order becomes (colour,
// feature, factor) -- still one exact coordinate step after the other -- and a row's state, e and the k values q_f = (X v_f)_r, kept together ([n][kp] doubles: one
// 128-byte line at k = 16, next to the pair that holds e), is fetched ONCE per level the row takes part in instead of once per level and factor.
// One workgroup per feature: its rows' q lines (8 lanes per row: whole lines), e and x land in LDS; then for f = 0 .. k - 1 every thread forms h for its rows
// (:310-317), the block adds (waves in order: reproducible), every thread takes the step from the same sums (:318-336) and corrects its rows' e (registers) and q_f
// (LDS) (:341-350); the lines go back.  Lists of up to `cap` rows (the level's longest, set by the launcher; the LDS holds cap x kp + 2 kp doubles + cap words).
__device__ __forceinline__ bool bad_number(double x);
template <bool UNIT>
__global__ __launch_bounds__(WG_THREADS) void als_level_allf_k(const uint32_t* __restrict__ feats, int n_feats, const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow,
                                                               const float* __restrict__ cval, double* __restrict__ V, int k, int kp, double alpha, const double* __restrict__ lam_mu,
                                                               const double* __restrict__ znorm, int64_t zstride, double* __restrict__ Q, double2* __restrict__ qe, int cap, int eil) {   // eil: e rides in the line's spare last slot (k < kp) instead of the pair table
  extern __shared__ double lds_allf[];
  double* sQ = lds_allf;                                  // [kp][cap]: factor-major (a wave's threads read neighbouring rows of one factor: no bank conflicts, no padding)
  double* sOld = sQ + (size_t)kp * cap;                   // [kp] the feature's coordinates as the level finds them
  double* sZ = sOld + kp;                                 // [kp] its standard normals (Gibbs)
  uint32_t* sRow = reinterpret_cast<uint32_t*>(sZ + kp);  // [cap]
  __shared__ double red[2][WG_THREADS / 64][2];           // (two sets, used alternately: ONE barrier per factor)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if ((int)blockIdx.x >= n_feats) return;
  const uint32_t j = feats[blockIdx.x];
  const int64_t b = col_ptr[j];
  const int len = (int)(col_ptr[j + 1] - b);
  constexpr int RPT = 4;                                  // rows per thread (cap <= RPT x 256): their e, x and h stay in registers
  double er[RPT]; float xr[RPT]; uint32_t rr[RPT];
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    const int i = tid + u * WG_THREADS;
    const int64_t tc = i < len ? b + i : 0;               // (a slot past the list: entry 0 of the matrix, read and never used -- a feature without rows still takes its step)
    rr[u] = crow[tc];
    xr[u] = UNIT ? 1.0f : cval[tc];
  }
  if (tid < k) {                                          // everything the k steps read from memory goes out now: nothing inside the factor loop waits for a load
    sOld[tid] = V[(size_t)j * kp + tid];
    sZ[tid] = znorm ? znorm[(size_t)tid * zstride + j] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < RPT; ++u) er[u] = eil ? 0.0 : qe[rr[u]].y;
#pragma unroll
  for (int u = 0; u < RPT; ++u) { const int i = tid + u * WG_THREADS; if (i < len) sRow[i] = rr[u]; }
  __syncthreads();
  const int LQ = kp >> 1;                                 // lanes per row: each takes two factors (16 bytes) of the row's line
  const int rpp = WG_THREADS / LQ;                        // rows per pass
  const int grow = tid / LQ, part = tid % LQ;
  for (int i0 = 0; i0 < len; i0 += 4 * rpp) {             // four passes' loads in flight
    double2 v2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q * rpp + grow, ic = (i < len && grow < rpp) ? i : 0;
      v2[q] = *reinterpret_cast<const double2*>(Q + (size_t)sRow[ic] * kp + 2 * part);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q * rpp + grow;
      if (i < len && grow < rpp) { sQ[(size_t)(2 * part) * cap + i] = v2[q].x; sQ[(size_t)(2 * part + 1) * cap + i] = v2[q].y; }
    }
  }
  __syncthreads();
  if (eil) {
#pragma unroll
    for (int u = 0; u < RPT; ++u) { const int i = tid + u * WG_THREADS; if (i < len) er[u] = sQ[(size_t)(kp - 1) * cap + i]; }
  }
  const bool gibbs = znorm != nullptr;
  for (int f = 0; f < k; ++f) {
    const double old = sOld[f];
    const double* __restrict__ qf = sQ + (size_t)f * cap;
    double mean = 0.0, var = 0.0, hk[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int i = tid + u * WG_THREADS;
      hk[u] = 0.0;
      if (i < len) {
        const float x = xr[u], xx = x * x;
        const double h = (double)x * qf[i] - (double)xx * old;   // :310-317
        hk[u] = h; mean += h * er[u]; var += h * h;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mean += __shfl_xor(mean, o); var += __shfl_xor(var, o); }
    if (lane == 0) { red[f & 1][wv][0] = mean; red[f & 1][wv][1] = var; }
    __syncthreads();
    // every thread takes the step itself from the same four pairs (the same bits): no second barrier to hand the result round
    double m = 0.0, vr = 0.0;
#pragma unroll
    for (int w = 0; w < WG_THREADS / 64; ++w) { m += red[f & 1][w][0]; vr += red[f & 1][w][1]; }
    const double lambda = lam_mu[2 * f], mu = lam_mu[2 * f + 1];
    m -= old * vr;                                         // :318
    vr = 1.0 / (lambda + alpha * vr);                      // :319
    m = -vr * (alpha * m - mu * lambda);                   // :320
    const double nv = bad_number(vr) ? 0.0 : (gibbs ? m + sqrt(vr) * sZ[f] : m);
    if (bad_number(nv)) continue;                          // CHECK_PARAM (:336): keep the old value, skip the corrections (uniform: no barrier is skipped)
    if (tid == 0) V[(size_t)j * kp + f] = nv;
    const double diff = old - nv;
    double* __restrict__ qw = sQ + (size_t)f * cap;
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int i = tid + u * WG_THREADS;
      if (i < len) { qw[i] -= (double)xr[u] * diff; er[u] -= hk[u] * diff; }   // :341-350
    }
  }
  if (eil) {
#pragma unroll
    for (int u = 0; u < RPT; ++u) { const int i = tid + u * WG_THREADS; if (i < len) sQ[(size_t)(kp - 1) * cap + i] = er[u]; }
  }
  __syncthreads();
  for (int i0 = 0; i0 < len; i0 += rpp) {
    const int i = i0 + grow;
    if (i < len && grow < rpp)
      *reinterpret_cast<double2*>(Q + (size_t)sRow[i] * kp + 2 * part) = make_double2(sQ[(size_t)(2 * part) * cap + i], sQ[(size_t)(2 * part + 1) * cap + i]);
  }
#pragma unroll
  for (int u = 0; u < RPT; ++u) { const int i = tid + u * WG_THREADS; if (i < len && !eil) qe[rr[u]].y = er[u]; }
}

// The same level with the rows' lines in REGISTERS (kp = 8 or 16, lists of up to 512 rows): thread t owns rows t and t + 256 -- their q lines (2 x kp doubles), e and x --
// and the LDS only stages the transposition (lines arrive 8 lanes per row, whole lines; 256 rows x (kp + 1) doubles per pass).  35 KB of LDS and <= 128 registers: FOUR
// workgroups per CU instead of the three (two, padded) the LDS-resident form fits -- on the i.i.d. law the colour classes hold ~1 000 features (FMX_COLOUR_DEBUG), more
// than 768 resident workgroups and fewer than 1 024: one round of workgroups per launch instead of two.  Same thread -> row map, same order of every sum as
// als_level_allf_k: bit for bit its results (tests/test_gpu_coloured.py).
#ifndef FMX_ALLF_AHEAD
#define FMX_ALLF_AHEAD 1
#endif
#ifndef FMX_ALLF_WPE
#define FMX_ALLF_WPE 2
#endif
#ifndef FMX_ALLF_KO
#define FMX_ALLF_KO 0   // diagnostic builds (profiles/probes/allf_knockouts.sh): 1 no factor loop, 2 no q lines moved, 4 no e moved
#endif
template <bool UNIT, int KP, bool EIL>   // EIL: e rides in the line's spare last slot (k < kp) instead of the pair table
__global__ __launch_bounds__(WG_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void als_level_allf_reg_k(
    const uint32_t* __restrict__ feats, int n_feats, const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow, const float* __restrict__ cval, double* __restrict__ V, int k,
    double alpha, const double* __restrict__ lam_mu, const double* __restrict__ znorm, int64_t zstride, double* __restrict__ Q, double2* __restrict__ qe) {
  constexpr int RPT = 2, TS = KP + 1, LQ = KP / 2, RPP = WG_THREADS / LQ, NSUB = WG_THREADS / RPP;   // NSUB = LQ sub-passes of RPP rows cover a pass of 256 rows
  __shared__ double sT[WG_THREADS * TS];
  __shared__ double sOld[KP], sZ[KP];
  __shared__ uint32_t sRow[RPT * WG_THREADS];
  __shared__ double red[2][WG_THREADS / 64][2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if ((int)blockIdx.x >= n_feats) return;
  const uint32_t j = feats[blockIdx.x];
  const int64_t b = col_ptr[j];
  const int len = (int)(col_ptr[j + 1] - b);
  double er[RPT]; float xr[RPT]; uint32_t rr[RPT];
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    const int i = tid + u * WG_THREADS;
    const int64_t tc = i < len ? b + i : 0;
    rr[u] = crow[tc];
    xr[u] = UNIT ? 1.0f : cval[tc];
  }
  if (tid < KP) {
    sOld[tid] = tid < k ? V[(size_t)j * KP + tid] : 0.0;
    sZ[tid] = (znorm && tid < k) ? znorm[(size_t)tid * zstride + j] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < RPT; ++u) er[u] = (EIL || (FMX_ALLF_KO & 4)) ? 0.5 : qe[rr[u]].y;
#pragma unroll
  for (int u = 0; u < RPT; ++u) sRow[tid + u * WG_THREADS] = rr[u];   // (slots past the list hold entry 0's row: loaded, never stored)
  __syncthreads();
  const int grow = tid / LQ, part = tid % LQ;
  const int npass = len > WG_THREADS ? 2 : 1;
  double q[RPT][KP];
  double2 v2[RPT][NSUB];
#pragma unroll
  for (int u = 0; u < RPT; ++u)
    if (u < npass) {
#pragma unroll
      for (int s = 0; s < NSUB; ++s) v2[u][s] = (FMX_ALLF_KO & 2) ? make_double2(1.0, 2.0) : *reinterpret_cast<const double2*>(Q + (size_t)sRow[u * WG_THREADS + s * RPP + grow] * KP + 2 * part);
    }
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    if (u < npass) {
      if (u > 0) __syncthreads();
#pragma unroll
      for (int s = 0; s < NSUB; ++s) { sT[(s * RPP + grow) * TS + 2 * part] = v2[u][s].x; sT[(s * RPP + grow) * TS + 2 * part + 1] = v2[u][s].y; }
      __syncthreads();
#pragma unroll
      for (int f = 0; f < KP; ++f) q[u][f] = sT[tid * TS + f];
    } else {
#pragma unroll
      for (int f = 0; f < KP; ++f) q[u][f] = 0.0;
    }
  }
  if (EIL) {
#pragma unroll
    for (int u = 0; u < RPT; ++u) er[u] = q[u][KP - 1];
  }
  const bool gibbs = znorm != nullptr;
#pragma unroll
  for (int f = 0; f < KP; ++f) {
    if (f < k && !(FMX_ALLF_KO & 1)) {
      const double old = sOld[f];
      double mean = 0.0, var = 0.0, hk[RPT];
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const int i = tid + u * WG_THREADS;
        hk[u] = 0.0;
        if (i < len) {
          const float x = xr[u], xx = x * x;
          const double h = (double)x * q[u][f] - (double)xx * old;   // :310-317
          hk[u] = h; mean += h * er[u]; var += h * h;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { mean += __shfl_xor(mean, o); var += __shfl_xor(var, o); }
      if (lane == 0) { red[f & 1][wv][0] = mean; red[f & 1][wv][1] = var; }
      __syncthreads();
      double m = 0.0, vr = 0.0;
#pragma unroll
      for (int w = 0; w < WG_THREADS / 64; ++w) { m += red[f & 1][w][0]; vr += red[f & 1][w][1]; }
      const double lambda = lam_mu[2 * f], mu = lam_mu[2 * f + 1];
      m -= old * vr;                                         // :318
      vr = 1.0 / (lambda + alpha * vr);                      // :319
      m = -vr * (alpha * m - mu * lambda);                   // :320
      const double nv = bad_number(vr) ? 0.0 : (gibbs ? m + sqrt(vr) * sZ[f] : m);
      if (!bad_number(nv)) {                                 // CHECK_PARAM (:336)
        if (tid == 0) V[(size_t)j * KP + f] = nv;
        const double diff = old - nv;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
          const int i = tid + u * WG_THREADS;
          if (i < len) { q[u][f] -= (double)xr[u] * diff; er[u] -= hk[u] * diff; }   // :341-350
        }
      }
    }
  }
  if (EIL) {
#pragma unroll
    for (int u = 0; u < RPT; ++u) q[u][KP - 1] = er[u];
  }
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    if (u < npass) {
      __syncthreads();
#pragma unroll
      for (int f = 0; f < KP; ++f) sT[tid * TS + f] = q[u][f];
      __syncthreads();
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        const int i = u * WG_THREADS + s * RPP + grow;
        if (i < len && (!(FMX_ALLF_KO & 2) || sT[(s * RPP + grow) * TS + 2 * part] == 1e300)) *reinterpret_cast<double2*>(Q + (size_t)sRow[i] * KP + 2 * part) = make_double2(sT[(s * RPP + grow) * TS + 2 * part], sT[(s * RPP + grow) * TS + 2 * part + 1]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < RPT; ++u) { const int i = tid + u * WG_THREADS; if (!EIL && i < len && (!(FMX_ALLF_KO & 4) || er[u] == 1e300)) qe[rr[u]].y = er[u]; }
}

// ONE WAVE per feature (kp = 8 or 16, lists of up to 64 x RPT rows, RPT <= 6).  The knock-outs of the 256-thread kernel (profiles/r05_allf_knockouts.txt) put 32 of a
// level's 48 us in the factor loop: four waves per feature each pay the wave sum, the hand-over through LDS, the barrier and the step (a division and a square
// root in fp64) -- about 250 instructions per wave and factor, issued by SIMDs that hold four such waves.  A list of 300 rows is five rows per lane of ONE wave:
// no barrier, no LDS in the loop, one wave sum (row-wise by DPP, the four rows by readlane: every lane ends with the same bits) and one step per factor, a
// quarter of the instructions; the feature's old coordinates and normals sit in lanes 0 .. kp - 1 and are read by readlane.  Lane t owns rows t, t + 64, ...
#define FMX_DPP64(x, ctrl) __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xF, 0xF, false))
__device__ __forceinline__ double lane_f64(double x, int lane) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane)); }
__device__ __forceinline__ double wave_allsum(double x) {
  x += FMX_DPP64(x, 0xB1);    // quad_perm [1,0,3,2]
  x += FMX_DPP64(x, 0x4E);    // quad_perm [2,3,0,1]
  x += FMX_DPP64(x, 0x141);   // row_half_mirror: the other quad of the eight
  x += FMX_DPP64(x, 0x140);   // row_mirror: the other eight of the row of 16
  return ((lane_f64(x, 0) + lane_f64(x, 16)) + lane_f64(x, 32)) + lane_f64(x, 48);
}
template <bool UNIT, int KP, int RPT, bool EIL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FMX_ALLF_WPE, FMX_ALLF_WPE))) void als_level_allf_wave_k(
    const uint32_t* __restrict__ feats, int n_feats, const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow, const float* __restrict__ cval, double* __restrict__ V, int k,
    double alpha, const double* __restrict__ lam_mu, const double* __restrict__ znorm, int64_t zstride, double* __restrict__ Q, double2* __restrict__ qe) {
  constexpr int TS = KP + 1, LQ = KP / 2, RPP = 64 / LQ, NSUB = LQ;   // a pass = 64 rows = NSUB loads of RPP rows (LQ lanes per row, 16 bytes each: whole lines)
  __shared__ double sT[64 * TS];
  __shared__ uint32_t sRow[RPT * 64];
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= n_feats) return;
  const uint32_t j = feats[blockIdx.x];
  const int64_t b = col_ptr[j];
  const int len = (int)(col_ptr[j + 1] - b);
  // Slots past the list's end (the launcher sizes RPT for the level's LONGEST list) read entry 0 of the matrix -- one hot line -- and count as rows with x = 0, q = 0,
  // e = 0: straight-line code, no branch per row anywhere; they are never stored.
  double er[RPT]; float xr[RPT];
  {
    uint32_t rr[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int i = lane + u * 64;
      const int64_t tc = i < len ? b + i : 0;
      rr[u] = crow[tc];
      xr[u] = i < len ? (UNIT ? 1.0f : cval[tc]) : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) { const double ev = (EIL || (FMX_ALLF_KO & 4)) ? 0.5 : qe[rr[u]].y; er[u] = lane + u * 64 < len ? ev : 0.0; }
#pragma unroll
    for (int u = 0; u < RPT; ++u) sRow[lane + u * 64] = rr[u];
  }
  const double vold = lane < k ? V[(size_t)j * KP + (lane < KP ? lane : 0)] : 0.0;                        // lane f: the feature's coordinate f as the level finds it
  const double vz = (znorm && lane < k) ? znorm[(size_t)(lane < KP ? lane : 0) * zstride + j] : 0.0;      //         and its standard normal (Gibbs)
  const int grow = lane / LQ, part = lane % LQ;
  double q[RPT][KP];
  // the passes' gathers run AHEAD of the transposition (the list's lines arrive in RPT dependent round trips otherwise): pass u + AHEAD goes out before pass u is unpacked
  constexpr int AHEAD = RPT < FMX_ALLF_AHEAD ? RPT : FMX_ALLF_AHEAD;   // (1, 2, 3, 6 ahead: the same sweep time to 3 % -- profiles/r05_allf_knockouts.txt)
  double2 v2[RPT][NSUB];
#define FMX_ALLF_ISSUE(u_)                                                                                                                         \
  _Pragma("unroll") for (int s = 0; s < NSUB; ++s)                                                                                                   \
    v2[u_][s] = (FMX_ALLF_KO & 2) ? make_double2(1.0, 2.0) : *reinterpret_cast<const double2*>(Q + (size_t)sRow[(u_) * 64 + s * RPP + grow] * KP + 2 * part)
#pragma unroll
  for (int u = 0; u < AHEAD; ++u) { FMX_ALLF_ISSUE(u); }
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    if (u + AHEAD < RPT) { FMX_ALLF_ISSUE(u + AHEAD); }
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise hoists every pass's gathers to the top: 192 registers of lines in flight beside the 192 they unpack into)
#pragma unroll
    for (int s = 0; s < NSUB; ++s) { sT[(s * RPP + grow) * TS + 2 * part] = v2[u][s].x; sT[(s * RPP + grow) * TS + 2 * part + 1] = v2[u][s].y; }
#pragma unroll
    for (int f = 0; f < KP; ++f) { const double t = sT[lane * TS + f]; q[u][f] = lane + u * 64 < len ? t : 0.0; }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef FMX_ALLF_ISSUE
  if (EIL) {   // (slots past the list's end were unpacked as zeros: e = 0 there too)
#pragma unroll
    for (int u = 0; u < RPT; ++u) er[u] = q[u][KP - 1];
  }
  const bool gibbs = znorm != nullptr;
  double vnew = vold;
#pragma unroll
  for (int f = 0; f < KP; ++f) {
    if (f < k && !(FMX_ALLF_KO & 1)) {
      const double old = lane_f64(vold, f);
      double mean = 0.0, var = 0.0;
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        const float x = xr[u], xx = x * x;
        const double h = (double)x * q[u][f] - (double)xx * old;   // :310-317
        mean += h * er[u]; var += h * h;
      }
      double m = wave_allsum(mean), vr = wave_allsum(var);
      const double lambda = lam_mu[2 * f], mu = lam_mu[2 * f + 1];
      m -= old * vr;                                         // :318
      vr = 1.0 / (lambda + alpha * vr);                      // :319
      m = -vr * (alpha * m - mu * lambda);                   // :320
      const double nv = bad_number(vr) ? 0.0 : (gibbs ? m + sqrt(vr) * lane_f64(vz, f) : m);
      if (!bad_number(nv)) {                                 // CHECK_PARAM (:336) -- the same in every lane: a scalar branch
        if (lane == f) vnew = nv;
        const double diff = old - nv;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
          const float x = xr[u], xx = x * x;
          const double h = (double)x * q[u][f] - (double)xx * old;   // (formed again rather than kept: the registers would cost a wave of occupancy)
          q[u][f] -= (double)x * diff; er[u] -= h * diff;            // :341-350
        }
      }
    }
  }
  if (lane < k) V[(size_t)j * KP + lane] = vnew;
  if (EIL) {
#pragma unroll
    for (int u = 0; u < RPT; ++u) q[u][KP - 1] = er[u];
  }
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
#pragma unroll
    for (int f = 0; f < KP; ++f) sT[lane * TS + f] = q[u][f];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      const int i = u * 64 + s * RPP + grow;
      if (i < len && (!(FMX_ALLF_KO & 2) || sT[(s * RPP + grow) * TS + 2 * part] == 1e300))
        *reinterpret_cast<double2*>(Q + (size_t)sRow[i] * KP + 2 * part) = make_double2(sT[(s * RPP + grow) * TS + 2 * part], sT[(s * RPP + grow) * TS + 2 * part + 1]);
    }
  }
#pragma unroll
  for (int u = 0; u < RPT; ++u) { const int i = lane + u * 64; if (!EIL && i < len && (!(FMX_ALLF_KO & 4) || er[u] == 1e300)) qe[sRow[i]].y = er[u]; }
}

// k < kp: the row's line has a spare last slot, and e rides there for the length of a feature-major sweep -- the 8 bytes of e otherwise cost a 128-byte line in and a
// partial line out per row and level, 43 % of a row's memory time (profiles/r05_allf_knockouts.txt (e)).  In from the pairs before the first level, back after the last.
__global__ void allf_e_enter_k(double* __restrict__ Q, const double2* __restrict__ qe, int64_t n, int kp) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) Q[(size_t)r * kp + (kp - 1)] = qe[r].y;
}
__global__ void allf_e_exit_k(const double* __restrict__ Q, double2* __restrict__ qe, int64_t n, int kp) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) qe[r].y = Q[(size_t)r * kp + (kp - 1)];
}

// ---- the COLOURED order (cfg.als_max_levels < 0) -----------------------------------------------------------------------------------------
// The exact schedule keeps the reference's feature ORDER: feature j waits for every earlier feature it shares a row with, which on i.i.d. columns is a chain
// of ~20 000 levels of ~50 features (10 M x 1 M: 310 K dependent launches per sweep of 16 factors: 5 M examples/s, profiles/r05_als_iid_sweep.txt).  A sweep that
// may choose its own order -- any order is a Gauss-Seidel pass for ALS and a valid scan for the Gibbs sampler, but not the reference's numbers on the same
// inputs -- only needs features of one level to share no row: a proper COLOURING of the "share a row" graph, a thousand-odd colours of a thousand features here.
// The sweep then visits the features in (colour, index) order, every step exact (no snapshot, no guard): the oracle reproduces it on the relabelled matrix.
// Colouring, deterministic (the plan, and with it the visiting order, must not depend on timing): rounds of
//   assign   every uncoloured feature reads the colours FIXED IN EARLIER ROUNDS of all features it shares a row with (a bitset per wave in LDS) and takes the
//            t-th free colour, t = hash(feature, round) mod COLOUR_SPREAD (neighbours that choose in the same round spread over several colours);
//   resolve  a feature that chose the colour of a LOWER-indexed neighbour of the same round gives it up and stays for the next round.
constexpr int COLOUR_MAX = 8192;       // bits of the per-wave set
constexpr int COLOUR_SPREAD = 8;
constexpr int COLOUR_TEAM_BELOW = 4096;   // fewer features than this in a launch: a workgroup walks each (colour_assign_k<true>)
__device__ __forceinline__ uint32_t colour_hash(uint32_t a, uint32_t b) {
  uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
  x ^= x >> 15; x *= 0xC2B2AE3Du; x ^= x >> 13;
  return x;
}
// TEAM = false: a wave per feature (crowded rounds: every SIMD has features to walk); TEAM = true: a WORKGROUP per feature -- a small class (the refit passes launch one
// class of ~1 000 features at a time, the late rounds a few thousand) is latency-bound with one wave walking 300 rows x 30 entries, four waves walk them in a quarter
// of the time.  Same forbidden set, same choice.
template <bool TEAM>
__global__ __launch_bounds__(WG_THREADS) void colour_assign_k(const uint32_t* __restrict__ act, int n_act, const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow,
                                                              const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const int* fixed,
                                                              int* chosen, int round, int spread, int* __restrict__ overflow, int scan) {   // (the refit passes run it with chosen == fixed; scan = 0: nothing is fixed yet -- round 0 without heavy columns --, no neighbour can forbid anything)
  constexpr int WORDS = COLOUR_MAX / 32, WPB = WG_THREADS / 64, PER = WORDS / 64;
  __shared__ uint32_t forb[TEAM ? 1 : WPB][WORDS];
  const int lane = threadIdx.x & 63, wv = TEAM ? 0 : (int)(threadIdx.x >> 6);
  const int step = TEAM ? (int)gridDim.x : (int)(gridDim.x * WPB);
  const int tl = TEAM ? (int)threadIdx.x : lane, tn = TEAM ? WG_THREADS : 64;   // this thread's place among the walkers of one feature
  for (int w = TEAM ? (int)blockIdx.x : (int)(blockIdx.x * WPB + (threadIdx.x >> 6)); w < n_act; w += step) {
    const uint32_t j = act[w];
    if (TEAM) { for (int q = threadIdx.x; q < WORDS; q += WG_THREADS) forb[0][q] = 0u; __syncthreads(); }
    else {
#pragma unroll
      for (int q = 0; q < PER; ++q) forb[wv][lane * PER + q] = 0u;
      __builtin_amdgcn_wave_barrier();
    }
    for (int64_t t = col_ptr[j] + tl; scan && t < col_ptr[j + 1]; t += tn) {
      const uint32_t r = crow[t];
      const int64_t ub = row_ptr[r], ue = row_ptr[r + 1];
      for (int64_t u = ub; u < ue; u += 8) {   // eight entries' columns, then their colours: independent loads in flight (a dependent pair per entry took 24 s of plan time at 10 M x 1 M)
        uint32_t k[8]; int c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) k[i] = col[u + i < ue ? u + i : ue - 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = fixed[k[i]];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (u + i < ue && c[i] >= 0 && k[i] != j) atomicOr(&forb[wv][c[i] >> 5], 1u << (c[i] & 31));   // (k == j: a feature being REFITTED may keep its own colour)
      }
    }
    if (TEAM) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    if (!TEAM || threadIdx.x < 64) {
      // the t-th free colour: lane l owns words l * PER .. (consecutive colours), an exclusive scan of the free counts finds the lane
      uint32_t fw[PER]; int nfree = 0;
#pragma unroll
      for (int q = 0; q < PER; ++q) { fw[q] = ~forb[wv][lane * PER + q]; nfree += __popc(fw[q]); }
      int incl = nfree;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
      const int total = __shfl(incl, 63);
      const int want = total > 0 ? (int)(colour_hash(j, (uint32_t)round) % (uint32_t)(total < spread ? total : spread)) : 0;
      const int before = incl - nfree;
      int mine = -1;
      if (total > 0 && want >= before && want < incl) {
        int left = want - before;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const int c = __popc(fw[q]);
          if (mine < 0 && left < c) {
            uint32_t wbits = fw[q];
            for (int z = 0; z < left; ++z) wbits &= wbits - 1;   // drop the lowest `left` free bits
            mine = (lane * PER + q) * 32 + (__ffs((int)wbits) - 1);
          }
          left -= c;
        }
      }
      // (exactly one lane holds the answer)
      const unsigned long long who = __ballot(mine >= 0);
      if (who == 0ull) { if (lane == 0) { chosen[j] = -1; atomicExch(overflow, 1); } }
      else if (mine >= 0) chosen[j] = mine;
    }
    if (TEAM) __syncthreads(); else __builtin_amdgcn_wave_barrier();
  }
}
// lose[w] = 1: the feature's choice collides with a lower-indexed feature of the same round
__global__ __launch_bounds__(WG_THREADS) void colour_resolve_k(const uint32_t* __restrict__ act, int n_act, const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow,
                                                               const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const int* __restrict__ fixed,
                                                               const int* __restrict__ chosen, int* __restrict__ lose) {
  const int lane = threadIdx.x & 63;
  const int waves = (int)(gridDim.x * (WG_THREADS / 64));
  for (int w = (int)(((int64_t)blockIdx.x * WG_THREADS + threadIdx.x) >> 6); w < n_act; w += waves) {
    const uint32_t j = act[w];
    const int c = chosen[j];
    bool bad = c < 0;
    for (int64_t t = col_ptr[j] + lane; t < col_ptr[j + 1] && !bad; t += 64) {
      const uint32_t r = crow[t];
      const int64_t ub = row_ptr[r], ue = row_ptr[r + 1];
      for (int64_t u = ub; u < ue; u += 8) {
        uint32_t k[8]; int fx[8], ch[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) k[i] = col[u + i < ue ? u + i : ue - 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) { fx[i] = fixed[k[i]]; ch[i] = chosen[k[i]]; }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (u + i < ue && k[i] < j && fx[i] < 0 && ch[i] == c) bad = true;   // (features fixed earlier never hold c: the assignment excluded their colours)
      }
    }
    if (lane == 0) lose[w] = 0;
    if (__any(bad) && lane == 0) lose[w] = 1;
  }
}
// The same test ROW by row, for crowded rounds: two features collide iff they share a row, so a row that looks at the colours its own unfixed entries chose finds every
// collision it hosts -- 30 gathers per row instead of the 8 700 per feature of the walk above (10 M x 1 M i.i.d.: 0.3 G against 8.7 G reads in round 0).  LPR lanes per
// row (columns ascend inside a row: the LATER entry of an equal pair is the higher index and loses); lose_f is per FEATURE, zeroed before, flag stores only.
template <int LPR>
__global__ __launch_bounds__(WG_THREADS) void colour_resolve_rows_k(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, int64_t n, const int* __restrict__ fixed,
                                                                    const int* __restrict__ chosen, int* __restrict__ lose_f) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, g = lane / LPR, li = lane % LPR;
  const int64_t waves = (int64_t)gridDim.x * (WG_THREADS / 64);
  for (int64_t r0 = (((int64_t)blockIdx.x * WG_THREADS + threadIdx.x) >> 6) * RPW; r0 < n; r0 += waves * RPW) {   // (uniform per wave: the shuffles below see all lanes)
    const int64_t row = r0 + g;
    int c = -1; uint32_t k = 0;
    if (row < n) {
      const int64_t ub = row_ptr[row];
      if (li < (int)(row_ptr[row + 1] - ub)) {
        k = col[ub + li];
        c = fixed[k] < 0 ? chosen[k] : -1;   // an entry that is choosing this round
      }
    }
    bool bad = false;
#pragma unroll
    for (int d = 1; d < LPR; ++d) {
      const int o = __shfl_up(c, d, LPR);
      if (li >= d && c >= 0 && o == c) bad = true;
    }
    if (bad) lose_f[k] = 1;
  }
}
__global__ void colour_lose_gather_k(const uint32_t* __restrict__ act, int n_act, const int* __restrict__ lose_f, const int* __restrict__ chosen, int* __restrict__ lose) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n_act) lose[w] = (lose_f[act[w]] != 0 || chosen[act[w]] < 0) ? 1 : 0;
}
// winners are fixed, losers form the next round's list (in this round's order: the list stays ascending)
__global__ void colour_commit_k(const uint32_t* __restrict__ act, int n_act, const int* __restrict__ lose, const int* __restrict__ chosen, int* __restrict__ fixed) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n_act && !lose[w]) fixed[act[w]] = chosen[act[w]];
}

// ---- per-factor cache q = X v_f (rows parallel; same ascending-feature association as :291-299) -------------------
__global__ void als_q_init_k(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const float* __restrict__ val,
                             int64_t n, const double* __restrict__ V, int kp, int f, double2* __restrict__ qe) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double acc = 0.0;
  for (int64_t t = row_ptr[r]; t < row_ptr[r + 1]; ++t) acc += (double)val[t] * V[(size_t)col[t] * kp + f];
  qe[r].x = acc;
}

// q of factor f out of the all-factor table, kept FACTOR-major (Q[kp][n]: one coalesced 8-byte read per row; the row-major table of rounds 1-3
// cost a 128-byte line per row and factor: 1.28 GB read per pick at configs[4], 0.30 ms x 16)
__global__ void als_q_pick_k(const double* __restrict__ Qf, int64_t n, double2* __restrict__ qe) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) { const double2 c = qe[r]; qe[r] = make_double2(Qf[r], c.y); }
}

__global__ void als_pack_k(const double* __restrict__ err, int64_t n, double2* __restrict__ qe) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) qe[r] = make_double2(0.0, err[r]);
}

__global__ void als_unpack_k(const double2* __restrict__ qe, int64_t n, double* __restrict__ err) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) err[r] = qe[r].y;
}

__device__ __forceinline__ bool bad_number(double x) { return isnan(x) || isinf(x); }

// SweepDyn (fmx_internal.h): what changes from factor to factor (and from call to call) in a sweep lives in device memory and every sweep
// kernel reads it from there, so that the launches of one factor's sweep are IDENTICAL for every factor and every call -- a deep-level plan
// (i.i.d. columns: 8 155 dependent launches per factor) is then captured once as a HIP graph and replayed (sweep_graph below).
__global__ void als_set_dyn_k(SweepDyn* d, int f, double alpha, double lambda, double mu, const double* znorm) {
  d->f = f; d->pad = 0; d->alpha = alpha; d->lambda = lambda; d->mu = mu; d->znorm = znorm;
}

// one wave per feature of the level.  The first ALS_KEEP entries per lane (512 per wave: almost every column) stay in
// registers between the two passes, so the rank-1 correction pass does not gather q/e again.
constexpr int ALS_KEEP = 8;
__global__ __launch_bounds__(WG_THREADS) void als_level_k(const uint32_t* __restrict__ feats, int n_feats, const int64_t* __restrict__ col_ptr,
                                                          const uint32_t* __restrict__ crow, const float* __restrict__ cval,
                                                          double* __restrict__ V, int kp, const SweepDyn* __restrict__ dyn, double2* __restrict__ qe) {
  const int f = dyn->f;
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  const double* __restrict__ znorm = dyn->znorm;
  const int lane = threadIdx.x & 63;
  const int wid = (int)(((int64_t)blockIdx.x * WG_THREADS + threadIdx.x) >> 6);
  if (wid >= n_feats) return;
  const uint32_t i = feats[wid];
  const int64_t b = col_ptr[i], e = col_ptr[i + 1];
  const double v_old = V[(size_t)i * kp + f];
  double v_mean = 0.0, v_var = 0.0;
  float kx[ALS_KEEP];
  uint32_t kr[ALS_KEEP];
  double2 kc[ALS_KEEP];
  // :310-317, entries lane, lane+64, ...  Loads are unconditional on a clamped index and selected afterwards: `in ? cval[t] : 0`
  // compiles to a branch with a wait at its join, i.e. three dependent round trips PER SLOT instead of two for all of them
  // (DESIGN.md section 6.2).  Slots 0-1 (128 entries: every column of one-column-per-field data) always, slots 2.. only for longer columns.
  auto load_slots = [&](int s0, int s1) {
#pragma unroll
    for (int s = s0; s < s1; ++s) {
      const int64_t t = b + lane + 64 * s;
      const int64_t tc = t < e ? t : 0;
      kx[s] = cval[tc];
      kr[s] = crow[tc];
    }
#pragma unroll
    for (int s = s0; s < s1; ++s) kc[s] = qe[kr[s]];
#pragma unroll
    for (int s = s0; s < s1; ++s) {
      if (b + lane + 64 * s >= e) { kx[s] = 0.f; kc[s] = make_double2(0.0, 0.0); }
    }
  };
#pragma unroll
  for (int s = 0; s < ALS_KEEP; ++s) { kx[s] = 0.f; kr[s] = 0u; kc[s] = make_double2(0.0, 0.0); }
  load_slots(0, 2);
  if (e - b > 128) load_slots(2, ALS_KEEP);
#pragma unroll
  for (int s = 0; s < ALS_KEEP; ++s) {
    const float xx = kx[s] * kx[s];
    const double h = (double)kx[s] * kc[s].x - (double)xx * v_old;  // x = 0 for the padding slots: h = 0
    v_mean += h * kc[s].y;
    v_var += h * h;
  }
  for (int64_t t = b + lane + 64 * ALS_KEEP; t < e; t += 64) {
    const float x = cval[t];
    const uint32_t r = crow[t];
    const float xx = x * x;
    const double2 c = qe[r];
    const double h = (double)x * c.x - (double)xx * v_old;
    v_mean += h * c.y;
    v_var += h * h;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v_mean += __shfl_xor(v_mean, off);
    v_var += __shfl_xor(v_var, off);
  }
  v_mean -= v_old * v_var;                               // :318
  v_var = 1.0 / (lambda + alpha * v_var);                // :319
  v_mean = -v_var * (alpha * v_mean - mu * lambda);      // :320
  // :323-333: ALS takes the mean; MCMC draws Rf_rnorm(v_mean, sqrt(v_var)) = v_mean + sqrt(v_var) * (the caller's standard normal)
  double v_new = bad_number(v_var) ? 0.0 : (znorm ? v_mean + sqrt(v_var) * znorm[i] : v_mean);
  if (bad_number(v_new)) return;                         // CHECK_PARAM (:336): keep the old value, skip the corrections
  if (lane == 0) V[(size_t)i * kp + f] = v_new;
  const double v_diff = v_old - v_new;
#pragma unroll
  for (int s = 0; s < ALS_KEEP; ++s) {  // :341-350 from the kept entries
    if (b + lane + 64 * s < e) {
      const float xx = kx[s] * kx[s];
      const double h = (double)kx[s] * kc[s].x - (double)xx * v_old;
      qe[kr[s]] = make_double2(kc[s].x - (double)kx[s] * v_diff, kc[s].y - h * v_diff);
    }
  }
  for (int64_t t = b + lane + 64 * ALS_KEEP; t < e; t += 64) {
    const float x = cval[t];
    const uint32_t r = crow[t];
    const float xx = x * x;
    const double2 c = qe[r];
    const double h = (double)x * c.x - (double)xx * v_old;
    qe[r] = make_double2(c.x - (double)x * v_diff, c.y - h * v_diff);
  }
}

// ---- deep exact plans: the level loop inside ONE launch -----------------------------------------------------------------------------------
// (Two forms.  This one orders the LEVELS by a counter of completed features; als_exact_flow_k below, the default, orders the STEPS by the rows' own records and is
// a third faster -- 2.62 against 3.71 us per level; this one stays for rows whose ranks do not fit the records' tags and as FMX_ALS_PERSIST=counter.)
// The reference's index order on columns without field structure is a CHAIN: 19 399 dependent levels of at most 109 features at 10 M x 1 M, 30 per row.  One
// launch per level costs ~6.5 us (a wave's four dependent memory rounds -- feature id, column bounds, entries, (q, e) pairs -- plus the kernel boundary) for a
// microsecond of work: 310 384 launches, 2.0 s per sweep.  Here a factor's whole sweep is one launch of PERSIST_WAVES co-resident one-wave workgroups: wave g takes the
// features g, g + NW, ... of every level, and the levels are ordered by ONE monotonic counter of completed features -- a feature of level l may start once
// done >= level_ptr[l] (every feature of the levels before it has stored its corrections).  What a step reads that another wave wrote in this launch -- the
// (q, e) pairs -- moves write-through: 16-byte sc1 stores, drained (s_waitcnt vmcnt(0)) before the wave's agent-scope add, and sc1 loads issued after the poll
// has matched; no fence on either side (MI355X_MICROARCH.md, visibility: sc1 stores and loads + an atomic counter are valid under any workgroup placement).
// Everything static -- the level's feature, its column bounds, rows and values, its own V entry (only this step writes it) -- is fetched BEFORE the wait.
// Same arithmetic, same order per feature as als_level_k / als_w_level_k: the same bits (tests/test_gpu_configs4.py).
// Every wait is bounded: ~4e6 polls (seconds; a legitimate wait is microseconds) raise the abort word (the line after the replicas; STICKY until the host has read it, so
// a give-up in an early factor's launch is still known after the last one), every other wait then ends, and the host reports the sweep failed (persist_check).
// Workgroups of ONE wave: the wave that stores is the wave that drains and signals, the wave that polls is the wave that loads (the hand-off form the guide lists: one
// lane of each storing workgroup signals for all that workgroup's stores; the polling wave loads after its poll has matched).
// The counter is kept in PERSIST_REPL replicas, each on a line of its own: a finishing wave adds to every replica with ONE instruction (one lane per replica), a
// waiting wave polls the replica gw % PERSIST_REPL -- one word polled by all 128 waves and added to by a hundred per level is a queue at one memory channel.
#ifndef FMX_PERSIST_REPL
#define FMX_PERSIST_REPL 32
#endif
constexpr int PERSIST_WAVES = 128, PERSIST_REPL = FMX_PERSIST_REPL, PERSIST_LINE_WORDS = 32, PERSIST_CTL_WORDS = (PERSIST_REPL + 1) * PERSIST_LINE_WORDS;   // + the abort word's line
__device__ __forceinline__ double2 pair_load_sc1(__amdgpu_buffer_rsrc_t r, uint32_t row) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(row * 16u), 0, 16);   // aux 16 = sc1: past this CU's L1
  return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}
__device__ __forceinline__ void pair_store_sc1(__amdgpu_buffer_rsrc_t r, uint32_t row, double2 c) {
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t v;
  v.x = (uint32_t)__double2loint(c.x); v.y = (uint32_t)__double2hiint(c.x); v.z = (uint32_t)__double2loint(c.y); v.w = (uint32_t)__double2hiint(c.y);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(row * 16u), 0, 16);                  // write-through
}

// The wave sum of als_level_k -- x += shfl_xor(x, 32), 16, 8, 4, 2, 1 -- with the same tree, hence the same bits, without the LDS crossbar: the halves and the
// rows meet through gfx950's v_permlane32_swap / v_permlane16_swap (both results added: a + b == b + a), the lanes of a row through DPP row rotations
// (after the 32-, 16- and 8-steps a lane's value depends on its index mod 8 only, so the lane 4 (2, 1) to its right holds what lane ^ 4 (2, 1) holds).
__device__ __forceinline__ double butterfly_allsum(double x) {
  {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
  {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
  }
  x += FMX_DPP64(x, 0x128);   // row_ror:8
  x += FMX_DPP64(x, 0x124);   // row_ror:4
  x += FMX_DPP64(x, 0x122);   // row_ror:2
  x += FMX_DPP64(x, 0x121);   // row_ror:1
  return x;
}

template <bool W>
__global__ __launch_bounds__(64) void als_exact_persist_k(const uint32_t* __restrict__ feats, const int64_t* __restrict__ level_ptr, int L,
                                                          const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow, const float* __restrict__ cval,
                                                          double* __restrict__ P, int kp, const SweepDyn* __restrict__ dyn, double2* qe, uint32_t qe_bytes,
                                                          unsigned int* ctl, int debug_skip) {
  const int f = W ? 0 : dyn->f;
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  const double* __restrict__ znorm = dyn->znorm;
  const int lane = threadIdx.x;
  const int gw = (int)blockIdx.x, NW = (int)gridDim.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(qe, 0, (int)qe_bytes, 0x00020000);
  unsigned int spins = 0;
  unsigned int* const my_ctr = ctl + (gw % PERSIST_REPL) * PERSIST_LINE_WORDS;
  unsigned int* const abort_w = ctl + PERSIST_REPL * PERSIST_LINE_WORDS;
#ifdef FMX_PERSIST_TIMING
  unsigned long long tP = 0, tG = 0, tC = 0, tD = 0, nF = 0, nPoll = 0, c0 = __builtin_amdgcn_s_memtime(), c1;
#define FMX_TP(acc) do { c1 = __builtin_amdgcn_s_memtime(); acc += c1 - c0; c0 = c1; } while (0)
#else
#define FMX_TP(acc) do { } while (0)
#endif
  // this wave's features, in order: positions gw, gw + NW, ... of every level.  (l, l0, j) walks them; false at the end of the plan
  int l = 0;
  int64_t l0 = level_ptr[0], j = l0 + gw - NW;
  auto advance = [&]() {
    j += NW;
    for (;;) {
      const int64_t l1 = level_ptr[l + 1];
      if (j < l1) return true;
      if (++l >= L) return false;
      l0 = l1;
      j = l0 + gw;
    }
  };
  // The static part of a step -- the feature, its column's bounds, rows and values, its own parameter (only this step writes it), its normal -- is three
  // dependent memory rounds (2.4 us: as long as everything else of a step together).  It is fetched one feature AHEAD, each round issued behind one of the current
  // step's own waits (the poll, the pair gather, the store drain), so it costs the chain nothing.
  struct Stat { int64_t l0, b, e; uint32_t i; double v_old, zi; float kx[ALS_KEEP]; uint32_t kr[ALS_KEEP]; };
  auto round1 = [&](Stat& st) { st.l0 = l0; st.i = feats[j]; };
  auto round2 = [&](Stat& st) {
    st.b = col_ptr[st.i]; st.e = col_ptr[st.i + 1];
    st.v_old = P[W ? (size_t)st.i : (size_t)st.i * kp + f];
    st.zi = znorm ? znorm[st.i] : 0.0;
  };
  auto round3 = [&](Stat& st) {
#pragma unroll
    for (int s = 0; s < ALS_KEEP; ++s) {
      const int64_t t = st.b + lane + 64 * s;
      const int64_t tc = t < st.e ? t : 0;   // (unconditional on a clamped index: a branch would put a wait at its join)
      st.kx[s] = cval[tc];
      st.kr[s] = crow[tc];
    }
  };
  Stat cur, nxt;
  bool have = advance();
  if (have) { round1(cur); round2(cur); round3(cur); }
  while (have) {
    FMX_TP(tD);
    const bool have_next = advance();
    if (have_next) round1(nxt);
    // ---- every feature of the levels before this one has stored its corrections.  ONE poll at a time (four in flight a quarter of a trip apart were tried: the
    // replicas' lines then queue and a level takes 4.2 us instead of 3.8), and a wave whose turn is levels away -- on average 76 of the 128 have no feature in the
    // level being swept -- sleeps by its distance instead of competing with the waves whose turn is next
    {
      const unsigned int want = (unsigned int)cur.l0;
      for (;;) {
#ifdef FMX_PERSIST_TIMING
        ++nPoll;
#endif
        const unsigned int seen = __hip_atomic_load(my_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen >= want) break;
        if ((++spins & 255u) == 0) {
          if (spins > (1u << 22)) __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        }
        const unsigned int away = want - seen;      // features still to finish before this wave's turn (a level holds at most a few hundred)
        if (away > 512u) __builtin_amdgcn_s_sleep(127);
        else if (away > 160u) __builtin_amdgcn_s_sleep(40);
        else __builtin_amdgcn_s_sleep(2);
      }
    }
    spins = 0;
    FMX_TP(tP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: the loads below stay below the poll)
    const int64_t b = cur.b, e = cur.e;
    const double v_old = cur.v_old;
    const int ns = (int)((e - b + 63) >> 6);   // live slots of the column (wave-uniform): the padding slots' zero terms add nothing, so they are not computed
    // ---- the pairs of the column's rows: entries lane, lane + 64, ... (:310-317)
    double2 kc[ALS_KEEP];
#pragma unroll
    for (int s = 0; s < ALS_KEEP; ++s) kc[s] = pair_load_sc1(rs, cur.kr[s]);
    if (have_next) round2(nxt);
#pragma unroll
    for (int s = 0; s < ALS_KEEP; ++s) if (b + lane + 64 * s >= e) { cur.kx[s] = 0.f; kc[s] = make_double2(0.0, 0.0); }
#ifdef FMX_PERSIST_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    FMX_TP(tG);
    double a_mean = 0.0, a_var = 0.0;
    if constexpr (W) {
#pragma unroll
      for (int s = 0; s < ALS_KEEP; ++s) {
        if (s < ns && b + lane + 64 * s < e) {
          const double x = (double)cur.kx[s];
          a_mean += kc[s].y * x - v_old * x * x;
          a_var += x * x;
        }
      }
      for (int64_t t = b + lane + 64 * ALS_KEEP; t < e; t += 64) {
        const double x = (double)cval[t];
        const double2 c = pair_load_sc1(rs, crow[t]);
        a_mean += c.y * x - v_old * x * x;
        a_var += x * x;
      }
    } else {
#pragma unroll
      for (int s = 0; s < ALS_KEEP; ++s) {
        if (s < ns) {
          const float xx = cur.kx[s] * cur.kx[s];
          const double h = (double)cur.kx[s] * kc[s].x - (double)xx * v_old;  // x = 0 for a lane past the column's end: h = 0
          a_mean += h * kc[s].y;
          a_var += h * h;
        }
      }
      for (int64_t t = b + lane + 64 * ALS_KEEP; t < e; t += 64) {
        const float x = cval[t];
        const float xx = x * x;
        const double2 c = pair_load_sc1(rs, crow[t]);
        const double h = (double)x * c.x - (double)xx * v_old;
        a_mean += h * c.y;
        a_var += h * h;
      }
    }
    if (have_next) round3(nxt);
    a_mean = butterfly_allsum(a_mean);
    a_var = butterfly_allsum(a_var);
    double v_new;
    if constexpr (W) {
      a_var = 1.0 / (lambda + alpha * a_var);
      a_mean = -a_var * (alpha * a_mean - mu * lambda);
      v_new = bad_number(a_var) ? 0.0 : (znorm ? a_mean + a_var * cur.zi : a_mean);   // (:239: the variance where a standard deviation belongs; kept)
    } else {
      a_mean -= v_old * a_var;                               // :318
      a_var = 1.0 / (lambda + alpha * a_var);                // :319
      a_mean = -a_var * (alpha * a_mean - mu * lambda);      // :320
      v_new = bad_number(a_var) ? 0.0 : (znorm ? a_mean + sqrt(a_var) * cur.zi : a_mean);
    }
    if (!bad_number(v_new)) {                                // CHECK_PARAM (:336): otherwise keep the old value, skip the corrections
      if (lane == 0) P[W ? (size_t)cur.i : (size_t)cur.i * kp + f] = v_new;
      const double v_diff = v_old - v_new;
#pragma unroll
      for (int s = 0; s < ALS_KEEP; ++s) {                   // :341-350 from the kept entries
        if (b + lane + 64 * s < e) {
          if constexpr (W) pair_store_sc1(rs, cur.kr[s], make_double2(kc[s].x, kc[s].y - (double)cur.kx[s] * v_diff));
          else {
            const float xx = cur.kx[s] * cur.kx[s];
            const double h = (double)cur.kx[s] * kc[s].x - (double)xx * v_old;
            pair_store_sc1(rs, cur.kr[s], make_double2(kc[s].x - (double)cur.kx[s] * v_diff, kc[s].y - h * v_diff));
          }
        }
      }
      for (int64_t t = b + lane + 64 * ALS_KEEP; t < e; t += 64) {
        const float x = cval[t];
        const uint32_t r = crow[t];
        const double2 c = pair_load_sc1(rs, r);
        if constexpr (W) pair_store_sc1(rs, r, make_double2(c.x, c.y - (double)x * v_diff));
        else {
          const float xx = x * x;
          const double h = (double)x * c.x - (double)xx * v_old;
          pair_store_sc1(rs, r, make_double2(c.x - (double)x * v_diff, c.y - h * v_diff));
        }
      }
    }
    // ---- drained, then counted: whoever reads done >= level_ptr[l + 1] finds this feature's pairs in memory
    FMX_TP(tC);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < PERSIST_REPL && !(debug_skip && gw == 0 && cur.l0 == level_ptr[0]))   // (test hook: wave 0's first feature is never counted)
      __hip_atomic_fetch_add(ctl + lane * PERSIST_LINE_WORDS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef FMX_PERSIST_TIMING
    ++nF;
#endif
    cur = nxt;
    have = have_next;
  }
#ifdef FMX_PERSIST_TIMING
  if ((gw == 0 || gw == 77) && lane == 0 && nF) printf("persist wave %d: %llu features, %.1f polls each; memtime ticks per feature: poll %.0f  gather %.0f  step+stores %.0f  drain+add+loop %.0f\n", gw, nF,
                                        (double)nPoll / nF, (double)tP / nF, (double)tG / nF, (double)tC / nF, (double)tD / nF);
#endif
#undef FMX_TP
}

// ---- the same sweep with NO counter: the rows' records order the steps themselves ----------------------------------------------------------
// What a level of the counter form costs is four dependent fabric trips (store drain -> add -> poll -> pair gather) and the slowest of its ~51 waves.  But a step
// does not need "every feature of the levels before": it needs, for each of its rows, the correction of the ONE feature that touched that row last -- in the
// reference's index order the column before it in the row.  So the row's state carries that fact with it.  A row's record is 32 bytes, four 8-byte words
//     { q.lo | tag << 32,  q.hi | tag << 32,  e.lo | tag << 32,  e.hi | tag << 32 }        tag = how many features have corrected this row in this launch
// and entry t of a column knows its RANK inside its row (als_rank: uint16 per entry, built once per plan).  The step loads its rows' records (two sc1 b128 loads per
// row) and takes a record when all four tags equal the entry's rank -- else it loads that record again; it stores the corrected record with tag rank + 1 (two sc1
// b128 stores) and goes on: no drain, no add, no poll of anything but the data.  Needs nothing of the memory system but that an aligned 8-byte word is stored and
// loaded whole (a record whose four tags agree with the rank is the predecessor's record, word by word: tags only grow inside a launch, and nobody else writes the
// row between its predecessor and this step), and that an sc1 store becomes visible to an sc1 load in the end -- what the counter form asks of its counter.
// The wave that holds the lowest unfinished feature of the plan can always run (its predecessors are all of lower levels, hence finished), so the launch ends; every
// wait is bounded like the counter form's.  Same arithmetic in the same order per feature: the same bits as als_level_k.
typedef uint32_t rec_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rec_u32x4_t rec_load(__amdgpu_buffer_rsrc_t r, uint32_t row, int half) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)(row * 32u + (uint32_t)half * 16u), 0, 16);   // aux 16 = sc1
}
__device__ __forceinline__ void rec_store(__amdgpu_buffer_rsrc_t r, uint32_t row, int half, double x, uint32_t tag) {
  rec_u32x4_t v;
  v.x = (uint32_t)__double2loint(x); v.y = tag; v.z = (uint32_t)__double2hiint(x); v.w = tag;
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(row * 32u + (uint32_t)half * 16u), 0, 16);
}
__device__ __forceinline__ double rec_value(rec_u32x4_t v) { return __hiloint2double((int)v.z, (int)v.x); }
__device__ __forceinline__ bool rec_is(rec_u32x4_t v, uint32_t tag) { return v.y == tag && v.w == tag; }

__global__ void als_rec_pack_k(const double2* __restrict__ qe, uint32_t* __restrict__ rec, int64_t n) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const double2 c = qe[r];
  rec_u32x4_t a, b;
  a.x = (uint32_t)__double2loint(c.x); a.y = 0u; a.z = (uint32_t)__double2hiint(c.x); a.w = 0u;
  b.x = (uint32_t)__double2loint(c.y); b.y = 0u; b.z = (uint32_t)__double2hiint(c.y); b.w = 0u;
  reinterpret_cast<rec_u32x4_t*>(rec)[2 * r] = a;
  reinterpret_cast<rec_u32x4_t*>(rec)[2 * r + 1] = b;
}
__global__ void als_rec_unpack_k(const uint32_t* __restrict__ rec, double2* __restrict__ qe, int64_t n) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const rec_u32x4_t a = reinterpret_cast<const rec_u32x4_t*>(rec)[2 * r], b = reinterpret_cast<const rec_u32x4_t*>(rec)[2 * r + 1];
  qe[r] = make_double2(rec_value(a), rec_value(b));
}
// rank of every column-major entry inside its ROW (rows strictly ascending in col, a column's rows ascending: the entry of (r, j) in column j by bisection)
__global__ void als_rank_k(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, int64_t n, const int64_t* __restrict__ col_ptr,
                           const uint32_t* __restrict__ crow, uint16_t* __restrict__ rank, int* __restrict__ longest) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t b = row_ptr[r], e = row_ptr[r + 1];
  if (e - b > 65535) { atomicMax(longest, 65536); return; }
  for (int64_t u = b; u < e; ++u) {
    const uint32_t j = col[u];
    int64_t lo = col_ptr[j], hi = col_ptr[j + 1] - 1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (crow[mid] < (uint32_t)r) lo = mid + 1; else hi = mid;
    }
    rank[lo] = (uint16_t)(u - b);
  }
}

#ifndef FMX_FLOW_SLEEP
#define FMX_FLOW_SLEEP 1
#endif
constexpr int FLOW_WAVES = 256;
template <bool W>
__global__ __launch_bounds__(64) void als_exact_flow_k(const uint32_t* __restrict__ feats, const int64_t* __restrict__ level_ptr, int L,
                                                       const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow, const float* __restrict__ cval,
                                                       const uint16_t* __restrict__ crank, double* __restrict__ P, int kp, const SweepDyn* __restrict__ dyn,
                                                       uint32_t* rec, uint32_t rec_bytes, unsigned int* abort_w, int debug_skip) {
  const int f = W ? 0 : dyn->f;
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  // (a pointer read from memory is a generic one to the compiler: its loads would be FLAT, and a flat load in flight makes every vector-memory wait a vmcnt(0) --
  // the polls below could not overlap.  It is device memory.)
  const __attribute__((address_space(1))) double* const znorm = (const __attribute__((address_space(1))) double*)dyn->znorm;
  const int lane = threadIdx.x;
  const int gw = (int)blockIdx.x, NW = (int)gridDim.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rec, 0, (int)rec_bytes, 0x00020000);
  unsigned int spins = 0;
#ifdef FMX_PERSIST_TIMING
  unsigned long long tG = 0, tC = 0, tD = 0, nF = 0, nPoll = 0, c0 = __builtin_amdgcn_s_memtime(), c1;
#define FMX_TP(acc) do { c1 = __builtin_amdgcn_s_memtime(); acc += c1 - c0; c0 = c1; } while (0)
#else
#define FMX_TP(acc) do { } while (0)
#endif
  // this wave's features: positions gw, gw + NW, ... of the WHOLE plan (not of every level: nothing here is level-synchronous, and a wave that held a feature of every
  // level would be the chain itself -- its own serial step, 4.9 us, for every level)
  const int64_t first = level_ptr[0], last = level_ptr[L];
  int64_t j = first + gw - NW;
  auto advance = [&]() { j += NW; return j < last; };
  // the static part of a step, fetched one feature ahead (als_exact_persist_k), with the entries' ranks
  struct Stat { int64_t j, b, e; uint32_t i; double v_old, zi; float kx[ALS_KEEP]; uint32_t kr[ALS_KEEP], kt[ALS_KEEP]; };
  auto round1 = [&](Stat& st) { st.j = j; st.i = feats[j]; };
  auto round2 = [&](Stat& st) {
    st.b = col_ptr[st.i]; st.e = col_ptr[st.i + 1];
    st.v_old = P[W ? (size_t)st.i : (size_t)st.i * kp + f];
    st.zi = znorm ? znorm[st.i] : 0.0;
  };
  auto round3 = [&](Stat& st) {
#pragma unroll
    for (int s = 0; s < ALS_KEEP; ++s) {
      const int64_t t = st.b + lane + 64 * s;
      const int64_t tc = t < st.e ? t : 0;
      st.kx[s] = cval[tc];
      st.kr[s] = crow[tc];
      st.kt[s] = crank[tc];
    }
  };
  // a record of the column's tail (entries past the kept ones; a column of more than 512): waited for lane by lane
  auto wait_record = [&](uint32_t row, uint32_t tag, rec_u32x4_t& a, rec_u32x4_t& b) -> bool {
    for (;;) {
      if constexpr (!W) a = rec_load(rs, row, 0);
      b = rec_load(rs, row, 1);
      if ((W || rec_is(a, tag)) && rec_is(b, tag)) return true;
      if ((++spins & 255u) == 0) {
        if (spins > (1u << 22)) __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  };
  Stat cur, nxt;
  bool have = advance();
  if (have) { round1(cur); round2(cur); round3(cur); }
  while (have) {
    FMX_TP(tD);
    const bool have_next = advance();
    if (have_next) round1(nxt);
    const int64_t b = cur.b, e = cur.e;
    const double v_old = cur.v_old;
    const int ns = (int)((e - b + 63) >> 6);
    // ---- the records of the column's rows, each taken once its tags say the row's previous feature has corrected it
    rec_u32x4_t ha[ALS_KEEP], hb[ALS_KEEP];
#pragma unroll
    for (int s = 0; s < ALS_KEEP; ++s) {
      if (s < ns) {
        if constexpr (!W) ha[s] = rec_load(rs, cur.kr[s], 0);
        hb[s] = rec_load(rs, cur.kr[s], 1);
      }
    }
    if (have_next) round2(nxt);
    unsigned int late = 0;
#pragma unroll
    for (int s = 0; s < ALS_KEEP; ++s)
      if (s < ns && b + lane + 64 * s < e && !((W || rec_is(ha[s], cur.kt[s])) && rec_is(hb[s], cur.kt[s]))) late |= 1u << s;
    while (__builtin_amdgcn_ballot_w64(late != 0u) != 0ull) {
#ifdef FMX_PERSIST_TIMING
      ++nPoll;
#endif
      if ((++spins & 255u) == 0) {
        if (spins > (1u << 22)) __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
      }
      __builtin_amdgcn_s_sleep(FMX_FLOW_SLEEP);
#pragma unroll
      for (int s = 0; s < ALS_KEEP; ++s) {
        if (late & (1u << s)) {
          if constexpr (!W) ha[s] = rec_load(rs, cur.kr[s], 0);
          hb[s] = rec_load(rs, cur.kr[s], 1);
        }
      }
#pragma unroll
      for (int s = 0; s < ALS_KEEP; ++s)
        if ((late & (1u << s)) && (W || rec_is(ha[s], cur.kt[s])) && rec_is(hb[s], cur.kt[s])) late &= ~(1u << s);
    }
    spins = 0;
    double2 kc[ALS_KEEP];
#pragma unroll
    for (int s = 0; s < ALS_KEEP; ++s) {
      if (s < ns && b + lane + 64 * s < e) kc[s] = make_double2(W ? 0.0 : rec_value(ha[s]), rec_value(hb[s]));
      else { cur.kx[s] = 0.f; kc[s] = make_double2(0.0, 0.0); }
    }
    FMX_TP(tG);
    double a_mean = 0.0, a_var = 0.0;
    if constexpr (W) {
#pragma unroll
      for (int s = 0; s < ALS_KEEP; ++s) {
        if (s < ns && b + lane + 64 * s < e) {
          const double x = (double)cur.kx[s];
          a_mean += kc[s].y * x - v_old * x * x;
          a_var += x * x;
        }
      }
      for (int64_t t = b + lane + 64 * ALS_KEEP; t < e; t += 64) {
        const double x = (double)cval[t];
        rec_u32x4_t qa, qb;
        if (!wait_record(crow[t], crank[t], qa, qb)) return;
        a_mean += rec_value(qb) * x - v_old * x * x;
        a_var += x * x;
      }
    } else {
#pragma unroll
      for (int s = 0; s < ALS_KEEP; ++s) {
        if (s < ns) {
          const float xx = cur.kx[s] * cur.kx[s];
          const double h = (double)cur.kx[s] * kc[s].x - (double)xx * v_old;
          a_mean += h * kc[s].y;
          a_var += h * h;
        }
      }
      for (int64_t t = b + lane + 64 * ALS_KEEP; t < e; t += 64) {
        const float x = cval[t];
        const float xx = x * x;
        rec_u32x4_t qa, qb;
        if (!wait_record(crow[t], crank[t], qa, qb)) return;
        const double h = (double)x * rec_value(qa) - (double)xx * v_old;
        a_mean += h * rec_value(qb);
        a_var += h * h;
      }
    }
    if (have_next) round3(nxt);
    a_mean = butterfly_allsum(a_mean);
    a_var = butterfly_allsum(a_var);
    double v_new;
    if constexpr (W) {
      a_var = 1.0 / (lambda + alpha * a_var);
      a_mean = -a_var * (alpha * a_mean - mu * lambda);
      v_new = bad_number(a_var) ? 0.0 : (znorm ? a_mean + a_var * cur.zi : a_mean);
    } else {
      a_mean -= v_old * a_var;
      a_var = 1.0 / (lambda + alpha * a_var);
      a_mean = -a_var * (alpha * a_mean - mu * lambda);
      v_new = bad_number(a_var) ? 0.0 : (znorm ? a_mean + sqrt(a_var) * cur.zi : a_mean);
    }
    // CHECK_PARAM (:336): a bad value keeps the old one and skips the corrections -- the records still move on (their tags), unchanged
    const bool keep = bad_number(v_new);
    if (!keep && lane == 0) P[W ? (size_t)cur.i : (size_t)cur.i * kp + f] = v_new;
    const double v_diff = v_old - v_new;
    const uint32_t bump = (debug_skip && cur.j == first) ? 0u : 1u;   // (test hook: the plan's first feature leaves its rows' tags where they were)
#pragma unroll
    for (int s = 0; s < ALS_KEEP; ++s) {
      if (s < ns && b + lane + 64 * s < e) {
        double nq = kc[s].x, ne = kc[s].y;
        if (!keep) {
          if constexpr (W) ne = kc[s].y - (double)cur.kx[s] * v_diff;
          else {
            const float xx = cur.kx[s] * cur.kx[s];
            const double h = (double)cur.kx[s] * kc[s].x - (double)xx * v_old;
            nq = kc[s].x - (double)cur.kx[s] * v_diff;
            ne = kc[s].y - h * v_diff;
          }
        }
        if constexpr (!W) rec_store(rs, cur.kr[s], 0, nq, cur.kt[s] + bump);
        rec_store(rs, cur.kr[s], 1, ne, cur.kt[s] + bump);
      }
    }
    for (int64_t t = b + lane + 64 * ALS_KEEP; t < e; t += 64) {
      const float x = cval[t];
      const uint32_t r = crow[t], tag = crank[t];
      rec_u32x4_t qa, qb;
      if (!wait_record(r, tag, qa, qb)) return;
      double nq = W ? 0.0 : rec_value(qa), ne = rec_value(qb);
      if (!keep) {
        if constexpr (W) ne = ne - (double)x * v_diff;
        else {
          const float xx = x * x;
          const double h = (double)x * nq - (double)xx * v_old;
          const double q0 = nq;
          nq = q0 - (double)x * v_diff;
          ne = ne - h * v_diff;
        }
      }
      if constexpr (!W) rec_store(rs, r, 0, nq, tag + bump);
      rec_store(rs, r, 1, ne, tag + bump);
    }
    FMX_TP(tC);
#ifdef FMX_PERSIST_TIMING
    ++nF;
#endif
    cur = nxt;
    have = have_next;
  }
#ifdef FMX_PERSIST_TIMING
  if ((gw == 0 || gw == 77) && lane == 0 && nF) printf("flow wave %d: %llu features, %.1f re-polls each; memtime ticks per feature: records %.0f  step+stores %.0f  loop %.0f\n", gw, nF,
                                        (double)nPoll / nF, (double)tG / nF, (double)tC / nF, (double)tD / nF);
#endif
#undef FMX_TP
}

// ---- w0 and w sweeps of the ALS learner (MCMC_ALS_Learner.h:162-270, ALS branch, the exact one-thread form) -------------
__global__ void als_residual_k(const double* __restrict__ yhat, const float* __restrict__ y, int64_t n, double2* __restrict__ qe,
                               const double* __restrict__ dp_y) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double e;
  if (dp_y == nullptr) e = yhat[r] - (double)y[r];  // calculate_error, REGRESSION, :520-527
  else e = (y[r] >= 0.0f) ? -fast_dpnorm(dp_y, -yhat[r]) : fast_dpnorm(dp_y, yhat[r]);  // CLASSIFICATION, ALS learner, :545-559
  qe[r] = make_double2(0.0, e);
}

// the same from predictions in ANOTHER row order (position i holds row order[i]: the forward pass ran on the block form's permuted CSR)
__global__ void als_residual_perm_k(const double* __restrict__ yhat_perm, const float* __restrict__ y, const uint32_t* __restrict__ order, int64_t n, double2* __restrict__ qe,
                                    const double* __restrict__ dp_y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = order[i];
  const double yh = yhat_perm[i];
  double e;
  if (dp_y == nullptr) e = yh - (double)y[r];
  else e = (y[r] >= 0.0f) ? -fast_dpnorm(dp_y, -yh) : fast_dpnorm(dp_y, yh);
  qe[r] = make_double2(0.0, e);
}

constexpr int ALS_SLAB = 4096;
__global__ __launch_bounds__(WG_THREADS) void als_w0_partial_k(const double2* __restrict__ qe, int64_t n, const double* __restrict__ scal,
                                                              double* __restrict__ partials) {
  __shared__ double red[WG_THREADS];
  const double w0 = scal[SC_W0];
  const int64_t base = (int64_t)blockIdx.x * ALS_SLAB;
  double acc = 0.0;
  for (int i = threadIdx.x; i < ALS_SLAB; i += WG_THREADS) {
    const int64_t r = base + i;
    if (r < n) acc += qe[r].y - w0;  // :166-167
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = WG_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

// one workgroup: finish the sum, set the new w0, leave (old - new) in partials[n_partials]
__global__ __launch_bounds__(WG_THREADS) void als_w0_final_k(double* __restrict__ partials, int64_t n_partials, int64_t n, double* __restrict__ scal,
                                                            double reg0, double alpha, double w0_mean_0, int sample, double znorm) {
  __shared__ double red[WG_THREADS];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n_partials; i += WG_THREADS) acc += partials[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = WG_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double err = red[0];
  const double w0_var = 1.0 / (reg0 + alpha * (double)n);               // :169
  const double w0_mean = -(alpha * err - w0_mean_0 * reg0) * w0_var;    // :170
  const double w0_old = scal[SC_W0];
  double w0_new = sample ? w0_mean + sqrt(w0_var) * znorm : w0_mean;   // MCMC: Rf_rnorm(w0_mean, sqrt(w0_var)), :174-175
  if (isnan(w0_new) || isinf(w0_new)) w0_new = w0_old;                  // CHECK_PARAM, :180
  scal[SC_W0] = w0_new;
  partials[n_partials] = w0_old - w0_new;
}

__global__ void als_shift_k(double2* __restrict__ qe, int64_t n, const double* __restrict__ diff) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) qe[r].y -= *diff;  // :184-187
}

// one wave per feature of the level: w sweep, :208-256 with one thread's residual
__global__ __launch_bounds__(WG_THREADS) void als_w_level_k(const uint32_t* __restrict__ feats, int n_feats, const int64_t* __restrict__ col_ptr,
                                                            const uint32_t* __restrict__ crow, const float* __restrict__ cval,
                                                            double* __restrict__ w, double2* __restrict__ qe, const SweepDyn* __restrict__ dyn) {
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  const double* __restrict__ znorm = dyn->znorm;
  const int lane = threadIdx.x & 63;
  const int wid = (int)(((int64_t)blockIdx.x * WG_THREADS + threadIdx.x) >> 6);
  if (wid >= n_feats) return;
  const uint32_t i = feats[wid];
  const int64_t b = col_ptr[i], e = col_ptr[i + 1];
  const double w_old = w[i];
  double w_mean = 0.0, w_var = 0.0;
  constexpr int WU = 2;  // entries per lane whose reads are in flight together (unconditional, clamped: see als_level_k)
  for (int64_t t0 = b + lane; t0 < e; t0 += 64 * WU) {
    float xs[WU]; uint32_t rs[WU]; double es[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) { const int64_t t = t0 + 64 * u, tc = t < e ? t : t0; xs[u] = cval[tc]; rs[u] = crow[tc]; }
#pragma unroll
    for (int u = 0; u < WU; ++u) es[u] = qe[rs[u]].y;
#pragma unroll
    for (int u = 0; u < WU; ++u) {  // same additions, same order as one entry per iteration
      if (t0 + 64 * u < e) {
        const double x = (double)xs[u];
        w_mean += es[u] * x - w_old * x * x;
        w_var += x * x;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    w_mean += __shfl_xor(w_mean, off);
    w_var += __shfl_xor(w_var, off);
  }
  w_var = 1.0 / (lambda + alpha * w_var);
  w_mean = -w_var * (alpha * w_mean - mu * lambda);
  // MCMC: Rf_rnorm(w_mean, w_var), :239 -- the variance sits where a standard deviation belongs; kept
  const double w_new = bad_number(w_var) ? 0.0 : (znorm ? w_mean + w_var * znorm[i] : w_mean);
  if (bad_number(w_new)) return;  // CHECK_PARAM: keep the old value, skip the corrections
  if (lane == 0) w[i] = w_new;
  const double w_diff = w_old - w_new;
  for (int64_t t0 = b + lane; t0 < e; t0 += 64 * WU) {
    float xs[WU]; uint32_t rs[WU]; double es[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) { const int64_t t = t0 + 64 * u, tc = t < e ? t : t0; xs[u] = cval[tc]; rs[u] = crow[tc]; }
#pragma unroll
    for (int u = 0; u < WU; ++u) es[u] = qe[rs[u]].y;
#pragma unroll
    for (int u = 0; u < WU; ++u)
      if (t0 + 64 * u < e) qe[rs[u]].y = es[u] - (double)xs[u] * w_diff;
  }
}

// ---- heavy columns and the approximate (grouped) sweep -----------------------------------------------------------------------
// als_sweep_k is the general form of the two kernels above: T threads own one feature (a wave, or a whole workgroup for
// columns of more than ALS_HEAVY entries: a Zipf head feature holds millions, and one wave walking it serialises the level),
// W picks the w update (:208-256) or the V update (:303-350), APPROX the grouped form:
//
// When a matrix needs far more levels than it has entries per row (i.i.d. or Zipf columns: thousands of levels of a few
// hundred features each -- the exact schedule is then a chain of thousands of dependent launches per factor), the sweeps can
// run in the reference's OWN approximate parallel form instead (solver/MCMC_ALS_Learner.h:200-268: every OpenMP thread sweeps
// its features against a private copy of the residual, the copies are merged afterwards).  Here every feature of a GROUP is
// such a thread: all of them read the (q, e) of the group's start, each takes its exact coordinate step against that snapshot,
// and the corrections are merged into the next snapshot by atomic adds.  Groups = the largest position a feature takes in any
// of its rows (for one-column-per-field data that IS the exact level, and the grouped sweep equals the exact one), after the
// features with long columns, which are stepped one by one first (see build_plan); groups run in ascending order.  cfg.als_max_levels switches it on; results are then reproducible up to the order of the atomic adds.
constexpr int ALS_HEAVY = 4096;

template <bool W, bool APPROX, int T>
__global__ __launch_bounds__(WG_THREADS) void als_sweep_k(const uint32_t* __restrict__ feats, int n_feats, const int64_t* __restrict__ col_ptr,
                                                          const uint32_t* __restrict__ crow, const float* __restrict__ cval, double* __restrict__ P, int kp,
                                                          const SweepDyn* __restrict__ dyn, const double2* qe_old, double2* qe_new) {
  const int f = dyn->f;
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  const double* __restrict__ znorm = dyn->znorm;
  __shared__ double red[2][WG_THREADS / 64];
  const int tid = threadIdx.x % T;
  const int wid = (int)(((int64_t)blockIdx.x * WG_THREADS + threadIdx.x) / T);
  if (wid >= n_feats) return;  // (T == 256: the whole workgroup leaves together)
  const uint32_t i = feats[wid];
  const int64_t b = col_ptr[i], e = col_ptr[i + 1];
  const size_t at = W ? (size_t)i : (size_t)i * kp + f;
  const double old = P[at];
  double mean = 0.0, var = 0.0;
  constexpr int UN = 4;
  for (int64_t t0 = b + tid; t0 < e; t0 += (int64_t)T * UN) {
    float x[UN]; double2 c[UN];
    uint32_t rr[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {  // unconditional on a clamped index, selected afterwards (see als_level_k)
      const int64_t t = t0 + (int64_t)u * T, tc = t < e ? t : t0;
      x[u] = cval[tc];
      rr[u] = crow[tc];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) c[u] = qe_old[rr[u]];
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (t0 + (int64_t)u * T >= e) { x[u] = 0.f; c[u] = make_double2(0.0, 0.0); }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (W) { const double xd = (double)x[u]; mean += c[u].y * xd - old * xd * xd; var += xd * xd; }
      else { const float xx = x[u] * x[u]; const double h = (double)x[u] * c[u].x - (double)xx * old; mean += h * c[u].y; var += h * h; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mean += __shfl_xor(mean, off); var += __shfl_xor(var, off); }
  if (T > 64) {
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = mean; red[1][wv] = var; }
    __syncthreads();
    mean = 0.0; var = 0.0;
    for (int q = 0; q < WG_THREADS / 64; ++q) { mean += red[0][q]; var += red[1][q]; }
  }
  double nv;
  if (W) {
    var = 1.0 / (lambda + alpha * var);
    mean = -var * (alpha * mean - mu * lambda);
    nv = bad_number(var) ? 0.0 : (znorm ? mean + var * znorm[i] : mean);      // (the variance as Rf_rnorm's sd: :239, kept)
  } else {
    mean -= old * var;
    var = 1.0 / (lambda + alpha * var);
    mean = -var * (alpha * mean - mu * lambda);
    nv = bad_number(var) ? 0.0 : (znorm ? mean + sqrt(var) * znorm[i] : mean);
  }
  if (bad_number(nv)) return;  // CHECK_PARAM
  if (tid == 0) P[at] = nv;
  const double diff = old - nv;
  for (int64_t t0 = b + tid; t0 < e; t0 += (int64_t)T * UN) {
    float xs[UN]; uint32_t rs[UN]; double2 cs[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) { const int64_t t = t0 + (int64_t)u * T, tc = t < e ? t : t0; xs[u] = cval[tc]; rs[u] = crow[tc]; }
#pragma unroll
    for (int u = 0; u < UN; ++u) cs[u] = qe_old[rs[u]];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t t = t0 + (int64_t)u * T;
      if (t >= e) continue;
      const float x = xs[u];
      const uint32_t r = rs[u];
      const double2 c = cs[u];
      double dq, de;
      if (W) { dq = 0.0; de = (double)x * diff; }
      else { const float xx = x * x; const double h = (double)x * c.x - (double)xx * old; dq = (double)x * diff; de = h * diff; }
      if (APPROX) {
        double* dst = reinterpret_cast<double*>(qe_new + r);
        if (!W) unsafeAtomicAdd(dst, -dq);
        unsafeAtomicAdd(dst + 1, -de);
      } else {
        qe_new[r] = make_double2(c.x - dq, c.y - de);
      }
    }
  }
}

// ---- very long columns, exact form: one column over MANY workgroups -----------------------------------------------------------
// A Zipf head feature holds a million entries; one workgroup walking it is milliseconds per pass, and such features sit in
// thousands of consecutive levels (a dependent chain of launches).  Columns of more than ALS_SPLIT (16 384) entries are therefore cut
// into segments of ALS_SPLIT_SEG entries: als_vh_partial_k sums a segment (the first pass of als_sweep_k), als_vh_value_k adds
// a feature's segment sums in segment order and takes the coordinate step, als_vh_apply_k applies the rank-1 corrections
// segment by segment.  Features of one level share no row, so the segments of a level never write the same (q, e).  Fixed
// geometry, fixed order: reproducible; the association of the two sums differs from the one-workgroup form (oracle parity 1e-10).
constexpr int64_t ALS_SPLIT = 16384, ALS_SPLIT_SEG = 8192;  // (65536 / 16384: Zipf exact sweep 4.0 s; one workgroup per column: 6.7 s)

template <bool W>
__global__ __launch_bounds__(WG_THREADS) void als_vh_partial_k(const uint32_t* __restrict__ vh, const uint32_t* __restrict__ seg_feat,
                                                               const int64_t* __restrict__ seg_b, const int64_t* __restrict__ seg_e, int64_t seg0,
                                                               const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow,
                                                               const float* __restrict__ cval, const double* __restrict__ P, int kp,
                                                               const SweepDyn* __restrict__ dyn, const double2* __restrict__ qe, double* __restrict__ partial) {
  const int f = dyn->f;
  __shared__ double red[2][WG_THREADS / 64];
  const int64_t sg = seg0 + blockIdx.x;
  const uint32_t i = vh[seg_feat[sg]];
  const int64_t b = col_ptr[i] + seg_b[sg], e = col_ptr[i] + seg_e[sg];
  const double old = P[W ? (size_t)i : (size_t)i * kp + f];
  double mean = 0.0, var = 0.0;
  constexpr int UN = 4;
  for (int64_t t0 = b + threadIdx.x; t0 < e; t0 += (int64_t)WG_THREADS * UN) {
    float x[UN]; uint32_t rr[UN]; double2 c[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) { const int64_t t = t0 + (int64_t)u * WG_THREADS, tc = t < e ? t : t0; x[u] = cval[tc]; rr[u] = crow[tc]; }
#pragma unroll
    for (int u = 0; u < UN; ++u) c[u] = qe[rr[u]];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (t0 + (int64_t)u * WG_THREADS >= e) continue;
      if (W) { const double xd = (double)x[u]; mean += c[u].y * xd - old * xd * xd; var += xd * xd; }
      else { const float xx = x[u] * x[u]; const double h = (double)x[u] * c[u].x - (double)xx * old; mean += h * c[u].y; var += h * h; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mean += __shfl_xor(mean, off); var += __shfl_xor(var, off); }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = mean; red[1][wv] = var; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ms = 0.0, vs = 0.0;
    for (int q = 0; q < WG_THREADS / 64; ++q) { ms += red[0][q]; vs += red[1][q]; }
    partial[2 * sg] = ms; partial[2 * sg + 1] = vs;
  }
}

template <bool W>
__global__ void als_vh_value_k(const uint32_t* __restrict__ vh, const uint32_t* __restrict__ seg_first, int64_t h0, int n, const double* __restrict__ partial,
                               double* __restrict__ P, int kp, const SweepDyn* __restrict__ dyn, double* __restrict__ v_old, double* __restrict__ v_diff) {
  const int f = dyn->f;
  const double alpha = dyn->alpha, lambda = dyn->lambda, mu = dyn->mu;
  const double* __restrict__ znorm = dyn->znorm;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int64_t hx = h0 + q;
  const uint32_t i = vh[hx];
  const size_t at = W ? (size_t)i : (size_t)i * kp + f;
  const double old = P[at];
  double mean = 0.0, var = 0.0;
  for (uint32_t sg = seg_first[hx]; sg < seg_first[hx + 1]; ++sg) { mean += partial[2 * (size_t)sg]; var += partial[2 * (size_t)sg + 1]; }
  double nv;
  if (W) {
    var = 1.0 / (lambda + alpha * var);
    mean = -var * (alpha * mean - mu * lambda);
    nv = bad_number(var) ? 0.0 : (znorm ? mean + var * znorm[i] : mean);      // (the variance as Rf_rnorm's sd: :239, kept)
  } else {
    mean -= old * var;
    var = 1.0 / (lambda + alpha * var);
    mean = -var * (alpha * mean - mu * lambda);
    nv = bad_number(var) ? 0.0 : (znorm ? mean + sqrt(var) * znorm[i] : mean);
  }
  v_old[hx] = old;
  if (bad_number(nv)) { v_diff[hx] = 0.0; return; }  // CHECK_PARAM: keep the old value, no corrections
  P[at] = nv;
  v_diff[hx] = old - nv;
}

template <bool W>
__global__ __launch_bounds__(WG_THREADS) void als_vh_apply_k(const uint32_t* __restrict__ vh, const uint32_t* __restrict__ seg_feat,
                                                             const int64_t* __restrict__ seg_b, const int64_t* __restrict__ seg_e, int64_t seg0,
                                                             const int64_t* __restrict__ col_ptr, const uint32_t* __restrict__ crow,
                                                             const float* __restrict__ cval, const double* __restrict__ v_old,
                                                             const double* __restrict__ v_diff, double2* __restrict__ qe) {
  const int64_t sg = seg0 + blockIdx.x;
  const uint32_t hx = seg_feat[sg];
  const double diff = v_diff[hx];
  if (diff == 0.0) return;  // nothing moves (or CHECK_PARAM refused the step)
  const uint32_t i = vh[hx];
  const double old = v_old[hx];
  const int64_t b = col_ptr[i] + seg_b[sg], e = col_ptr[i] + seg_e[sg];
  constexpr int UN = 4;
  for (int64_t t0 = b + threadIdx.x; t0 < e; t0 += (int64_t)WG_THREADS * UN) {
    float xs[UN]; uint32_t rs[UN]; double2 cs[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) { const int64_t t = t0 + (int64_t)u * WG_THREADS, tc = t < e ? t : t0; xs[u] = cval[tc]; rs[u] = crow[tc]; }
#pragma unroll
    for (int u = 0; u < UN; ++u) cs[u] = qe[rs[u]];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (t0 + (int64_t)u * WG_THREADS >= e) continue;
      const float x = xs[u];
      const double2 c = cs[u];
      double dq, de;
      if (W) { dq = 0.0; de = (double)x * diff; }
      else { const float xx = x * x; const double h = (double)x * c.x - (double)xx * old; dq = (double)x * diff; de = h * diff; }
      qe[rs[u]] = make_double2(c.x - dq, c.y - de);
    }
  }
}

// largest position (0-based) each feature takes inside a row: the groups of the approximate sweep
__global__ void als_maxpos_k(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, int64_t n, int* __restrict__ level) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t b = row_ptr[r];
  for (int64_t t = b; t < row_ptr[r + 1]; ++t) atomicMax(&level[col[t]], (int)(t - b));
}

// the level plan depends on the matrix only: built once, kept in the fmx_matrix
static int build_plan(fmx_matrix* m, hipStream_t stream, int max_levels = 0) {
  if (m->als_force_exact) max_levels = 0;  // an approximate sweep of this matrix raised the residual once: exact from then on
  if (m->als_feats && m->als_plan_cap == max_levels) {
    if (!m->als_tiled_tried) { m->als_tiled_tried = 1; FMX_TRY(als_tiled_build(m, stream)); }  // (the values changed since: the tiled lists hold copies)
    return FMX_OK;
  }
  als_tiled_free(m); m->als_tiled_tried = 0;
  (void)hipFree(m->als_feats); m->als_feats = nullptr;
  (void)hipFree(m->als_level_ptr_dev); m->als_level_ptr_dev = nullptr;
  (void)hipFree(m->als_rank); m->als_rank = nullptr; m->als_rank_state = 0;
  (void)hipFree(m->als_heavy); m->als_heavy = nullptr;
  (void)hipFree(m->als_vh); (void)hipFree(m->als_vh_seg0); (void)hipFree(m->als_vseg_feat); (void)hipFree(m->als_vseg_b); (void)hipFree(m->als_vseg_e); (void)hipFree(m->als_vh_work);
  m->als_vh = nullptr; m->als_vh_seg0 = nullptr; m->als_vseg_feat = nullptr; m->als_vseg_b = nullptr; m->als_vseg_e = nullptr; m->als_vh_work = nullptr;
  m->als_n_vh = 0; m->als_n_vseg = 0;
  const uint32_t p = m->p;
  int *d_level = nullptr, *d_changed = nullptr;
  FMX_HIP(hipMalloc(&d_level, (size_t)p * sizeof(int)));
  FMX_HIP(hipMalloc(&d_changed, sizeof(int)));
  FMX_HIP(hipMemsetAsync(d_level, 0, (size_t)p * sizeof(int), stream));  // features that never occur stay at level 0
  bool approx = false;
  bool coloured = false;
  // the exact schedule by a frontier walk; cap > 0: give up after cap (+ one batch of) rounds -- a deep chain
  auto frontier_walk = [&](int cap, bool& gave_up) -> int {
    // frontier walk: rounds are enqueued CHECK at a time (an empty frontier makes a round a no-op), then the number of features
    // placed so far is read back; done when every occurring feature has its level
    struct Tmp {
      int *pos = nullptr, *ready = nullptr, *cnt = nullptr;
      uint32_t *f0 = nullptr, *f1 = nullptr, *heavy = nullptr;
      unsigned long long* assigned = nullptr;
      ~Tmp() { (void)hipFree(pos); (void)hipFree(ready); (void)hipFree(cnt); (void)hipFree(f0); (void)hipFree(f1); (void)hipFree(assigned); (void)hipFree(heavy); }
    } w;
    FMX_HIP(hipMalloc(&w.pos, (size_t)m->n * sizeof(int)));
    FMX_HIP(hipMalloc(&w.ready, (size_t)p * sizeof(int)));
    FMX_HIP(hipMalloc(&w.cnt, 5 * sizeof(int)));  // three rotating frontier counts, the heavy list's count, its done flag
    FMX_HIP(hipMalloc(&w.f0, (size_t)p * sizeof(uint32_t)));
    FMX_HIP(hipMalloc(&w.f1, (size_t)p * sizeof(uint32_t)));
    FMX_HIP(hipMalloc(&w.assigned, sizeof(unsigned long long)));
    FMX_HIP(hipMemsetAsync(w.pos, 0, (size_t)m->n * sizeof(int), stream));
    FMX_HIP(hipMemsetAsync(w.ready, 0, (size_t)p * sizeof(int), stream));
    FMX_HIP(hipMemsetAsync(w.cnt, 0, 5 * sizeof(int), stream));
    FMX_HIP(hipMemsetAsync(w.assigned, 0, sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(level_heads_k, dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, stream, m->row_ptr, m->col, m->n, m->col_ptr, w.ready, w.f0, w.cnt);
    std::vector<int64_t> cph((size_t)p + 1);
    FMX_HIP(hipMemcpyAsync(cph.data(), m->col_ptr, cph.size() * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    FMX_HIP(hipStreamSynchronize(stream));
    unsigned long long occurring = 0, n_heavy = 0;
    for (uint32_t j = 0; j < p; ++j) {
      occurring += cph[(size_t)j + 1] > cph[(size_t)j] ? 1 : 0;
      n_heavy += cph[(size_t)j + 1] - cph[(size_t)j] > LEVEL_HEAVY ? 1 : 0;
    }
    FMX_HIP(hipMalloc(&w.heavy, (n_heavy ? n_heavy : 1) * sizeof(uint32_t)));
    const int CHECK = 64;
    int round = 0;
    for (;;) {
      for (int q = 0; q < CHECK; ++q, ++round)
      {
        hipLaunchKernelGGL(level_round_k, dim3(512), dim3(WG_THREADS), 0, stream, round % 2 ? w.f1 : w.f0, round % 2 ? w.f0 : w.f1, w.cnt, round, m->row_ptr, m->col,
                           m->col_ptr, m->crow, w.pos, w.ready, d_level, w.assigned, w.heavy, w.cnt + 3);
        if (n_heavy)
          hipLaunchKernelGGL(level_heavy_k, dim3(512), dim3(WG_THREADS), 0, stream, w.heavy, w.cnt + 3, round % 2 ? w.f0 : w.f1, w.cnt, round, m->row_ptr, m->col,
                             m->col_ptr, m->crow, w.pos, w.ready, w.cnt + 4);
      }
      unsigned long long done = 0;
      FMX_HIP(hipMemcpyAsync(&done, w.assigned, sizeof(done), hipMemcpyDeviceToHost, stream));
      FMX_HIP(hipStreamSynchronize(stream));
      // (`assigned` counts the frontier each round STARTED with: the last round's features are placed once the count is complete)
      if (done >= occurring) break;
      if (cap > 0 && round >= cap + CHECK) { gave_up = true; break; }
      FMX_CHECK((int64_t)round <= (int64_t)p + 2 * CHECK, FMX_ERR_STATE, "level scheduling did not converge (a row holds a column twice?)");
    }
    return FMX_OK;
  };
  if (max_levels < 0 && m->n > 0 && m->nnz > 0) {
    // cfg.als_max_levels < 0: the sweep may choose its own feature order -- levels = the colours of a proper colouring of the "share a row" graph (kernels above).
    // Falls through to the exact schedule below where it does not apply (a column too long for one wave per feature, more colours than the kernel's set).
    // The levels of the EXACT schedule are themselves a proper colouring -- in the reference's own feature order -- and on field-structured data (one column per field
    // and row) there are as many as fields: taken when the walk finishes within COLOUR_EXACT_ROUNDS rounds (a speculative colouring spreads such data over a
    // thousand classes; the sweep's numbers are then the reference's for -1).  A deep chain (i.i.d. columns: 19 399 levels at 10 M x 1 M) gives up after two read-backs.
    {
      constexpr int COLOUR_EXACT_ROUNDS = 64;
      bool gave_up = false;
      FMX_TRY(frontier_walk(COLOUR_EXACT_ROUNDS, gave_up));
      if (!gave_up) coloured = true;
      else FMX_HIP(hipMemsetAsync(d_level, 0, (size_t)p * sizeof(int), stream));
    }
    if (!coloured) {
    std::vector<int64_t> cph((size_t)p + 1);
    FMX_HIP(hipMemcpy(cph.data(), m->col_ptr, cph.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    // Columns too long for one wave per feature (the heads of a skewed distribution: they meet almost every row, and each other) take a colour of their own each,
    // first, in index order -- an exact pass over the heavy features, as the approximate form orders them -- and the light features colour around them.
    std::vector<uint32_t> act, heavy_first;
    for (uint32_t j = 0; j < p; ++j) {
      const int64_t len = cph[(size_t)j + 1] - cph[(size_t)j];
      if (len > LEVEL_HEAVY) heavy_first.push_back(j);
      else if (len > 0) act.push_back(j);
    }
    const bool too_long = heavy_first.size() > (size_t)COLOUR_MAX / 2;
    if (!too_long && !(act.empty() && heavy_first.empty())) {
      struct Tmp {
        int *fixed = nullptr, *chosen = nullptr, *lose = nullptr, *overflow = nullptr, *lose_f = nullptr; uint32_t* act = nullptr;
        ~Tmp() { (void)hipFree(fixed); (void)hipFree(chosen); (void)hipFree(lose); (void)hipFree(overflow); (void)hipFree(act); (void)hipFree(lose_f); }
      } w;
      const size_t n_occ = act.size() + heavy_first.size();
      FMX_HIP(hipMalloc(&w.fixed, (size_t)p * sizeof(int))); FMX_HIP(hipMalloc(&w.chosen, (size_t)p * sizeof(int))); FMX_HIP(hipMalloc(&w.lose, (n_occ ? n_occ : 1) * sizeof(int)));
      FMX_HIP(hipMalloc(&w.overflow, sizeof(int))); FMX_HIP(hipMalloc(&w.act, (n_occ ? n_occ : 1) * sizeof(uint32_t)));
      FMX_HIP(hipMalloc(&w.lose_f, (size_t)p * sizeof(int)));
      const char* rows_env = getenv("FMX_COLOUR_ROWS");   // =0: the collisions of every round by the feature walk (read per plan: the tests compare the plans)
      const bool rows_ok = m->max_row_len <= 64 && !(rows_env && rows_env[0] == '0');
      {
        std::vector<int> init(p, -1);
        for (size_t i = 0; i < heavy_first.size(); ++i) init[heavy_first[i]] = (int)i;
        FMX_HIP(hipMemcpy(w.fixed, init.data(), (size_t)p * sizeof(int), hipMemcpyHostToDevice));
      }
      FMX_HIP(hipMemsetAsync(w.chosen, 0xFF, (size_t)p * sizeof(int), stream));
      FMX_HIP(hipMemsetAsync(w.overflow, 0, sizeof(int), stream));
      std::vector<int> h_lose;
      bool failed = false;
      for (int round = 0; !act.empty(); ++round) {
        const int na = (int)act.size();
        FMX_HIP(hipMemcpyAsync(w.act, act.data(), (size_t)na * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        const unsigned grid = (unsigned)(na < 4 * 2048 ? (na + 3) / 4 : 2048);
        // a colour holds at most n / (rows per feature) features: with S colours on offer a round fixes at most S classes' worth -- S grows with the crowd (150 rounds at S = 8)
        const int spread = na / 2048 < COLOUR_SPREAD ? COLOUR_SPREAD : (na / 2048 > 1024 ? 1024 : na / 2048);
        const int scan = (round == 0 && heavy_first.empty()) ? 0 : 1;
        if (na < COLOUR_TEAM_BELOW) hipLaunchKernelGGL((colour_assign_k<true>), dim3((unsigned)na), dim3(WG_THREADS), 0, stream, (const uint32_t*)w.act, na, (const int64_t*)m->col_ptr, (const uint32_t*)m->crow,
                                                       (const int64_t*)m->row_ptr, (const uint32_t*)m->col, (const int*)w.fixed, w.chosen, round, spread, w.overflow, scan);
        else hipLaunchKernelGGL((colour_assign_k<false>), dim3(grid), dim3(WG_THREADS), 0, stream, (const uint32_t*)w.act, na, (const int64_t*)m->col_ptr, (const uint32_t*)m->crow,
                                (const int64_t*)m->row_ptr, (const uint32_t*)m->col, (const int*)w.fixed, w.chosen, round, spread, w.overflow, scan);
        // a crowded round asks the ROWS for its collisions (30 reads per row) -- a feature's walk reads 30 per row of its list: cheaper once few features are left
        if (rows_ok && (double)na * ((double)m->nnz / (double)(p ? p : 1)) > (double)m->n) {
          FMX_HIP(hipMemsetAsync(w.lose_f, 0, (size_t)p * sizeof(int), stream));
          if (m->max_row_len <= 32) hipLaunchKernelGGL((colour_resolve_rows_k<32>), dim3(4096), dim3(WG_THREADS), 0, stream, (const int64_t*)m->row_ptr, (const uint32_t*)m->col, m->n, (const int*)w.fixed, (const int*)w.chosen, w.lose_f);
          else hipLaunchKernelGGL((colour_resolve_rows_k<64>), dim3(4096), dim3(WG_THREADS), 0, stream, (const int64_t*)m->row_ptr, (const uint32_t*)m->col, m->n, (const int*)w.fixed, (const int*)w.chosen, w.lose_f);
          hipLaunchKernelGGL(colour_lose_gather_k, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, stream, (const uint32_t*)w.act, na, (const int*)w.lose_f, (const int*)w.chosen, w.lose);
        } else
        hipLaunchKernelGGL(colour_resolve_k, dim3(grid), dim3(WG_THREADS), 0, stream, (const uint32_t*)w.act, na, (const int64_t*)m->col_ptr, (const uint32_t*)m->crow,
                           (const int64_t*)m->row_ptr, (const uint32_t*)m->col, (const int*)w.fixed, (const int*)w.chosen, w.lose);
        hipLaunchKernelGGL(colour_commit_k, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, stream, (const uint32_t*)w.act, na, (const int*)w.lose, (const int*)w.chosen, w.fixed);
        h_lose.resize((size_t)na);
        int h_over = 0;
        FMX_HIP(hipMemcpyAsync(h_lose.data(), w.lose, (size_t)na * sizeof(int), hipMemcpyDeviceToHost, stream));
        FMX_HIP(hipMemcpyAsync(&h_over, w.overflow, sizeof(int), hipMemcpyDeviceToHost, stream));
        FMX_HIP(hipStreamSynchronize(stream));
        if (h_over || round > 4096) { failed = true; break; }
        if (getenv("FMX_COLOUR_DEBUG")) fprintf(stderr, "colouring round %d: %d features choosing\n", round, na);
        std::vector<uint32_t> next;
        for (int i = 0; i < na; ++i) if (h_lose[(size_t)i]) next.push_back(act[(size_t)i]);
        act.swap(next);
      }
      // The spread that makes the rounds converge also spreads the colours (one-column-per-field data: 240 colours where 30 do).  REFIT, class by class from the
      // highest colour down: the features of one colour share no row, so all of them may move to their smallest free colour at once -- nothing they read changes
      // in that launch -- and the colouring stays proper, never grows, and usually loses most of its upper classes (four passes).
      const char* refit_env = getenv("FMX_COLOUR_REFIT");   // passes of the refit (default 4; read per plan: profiles/r05_colour_refit.txt)
      const int refit_passes = refit_env ? atoi(refit_env) : 4;
      for (int pass = 0; pass < refit_passes && !failed; ++pass) {
        std::vector<int> col_now(p);
        FMX_HIP(hipMemcpy(col_now.data(), w.fixed, (size_t)p * sizeof(int), hipMemcpyDeviceToHost));
        int top = -1;
        for (uint32_t j = 0; j < p; ++j) if (col_now[j] > top) top = col_now[j];
        std::vector<std::vector<uint32_t>> cls((size_t)(top + 1));
        for (uint32_t j = 0; j < p; ++j) if (col_now[j] >= 0 && cph[(size_t)j + 1] - cph[(size_t)j] <= LEVEL_HEAVY) cls[(size_t)col_now[j]].push_back(j);   // (the heavy features keep their own colours)
        std::vector<uint32_t> flat; std::vector<size_t> at((size_t)top + 2, 0);
        for (int c = 0; c <= top; ++c) { at[(size_t)c] = flat.size(); flat.insert(flat.end(), cls[(size_t)c].begin(), cls[(size_t)c].end()); }
        at[(size_t)top + 1] = flat.size();
        if (flat.empty()) break;
        FMX_HIP(hipMemcpyAsync(w.act, flat.data(), flat.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));   // (w.act holds every occurring feature: flat is no longer)
        for (int c = top; c >= 1; --c) {
          const int na = (int)(at[(size_t)c + 1] - at[(size_t)c]);
          if (na == 0) continue;
          const unsigned grid = (unsigned)(na < 4 * 2048 ? (na + 3) / 4 : 2048);
          if (na < COLOUR_TEAM_BELOW) hipLaunchKernelGGL((colour_assign_k<true>), dim3((unsigned)na), dim3(WG_THREADS), 0, stream, (const uint32_t*)(w.act + at[(size_t)c]), na, (const int64_t*)m->col_ptr,
                                                         (const uint32_t*)m->crow, (const int64_t*)m->row_ptr, (const uint32_t*)m->col, (const int*)w.fixed, w.fixed, 0, 1, w.overflow, 1);
          else hipLaunchKernelGGL((colour_assign_k<false>), dim3(grid), dim3(WG_THREADS), 0, stream, (const uint32_t*)(w.act + at[(size_t)c]), na, (const int64_t*)m->col_ptr, (const uint32_t*)m->crow,
                                  (const int64_t*)m->row_ptr, (const uint32_t*)m->col, (const int*)w.fixed, w.fixed, 0, 1, w.overflow, 1);
        }
        int h_over = 0;
        FMX_HIP(hipMemcpyAsync(&h_over, w.overflow, sizeof(int), hipMemcpyDeviceToHost, stream));
        FMX_HIP(hipStreamSynchronize(stream));
        if (h_over) failed = true;
      }
      if (!failed) {
        // colours -> levels 0 .. L - 1 without gaps (ascending colour); features that never occur stay at level 0
        std::vector<int> col_of(p);
        FMX_HIP(hipMemcpy(col_of.data(), w.fixed, (size_t)p * sizeof(int), hipMemcpyDeviceToHost));
        std::vector<int> remap((size_t)COLOUR_MAX, -1);
        for (uint32_t j = 0; j < p; ++j) if (col_of[j] >= 0) remap[(size_t)col_of[j]] = 0;
        int nl = 0;
        for (int c = 0; c < COLOUR_MAX; ++c) if (remap[(size_t)c] == 0) remap[(size_t)c] = nl++;
        for (uint32_t j = 0; j < p; ++j) col_of[j] = col_of[j] >= 0 ? remap[(size_t)col_of[j]] : 0;
        FMX_HIP(hipMemcpy(d_level, col_of.data(), (size_t)p * sizeof(int), hipMemcpyHostToDevice));
        if (getenv("FMX_COLOUR_DEBUG")) {   // the classes' sizes: a launch of the feature-major form is as long as its rounds of resident workgroups
          std::vector<int> sz((size_t)(nl > 0 ? nl : 1), 0);
          for (uint32_t j = 0; j < p; ++j) if (cph[(size_t)j + 1] > cph[(size_t)j]) sz[(size_t)col_of[j]]++;
          fprintf(stderr, "colour classes (%d):", nl);
          for (int c = 0; c < nl; c += (nl > 64 ? nl / 64 : 1)) fprintf(stderr, " %d", sz[(size_t)c]);
          fprintf(stderr, "\n");
        }
        coloured = true;
      }
    }
    }
  }
  const char* lv_env = getenv("FMX_ALS_LEVELS");  // FMX_ALS_LEVELS=relax: the round-1 builder (read per call: the tests compare the two)
  const bool relax = lv_env && lv_env[0] == 'r';
  if (coloured) {
  } else if (relax) {
    // monotone relaxation to the fixed point; every sweep propagates along whole rows.  The "changed" flag is read back once per
    // CHECK sweeps.  max_levels > 0: give up after that many sweeps (a deep chain: i.i.d. or Zipf columns) and fall back to the
    // grouped sweep, whose groups need one pass.
    const int CHECK = 8;
    int64_t sweeps = 0;
    for (;;) {
      int h = 0;
      FMX_HIP(hipMemsetAsync(d_changed, 0, sizeof(int), stream));
      for (int q = 0; q < CHECK; ++q)
        if (m->n > 0) hipLaunchKernelGGL(level_relax_k, dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, stream, m->row_ptr, m->col, m->n, d_level, d_changed);
      FMX_HIP(hipMemcpyAsync(&h, d_changed, sizeof(int), hipMemcpyDeviceToHost, stream));
      FMX_HIP(hipStreamSynchronize(stream));
      if (!h) break;
      sweeps += CHECK;
      if (max_levels > 0 && sweeps >= max_levels) { approx = true; break; }
      FMX_CHECK(sweeps <= (int64_t)p + CHECK, FMX_ERR_STATE, "level scheduling did not converge");
    }
  } else if (m->n > 0 && m->nnz > 0) {
    bool gave_up = false;
    FMX_TRY(frontier_walk(max_levels > 0 ? max_levels : 0, gave_up));
    if (gave_up) approx = true;
  }
  std::vector<int> level(p);
  FMX_HIP(hipMemcpy(level.data(), d_level, (size_t)p * sizeof(int), hipMemcpyDeviceToHost));
  if (!approx && max_levels > 0) {  // converged, but with more levels than asked for?
    int L = 0;
    for (uint32_t j = 0; j < p; ++j) if (level[j] + 1 > L) L = level[j] + 1;
    approx = L > max_levels;
  }
  if (approx) {
    FMX_HIP(hipMemsetAsync(d_level, 0, (size_t)p * sizeof(int), stream));
    if (m->n > 0) hipLaunchKernelGGL(als_maxpos_k, dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, stream, m->row_ptr, m->col, m->n, d_level);
    FMX_HIP(hipMemcpyAsync(level.data(), d_level, (size_t)p * sizeof(int), hipMemcpyDeviceToHost, stream));
    FMX_HIP(hipStreamSynchronize(stream));
  }
  (void)hipFree(d_level); (void)hipFree(d_changed);
  std::vector<int64_t> cp((size_t)p + 1);
  FMX_HIP(hipMemcpy(cp.data(), m->col_ptr, cp.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (approx) {
    // Features with long columns meet almost every row, and each other: stepping them against one snapshot overshoots (measured:
    // a Zipf(1.05) matrix diverges within two sweeps).  They go FIRST, one group each, in index order -- an exact Gauss-Seidel
    // pass over the heavy features -- and the many rare features follow in their position groups.
    int H = 0;
    for (uint32_t j = 0; j < p; ++j) if (cp[(size_t)j + 1] - cp[(size_t)j] > ALS_HEAVY) ++H;
    int h_at = 0;
    for (uint32_t j = 0; j < p; ++j) {
      if (cp[(size_t)j + 1] - cp[(size_t)j] > ALS_HEAVY) level[j] = h_at++;
      else level[j] += H;
    }
  }
  int L = 0;
  for (uint32_t j = 0; j < p; ++j) if (level[j] + 1 > L) L = level[j] + 1;
  // per level: the light features (one wave each), the heavy ones (one workgroup each) and -- exact plan only -- the very long
  // columns, cut into segments (one workgroup per segment: als_vh_*_k); ascending index inside a level
  const char* split_env = getenv("FMX_ALS_SPLIT");  // read per call: the tests compare the two forms
  const bool split_ok = !(split_env && split_env[0] == '0');
  std::vector<std::vector<uint32_t>> light((size_t)L), heavy((size_t)L), vheavy((size_t)L);
  for (uint32_t j = 0; j < p; ++j) {
    const int64_t len = cp[(size_t)j + 1] - cp[(size_t)j];
    (len > ALS_SPLIT && !approx && split_ok ? vheavy : len > ALS_HEAVY ? heavy : light)[(size_t)level[j]].push_back(j);
  }
  std::vector<uint32_t> fl, fh, fv, seg_first(1, 0u), seg_feat;
  std::vector<int64_t> seg_b, seg_e;
  m->als_level_ptr.assign((size_t)L + 1, 0);
  m->als_heavy_ptr.assign((size_t)L + 1, 0);
  m->als_vh_ptr.assign((size_t)L + 1, 0);
  m->als_vseg_ptr.assign((size_t)L + 1, 0);
  for (int l = 0; l < L; ++l) {
    fl.insert(fl.end(), light[(size_t)l].begin(), light[(size_t)l].end());
    fh.insert(fh.end(), heavy[(size_t)l].begin(), heavy[(size_t)l].end());
    for (uint32_t j : vheavy[(size_t)l]) {
      const int64_t len = cp[(size_t)j + 1] - cp[(size_t)j];
      for (int64_t b = 0; b < len; b += ALS_SPLIT_SEG) {
        seg_feat.push_back((uint32_t)fv.size());
        seg_b.push_back(b);
        seg_e.push_back(b + ALS_SPLIT_SEG < len ? b + ALS_SPLIT_SEG : len);
      }
      fv.push_back(j);
      seg_first.push_back((uint32_t)seg_feat.size());
    }
    m->als_level_ptr[(size_t)l + 1] = (int64_t)fl.size();
    m->als_heavy_ptr[(size_t)l + 1] = (int64_t)fh.size();
    m->als_vh_ptr[(size_t)l + 1] = (int64_t)fv.size();
    m->als_vseg_ptr[(size_t)l + 1] = (int64_t)seg_feat.size();
  }
  FMX_HIP(hipMalloc(&m->als_feats, (fl.size() ? fl.size() : 1) * sizeof(uint32_t)));
  if (!fl.empty()) FMX_HIP(hipMemcpy(m->als_feats, fl.data(), fl.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  FMX_HIP(hipMalloc(&m->als_heavy, (fh.size() ? fh.size() : 1) * sizeof(uint32_t)));
  if (!fh.empty()) FMX_HIP(hipMemcpy(m->als_heavy, fh.data(), fh.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  m->als_n_vh = (int64_t)fv.size(); m->als_n_vseg = (int64_t)seg_feat.size();
  if (!fv.empty()) {
    FMX_HIP(hipMalloc(&m->als_vh, fv.size() * sizeof(uint32_t)));
    FMX_HIP(hipMalloc(&m->als_vh_seg0, seg_first.size() * sizeof(uint32_t)));
    FMX_HIP(hipMalloc(&m->als_vseg_feat, seg_feat.size() * sizeof(uint32_t)));
    FMX_HIP(hipMalloc(&m->als_vseg_b, seg_b.size() * sizeof(int64_t)));
    FMX_HIP(hipMalloc(&m->als_vseg_e, seg_e.size() * sizeof(int64_t)));
    FMX_HIP(hipMalloc(&m->als_vh_work, (2 * seg_feat.size() + 2 * fv.size()) * sizeof(double)));
    FMX_HIP(hipMemcpy(m->als_vh, fv.data(), fv.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    FMX_HIP(hipMemcpy(m->als_vh_seg0, seg_first.data(), seg_first.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    FMX_HIP(hipMemcpy(m->als_vseg_feat, seg_feat.data(), seg_feat.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    FMX_HIP(hipMemcpy(m->als_vseg_b, seg_b.data(), seg_b.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    FMX_HIP(hipMemcpy(m->als_vseg_e, seg_e.data(), seg_e.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  }
  m->als_level_maxlen.assign((size_t)L, 0);
  for (int l = 0; l < L; ++l)
    for (uint32_t j : light[(size_t)l]) { const int64_t len = cp[(size_t)j + 1] - cp[(size_t)j]; if (len > m->als_level_maxlen[(size_t)l]) m->als_level_maxlen[(size_t)l] = len; }
  m->als_approx = approx ? 1 : 0;
  m->als_coloured = coloured ? 1 : 0;
  m->als_plan_cap = max_levels;
  m->als_level_of.assign(level.begin(), level.end());
  m->als_tiled_tried = 1;
  return als_tiled_build(m, stream);  // wide levels of an exact plan: the row-tiled form (fm_als_tiled.hip)
}

// One pass over all features of the plan for the w sweep (W) or one factor of the V sweep: levels (exact) or groups (approximate)
// in ascending order; the factor, alpha, lambda, mu and the normals come from *dyn (device).  qe_new: second (q, e) array of the
// approximate form (the merged corrections land there; it is copied over the snapshot after every group), unused by the exact form.
template <bool W>
static void sweep_features(fmx_engine* e, fmx_matrix* m, double2* d_qe, double2* d_qe_new, const SweepDyn* dyn, bool profile = true) {
  const std::vector<int64_t>& lp = m->als_level_ptr;
  const std::vector<int64_t>& hp = m->als_heavy_ptr;
  const int L = (int)lp.size() - 1;
  double* P = W ? e->dw : e->dV;
  bool synced = true;  // approximate form: d_qe_new holds what d_qe holds (the caller copied it)
  for (int l = 0; l < L; ++l) {
    const int64_t cnt = lp[(size_t)l + 1] - lp[(size_t)l], hcnt = hp[(size_t)l + 1] - hp[(size_t)l];
    const int64_t vcnt = m->als_vh_ptr.empty() ? 0 : m->als_vh_ptr[(size_t)l + 1] - m->als_vh_ptr[(size_t)l];  // (the approximate plan has none)
    if (cnt + hcnt + vcnt == 0) continue;
    const uint32_t* lf = m->als_feats + lp[(size_t)l];
    const uint32_t* hf = m->als_heavy + hp[(size_t)l];
    const dim3 gl((unsigned)((cnt * 64 + WG_THREADS - 1) / WG_THREADS)), gh((unsigned)hcnt), blk(WG_THREADS);
    struct ProfEnd { fmx_engine* e; bool on; ~ProfEnd() { if (on) prof_end(e); } } prof_guard{e, profile};
    if (profile) prof_begin(e, FMX_KERNEL_ALS_SWEEP);  // one level (or group) of one factor: the unit bench.py --solver als prices
    if (!m->als_approx && m->als_tiled) {
      bool done = false;
      bool last = true;   // no later level of this sweep holds a feature?
      for (int l2 = l + 1; l2 < L && last; ++l2)
        last = lp[(size_t)l2 + 1] - lp[(size_t)l2] + hp[(size_t)l2 + 1] - hp[(size_t)l2] + (m->als_vh_ptr.empty() ? 0 : m->als_vh_ptr[(size_t)l2 + 1] - m->als_vh_ptr[(size_t)l2]) == 0;
      if (als_tiled_level<W>(e, m, l, last, d_qe, dyn, &done) == FMX_OK && done) continue;
    }
    if (!m->als_approx) {
      if (cnt > 0) {
        if (W) hipLaunchKernelGGL(als_w_level_k, gl, blk, 0, e->stream, lf, (int)cnt, m->col_ptr, m->crow, m->cval, e->dw, d_qe, dyn);
        else hipLaunchKernelGGL(als_level_k, gl, blk, 0, e->stream, lf, (int)cnt, m->col_ptr, m->crow, m->cval, e->dV, e->kp64, dyn, d_qe);
      }
      // (a heavy feature shares rows with nearly everything: it is alone in its level, or with a few other heavy ones)
      if (hcnt > 0) hipLaunchKernelGGL((als_sweep_k<W, false, WG_THREADS>), gh, blk, 0, e->stream, hf, (int)hcnt, m->col_ptr, m->crow, m->cval, P, e->kp64, dyn,
                                       (const double2*)d_qe, d_qe);
      if (vcnt > 0) {  // the very long columns of the level, over many workgroups each
        const int64_t v0 = m->als_vh_ptr[(size_t)l], s0 = m->als_vseg_ptr[(size_t)l], ns = m->als_vseg_ptr[(size_t)l + 1] - s0;
        double* partial = m->als_vh_work;
        double* v_old = m->als_vh_work + 2 * m->als_n_vseg;
        double* v_diff = v_old + m->als_n_vh;
        hipLaunchKernelGGL((als_vh_partial_k<W>), dim3((unsigned)ns), blk, 0, e->stream, m->als_vh, m->als_vseg_feat, m->als_vseg_b, m->als_vseg_e, s0, m->col_ptr,
                           m->crow, m->cval, (const double*)P, e->kp64, dyn, (const double2*)d_qe, partial);
        hipLaunchKernelGGL((als_vh_value_k<W>), dim3((unsigned)((vcnt + 63) / 64)), dim3(64), 0, e->stream, m->als_vh, m->als_vh_seg0, v0, (int)vcnt, (const double*)partial, P,
                           e->kp64, dyn, v_old, v_diff);
        hipLaunchKernelGGL((als_vh_apply_k<W>), dim3((unsigned)ns), blk, 0, e->stream, m->als_vh, m->als_vseg_feat, m->als_vseg_b, m->als_vseg_e, s0, m->col_ptr,
                           m->crow, m->cval, (const double*)v_old, (const double*)v_diff, d_qe);
      }
    } else if (cnt + hcnt == 1) {
      // a group of one: its step against "the snapshot" is the exact step -- in place, no merge (the heavy features' pass)
      if (hcnt) hipLaunchKernelGGL((als_sweep_k<W, false, WG_THREADS>), gh, blk, 0, e->stream, hf, 1, m->col_ptr, m->crow, m->cval, P, e->kp64, dyn,
                                   (const double2*)d_qe, d_qe);
      else hipLaunchKernelGGL((als_sweep_k<W, false, 64>), gl, blk, 0, e->stream, lf, 1, m->col_ptr, m->crow, m->cval, P, e->kp64, dyn,
                              (const double2*)d_qe, d_qe);
      synced = false;
    } else {
      if (!synced) { (void)hipMemcpyAsync(d_qe_new, d_qe, (size_t)m->n * sizeof(double2), hipMemcpyDeviceToDevice, e->stream); synced = true; }
      if (cnt > 0) hipLaunchKernelGGL((als_sweep_k<W, true, 64>), gl, blk, 0, e->stream, lf, (int)cnt, m->col_ptr, m->crow, m->cval, P, e->kp64, dyn,
                                      (const double2*)d_qe, d_qe_new);
      if (hcnt > 0) hipLaunchKernelGGL((als_sweep_k<W, true, WG_THREADS>), gh, blk, 0, e->stream, hf, (int)hcnt, m->col_ptr, m->crow, m->cval, P, e->kp64, dyn,
                                       (const double2*)d_qe, d_qe_new);
      (void)hipMemcpyAsync(d_qe, d_qe_new, (size_t)m->n * sizeof(double2), hipMemcpyDeviceToDevice, e->stream);  // the next group's snapshot
    }
  }
}

// the device struct the sweep kernels read their per-factor values from
static SweepDyn* sweep_dyn(fmx_engine* e) {
  if (!e->als_dyn && hipMalloc(&e->als_dyn, sizeof(SweepDyn)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return reinterpret_cast<SweepDyn*>(e->als_dyn);
}
static void set_dyn(fmx_engine* e, SweepDyn* dyn, int f, double alpha, double lambda, double mu, const double* znorm) {
  hipLaunchKernelGGL(als_set_dyn_k, dim3(1), dim3(1), 0, e->stream, dyn, f, alpha, lambda, mu, znorm);
}

// ---- deep plans as a HIP graph ---------------------------------------------------------------------------------------------------
// An exact plan over i.i.d. columns has thousands of levels of a few hundred features (4 M x 1 M, 30 per row: 8 155 levels): one
// sweep of one factor is 8 155 dependent launches of a microsecond each, and eager launches cost the host 3-4 us apiece -- the sweep
// is HOST-bound (0.65 s for 16 factors = 5 us per level; a same-stream kernel boundary is 1.5 us on the device).  Since every
// launch of a factor's sweep is identical for all factors and calls (SweepDyn), the sequence is captured ONCE per (plan, buffers)
// and replayed: the device walks it at its own pace.  Same kernels, same order: same bits (tests/test_gpu_configs4.py).
// MEASURED (profiles/r03_als_graph_probe.txt): 8 155 nodes capture and instantiate in 13 ms, and the replayed sweep takes the SAME
// 0.655 s as the eager one -- the sweep was never host-bound: the host enqueues a launch in 3.5 us, a level takes 5 us on the device
// (its wave's chain of four dependent memory rounds: feature id -> column bounds -> entries -> (q, e) pairs), so the host runs
// ahead either way.  The replay is therefore OFF unless FMX_ALS_GRAPH=1 asks for it; the record stays as the answer to "would a
// graph (or a persistent level loop, whose grid barrier costs more than the 1.5 us kernel boundary) help": no.
constexpr int ALS_GRAPH_MIN_LEVELS = 64;
struct AlsGraph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  uint64_t matrix_uid = 0;
  const void *qe = nullptr, *feats = nullptr;
  int plan_cap = -2, levels = 0;
};
void als_graph_free(void* p) {
  AlsGraph* g = reinterpret_cast<AlsGraph*>(p);
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
}
template <bool W>
static bool sweep_graph_wanted(const fmx_matrix* m) {
  if (m->als_approx || m->als_tiled) return false;   // (a tiled level folds e->als_qnext and sizes its workspace on the host: not capturable)
  const char* v = getenv("FMX_ALS_GRAPH");
  return v && v[0] == '1' && (int)m->als_level_ptr.size() - 1 >= ALS_GRAPH_MIN_LEVELS;
}
// the replayable form of sweep_features<W>(e, m, d_qe, nullptr, dyn): null when capture is not possible (the caller then launches eagerly)
template <bool W>
static hipGraphExec_t sweep_graph(fmx_engine* e, fmx_matrix* m, double2* d_qe, const SweepDyn* dyn) {
  void*& slot = W ? e->als_graph_w : e->als_graph_v;
  AlsGraph* g = reinterpret_cast<AlsGraph*>(slot);
  const int L = (int)m->als_level_ptr.size() - 1;
  if (g && g->exec && g->matrix_uid == m->uid && g->qe == d_qe && g->feats == m->als_feats && g->plan_cap == m->als_plan_cap && g->levels == L) return g->exec;
  als_graph_free(g);
  slot = nullptr;
  g = new AlsGraph();
  const bool verbose = getenv("FMX_ALS_GRAPH_VERBOSE") != nullptr;
  timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  hipError_t err = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal);
  if (err != hipSuccess) {
    if (verbose) fprintf(stderr, "fmx: als graph: begin capture failed: %s\n", hipGetErrorString(err));
    (void)hipGetLastError(); delete g; return nullptr;
  }
  sweep_features<W>(e, m, d_qe, nullptr, dyn, /*profile=*/false);
  err = hipStreamEndCapture(e->stream, &g->graph);
  if (err == hipSuccess && g->graph != nullptr) err = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (err != hipSuccess || g->graph == nullptr) {
    if (verbose) fprintf(stderr, "fmx: als graph: capture / instantiate failed: %s\n", hipGetErrorString(err));
    (void)hipGetLastError();
    als_graph_free(g);
    return nullptr;
  }
  if (verbose) {
    size_t nodes = 0;
    (void)hipGraphGetNodes(g->graph, nullptr, &nodes);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    fprintf(stderr, "fmx: als graph (%s sweep): %zu nodes for %d levels, captured and instantiated in %.1f ms\n", W ? "w" : "V", nodes, L,
            1e3 * ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec)));
  }
  g->matrix_uid = m->uid; g->qe = d_qe; g->feats = m->als_feats; g->plan_cap = m->als_plan_cap; g->levels = L;
  slot = g;
  return g->exec;
}

// does the feature-major form (cfg.als_max_levels = -2) run on this plan?  A coloured plan of light lists of at most 1 024 rows; otherwise the sweep nests factor outer, as -1
constexpr uint64_t ALLF_TRUSTED = 0xA11FA11FA11FA11Full;   // e->als_q_trusted of a row-major table: the matrix uid under this mask (a block-form table carries its plan's uid)
// Levels whose lists the register kernels do not take (kp other than 8 / 16, or more than 512 rows) keep the rows' lines in LDS: (cap kp + 2 kp) doubles + cap row ids, which must fit
// the 150 KB the kernel is allowed -- 1 024 rows at kp <= 16, 577 at kp = 32, 296 at kp = 64, 148 at kp = 128 (the bound used to ignore kp, and a level that did not fit failed
// at its launch, in the middle of a sweep).  A plan with such a level nests factor outer, as -1.
constexpr size_t ALLF_LDS_LIMIT = 150 * 1024;
static size_t allf_lds_bytes(int64_t maxlen, int kp) {
  const size_t cap = (size_t)((maxlen + 63) / 64 * 64);
  return (cap * (size_t)kp + 2 * (size_t)kp) * sizeof(double) + cap * sizeof(uint32_t);
}
static bool allf_applies(const fmx_engine* e, const fmx_matrix* m) {
  if (!m->als_coloured || m->als_plan_cap != -2 || e->k <= 0) return false;
  if (m->als_heavy_ptr.empty() || m->als_heavy_ptr.back() != 0 || (!m->als_vh_ptr.empty() && m->als_vh_ptr.back() != 0)) return false;
  const bool in_regs = e->kp64 == 8 || e->kp64 == 16;
  for (int64_t v : m->als_level_maxlen) {
    if (v > 1024) return false;
    if (!(in_regs && v <= 512) && allf_lds_bytes(v, e->kp64) > ALLF_LDS_LIMIT) return false;
  }
  return true;
}

// Does the persistent form take this plan?  A deep exact plan of light columns only (no workgroup-wide or segmented columns, no tiled levels), a pair table a
// buffer descriptor can address.  FMX_ALS_PERSIST=0 keeps one launch per level (the tests compare the two: the same bits).
constexpr int PERSIST_MIN_LEVELS = 64;
static std::atomic<int> g_stall_next_persistent_sweep{0};
void debug_stall_next_persistent_sweep() { g_stall_next_persistent_sweep.store(1); }
static bool persist_applies(const fmx_matrix* m) {
  const char* v = getenv("FMX_ALS_PERSIST");
  if (v && v[0] == '0') return false;
  if (m->als_approx || m->als_tiled) return false;
  if ((int)m->als_level_ptr.size() - 1 < PERSIST_MIN_LEVELS) return false;
  if (!m->als_heavy_ptr.empty() && m->als_heavy_ptr.back() != 0) return false;
  if (!m->als_vh_ptr.empty() && m->als_vh_ptr.back() != 0) return false;
  // NARROW levels only: a wave takes a level's positions g, g + 128, ... one after the other, so a level of a thousand features (a coloured plan's classes) would be eight
  // dependent steps per wave where one launch per level runs them side by side
  int64_t widest = 0;
  for (size_t l = 0; l + 1 < m->als_level_ptr.size(); ++l) widest = std::max(widest, m->als_level_ptr[l + 1] - m->als_level_ptr[l]);
  if (widest > 2 * PERSIST_WAVES) return false;
  return (uint64_t)m->n * sizeof(double2) <= 0xFFFFFFF0ull && m->als_level_ptr.back() < (int64_t)0xFFFFFFFFll;
}
// A persistent sweep's waves wait for one another: every one of them must be RUNNING.  No more one-wave workgroups than the device holds of this kernel at once (its
// registers decide: als_exact_flow_k<false> takes 227, eight waves per CU -- 256 of them fit a whole MI355X eight times over, a 32-CU partition exactly); both forms
// work with any number of waves.  (Cached per device; 0 from the runtime = unknown: as asked.)
template <bool FLOW, bool W>   // (one cache per kernel: the two W variants of a form have the same function type)
static int resident_waves(const fmx_engine* e, int want) {
  constexpr int MAX_DEV = 64;
  static int cap[MAX_DEV] = {};
  const int slot = (e->cfg.device >= 0 && e->cfg.device < MAX_DEV) ? e->cfg.device : 0;
  if (cap[slot] == 0) {
    int cus = 0, per = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->cfg.device) != hipSuccess) cus = 0;
    hipError_t st;
    if constexpr (FLOW) st = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, als_exact_flow_k<W>, 64, 0);
    else st = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, als_exact_persist_k<W>, 64, 0);
    if (st != hipSuccess) per = 0;
    (void)hipGetLastError();
    cap[slot] = cus > 0 && per > 0 ? cus * per : -1;
  }
  return cap[slot] > 0 && cap[slot] < want ? cap[slot] : want;
}
// the record-ordered form (als_exact_flow_k) where the rows' ranks fit its tags; FMX_ALS_PERSIST=counter keeps the counter form (als_exact_persist_k)
static int flow_prepare(fmx_engine* e, fmx_matrix* m, bool& ok) {
  ok = false;
  const char* v = getenv("FMX_ALS_PERSIST");
  if (v && v[0] == 'c') return FMX_OK;
  // (the ranks are those of the reference's INDEX order -- the feature before an entry's in its row is the column before it; a coloured plan visits in another order)
  if ((uint64_t)m->n * 32ull > 0xFFFFFFF0ull || !m->rows_sorted || m->als_coloured) return FMX_OK;
  if (m->als_rank_state == 0) {
    int* d_longest = nullptr;
    FMX_HIP(hipMalloc(&d_longest, sizeof(int)));
    FMX_HIP(hipMemsetAsync(d_longest, 0, sizeof(int), e->stream));
    FMX_HIP(hipMalloc(&m->als_rank, (size_t)(m->nnz ? m->nnz : 1) * sizeof(uint16_t)));
    hipLaunchKernelGGL(als_rank_k, dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, e->stream, (const int64_t*)m->row_ptr, (const uint32_t*)m->col, m->n,
                       (const int64_t*)m->col_ptr, (const uint32_t*)m->crow, m->als_rank, d_longest);
    int longest = 0;
    FMX_HIP(hipMemcpyAsync(&longest, d_longest, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    FMX_HIP(hipStreamSynchronize(e->stream));
    (void)hipFree(d_longest);
    m->als_rank_state = longest > 65535 ? -1 : 1;
    if (m->als_rank_state < 0) { (void)hipFree(m->als_rank); m->als_rank = nullptr; }
  }
  if (m->als_rank_state < 0) return FMX_OK;
  if (e->als_rec_rows < m->n) {
    (void)hipFree(e->als_rec); e->als_rec = nullptr; e->als_rec_rows = 0;
    FMX_HIP(hipMalloc(&e->als_rec, (size_t)m->n * 32));
    e->als_rec_rows = m->n;
  }
  ok = true;
  return FMX_OK;
}
template <bool W>
static int sweep_persist(fmx_engine* e, fmx_matrix* m, double2* d_qe, const SweepDyn* dyn) {
  const int L = (int)m->als_level_ptr.size() - 1;
  if (!m->als_level_ptr_dev) {
    FMX_HIP(hipMalloc(&m->als_level_ptr_dev, m->als_level_ptr.size() * sizeof(int64_t)));
    FMX_HIP(hipMemcpy(m->als_level_ptr_dev, m->als_level_ptr.data(), m->als_level_ptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  }
  if (!e->als_persist_ctl) {
    FMX_HIP(hipMalloc(&e->als_persist_ctl, PERSIST_CTL_WORDS * sizeof(unsigned int)));
    FMX_HIP(hipMemsetAsync(e->als_persist_ctl, 0, PERSIST_CTL_WORDS * sizeof(unsigned int), e->stream));
  }
  const int debug_skip = g_stall_next_persistent_sweep.exchange(0) > 0 ? 1 : 0;
  bool flow = false;
  FMX_TRY(flow_prepare(e, m, flow));
  if (flow) {
    // (q, e) pairs -> tagged records (tag 0: nobody has corrected the row in this launch), the sweep, records -> pairs: two streaming passes of 48 bytes per row
    const unsigned grid = (unsigned)((m->n + 255) / 256);
    prof_begin(e, FMX_KERNEL_ALS_SWEEP);
    hipLaunchKernelGGL(als_rec_pack_k, dim3(grid), dim3(256), 0, e->stream, (const double2*)d_qe, e->als_rec, m->n);
    const char* fw = getenv("FMX_ALS_FLOW_WAVES");
    const int flow_waves = resident_waves<true, W>(e, fw && atoi(fw) > 0 && atoi(fw) <= 2048 ? atoi(fw) : FLOW_WAVES);
    hipLaunchKernelGGL((als_exact_flow_k<W>), dim3((unsigned)flow_waves), dim3(64), 0, e->stream, (const uint32_t*)m->als_feats, (const int64_t*)m->als_level_ptr_dev, L,
                       (const int64_t*)m->col_ptr, (const uint32_t*)m->crow, (const float*)m->cval, (const uint16_t*)m->als_rank, W ? e->dw : e->dV, e->kp64, dyn,
                       e->als_rec, (uint32_t)((uint64_t)m->n * 32ull), e->als_persist_ctl + PERSIST_REPL * PERSIST_LINE_WORDS, debug_skip);
    hipLaunchKernelGGL(als_rec_unpack_k, dim3(grid), dim3(256), 0, e->stream, (const uint32_t*)e->als_rec, d_qe, m->n);
    prof_end(e);
    FMX_HIP(hipGetLastError());
    return FMX_OK;
  }
  // the counter's replicas, every launch.  The abort word (the line after them) is STICKY: a launch that gave up must still be known when the sweep's last factor has
  // run (persist_check reads and clears it); the launches after it leave at their first look at it
  FMX_HIP(hipMemsetAsync(e->als_persist_ctl, 0, (size_t)PERSIST_REPL * PERSIST_LINE_WORDS * sizeof(unsigned int), e->stream));
  prof_begin(e, FMX_KERNEL_ALS_SWEEP);
  hipLaunchKernelGGL((als_exact_persist_k<W>), dim3((unsigned)resident_waves<false, W>(e, PERSIST_WAVES)), dim3(64), 0, e->stream, (const uint32_t*)m->als_feats, (const int64_t*)m->als_level_ptr_dev, L,
                     (const int64_t*)m->col_ptr, (const uint32_t*)m->crow, (const float*)m->cval, W ? e->dw : e->dV, e->kp64, dyn, d_qe,
                     (uint32_t)((uint64_t)m->n * sizeof(double2)), e->als_persist_ctl, debug_skip);
  prof_end(e);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}
// after the sweeps of a call: did a wait of the persistent form give up?  (never seen; a lost update would otherwise be silent)
static int persist_check(fmx_engine* e) {
  if (!e->als_persist_ctl) return FMX_OK;
  unsigned int ctl[2] = {0, 0};
  FMX_HIP(hipMemcpyAsync(ctl, e->als_persist_ctl, sizeof(unsigned int), hipMemcpyDeviceToHost, e->stream));
  FMX_HIP(hipMemcpyAsync(ctl + 1, e->als_persist_ctl + PERSIST_REPL * PERSIST_LINE_WORDS, sizeof(unsigned int), hipMemcpyDeviceToHost, e->stream));
  FMX_HIP(hipStreamSynchronize(e->stream));
  if (ctl[1] != 0) (void)hipMemset(e->als_persist_ctl + PERSIST_REPL * PERSIST_LINE_WORDS, 0, sizeof(unsigned int));   // reported once: the next sweep starts clean
  FMX_CHECK(ctl[1] == 0, FMX_ERR_HIP, "the persistent sweep gave up waiting (its workgroups were not all running?): V and the residual are part-way through a sweep");
  return FMX_OK;
}

// one sweep of the w coordinates or of one factor, by replay when the plan is deep
template <bool W>
static void sweep_once(fmx_engine* e, fmx_matrix* m, double2* d_qe, double2* d_qe_new, SweepDyn* dyn) {
  if (!d_qe_new && persist_applies(m) && sweep_persist<W>(e, m, d_qe, dyn) == FMX_OK) return;
  if (sweep_graph_wanted<W>(m)) {
    hipGraphExec_t x = sweep_graph<W>(e, m, d_qe, dyn);
    if (x && hipGraphLaunch(x, e->stream) == hipSuccess) return;
    (void)hipGetLastError();
  }
  sweep_features<W>(e, m, d_qe, d_qe_new, dyn);
}

// the approximate form's second (q, e) array: allocated on first use, kept in the engine
static double2* approx_buffer(fmx_engine* e, fmx_matrix* m) {
  if (!m->als_approx) return nullptr;
  if (e->als_qe_new_rows < m->n) {
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(e->als_qe_new); e->als_qe_new = nullptr; e->als_qe_new_rows = 0;
    if (hipMalloc(&e->als_qe_new, (size_t)m->n * sizeof(double2)) != hipSuccess) return nullptr;
    e->als_qe_new_rows = m->n;
  }
  return reinterpret_cast<double2*>(e->als_qe_new);
}

// sum of e^2 over the rows: partial sums on the device (mcmc_sumsq_partial_k), the rest on the host
static int residual_sumsq(fmx_engine* e, const double2* d_qe, int64_t n, double* out);

// (q, e) pairs of a learner's loop or a device-resident sweep: kept in the engine, grow-only (a stable address: the replayed graphs
// of deep plans are captured against it)
static double2* sweep_pairs(fmx_engine* e, int64_t n) {
  if (e->als_qe_rows < n) {
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(e->als_qe); e->als_qe = nullptr; e->als_qe_rows = 0;
    if (hipMalloc(&e->als_qe, (size_t)n * sizeof(double2)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    e->als_qe_rows = n;
  }
  return reinterpret_cast<double2*>(e->als_qe);
}

// the n x kp table of every factor's q (grow-only), or null where there is no room for it (the sweep then falls back to one gather pass per factor)
static double* q_table(fmx_engine* e, fmx_matrix* m) {
  const size_t need = (size_t)m->n * e->kp64;
  if (e->als_Q_elems < need) {
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(e->als_Q); e->als_Q = nullptr; e->als_Q_elems = 0; e->als_q_have = 0; e->als_q_trusted = 0;
    if (hipMalloc(&e->als_Q, need * sizeof(double)) == hipSuccess) e->als_Q_elems = need;
    else (void)hipGetLastError();
  }
  return e->als_Q;
}

// V sweep over all factors on the interleaved (q, e) pairs
static int v_sweep_enqueue(fmx_engine* e, fmx_matrix* m, double2* d_qe, double alpha, const double* h_lambda, const double* h_mu,
                           const double* d_znorm = nullptr) {
  const unsigned row_grid = (unsigned)((m->n + 255) / 256);
  double2* d_qe_new = approx_buffer(e, m);
  SweepDyn* dyn = sweep_dyn(e);
  FMX_CHECK(dyn != nullptr, FMX_ERR_HIP, "out of device memory");
  // q_f = X v_f only depends on column f of V, which no other factor's sweep touches: all k of them come out of ONE
  // row-gather pass (the forward kernel on the fp64 tables) instead of one gather per nonzero per factor.  The n x kp table lives
  // in the engine (grow-only): a sweep allocates nothing once the first one has run.
  double* d_Q = q_table(e, m);
  // cfg.als_max_levels = -2 on a coloured plan of light lists: the FEATURE-MAJOR order -- all k factors of a feature while its rows' state is in LDS (als_level_allf_k)
  if (allf_applies(e, m) && !d_qe_new) {
    double* d_Qr = q_table(e, m);
    FMX_CHECK(d_Qr != nullptr, FMX_ERR_HIP, "out of device memory: the feature-major sweep keeps an n x kp table of doubles (%.1f GB)", (double)m->n * e->kp64 * 8e-9);
    {
      RowsArgs a{};
      a.row_ptr = m->row_ptr; a.col = m->col; a.val = m->val; a.r0 = 0; a.nrows = m->n;
      a.V = e->dV; a.w = e->dw; a.vs = e->kp64; a.ws = 1; a.scal = e->scal; a.yhat = nullptr; a.qout = d_Qr; a.qout_t = 0; a.link = FMX_LINK_NONE;   // q of every factor, ROW-major: [n][kp]
      a.unit = m->unit_values; a.no_w = 1;
      const bool trusted = e->als_q_trusted == (m->uid ^ ALLF_TRUSTED);   // the learner's own forward pass of this iteration left the table (launch_als_train): V untouched since
      // ... or carried from the previous sweep (opt-in, fmx_als_carry_q): the sweep corrects every q_f of every row as it goes, so the table it leaves IS X v_f of the
      // new V (to rounding); valid while V is bit for bit what that sweep left (64-bit fingerprint) and for 64 sweeps at most
      bool carried = false;
      const uint64_t carry_key = (m->uid ^ ALLF_TRUSTED) ^ ((uint64_t)m->value_generation << 40);   // (this matrix, these values)
      if (!trusted && e->als_carry_q && e->als_q_have && e->als_q_plan == carry_key && e->als_q_age < 64) {
        uint64_t hsh = 0;
        FMX_TRY(als_vhash(e, &hsh));
        carried = hsh == e->als_q_hash;
      }
      const int carried_age = carried ? e->als_q_age : 0;
      e->als_q_have = 0; e->als_q_trusted = 0;
      // (a failing forward pass is an error of the sweep, not a reason to change the nesting: fmx_als_plan_info has told the caller "feature-major" -- ADVICE r5)
      if (!(trusted || carried)) FMX_TRY(launch_rows_forward(e, a, false, true));
      {
        std::vector<double> lm((size_t)2 * e->k, 0.0);
        for (int f = 0; f < e->k; ++f) { lm[(size_t)2 * f] = h_lambda ? h_lambda[f] : 0.0; lm[(size_t)2 * f + 1] = h_mu ? h_mu[f] : 0.0; }
        if (!e->als_lam_mu) FMX_HIP(hipMalloc(&e->als_lam_mu, (size_t)2 * 1024 * sizeof(double)));
        FMX_HIP(hipMemcpyAsync(e->als_lam_mu, lm.data(), lm.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
        FMX_HIP(hipStreamSynchronize(e->stream));   // (lm is a local)
        // (per sweep, not once per process: the attribute belongs to the CURRENT device's copy of the kernel, and engines may live on several devices)
        FMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&als_level_allf_k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ALLF_LDS_LIMIT));
        FMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&als_level_allf_k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ALLF_LDS_LIMIT));
        const int L = (int)m->als_level_ptr.size() - 1;
        const char* form_env = getenv("FMX_ALS_ALLF_FORM");   // 1: the LDS-resident kernel everywhere, 2: no one-wave kernel (read per call: the tests compare the forms)
        const int form = form_env ? atoi(form_env) : 0;
        const bool in_regs = (e->kp64 == 8 || e->kp64 == 16) && form != 1;
        // k < kp: e rides in the spare last slot of the rows' lines for the length of the sweep (allf_e_enter_k above); FMX_ALS_ALLF_EIL=0: in the pair table as at k = kp (tests: the same bits)
        const char* eil_env = getenv("FMX_ALS_ALLF_EIL");
        const bool eil = e->k < e->kp64 && !(eil_env && eil_env[0] == '0');
        const unsigned row_grid = (unsigned)((m->n + 255) / 256);
        if (eil) hipLaunchKernelGGL(allf_e_enter_k, dim3(row_grid), dim3(256), 0, e->stream, d_Qr, (const double2*)d_qe, m->n, e->kp64);
        for (int l = 0; l < L; ++l) {
          const int64_t l0 = m->als_level_ptr[(size_t)l], cnt = m->als_level_ptr[(size_t)l + 1] - l0;
          if (cnt == 0) continue;
          const int cap = (int)((m->als_level_maxlen[(size_t)l] + 63) / 64 * 64);
          const size_t lds = allf_lds_bytes(m->als_level_maxlen[(size_t)l], e->kp64);
          prof_begin(e, FMX_KERNEL_ALS_SWEEP);
#define FMX_ALLF_ARGS (const uint32_t*)(m->als_feats + l0), (int)cnt, (const int64_t*)m->col_ptr, (const uint32_t*)m->crow, (const float*)m->cval, e->dV, e->k, alpha, \
                      (const double*)e->als_lam_mu, d_znorm, (int64_t)m->p, d_Qr, d_qe
          if (in_regs && form != 2 && cap <= 384) {   // one wave per feature
#define FMX_ALLF_WAVE(U, KPv, R, E) hipLaunchKernelGGL((als_level_allf_wave_k<U, KPv, R, E>), dim3((unsigned)cnt), dim3(64), 0, e->stream, FMX_ALLF_ARGS)
#define FMX_ALLF_WAVE_R(U, KPv, E) do { if (cap <= 128) FMX_ALLF_WAVE(U, KPv, 2, E); else if (cap <= 256) FMX_ALLF_WAVE(U, KPv, 4, E); else FMX_ALLF_WAVE(U, KPv, 6, E); } while (0)
#define FMX_ALLF_WAVE_E(U, KPv) do { if (eil) FMX_ALLF_WAVE_R(U, KPv, true); else FMX_ALLF_WAVE_R(U, KPv, false); } while (0)
            if (e->kp64 == 16) { if (m->unit_values) FMX_ALLF_WAVE_E(true, 16); else FMX_ALLF_WAVE_E(false, 16); }
            else { if (m->unit_values) FMX_ALLF_WAVE_E(true, 8); else FMX_ALLF_WAVE_E(false, 8); }
#undef FMX_ALLF_WAVE_E
#undef FMX_ALLF_WAVE_R
#undef FMX_ALLF_WAVE
          } else if (in_regs && m->als_level_maxlen[(size_t)l] <= 512) {   // the rows' lines in registers: four workgroups per CU
#define FMX_ALLF_REG(U, KPv, E) hipLaunchKernelGGL((als_level_allf_reg_k<U, KPv, E>), dim3((unsigned)cnt), dim3(WG_THREADS), 0, e->stream, FMX_ALLF_ARGS)
#define FMX_ALLF_REG_E(U, KPv) do { if (eil) FMX_ALLF_REG(U, KPv, true); else FMX_ALLF_REG(U, KPv, false); } while (0)
            if (e->kp64 == 16) { if (m->unit_values) FMX_ALLF_REG_E(true, 16); else FMX_ALLF_REG_E(false, 16); }
            else { if (m->unit_values) FMX_ALLF_REG_E(true, 8); else FMX_ALLF_REG_E(false, 8); }
#undef FMX_ALLF_REG_E
#undef FMX_ALLF_REG
          } else if (m->unit_values)
            hipLaunchKernelGGL((als_level_allf_k<true>), dim3((unsigned)cnt), dim3(WG_THREADS), lds, e->stream, (const uint32_t*)(m->als_feats + l0), (int)cnt, (const int64_t*)m->col_ptr,
                               (const uint32_t*)m->crow, (const float*)m->cval, e->dV, e->k, e->kp64, alpha, (const double*)e->als_lam_mu, d_znorm, (int64_t)m->p, d_Qr, d_qe, cap, eil ? 1 : 0);
          else
            hipLaunchKernelGGL((als_level_allf_k<false>), dim3((unsigned)cnt), dim3(WG_THREADS), lds, e->stream, (const uint32_t*)(m->als_feats + l0), (int)cnt, (const int64_t*)m->col_ptr,
                               (const uint32_t*)m->crow, (const float*)m->cval, e->dV, e->k, e->kp64, alpha, (const double*)e->als_lam_mu, d_znorm, (int64_t)m->p, d_Qr, d_qe, cap, eil ? 1 : 0);
#undef FMX_ALLF_ARGS
          prof_end(e);
          FMX_HIP(hipGetLastError());   // (per level: a launch that fails must not be followed by the levels after it)
        }
        if (eil) hipLaunchKernelGGL(allf_e_exit_k, dim3(row_grid), dim3(256), 0, e->stream, (const double*)d_Qr, d_qe, m->n, e->kp64);
        if (e->als_carry_q) {   // the table now holds X v_f of the new V, every factor (at k < kp the spare slot holds e: the next sweep's enter overwrites it)
          uint64_t hsh = 0;
          FMX_TRY(als_vhash(e, &hsh));
          e->als_q_hash = hsh; e->als_q_plan = carry_key; e->als_q_age = carried_age + 1; e->als_q_have = 1;
        }
        FMX_HIP(hipGetLastError());
        return FMX_OK;
      }
    }
  }
  const uint32_t* colP = nullptr; const float* valP = nullptr;
  e->als_q_level0 = 0;
  if (d_Q && !d_qe_new) FMX_TRY(als_order_prepare(e, m, &colP, &valP, nullptr));   // the block form: q in level 0's array order (the forward runs on the permuted CSR)
  // q carried from the previous sweep (opt-in, block form): valid if it belongs to this plan and V is bit for bit what that sweep left
  constexpr int ALS_CARRY_REFRESH = 64;
  const bool carry = e->als_carry_q && colP != nullptr && d_Q != nullptr;
  bool reuse = false;
  if (carry && e->als_q_have && e->als_q_plan == als_order_plan_uid(m) && e->als_q_age < ALS_CARRY_REFRESH) {
    uint64_t hsh = 0;
    FMX_TRY(als_vhash(e, &hsh));
    reuse = hsh == e->als_q_hash;
  }
  const bool reuse_carried = reuse;   // (a table the learner's forward pass built a moment ago is fresh: its age starts at 1)
  // ... or built a moment ago by the learner's own forward pass (launch_als_train: the pass that computes y_hat leaves q beside it; nothing between it and this
  // sweep touches V)
  if (colP && d_Q && e->als_q_trusted != 0 && e->als_q_trusted == als_order_plan_uid(m)) reuse = true;
  e->als_q_trusted = 0;
  e->als_q_have = 0;
  if (d_Q && !reuse) {
    RowsArgs a{};
    a.row_ptr = m->row_ptr; a.col = colP ? colP : m->col; a.val = colP ? valP : m->val; a.r0 = 0; a.nrows = m->n;
    a.V = e->dV; a.w = e->dw; a.vs = e->kp64; a.ws = 1; a.scal = e->scal; a.yhat = nullptr; a.qout = d_Q; a.qout_t = m->n; a.link = FMX_LINK_NONE;
    a.unit = m->unit_values;
    a.no_w = 1;
    if (launch_rows_forward(e, a, false, true) != FMX_OK) d_Q = nullptr;
  }
  // a COMPLETE tiled plan (every level tiled, every row in every level: one-column-per-field data): the pairs travel in the list order of the level that
  // consumes them next (fm_als_tiled.hip, the level-order form) -- per level one streaming sums + step kernel and one correct-and-permute kernel
  if (d_Q && !d_qe_new && als_order_ready(m)) {
    bool ok = false;
    FMX_TRY(als_order_enter(e, m, d_qe, d_Q, &ok));
    if (ok) {
      const int S = als_order_levels(m);
      for (int f = 0; f < e->k; ++f) {
        set_dyn(e, dyn, f, alpha, h_lambda ? h_lambda[f] : 0.0, h_mu ? h_mu[f] : 0.0, d_znorm ? d_znorm + (size_t)f * m->p : nullptr);
        for (int s = 0; s < S; ++s) {
          prof_begin(e, FMX_KERNEL_ALS_SWEEP);   // one level of one factor: the unit bench.py --solver als prices
          const double* d_q = e->als_q_level0 ? ((s == 0 && f > 0) ? d_Q + (size_t)f * m->n : nullptr)                      // block form: q enters at the first level
                                              : ((s == S - 1 && f + 1 < e->k) ? d_Q + (size_t)(f + 1) * m->n : nullptr);   // tile form: the next q leaves the last
          const int st = als_order_level(e, m, s, dyn, d_q, (carry && e->als_q_level0 && s == 0 && f > 0) ? d_Q + (size_t)(f - 1) * m->n : nullptr);
          prof_end(e);
          FMX_TRY(st);
        }
      }
      FMX_TRY(als_order_exit(e, m, d_qe, (carry && e->als_q_level0) ? d_Q + (size_t)(e->k - 1) * m->n : nullptr));
      if (carry && e->als_q_level0) {   // the table now holds X v_f of the new V, every factor
        uint64_t hsh = 0;
        FMX_TRY(als_vhash(e, &hsh));
        e->als_q_hash = hsh; e->als_q_plan = als_order_plan_uid(m); e->als_q_age = reuse_carried ? e->als_q_age + 1 : 1; e->als_q_have = 1;
      }
      return FMX_OK;
    }
  }
  bool picked = false;   // the previous factor's last correction pass already stored this factor's q (tiled form)
  e->als_vf_slot = -1;
  for (int f = 0; f < e->k; ++f) {
    if (picked) {}
    else if (d_Q) hipLaunchKernelGGL(als_q_pick_k, dim3(row_grid), dim3(256), 0, e->stream, (const double*)(d_Q + (size_t)f * m->n), m->n, d_qe);
    else hipLaunchKernelGGL(als_q_init_k, dim3(row_grid), dim3(256), 0, e->stream, m->row_ptr, m->col, m->val, m->n, e->dV, e->kp64, f, d_qe);
    const double lambda = h_lambda ? h_lambda[f] : 0.0, mu = h_mu ? h_mu[f] : 0.0;
    if (d_qe_new) (void)hipMemcpyAsync(d_qe_new, d_qe, (size_t)m->n * sizeof(double2), hipMemcpyDeviceToDevice, e->stream);  // q changed: resynchronise the pair
    set_dyn(e, dyn, f, alpha, lambda, mu, d_znorm ? d_znorm + (size_t)f * m->p : nullptr);
    e->als_vf_slot = -1;   // (another factor: nothing gathered ahead is valid)
    e->als_qnext = (d_Q && f + 1 < e->k) ? d_Q + (size_t)(f + 1) * m->n : nullptr;
    const bool offered = e->als_qnext != nullptr;
    sweep_once<false>(e, m, d_qe, d_qe_new, dyn);
    picked = offered && e->als_qnext == nullptr;
    e->als_qnext = nullptr;
  }
  e->als_vf_slot = -1;
  if (!d_qe_new && persist_applies(m)) FMX_TRY(persist_check(e));
  return FMX_OK;
}

// ---- the approximate form, guarded ------------------------------------------------------------------------------------------------
// Features of a group step against ONE snapshot of the residual; correlated features then overshoot together, and on Zipf columns
// the sweep diverges (sum e^2 1e6 -> 1e22 in one sweep, profiles/r02_als_levels.txt).  So an approximate sweep is run under a guard:
// V (or w) and the residual are kept aside, the sweep runs, and if the residual's sum of squares went UP (ALS: at all; MCMC, whose
// draws add variance of their own: tenfold, or not finite) everything is put back, the matrix is marked exact-only and the sweep
// is run again through the exact level schedule.  cfg.als_max_levels is therefore a request, not a risk.
// What is compared is the objective the coordinate steps descend, alpha sum e^2 + sum_f lambda_f sum_j (theta_fj - mu_f)^2, not the residual alone: with a
// prior (lambda > 0) or a warm start beyond the regularised optimum a correct sweep lowers the objective while RAISING sum e^2, and the residual alone
// would demote the matrix to the exact schedule for good after one benign sweep.  The demotion is visible to the caller: fmx_als_plan_info
// reports approximate = 0 from then on.
static int penalty_sum(fmx_engine* e, bool w, const double* h_lambda, const double* h_mu, double* out);
struct ApproxGuard {
  fmx_engine* e; fmx_matrix* m; double2* d_qe; bool w; bool gibbs;
  double alpha = 1.0;
  const double *h_lambda = nullptr, *h_mu = nullptr;   // [k] (V sweep) or [1] (w sweep); null: no prior
  double ss0 = 0.0;
  bool armed = false;
  int objective(double* out) {
    double ss = 0.0, pen = 0.0;
    FMX_TRY(residual_sumsq(e, d_qe, m->n, &ss));
    FMX_TRY(penalty_sum(e, w, h_lambda, h_mu, &pen));
    *out = alpha * ss + pen;
    return FMX_OK;
  }
  int begin() {
    if (!m->als_approx) return FMX_OK;
    const size_t pv = w ? (size_t)e->p : (size_t)e->p * e->kp64;
    const size_t need = pv + 2 * (size_t)m->n;
    if (e->als_backup_elems < need) {
      FMX_HIP(hipStreamSynchronize(e->stream));
      (void)hipFree(e->als_backup); e->als_backup = nullptr; e->als_backup_elems = 0;
      FMX_HIP(hipMalloc(&e->als_backup, need * sizeof(double)));
      e->als_backup_elems = need;
    }
    FMX_TRY(objective(&ss0));
    FMX_HIP(hipMemcpyAsync(e->als_backup, w ? e->dw : e->dV, pv * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    FMX_HIP(hipMemcpyAsync(e->als_backup + pv, d_qe, (size_t)m->n * sizeof(double2), hipMemcpyDeviceToDevice, e->stream));
    armed = true;
    return FMX_OK;
  }
  // *redo = true: the state was put back and the plan is exact now -- run the sweep again
  int end(bool* redo) {
    *redo = false;
    if (!armed) return FMX_OK;
    double ss1 = 0.0;
    FMX_TRY(objective(&ss1));
    const bool bad = !(ss1 == ss1) || std::isinf(ss1) || ss1 > ss0 * (gibbs ? 10.0 : 1.0 + 1e-9);
    if (!bad) return FMX_OK;
    const size_t pv = w ? (size_t)e->p : (size_t)e->p * e->kp64;
    FMX_HIP(hipMemcpyAsync(w ? e->dw : e->dV, e->als_backup, pv * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    FMX_HIP(hipMemcpyAsync(d_qe, e->als_backup + pv, (size_t)m->n * sizeof(double2), hipMemcpyDeviceToDevice, e->stream));
    FMX_HIP(hipStreamSynchronize(e->stream));
    m->als_force_exact = 1;
    FMX_TRY(build_plan(m, e->stream, 0));
    *redo = true;
    return FMX_OK;
  }
};

static int v_sweep_guarded(fmx_engine* e, fmx_matrix* m, double2* d_qe, double alpha, const double* h_lambda, const double* h_mu, const double* d_znorm = nullptr) {
  ApproxGuard g{e, m, d_qe, false, d_znorm != nullptr, alpha, h_lambda, h_mu};
  FMX_TRY(g.begin());
  FMX_TRY(v_sweep_enqueue(e, m, d_qe, alpha, h_lambda, h_mu, d_znorm));
  bool redo = false;
  FMX_TRY(g.end(&redo));
  if (redo) FMX_TRY(v_sweep_enqueue(e, m, d_qe, alpha, h_lambda, h_mu, d_znorm));
  return FMX_OK;
}

static int w_sweep_guarded(fmx_engine* e, fmx_matrix* m, double2* d_qe, double alpha, double lambda, double mu, const double* d_znorm) {
  SweepDyn* dyn = sweep_dyn(e);
  FMX_CHECK(dyn != nullptr, FMX_ERR_HIP, "out of device memory");
  ApproxGuard g{e, m, d_qe, true, d_znorm != nullptr, alpha, &lambda, &mu};
  FMX_TRY(g.begin());
  for (int pass = 0; pass < 2; ++pass) {
    double2* d_qe_new = approx_buffer(e, m);
    if (d_qe_new) FMX_HIP(hipMemcpyAsync(d_qe_new, d_qe, (size_t)m->n * sizeof(double2), hipMemcpyDeviceToDevice, e->stream));
    set_dyn(e, dyn, 0, alpha, lambda, mu, d_znorm);
    e->als_vf_slot = -1; e->als_qnext = nullptr;
    bool blocks_done = false;
    if (!d_qe_new) FMX_TRY(als_order_w_sweep(e, m, d_qe, dyn, &blocks_done));   // a complete plan whose lists fit a block: one kernel per level (fm_als_blocks.hip)
    if (!blocks_done) {
      sweep_once<true>(e, m, d_qe, d_qe_new, dyn);
      if (!d_qe_new && persist_applies(m)) FMX_TRY(persist_check(e));
    }
    e->als_vf_slot = -1;
    bool redo = false;
    if (pass == 0) FMX_TRY(g.end(&redo));
    if (!redo) break;
  }
  return FMX_OK;
}

// MCMC_ALS_Learner::learn for the ALS learner (:91-156): per iteration a fresh forward, the residual of the task
// (e = y_hat - y, or the probit-table ratio for CLASSIFICATION, :520-562), the w0
// update, the w sweep; with_v adds the V sweep the shipped update_all leaves out (SURVEY A-1).  init() fixes alpha = 1,
// w0_mean_0 = 0 and all lambda / mu = 0 (A-7), so the R-side solver parameters do not enter.
int als_plan_info(fmx_engine* e, fmx_matrix* m, int64_t* levels, int64_t* largest, int32_t* approx, int32_t* level_of) {
  FMX_CHECK(m->rows_sorted, FMX_ERR_INVALID, "the ALS sweeps need every row's columns strictly ascending (as R's dgCMatrix rows are)");
  FMX_TRY(build_full_csc(m, e->stream));
  FMX_TRY(build_plan(m, e->stream, e->cfg.als_max_levels));
  const int64_t L = (int64_t)m->als_level_ptr.size() - 1;
  int64_t big = 0;
  for (int64_t l = 0; l < L; ++l) {
    const int64_t c = m->als_level_ptr[(size_t)l + 1] - m->als_level_ptr[(size_t)l] + m->als_heavy_ptr[(size_t)l + 1] - m->als_heavy_ptr[(size_t)l] +
                      (m->als_vh_ptr.empty() ? 0 : m->als_vh_ptr[(size_t)l + 1] - m->als_vh_ptr[(size_t)l]);
    if (c > big) big = c;
  }
  if (levels) *levels = L;
  if (largest) *largest = big;
  if (approx) *approx = m->als_approx ? 1 : (m->als_coloured ? (allf_applies(e, m) ? 3 : 2) : 0);   // 2: exact steps in a COLOURED feature order (cfg.als_max_levels < 0); 3: and nested feature-major (-2)
  if (level_of) for (size_t j = 0; j < m->als_level_of.size(); ++j) level_of[j] = m->als_level_of[j];
  return FMX_OK;
}

int launch_als_train(fmx_engine* e, fmx_matrix* m, int max_iter, int with_v) {
  FMX_CHECK(m->rows_sorted, FMX_ERR_INVALID, "the ALS sweeps need every row's columns strictly ascending (as R's dgCMatrix rows are)");
  const double* dp_y = nullptr;
  if (e->cfg.task == FMX_TASK_CLASSIFICATION) {
    FMX_TRY(ensure_probit(e));
    dp_y = e->probit + PN_POINTS + 1;
  }
  FMX_TRY(build_full_csc(m, e->stream));
  FMX_TRY(build_plan(m, e->stream, e->cfg.als_max_levels));
  const int64_t n = m->n;
  const unsigned row_grid = (unsigned)((n + 255) / 256);
  const int64_t np = (n + ALS_SLAB - 1) / ALS_SLAB;
  double *d_yhat = nullptr, *d_part = nullptr;
  double2* d_qe = sweep_pairs(e, n);
  FMX_CHECK(d_qe != nullptr, FMX_ERR_HIP, "out of device memory");
  FMX_HIP(hipMalloc(&d_yhat, (size_t)n * sizeof(double)));
  if (hipMalloc(&d_part, ((size_t)np + 1) * sizeof(double)) != hipSuccess) {
    (void)hipFree(d_yhat); (void)hipFree(d_part);
    set_error("out of device memory"); return FMX_ERR_HIP;
  }
  int st = FMX_OK;
  // "the forward pass of this iteration left q beside y_hat" is a promise for the V sweep of the SAME iteration only: whatever way this call ends -- an error between
  // the two included -- the flag does not outlive it
  struct TrustGuard { fmx_engine* e; ~TrustGuard() { e->als_q_trusted = 0; } } trust_guard{e};
  for (int it = 0; it < max_iter && st == FMX_OK; ++it) {
    RowsArgs a{};
    a.row_ptr = m->row_ptr; a.col = m->col; a.val = m->val; a.r0 = 0; a.nrows = n;
    a.V = e->dV; a.w = e->dw; a.vs = e->kp64; a.ws = 1; a.scal = e->scal; a.yhat = d_yhat; a.link = FMX_LINK_NONE;
    // With a V sweep to follow on a block-form plan, this forward pass runs on the plan's copy of the CSR (rows in level 0's array order) and leaves the q of every
    // factor beside y_hat: nothing between here and the V sweep touches V, so the sweep's own pass over the matrix (5.5 of an iteration's 58 ms) is saved.
    const uint32_t *colP = nullptr, *row0 = nullptr; const float* valP = nullptr;
    double* d_Q = nullptr;
    static const bool share = [] { const char* v = getenv("FMX_ALS_SHARE_FORWARD"); return !(v && v[0] == '0'); }();
    const bool fmajor = share && with_v && allf_applies(e, m);   // a feature-major plan (-2): the sweep wants q ROW-major, rows in the matrix's own order
    if (fmajor) {
      d_Q = q_table(e, m);
      if (d_Q) { a.unit = m->unit_values; a.qout = d_Q; a.qout_t = 0; }
    } else if (share && with_v && e->k > 0 && !m->als_approx) {
      if (als_order_prepare(e, m, &colP, &valP, &row0) != FMX_OK) colP = nullptr;
      if (colP) d_Q = q_table(e, m);
      if (!d_Q) colP = nullptr;
    }
    if (colP) { a.col = colP; a.val = valP; a.unit = m->unit_values; a.qout = d_Q; a.qout_t = n; }   // (one-hot matrices keep no copy of the values: valP is null and must not be read)
    st = launch_rows_forward(e, a, false, true);  // fm->predict_batch(train, train_err), :100
    if (st != FMX_OK) break;
    if (colP) {
      hipLaunchKernelGGL(als_residual_perm_k, dim3(row_grid), dim3(256), 0, e->stream, (const double*)d_yhat, (const float*)m->y, row0, n, d_qe, dp_y);
      e->als_q_trusted = als_order_plan_uid(m);
    } else {
      hipLaunchKernelGGL(als_residual_k, dim3(row_grid), dim3(256), 0, e->stream, d_yhat, m->y, n, d_qe, dp_y);
      if (fmajor && d_Q) e->als_q_trusted = m->uid ^ ALLF_TRUSTED;
    }
    if (e->hyper.k0) {
      hipLaunchKernelGGL(als_w0_partial_k, dim3((unsigned)np), dim3(WG_THREADS), 0, e->stream, d_qe, n, e->scal, d_part);
      hipLaunchKernelGGL(als_w0_final_k, dim3(1), dim3(WG_THREADS), 0, e->stream, d_part, np, n, e->scal, e->hyper.reg0, 1.0, 0.0, 0, 0.0);
      hipLaunchKernelGGL(als_shift_k, dim3(row_grid), dim3(256), 0, e->stream, d_qe, n, d_part + np);
    }
    if (e->hyper.k1 && st == FMX_OK) st = w_sweep_guarded(e, m, d_qe, 1.0, 0.0, 0.0, nullptr);
    if (with_v && e->k > 0 && st == FMX_OK) st = v_sweep_guarded(e, m, d_qe, 1.0, nullptr, nullptr);
  }
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  (void)hipFree(d_yhat); (void)hipFree(d_part);
  FMX_TRY(st);
  FMX_CHECK(err == hipSuccess, FMX_ERR_HIP, "ALS training failed: %s", hipGetErrorString(err));
  return FMX_OK;
}

// ---- the MCMC learner (MCMC_Learner: do_sample, do_multilevel; :565-576) ------------------------------------------------
__global__ __launch_bounds__(WG_THREADS) void mcmc_sumsq_partial_k(const double2* __restrict__ qe, int64_t n, double* __restrict__ partials) {
  __shared__ double red[WG_THREADS];
  const int64_t base = (int64_t)blockIdx.x * ALS_SLAB;
  double acc = 0.0;
  for (int i = threadIdx.x; i < ALS_SLAB; i += WG_THREADS) {
    const int64_t r = base + i;
    if (r < n) acc += qe[r].y * qe[r].y;  // update_alpha, :370-373
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = WG_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

static int residual_sumsq(fmx_engine* e, const double2* d_qe, int64_t n, double* out) {
  const int64_t np = (n + ALS_SLAB - 1) / ALS_SLAB;
  double* d_part = nullptr;
  FMX_HIP(hipMalloc(&d_part, (size_t)(np > 0 ? np : 1) * sizeof(double)));
  std::vector<double> h((size_t)np);
  int st = FMX_OK;
  if (np > 0) {
    hipLaunchKernelGGL(mcmc_sumsq_partial_k, dim3((unsigned)np), dim3(WG_THREADS), 0, e->stream, d_qe, n, d_part);
    if (hipMemcpyAsync(h.data(), d_part, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess) { set_error("residual sum of squares failed"); st = FMX_ERR_HIP; }
  }
  (void)hipFree(d_part);
  double s = 0.0;
  for (double v : h) s += v;
  *out = s;
  return st;
}

// per slab of w: sum w and sum (w - mu)^2 (update_w_lambda :423-427, update_w_mu :392-395)
__global__ __launch_bounds__(WG_THREADS) void mcmc_wstats_partial_k(const double* __restrict__ w, int64_t p, double mu, double* __restrict__ partials) {
  __shared__ double r1[WG_THREADS], r2[WG_THREADS];
  const int64_t base = (int64_t)blockIdx.x * ALS_SLAB;
  double a1 = 0.0, a2 = 0.0;
  for (int i = threadIdx.x; i < ALS_SLAB; i += WG_THREADS) {
    const int64_t j = base + i;
    if (j < p) { a1 += w[j]; a2 += (w[j] - mu) * (w[j] - mu); }
  }
  r1[threadIdx.x] = a1; r2[threadIdx.x] = a2;
  __syncthreads();
  for (int off = WG_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) { r1[threadIdx.x] += r1[threadIdx.x + off]; r2[threadIdx.x] += r2[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = r1[0]; partials[2 * blockIdx.x + 1] = r2[0]; }
}

// util/Random.h:20-93 on libc rand(), host side: the reference draws its truncated normals from rand(), row by row
static double h_runif() { return rand() / ((double)RAND_MAX + 1); }
static double h_rexp() { return -std::log(1 - h_runif()); }
static double h_rnorm() {  // Leva's ratio-of-uniforms, Random.h:31-49
  double u, v, abs_v, x, y, Q;
  do {
    do { u = h_runif(); } while (u == 0.0);
    v = 1.7156 * (h_runif() - 0.5);
    abs_v = v < 0 ? -v : v;
    x = u - 0.449871;
    y = abs_v + 0.386595;
    Q = x * x + y * (0.19600 * y - 0.25472 * x);
    if (Q < 0.27597) break;
  } while ((Q > 0.27846) || ((v * v) > (-4.0 * u * u * std::log(u))));
  return v / u;
}
static double h_trnorm_left(double left) {  // Random.h:52-76
  if (left < 0.0) {
    for (;;) { const double r = h_rnorm(); if (r >= left) return r; }
  }
  const double alpha_star = 0.5 * (left + std::sqrt(left * left + 4.0));
  for (;;) {
    const double z = h_rexp() / alpha_star + left;
    double d = z - alpha_star;
    d = std::exp(-(d * d) / 2);
    const double u = h_runif();
    if (u < d) return z;
  }
}

// MCMC_ALS_Learner::learn + update_all for the MCMC learner, one attribute group (core/Data.h:10-26).  R's generator is
// not available to a library: the draws the reference takes from it come pre-drawn from the caller, in call order (see
// fmx.h).  The CLASSIFICATION residual subtracts truncated normals drawn from libc rand() row by row (:529-542), which is a
// serial stream by construction: y_hat goes to the host, the draws are made there, the residual comes back.  V is never
// updated, as shipped (SURVEY A-1).
int launch_mcmc_train(fmx_engine* e, fmx_matrix* m, int max_iter, const double* h_gammas, const double* h_normals, double* h_state, const double* h_state_in) {
  FMX_CHECK(m->rows_sorted, FMX_ERR_INVALID, "the ALS sweeps need every row's columns strictly ascending (as R's dgCMatrix rows are)");
  FMX_TRY(build_full_csc(m, e->stream));
  FMX_TRY(build_plan(m, e->stream, e->cfg.als_max_levels));
  const int64_t n = m->n;
  const int64_t p = (int64_t)e->p;
  const unsigned row_grid = (unsigned)((n + 255) / 256);
  const int64_t np = (n + ALS_SLAB - 1) / ALS_SLAB, npw = (p + ALS_SLAB - 1) / ALS_SLAB;
  const int64_t part_cap = (np > 2 * npw ? np : 2 * npw) + 1;
  double *d_yhat = nullptr, *d_part = nullptr, *d_z = nullptr;
  double2* d_qe = sweep_pairs(e, n);
  FMX_CHECK(d_qe != nullptr, FMX_ERR_HIP, "out of device memory");
  auto cleanup = [&]() { (void)hipFree(d_yhat); (void)hipFree(d_part); (void)hipFree(d_z); };
  if (hipMalloc(&d_yhat, (size_t)n * sizeof(double)) != hipSuccess ||
      hipMalloc(&d_part, (size_t)part_cap * sizeof(double)) != hipSuccess || hipMalloc(&d_z, (size_t)p * sizeof(double)) != hipSuccess) {
    cleanup(); set_error("out of device memory"); return FMX_ERR_HIP;
  }
  std::vector<double> h_part((size_t)part_cap), h_y;
  std::vector<float> h_lab;
  const bool cls = e->cfg.task == FMX_TASK_CLASSIFICATION;
  if (cls) {
    h_y.resize((size_t)n); h_lab.resize((size_t)n);
    if (hipMemcpy(h_lab.data(), m->y, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { cleanup(); set_error("label download failed"); return FMX_ERR_HIP; }
  }
  const double alpha_0 = 1.0, gamma_0 = 1.0, beta_0 = 1.0, mu_0 = 0.0, w0_mean_0 = 0.0;  // init(), :59-90 (SURVEY A-7)
  double alpha = 1.0, w_lambda = 0.0, w_mu = 0.0;
  if (h_state_in) { alpha = h_state_in[0]; w_lambda = h_state_in[1]; w_mu = h_state_in[2]; }  // a chain continued (fmx_mcmc_train_from)
  auto bad = [](double x) { return std::isnan(x) || std::isinf(x); };
  int st = FMX_OK;
#define MC_HIP(call) do { if (st == FMX_OK) { hipError_t _e = (call); if (_e != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(_e)); st = FMX_ERR_HIP; } } } while (0)
  for (int it = 0; it < max_iter && st == FMX_OK; ++it) {
    const double* G = h_gammas + (size_t)it * 2;
    const double* Z = h_normals + (size_t)it * (2 + (size_t)p);
    RowsArgs a{};
    a.row_ptr = m->row_ptr; a.col = m->col; a.val = m->val; a.r0 = 0; a.nrows = n;
    a.V = e->dV; a.w = e->dw; a.vs = e->kp64; a.ws = 1; a.scal = e->scal; a.yhat = d_yhat; a.link = FMX_LINK_NONE;
    st = launch_rows_forward(e, a, false, true);
    if (st != FMX_OK) break;
    if (!cls) {
      hipLaunchKernelGGL(als_residual_k, dim3(row_grid), dim3(256), 0, e->stream, d_yhat, m->y, n, d_qe, (const double*)nullptr);
    } else {  // calculate_error with do_sample, one thread, rows in order
      MC_HIP(hipMemcpyAsync(h_y.data(), d_yhat, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
      MC_HIP(hipStreamSynchronize(e->stream));
      for (int64_t i = 0; i < n && st == FMX_OK; ++i) {
        const double ev = h_y[(size_t)i];
        h_y[(size_t)i] = (h_lab[(size_t)i] >= 0.0f) ? ev - h_trnorm_left(ev) : ev - (-h_trnorm_left(-ev));
      }
      MC_HIP(hipMemcpyAsync(d_yhat, h_y.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, e->stream));
      hipLaunchKernelGGL(als_pack_k, dim3(row_grid), dim3(256), 0, e->stream, d_yhat, n, d_qe);
    }
    // update_alpha, :359-380
    hipLaunchKernelGGL(mcmc_sumsq_partial_k, dim3((unsigned)np), dim3(WG_THREADS), 0, e->stream, d_qe, n, d_part);
    MC_HIP(hipMemcpyAsync(h_part.data(), d_part, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    MC_HIP(hipStreamSynchronize(e->stream));
    if (st != FMX_OK) break;
    {
      double gamma_n = gamma_0;
      for (int64_t i = 0; i < np; ++i) gamma_n += h_part[(size_t)i];
      (void)alpha_0;
      const double a_new = (2.0 / gamma_n) * G[0];  // Rf_rgamma((alpha_0 + n) / 2, 2 / gamma_n)
      if (!bad(a_new)) alpha = a_new;
    }
    if (e->hyper.k0) {  // update_w0, :160-188
      hipLaunchKernelGGL(als_w0_partial_k, dim3((unsigned)np), dim3(WG_THREADS), 0, e->stream, d_qe, n, e->scal, d_part);
      hipLaunchKernelGGL(als_w0_final_k, dim3(1), dim3(WG_THREADS), 0, e->stream, d_part, np, n, e->scal, e->hyper.reg0, alpha, w0_mean_0, 1, Z[0]);
      hipLaunchKernelGGL(als_shift_k, dim3(row_grid), dim3(256), 0, e->stream, d_qe, n, d_part + np);
    }
    if (e->hyper.k1) {
      hipLaunchKernelGGL(mcmc_wstats_partial_k, dim3((unsigned)npw), dim3(WG_THREADS), 0, e->stream, e->dw, p, w_mu, d_part);
      MC_HIP(hipMemcpyAsync(h_part.data(), d_part, (size_t)npw * 2 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
      MC_HIP(hipStreamSynchronize(e->stream));
      if (st != FMX_OK) break;
      double sum_w = 0.0, sum_sq = 0.0;
      for (int64_t i = 0; i < npw; ++i) { sum_w += h_part[(size_t)(2 * i)]; sum_sq += h_part[(size_t)(2 * i + 1)]; }
      {  // update_w_lambda, :415-445
        const double s_ = sum_sq + beta_0 * (w_mu - mu_0) * (w_mu - mu_0) + gamma_0;
        const double l_new = (2.0 / s_) * G[1];  // Rf_rgamma((alpha_0 + p + 1) / 2, 2 / s)
        if (!bad(l_new)) w_lambda = l_new;
      }
      {  // update_w_mu, :383-412
        const double mean = (sum_w + beta_0 * mu_0) / ((double)p + beta_0);
        const double var = 1.0 / (((double)p + beta_0) * w_lambda);
        const double mu_new = mean + std::sqrt(var) * Z[1];
        if (!bad(mu_new)) w_mu = mu_new;
      }
      MC_HIP(hipMemcpyAsync(d_z, Z + 2, (size_t)p * sizeof(double), hipMemcpyHostToDevice, e->stream));
      if (st == FMX_OK) st = w_sweep_guarded(e, m, d_qe, alpha, w_lambda, w_mu, (const double*)d_z);  // update_w, :190-270
    }
  }
#undef MC_HIP
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  cleanup();
  FMX_TRY(st);
  FMX_CHECK(err == hipSuccess, FMX_ERR_HIP, "MCMC training failed: %s", hipGetErrorString(err));
  if (h_state) { h_state[0] = alpha; h_state[1] = w_lambda; h_state[2] = w_mu; }
  return FMX_OK;
}

// per slab of features: sum (v_fj - mu)^2 of factor f (update_v_lambda, :489-493)
__global__ __launch_bounds__(WG_THREADS) void mcmc_vstats_partial_k(const double* __restrict__ V, int kp, int f, int64_t p, double mu, double* __restrict__ partials) {
  __shared__ double red[WG_THREADS];
  const int64_t base = (int64_t)blockIdx.x * ALS_SLAB;
  double a = 0.0;
  for (int i = threadIdx.x; i < ALS_SLAB; i += WG_THREADS) {
    const int64_t j = base + i;
    if (j < p) { const double d = V[(size_t)j * kp + f] - mu; a += d * d; }
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int off = WG_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

// update_v_lambda + update_v_mu (solver/MCMC_ALS_Learner.h:448-517), one attribute group; the O(p k) sums run on the device,
// the k scalar draws on the host (the caller's standard variates).  Shipped indexing of update_v_mu kept (A-8).
int launch_mcmc_v_hyper(fmx_engine* e, const double* h_gammas, const double* h_normals, double* v_lambda, double* v_mu, int sample) {
  const int64_t p = (int64_t)e->p;
  const int k = e->k;
  if (k == 0) return FMX_OK;
  const int64_t npw = (p + ALS_SLAB - 1) / ALS_SLAB;
  double* d_part = nullptr;
  FMX_HIP(hipMalloc(&d_part, (size_t)npw * sizeof(double)));
  std::vector<double> h_part((size_t)npw), v0((size_t)e->kp64);
  const double alpha_0 = 1.0, gamma_0 = 1.0, beta_0 = 1.0, mu_0 = 0.0;
  auto bad = [](double x) { return std::isnan(x) || std::isinf(x); };
  int st = FMX_OK;
  for (int f = 0; f < k && st == FMX_OK; ++f) {
    hipLaunchKernelGGL(mcmc_vstats_partial_k, dim3((unsigned)npw), dim3(WG_THREADS), 0, e->stream, e->dV, e->kp64, f, p, v_mu[f], d_part);
    if (hipMemcpyAsync(h_part.data(), d_part, (size_t)npw * sizeof(double), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess) { set_error("v statistics failed"); st = FMX_ERR_HIP; break; }
    double g = 0.0;
    for (int64_t i = 0; i < npw; ++i) g += h_part[(size_t)i];
    g += beta_0 * (v_mu[f] - mu_0) * (v_mu[f] - mu_0) + gamma_0;
    const double a = alpha_0 + (double)p + 1.0;
    const double l_new = sample ? (2.0 / g) * h_gammas[f] : a / g;
    if (!bad(l_new)) v_lambda[f] = l_new;
  }
  if (st == FMX_OK && hipMemcpy(v0.data(), e->dV, (size_t)e->kp64 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { set_error("download of V row 0 failed"); st = FMX_ERR_HIP; }
  for (int f = 0; f < k && st == FMX_OK; ++f) {
    double m = 0.0;
    for (int64_t i = 0; i < p; ++i) m += v0[(size_t)f];  // sic: sum over i of v(f, attr_group[i]) = v(f, 0), p times (:462)
    m = (m + beta_0 * mu_0) / ((double)p + beta_0);
    const double var = 1.0 / (((double)p + beta_0) * v_lambda[f]);
    const double mu_new = sample ? m + std::sqrt(var) * h_normals[f] : m;
    if (!bad(mu_new)) v_mu[f] = mu_new;
  }
  (void)hipFree(d_part);
  return st;
}

// sum_f lambda_f sum_j (theta_fj - mu_f)^2 over the V table (all factors) or over w: the prior's part of the objective an ALS sweep descends
static int penalty_sum(fmx_engine* e, bool w, const double* h_lambda, const double* h_mu, double* out) {
  *out = 0.0;
  if (!h_lambda) return FMX_OK;
  const int64_t p = (int64_t)e->p;
  const int64_t npw = (p + ALS_SLAB - 1) / ALS_SLAB;
  const int nf = w ? 1 : e->k;
  bool any = false;
  for (int f = 0; f < nf; ++f) any = any || h_lambda[f] != 0.0;
  if (!any || p == 0) return FMX_OK;
  double* d_part = nullptr;
  FMX_HIP(hipMalloc(&d_part, (size_t)npw * 2 * sizeof(double)));
  std::vector<double> h((size_t)npw * 2);
  int st = FMX_OK;
  double total = 0.0;
  for (int f = 0; f < nf && st == FMX_OK; ++f) {
    if (h_lambda[f] == 0.0) continue;
    const double mu = h_mu ? h_mu[f] : 0.0;
    if (w) hipLaunchKernelGGL(mcmc_wstats_partial_k, dim3((unsigned)npw), dim3(WG_THREADS), 0, e->stream, e->dw, p, mu, d_part);
    else hipLaunchKernelGGL(mcmc_vstats_partial_k, dim3((unsigned)npw), dim3(WG_THREADS), 0, e->stream, e->dV, e->kp64, f, p, mu, d_part);
    if (hipMemcpyAsync(h.data(), d_part, (size_t)npw * (w ? 2 : 1) * sizeof(double), hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess) { set_error("prior term of the ALS objective failed"); st = FMX_ERR_HIP; break; }
    double sq = 0.0;
    for (int64_t i = 0; i < npw; ++i) sq += w ? h[(size_t)(2 * i + 1)] : h[(size_t)i];   // (w statistics: {sum w, sum (w - mu)^2} per slab)
    total += h_lambda[f] * sq;
  }
  (void)hipFree(d_part);
  *out = total;
  return st;
}

int launch_als_vsweep_device(fmx_engine* e, fmx_matrix* m, double* d_error, double alpha, const double* h_lambda, const double* h_mu, const double* d_znorm) {
  double2* qe = sweep_pairs(e, m->n);
  FMX_CHECK(qe != nullptr, FMX_ERR_HIP, "out of device memory");
  return launch_als_vsweep(e, m, d_error, reinterpret_cast<double*>(qe), alpha, h_lambda, h_mu, d_znorm);
}

int launch_als_vsweep(fmx_engine* e, fmx_matrix* m, double* d_error, double* d_qe_raw, double alpha, const double* h_lambda, const double* h_mu,
                      const double* d_znorm) {
  FMX_CHECK(m->rows_sorted, FMX_ERR_INVALID, "the ALS sweep needs every row's columns strictly ascending (as R's dgCMatrix rows are)");
  FMX_TRY(build_full_csc(m, e->stream));
  FMX_TRY(build_plan(m, e->stream, e->cfg.als_max_levels));
  double2* d_qe = reinterpret_cast<double2*>(d_qe_raw);
  const unsigned row_grid = (unsigned)((m->n + 255) / 256);
  hipLaunchKernelGGL(als_pack_k, dim3(row_grid), dim3(256), 0, e->stream, d_error, m->n, d_qe);
  FMX_TRY(v_sweep_guarded(e, m, d_qe, alpha, h_lambda, h_mu, d_znorm));
  hipLaunchKernelGGL(als_unpack_k, dim3(row_grid), dim3(256), 0, e->stream, d_qe, m->n, d_error);
  hipError_t err = hipGetLastError();
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  FMX_CHECK(err == hipSuccess, FMX_ERR_HIP, "ALS sweep failed: %s", hipGetErrorString(err));
  return FMX_OK;
}

}  // namespace fmx
