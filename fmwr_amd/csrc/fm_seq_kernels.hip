// Sequential-exact learners (FMX_MODE_SEQUENTIAL): the reference's algorithm as it is written --
// one example per update, examples in the reference's visiting order, fp64 parameters -- executed by ONE
// wavefront so that every example sees the parameters the previous one left.
//
//   SGD  : solver/SGD_Learner.h:88-138   (L2 lazy decay or cumulative L1 penalty)
//   FTRL : solver/FTRL_Learner.h:74-116 + calculate_param :158-202
//   TDAP : solver/TDAP_Learner.h:87-146 + calculate_param :189-233 (its z_w[i] indexing bug kept, SURVEY A-6)
//
// Lane mapping inside the wave: the forward and the V update put factor f on lane f (f + 64 for k > 64)
// and walk the row's nonzeros in row order, so sum_f / sum_sqr_f are accumulated in exactly the
// reference's association (core/Model.h:83-97); the linear term and the pairwise term are summed in the
// reference's order too.  With -ffp-contract=off the only arithmetic difference to the CPU code is the
// device exp().  The w update puts nonzero t on lane t (rows whose columns are not strictly ascending may
// hold a column twice and take a one-lane serial path instead).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <type_traits>

#include <rocprim/rocprim.hpp>

#include "fmx_internal.h"

namespace fmx {

struct SeqArgs {
  const int64_t* row_ptr;
  const uint32_t* col;
  const float* val;
  const float* y;
  const int64_t* order;
  int64_t count;
  const int64_t* ex_b;  // per example: first entry, length, label (gathered by seq_prepare_k so the walker reads them in order)
  const int* ex_len;
  const float* ex_y;
  double *V, *w, *sV, *sw, *nV, *nw;
  double *t1V, *t1w, *t2V, *t2w, *t3V, *t3w;  // TDAP: nu, delta, h (u in nV/nw, z in sV/sw)
  double* scal;
  int k, kp;
  int sorted_rows;  // every row strictly ascending in col => no duplicate column inside a row
};

__device__ __forceinline__ double bcast(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t bcast(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ float bcast(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// Wave sums without the LDS crossbar (the reassociated learner's workers: fifteen waves' ds_bpermutes were what made every LDS poll of the workgroup a ~500-clock
// trip).  x + x[lane ^ 32] and x + x[lane ^ 16] through gfx950's v_permlane32_swap / v_permlane16_swap (both results added: a + b == b + a, the bits of
// x += __shfl_xor(x, 32 / 16)); the whole butterfly 32, 16, 8, 4, 2, 1 with DPP row rotations for the last four (after the 32-, 16- and 8-steps a lane's value
// depends on its index mod 8 only, so the lane 4 (2, 1) to its right holds what lane ^ 4 (2, 1) holds): the same tree, the same bits as the __shfl_xor loop.
#define FMX_SEQ_DPP64(x, ctrl) __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xF, 0xF, false))
__device__ __forceinline__ double xor32_add(double x) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double xor16_add(double x) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double seq_butterfly_allsum(double x) {
  x = xor32_add(x);
  x = xor16_add(x);
  x += FMX_SEQ_DPP64(x, 0x128);   // row_ror:8
  x += FMX_SEQ_DPP64(x, 0x124);   // row_ror:4
  x += FMX_SEQ_DPP64(x, 0x122);   // row_ror:2
  x += FMX_SEQ_DPP64(x, 0x121);   // row_ror:1
  return x;
}

__device__ __forceinline__ double seq_grad_mult(const Hyper& h, double y_hat, float y) {
  if (h.task == FMX_TASK_REGRESSION) {
    y_hat = fmin(h.max_t, y_hat);
    y_hat = fmax(h.min_t, y_hat);
    return -((double)y - y_hat);
  }
  return -(double)y * (1.0 - 1.0 / (1.0 + exp(-(double)y * y_hat)));
}

__device__ __forceinline__ void seq_penalty(double& theta, double u, double& q) {  // SGD_Learner.h:195-204
  const double old = theta;
  if (theta > 0) theta = fmax(0.0, old - (u + q));
  else if (theta < 0) theta = fmin(0.0, old + (u - q));
  q += theta - old;
}

__device__ __forceinline__ double seq_prox(double z, double n, double l1, double l2, double alpha, double beta) {
  if (fabs(z) <= l1) return 0.0;
  const double sign = z < 0.0 ? -1.0 : 1.0;
  return -(z - sign * l1) / ((beta + sqrt(n)) / alpha + l2);
}

// One coordinate's TDAP accumulation (TDAP_Learner.h:97-105): returns nothing, updates the five state values.
__device__ __forceinline__ void tdap_coord(double g, double theta, double alpha, double egamma, double& u, double& nu, double& delta,
                                           double& h, double& z) {
  const double u_old = u;
  u += g * g;
  nu += g;
  const double sigma = (sqrt(u) - sqrt(u_old)) / alpha;
  delta = egamma * (delta + sigma);
  h = egamma * (h + sigma * theta);
  z = nu - h;
}

__device__ __forceinline__ double tdap_prox(double z, double delta, double l1, double l2) {  // TDAP_Learner.h:208-213
  if (fabs(z) <= l1) return 0.0;
  const double sign = z < 0.0 ? -1.0 : 1.0;
  return -(z - sign * l1) / (delta + l2);
}

constexpr int FI = 2;  // factor slots per lane: k <= 128
constexpr int UB = 8;  // nonzeros whose V loads are issued together on the no-duplicate path

template <int KIND>
__device__ __forceinline__ void seq_w_one(const SeqArgs& a, const Hyper& h, uint32_t c, double x, double mult, double uw) {
  if constexpr (KIND == UPD_TDAP) {  // TDAP_Learner.h:110-127
    double u = a.nw[c], nu = a.t1w[c], dl = a.t2w[c], hh = a.t3w[c], z;
    tdap_coord(mult * x, a.w[c], h.alpha_w, h.egamma, u, nu, dl, hh, z);
    a.nw[c] = u; a.t1w[c] = nu; a.t2w[c] = dl; a.t3w[c] = hh; a.sw[c] = z;
  } else if constexpr (KIND == UPD_FTRL) {  // FTRL_Learner.h:90-97
    const double g = mult * x;
    const double n_old = a.nw[c];
    const double n_new = n_old + g * g;
    a.nw[c] = n_new;
    const double delta = (sqrt(n_new) - sqrt(n_old)) / h.alpha_w;
    a.sw[c] += g - delta * a.w[c];
  } else {  // SGD_Learner.h:114-120
    double wv = a.w[c];
    wv -= h.lr * mult * x;
    if constexpr (KIND == UPD_SGD_L1) { double q = a.sw[c]; seq_penalty(wv, uw, q); a.sw[c] = q; }
    else wv -= h.lr * h.regw * wv;
    a.w[c] = wv;
  }
}

template <int KIND>
__device__ __forceinline__ void seq_v_one(const SeqArgs& a, const Hyper& h, size_t at, double vv, double sum, double x,
                                          double mult, double uv) {
  if constexpr (KIND == UPD_TDAP) {  // TDAP_Learner.h:133-141
    const double g = mult * (sum * x - vv * x * x);
    double u = a.nV[at], nu = a.t1V[at], dl = a.t2V[at], hh = a.t3V[at], z;
    tdap_coord(g, vv, h.alpha_v, h.egamma, u, nu, dl, hh, z);
    a.nV[at] = u; a.t1V[at] = nu; a.t2V[at] = dl; a.t3V[at] = hh; a.sV[at] = z;
  } else if constexpr (KIND == UPD_FTRL) {  // FTRL_Learner.h:106-111
    const double g = mult * (sum * x - vv * x * x);
    const double n_old = a.nV[at];
    const double n_new = n_old + g * g;
    a.nV[at] = n_new;
    const double delta = (sqrt(n_new) - sqrt(n_old)) / h.alpha_v;
    a.sV[at] += g - delta * vv;
  } else {  // SGD_Learner.h:128-136
    const double grad = sum * x - vv * x * x;
    vv -= h.lr * mult * grad;
    if constexpr (KIND == UPD_SGD_L1) { double q = a.sV[at]; seq_penalty(vv, uv, q); a.sV[at] = q; }
    else vv -= h.lr * h.regv * vv;
    a.V[at] = vv;
  }
}

// ---- fast example: rows of at most FAST_NZ strictly ascending columns, k <= 64 ---------------------------------------
// All gathers of the example (V rows, optimizer state, w) are issued as ONE batch into registers, the forward runs from
// registers in the reference's order, and every coordinate is updated from its registers and stored once.  With no column
// repeated inside the row, a coordinate's FTRL/TDAP prox depends only on its own state, so it is applied right after the
// accumulation instead of in a second pass (same values as calculate_param).
constexpr int FAST_NZ = 32;

template <int KIND> struct SeqState { static constexpr int N = KIND == UPD_SGD_L2 ? 0 : KIND == UPD_SGD_L1 ? 1 : KIND == UPD_FTRL ? 2 : 4; };

// one coordinate's step on registers; `grad` is x for w and (sum*x - theta*x*x) for V.  st: L1 {q}; FTRL {z, n}; TDAP {u, nu, delta, h}
// returns the z value for TDAP (stored by the caller), 0 otherwise
template <int KIND>
__device__ __forceinline__ double coord_seq(const Hyper& h, bool is_w, bool accumulate, double& theta, double grad, double mult,
                                            double u_pen, double* st) {
  const double alpha = is_w ? h.alpha_w : h.alpha_v, beta = is_w ? h.beta_w : h.beta_v;
  const double l1 = is_w ? h.l1w : h.l1v, l2 = is_w ? h.l2w : h.l2v, reg = is_w ? h.regw : h.regv;
  if constexpr (KIND == UPD_SGD_L2) {
    theta -= h.lr * mult * grad;
    theta -= h.lr * reg * theta;
    return 0.0;
  } else if constexpr (KIND == UPD_SGD_L1) {
    theta -= h.lr * mult * grad;
    seq_penalty(theta, u_pen, st[0]);
    return 0.0;
  } else if constexpr (KIND == UPD_FTRL) {
    if (accumulate) {
      const double g = mult * grad;
      const double n_old = st[1];
      st[1] += g * g;
      const double delta = (sqrt(st[1]) - sqrt(n_old)) / alpha;
      st[0] += g - delta * theta;
    }
    theta = seq_prox(st[0], st[1], l1, l2, alpha, beta);
    return 0.0;
  } else {
    double z = 0.0;
    tdap_coord(mult * grad, theta, alpha, h.egamma, st[0], st[1], st[2], st[3], z);
    if (!is_w) theta = tdap_prox(z, st[2], l1, l2);
    return z;
  }
}

template <int KIND>
__device__ __forceinline__ double* seq_state_ptr(const SeqArgs& a, bool is_w, int j) {
  if constexpr (KIND == UPD_SGD_L1) return is_w ? a.sw : a.sV;
  if constexpr (KIND == UPD_FTRL) return j == 0 ? (is_w ? a.sw : a.sV) : (is_w ? a.nw : a.nV);
  if constexpr (KIND == UPD_TDAP) return j == 0 ? (is_w ? a.nw : a.nV) : j == 1 ? (is_w ? a.t1w : a.t1V) : j == 2 ? (is_w ? a.t2w : a.t2V) : (is_w ? a.t3w : a.t3V);
  return nullptr;
}

__global__ void seq_prepare_k(const int64_t* __restrict__ order, int64_t count, const int64_t* __restrict__ row_ptr, const float* __restrict__ y,
                              int64_t* __restrict__ ex_b, int* __restrict__ ex_len, float* __restrict__ ex_y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int64_t r = order[i];
  ex_b[i] = row_ptr[r];
  ex_len[i] = (int)(row_ptr[r + 1] - row_ptr[r]);
  ex_y[i] = y[r];
}

template <int KIND>
__global__ __launch_bounds__(64) void fm_seq_learn_k(SeqArgs a, Hyper h) {
  const int lane = threadIdx.x;
  const int k = a.k, kp = a.kp;
  const bool k0 = h.k0 != 0, k1 = h.k1 != 0;
  double w0 = a.scal[SC_W0], z0 = a.scal[SC_Z0], n0 = a.scal[SC_N0], uw = a.scal[SC_UW], uv = a.scal[SC_UV];
  double t_nu = a.scal[SC_T_NU], t_delta = a.scal[SC_T_DELTA], t_h = a.scal[SC_T_H];

  constexpr int NS = SeqState<KIND>::N;
  const bool fast_ok = a.sorted_rows && k <= 64;
  // the first 64 entries of the NEXT example are fetched while the current one is processed
  uint32_t ncol = 0u;
  float nx = 0.f;
  if (a.count > 0 && lane < a.ex_len[0]) { ncol = a.col[a.ex_b[0] + lane]; nx = a.val[a.ex_b[0] + lane]; }

  for (int64_t ex = 0; ex < a.count; ++ex) {
    const int64_t b = a.ex_b[ex];
    const int len = a.ex_len[ex];
    const float yv = a.ex_y[ex];
    const uint32_t ccol = ncol;
    const float cx = nx;
    if (ex + 1 < a.count) {
      const int64_t nb = a.ex_b[ex + 1];
      const bool nv = lane < a.ex_len[ex + 1];
      ncol = nv ? a.col[nb + lane] : 0u;
      nx = nv ? a.val[nb + lane] : 0.f;
    }
    if constexpr (KIND == UPD_SGD_L1) { uw += h.lr * h.regw; uv += h.lr * h.regv; }  // SGD_Learner.h:92-97

    if (fast_ok && len <= FAST_NZ) {
      const bool tv = lane < len;            // this lane owns nonzero `lane` (w side)
      const bool fv = lane < k;              // this lane owns factor `lane` (V side)
      const uint32_t mycol = tv ? ccol : 0u;
      const double myx = tv ? (double)cx : 0.0;
      // ---- one batch of gathers.  Every load is UNCONDITIONAL (idle lanes / slots read a valid dummy address: column 0,
      // factor 0) -- loads under divergent control flow make the compiler wait vmcnt(0) before every later store, and vmcnt
      // counts stores too, so each store would wait for the previous one to complete.
      const int fl = fv ? lane : 0;
      double myw = a.w[mycol];
      double stw[NS > 0 ? NS : 1];
#pragma unroll
      for (int j = 0; j < NS; ++j) stw[j] = seq_state_ptr<KIND>(a, true, j)[mycol];
      double vv[FAST_NZ];
      double stv[NS > 0 ? NS : 1][FAST_NZ];
#pragma unroll
      for (int u = 0; u < FAST_NZ; ++u) {
        const size_t at = (size_t)bcast(mycol, u) * kp + fl;
        vv[u] = a.V[at];
#pragma unroll
        for (int j = 0; j < NS; ++j) stv[j][u] = seq_state_ptr<KIND>(a, false, j)[at];
      }
      // ---- forward from registers, core/Model.h:75-103 (row order; linear term first, then the factors in order)
      double s1 = 0.0, q1 = 0.0;
      double pred = k0 ? w0 : 0.0;
      const double wlin = k1 ? myw : 0.0;
#pragma unroll
      for (int u = 0; u < FAST_NZ; ++u) {  // slots beyond the row carry x = 0: they add +0.0, which changes nothing
        const double xu = bcast(myx, u);
        pred += bcast(wlin, u) * xu;
        const double tmp = vv[u] * xu;
        s1 += tmp;
        q1 += tmp * tmp;
      }
      for (int fl = 0; fl < k; ++fl) {
        const double sf = bcast(s1, fl), qf = bcast(q1, fl);
        pred += 0.5 * (sf * sf - qf);
      }
      const double mult = seq_grad_mult(h, pred, yv);
      // ---- w0
      if (k0) {
        if constexpr (KIND == UPD_TDAP) tdap_coord(mult, w0, h.alpha_w, h.egamma, n0, t_nu, t_delta, t_h, z0);
        else if constexpr (KIND == UPD_FTRL) {
          const double n_old = n0;
          n0 += mult * mult;
          const double delta = (sqrt(n0) - sqrt(n_old)) / h.alpha_w;
          z0 += mult - delta * w0;
        } else w0 -= h.lr * (mult + h.reg0 * w0);
      }
      if constexpr (KIND == UPD_FTRL) w0 = -z0 * h.alpha_w / (h.beta_w + sqrt(n0));
      if constexpr (KIND == UPD_TDAP) w0 = -z0 / t_delta;
      // ---- w (lane = nonzero)
      if (tv && (k1 || KIND == UPD_FTRL)) {
        if constexpr (KIND == UPD_TDAP) {
          const double z = coord_seq<KIND>(h, true, true, myw, myx, mult, uw, stw);
          a.sw[mycol] = z;
        } else {
          coord_seq<KIND>(h, true, k1, myw, myx, mult, uw, stw);
          a.w[mycol] = myw;
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) if (k1) seq_state_ptr<KIND>(a, true, j)[mycol] = stw[j];
      }
      // ---- V (lane = factor)
#pragma unroll
      for (int u = 0; u < FAST_NZ; ++u) {
        const double xu = bcast(myx, u);
        const size_t at = (size_t)bcast(mycol, u) * kp + lane;
        if (u < len && fv) {
          double th = vv[u];
          double st[NS > 0 ? NS : 1];
#pragma unroll
          for (int j = 0; j < NS; ++j) st[j] = stv[j][u];
          const double grad = s1 * xu - th * xu * xu;
          const double z = coord_seq<KIND>(h, false, true, th, grad, mult, uv, st);
          a.V[at] = th;
          if constexpr (KIND == UPD_TDAP) a.sV[at] = z;
#pragma unroll
          for (int j = 0; j < NS; ++j) seq_state_ptr<KIND>(a, false, j)[at] = st[j];
        }
      }
      if constexpr (KIND == UPD_TDAP) {  // w prox reads z_w by POSITION (TDAP_Learner.h:207): needs every z_w of the example stored
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (tv) a.w[mycol] = tdap_prox(a.sw[lane], k1 ? stw[2] : a.t2w[mycol], h.l1w, h.l2w);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      continue;
    }

    // ---------------------------------------------------------------- forward, core/Model.h:75-103
    double s[FI], q[FI];
#pragma unroll
    for (int i = 0; i < FI; ++i) { s[i] = 0.0; q[i] = 0.0; }
    double pred = k0 ? w0 : 0.0;
    for (int c = 0; c < len; c += 64) {
      const int t = c + lane;
      const bool valid = t < len;
      const uint32_t mycol = valid ? a.col[b + t] : 0u;
      const float myx = valid ? a.val[b + t] : 0.f;
      const double myw = (valid && k1) ? a.w[mycol] : 0.0;
      const int n_in = (len - c < 64) ? len - c : 64;
#pragma unroll 4
      for (int u = 0; u < n_in; ++u) {
        const uint32_t cu = bcast(mycol, u);
        const double xu = (double)bcast(myx, u);
        if (k1) pred += bcast(myw, u) * xu;
#pragma unroll
        for (int i = 0; i < FI; ++i) {
          const int f = lane + 64 * i;
          if (f < k) {
            const double tmp = a.V[(size_t)cu * kp + f] * xu;
            s[i] += tmp;
            q[i] += tmp * tmp;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      const int nf = (k - 64 * i < 64) ? k - 64 * i : 64;
      for (int fl = 0; fl < nf; ++fl) {
        const double sf = bcast(s[i], fl), qf = bcast(q[i], fl);
        pred += 0.5 * (sf * sf - qf);
      }
    }
    const double mult = seq_grad_mult(h, pred, yv);

    // ---------------------------------------------------------------- w0
    if (k0) {
      if constexpr (KIND == UPD_TDAP) {  // TDAP_Learner.h:96-106
        tdap_coord(mult, w0, h.alpha_w, h.egamma, n0, t_nu, t_delta, t_h, z0);
      } else if constexpr (KIND == UPD_FTRL) {  // FTRL_Learner.h:80-86
        const double n_old = n0;
        n0 += mult * mult;
        const double delta = (sqrt(n0) - sqrt(n_old)) / h.alpha_w;
        z0 += mult - delta * w0;
      } else {
        w0 -= h.lr * (mult + h.reg0 * w0);  // SGD_Learner.h:106-109
      }
    }

    // ---------------------------------------------------------------- w
    if (k1) {
      if (a.sorted_rows) {
        for (int c = 0; c < len; c += 64) {
          const int t = c + lane;
          if (t < len) seq_w_one<KIND>(a, h, a.col[b + t], (double)a.val[b + t], mult, uw);
        }
      } else if (lane == 0) {
        for (int t = 0; t < len; ++t) seq_w_one<KIND>(a, h, a.col[b + t], (double)a.val[b + t], mult, uw);
      }
    }

    // ---------------------------------------------------------------- V
    for (int c = 0; c < len; c += 64) {
      const int t = c + lane;
      const bool valid = t < len;
      const uint32_t mycol = valid ? a.col[b + t] : 0u;
      const float myx = valid ? a.val[b + t] : 0.f;
      const int n_in = (len - c < 64) ? len - c : 64;
      if (a.sorted_rows) {
        for (int u0 = 0; u0 < n_in; u0 += UB) {
          double vv[UB][FI];
          uint32_t cu[UB];
          double xu[UB];
#pragma unroll
          for (int j = 0; j < UB; ++j) {
            const int u = (u0 + j < n_in) ? u0 + j : u0;
            cu[j] = bcast(mycol, u);
            xu[j] = (double)bcast(myx, u);
#pragma unroll
            for (int i = 0; i < FI; ++i) {
              const int f = lane + 64 * i;
              vv[j][i] = (f < k) ? a.V[(size_t)cu[j] * kp + f] : 0.0;
            }
          }
#pragma unroll
          for (int j = 0; j < UB; ++j) {
            if (u0 + j < n_in) {
#pragma unroll
              for (int i = 0; i < FI; ++i) {
                const int f = lane + 64 * i;
                if (f < k) seq_v_one<KIND>(a, h, (size_t)cu[j] * kp + f, vv[j][i], s[i], xu[j], mult, uv);
              }
            }
          }
        }
      } else {
        for (int u = 0; u < n_in; ++u) {
          const uint32_t cu = bcast(mycol, u);
          const double xu = (double)bcast(myx, u);
#pragma unroll
          for (int i = 0; i < FI; ++i) {
            const int f = lane + 64 * i;
            if (f < k) {
              const size_t at = (size_t)cu * kp + f;
              seq_v_one<KIND>(a, h, at, a.V[at], s[i], xu, mult, uv);
            }
          }
        }
      }
    }

    // ---------------------------------------------------------------- FTRL calculate_param, :158-202
    if constexpr (KIND == UPD_FTRL) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      w0 = -z0 * h.alpha_w / (h.beta_w + sqrt(n0));
      for (int c = 0; c < len; c += 64) {
        const int t = c + lane;
        const bool valid = t < len;
        const uint32_t mycol = valid ? a.col[b + t] : 0u;
        if (valid) a.w[mycol] = seq_prox(a.sw[mycol], a.nw[mycol], h.l1w, h.l2w, h.alpha_w, h.beta_w);
        const int n_in = (len - c < 64) ? len - c : 64;
        for (int u = 0; u < n_in; ++u) {
          const uint32_t cu = bcast(mycol, u);
#pragma unroll
          for (int i = 0; i < FI; ++i) {
            const int f = lane + 64 * i;
            if (f < k) {
              const size_t at = (size_t)cu * kp + f;
              a.V[at] = seq_prox(a.sV[at], a.nV[at], h.l1v, h.l2v, h.alpha_v, h.beta_v);
            }
          }
        }
      }
    }
    // ---------------------------------------------------------------- TDAP calculate_param, :189-233
    if constexpr (KIND == UPD_TDAP) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      w0 = -z0 / t_delta;
      for (int c = 0; c < len; c += 64) {
        const int t = c + lane;
        const bool valid = t < len;
        const uint32_t mycol = valid ? a.col[b + t] : 0u;
        // sic: z_w is indexed by the POSITION inside the row, not by the column (TDAP_Learner.h:207, SURVEY A-6)
        if (valid) a.w[mycol] = tdap_prox(a.sw[t], a.t2w[mycol], h.l1w, h.l2w);
        const int n_in = (len - c < 64) ? len - c : 64;
        for (int u = 0; u < n_in; ++u) {
          const uint32_t cu = bcast(mycol, u);
#pragma unroll
          for (int i = 0; i < FI; ++i) {
            const int f = lane + 64 * i;
            if (f < k) {
              const size_t at = (size_t)cu * kp + f;
              a.V[at] = tdap_prox(a.sV[at], a.t2V[at], h.l1v, h.l2v);
            }
          }
        }
      }
    }
    // the next example must see these stores (other lanes of this wave wrote them)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }

  if (lane == 0) {
    a.scal[SC_W0] = w0; a.scal[SC_Z0] = z0; a.scal[SC_N0] = n0; a.scal[SC_UW] = uw; a.scal[SC_UV] = uv;
    a.scal[SC_T_NU] = t_nu; a.scal[SC_T_DELTA] = t_delta; a.scal[SC_T_H] = t_h;
  }
}

// ---- windowed learner ------------------------------------------------------------------------------------------------
// The same algorithm, the same visiting order, the same arithmetic in the same association -- but consecutive examples
// that share no feature touch disjoint parameters, so only the w0 chain orders them.  A workgroup of NW waves takes a
// GROUP of up to NW consecutive examples that are pairwise feature-disjoint (one example per wave):
//   A  every wave gathers its example's parameters and forms the terms of y_hat (w_j x_j in row order, then
//      0.5 (s_f^2 - q_f) in factor order) -- everything of the forward except the running sum that starts at w0;
//   S  wave 0 runs the scalar chain example by example in order: pred = w0 + terms (the reference's association,
//      core/Model.h:77-100), the gradient multiplier, the w0 step -- the only truly sequential work;
//   C  every wave applies its example's update from its registers.
// A group ends at the first example that shares a feature with an earlier member (it then reads what that member wrote).
// `conf[t]`, the last earlier example of the launch sharing a feature with t, comes from a stable sort of the launch's
// (feature, example) pairs; TDAP examples holding a feature id < NZ run alone (its w prox reads z_w by POSITION, A-6).
// Bitwise the same results as the one-wave kernel above (tests/test_gpu_seq_window.py).
// NZ, the entries a row may hold, is 32, or 64 when k <= 32 (rows of 33..64 entries, e.g. Criteo's 39: the lane layout below
// keeps NZ / Q slots per lane, so the longer rows still fit the registers).
constexpr int WIN_NZ_MAX = 64;
// Lane mapping of a wave: KL lanes per factor block (16 / 32 / 64 for k <= 16 / 32 / 64) and Q = 64 / KL blocks, block q
// holding the nonzeros u = q, q + Q, q + 2Q ...: with k = 16 all 64 lanes gather and update (8 coordinates each instead
// of 32 on 16 lanes), and a lane keeps NZ / Q x (1 + state) doubles in registers, which is what sets the number of
// waves (= examples per group) a workgroup can hold.
template <int KIND, int KL, int NZ> struct SeqWin {
  static constexpr int Q = 64 / KL;
  static constexpr int SL = NZ / Q;  // nonzero slots per lane
  static constexpr int TERMS = NZ + 64;
  // 256 (512) VGPRs per lane at 8 (4) waves per workgroup; a lane holds SL x (1 + state) doubles of parameters, with
  // several blocks also SL columns and x values, on top of ~120 registers of everything else
  static constexpr int REGS = 2 * SL * (1 + SeqState<KIND>::N) + (Q > 1 ? 3 * SL : 0) + 120;
#ifdef FMX_SEQ_NW16   // diagnostic build (profiles/variant_build.sh): 16 waves where the shape's registers are fewest -- the chain wave then has 15 examples per step to hide the workers' gathers behind
  static constexpr int NW = (KIND == UPD_SGD_L2 && KL == 16 && NZ == 32) ? 16 : (REGS > 270 ? 4 : 8);
#elif defined(FMX_SEQ_NW)   // diagnostic build: any number of waves for that shape (-DFMX_SEQ_NW=10 / 12 / 14)
  static constexpr int NW = (KIND == UPD_SGD_L2 && KL == 16 && NZ == 32) ? FMX_SEQ_NW : (REGS > 270 ? 4 : 8);
#else
  static constexpr int NW = REGS > 270 ? 4 : 8;
#endif
};

struct WinArgs {
  const uint2* packed;  // [count][NZ] (column, x bits), idle slots (0, +0.0f)
  const int* ex_len;
  const float* ex_y;
  const int* conf;      // [count] last earlier example sharing a feature (-1: none; t itself: must run alone)
  int count;
  int debug_lose;       // test hook (fmx_debug_lose_next_seq_multiplier): the reassociated learner's worker never sees the multiplier tagged with this value (0: off)
};

__global__ void seq_pack_k(const int64_t* __restrict__ ex_b, const int* __restrict__ ex_len, int count, int nz, const uint32_t* __restrict__ col,
                           const float* __restrict__ val, int tdap, uint2* __restrict__ packed, uint32_t* __restrict__ keys,
                           uint32_t* __restrict__ vals, int* __restrict__ conf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count * nz) return;
  const int t = i / nz, u = i % nz;
  const bool have = u < ex_len[t];
  const uint32_t c = have ? col[ex_b[t] + u] : 0u;
  packed[i] = make_uint2(c, have ? __float_as_uint(val[ex_b[t] + u]) : 0u);
  keys[i] = have ? c : 0xFFFFFFFFu;
  vals[i] = (uint32_t)t;
  if (tdap && have && c < (uint32_t)nz) atomicMax(conf + t, t);  // z_w[c] is what other examples read by position
}

__global__ void seq_conf_k(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int n, int* __restrict__ conf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= 0 || i >= n) return;
  if (keys[i] != 0xFFFFFFFFu && keys[i] == keys[i - 1]) atomicMax(conf + vals[i], (int)vals[i - 1]);  // stable sort: vals ascend inside a key
}

template <int KIND, int KL, int NZ>
__device__ __forceinline__ void seq_window_body(const SeqArgs& a, const WinArgs& wa, const Hyper& h) {
  constexpr int NW = SeqWin<KIND, KL, NZ>::NW;
  constexpr int Q = SeqWin<KIND, KL, NZ>::Q;
  constexpr int SL = SeqWin<KIND, KL, NZ>::SL;
  constexpr int WIN_TERMS = SeqWin<KIND, KL, NZ>::TERMS;
  constexpr int NS = SeqState<KIND>::N;
  __shared__ double terms[NW][WIN_TERMS];
  __shared__ double s_mult[NW], s_uw[NW], s_uv[NW];
  __shared__ float s_y[NW];
  __shared__ int s_cand[NW], s_need[NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = a.k, kp = a.kp;
  const bool k0 = h.k0 != 0, k1 = h.k1 != 0;
  const int k16 = (k + 15) & ~15;  // the chain adds the pairwise terms sixteen at a time (two eights)
  // the scalar chain lives in wave 0
  double w0 = a.scal[SC_W0], z0 = a.scal[SC_Z0], n0 = a.scal[SC_N0], uw = a.scal[SC_UW], uv = a.scal[SC_UV];
  double t_nu = a.scal[SC_T_NU], t_delta = a.scal[SC_T_DELTA], t_h = a.scal[SC_T_H];

  // this wave's candidate example of the group starting at g: its metadata is fetched one group ahead
  // Loaded unconditionally, on indices clamped into the piece: a wave past the end carries the last example's metadata, and every use
  // below is guarded by its own range test anyway.  (Inside `if (tt < count)` the loads sat behind a branch whose join waits for
  // them -- the "one group ahead" fetch was a round trip in FRONT of every group's gathers instead of one hidden behind them.)
  struct Meta { int conf_t, conf_g, len; uint2 en; float y; };
  auto fetch = [&](int gg) {
    Meta mt;
    const int last = wa.count - 1;
    const int tt = gg + wave < last ? gg + wave : last;
    mt.conf_t = wa.conf[tt];
    mt.conf_g = wa.conf[gg < last ? gg : last];
    mt.len = wa.ex_len[tt];
    mt.en = wa.packed[(size_t)tt * NZ + (lane & (NZ - 1))];
    mt.y = wa.ex_y[tt];
    return mt;
  };
  int g = 0;
  int unfenced = 0;  // stores of the examples >= unfenced may still be in flight (no fence since)
#ifdef FMX_SEQ_TIMING
  unsigned long long tA = 0, tS = 0, tC = 0, tV = 0, tF = 0, nG = 0, nF = 0, t0, t1;
#define FMX_T(acc) do { t1 = __builtin_amdgcn_s_memtime(); acc += t1 - t0; t0 = t1; } while (0)
#else
#define FMX_T(acc) do { } while (0)
#endif
  Meta cur = fetch(0);
  if (lane == 0) s_cand[wave] = (wave < wa.count) && (wave == 0 || (cur.conf_t < 0 && cur.conf_g != 0));
  __syncthreads();

  while (g < wa.count) {
#ifdef FMX_SEQ_TIMING
    t0 = __builtin_amdgcn_s_memtime(); ++nG;
#endif
    int G = 0;  // the group: the leading candidates (all flags read at once: an early-exit loop is NW dependent LDS round trips)
    {
      int c[NW];
#pragma unroll
      for (int i = 0; i < NW; ++i) c[i] = s_cand[i];
      bool run = true;
#pragma unroll
      for (int i = 0; i < NW; ++i) { run = run && c[i] != 0; G += run ? 1 : 0; }
    }
    const bool mine = wave < G;
    const int gn = g + G;             // where the next group starts: known now, so its metadata can travel during this group
    const Meta nxt = fetch(gn);

    // ---------------------------------------------------------------- A: gathers and the terms of y_hat
    const int fq = lane / KL, ff = lane % KL;  // this lane's nonzero block and factor (V side)
    const bool fv = ff < k;
    const int fl = fv ? ff : 0;
    const int len = cur.len;
    const bool tv = lane < len;                // this lane's nonzero (w side)
    const uint32_t mycol = tv ? cur.en.x : 0u;
    const double myx = tv ? (double)__uint_as_float(cur.en.y) : 0.0;
    double myw = 0.0, s1 = 0.0, q1 = 0.0;
    double stw[NS > 0 ? NS : 1];
    uint32_t cu[Q > 1 ? SL : 1];  // with one block the slot's column and x come straight from the owning lane (readlane)
    double xu[Q > 1 ? SL : 1], vv[SL];
    double stv[NS > 0 ? NS : 1][SL];
    if (mine) {
      myw = a.w[mycol];
#pragma unroll
      for (int j = 0; j < NS; ++j) stw[j] = seq_state_ptr<KIND>(a, true, j)[mycol];
#pragma unroll
      for (int j = 0; j < SL; ++j) {  // slot j of this lane is nonzero u = j * Q + fq (idle slots: column 0, x = 0)
        uint32_t cj;
        if constexpr (Q == 1) cj = bcast(mycol, j);
        else { cj = cu[j] = (uint32_t)__shfl((int)mycol, j * Q + fq); xu[j] = __shfl(myx, j * Q + fq); }
        const size_t at = (size_t)cj * kp + fl;
        vv[j] = a.V[at];
#pragma unroll
        for (int n = 0; n < NS; ++n) stv[n][j] = seq_state_ptr<KIND>(a, false, n)[at];
      }
#pragma unroll
      for (int u = 0; u < NZ; ++u) {  // core/Model.h:83-97: every factor's sums run over the nonzeros in row order
        double tmp;
        if constexpr (Q == 1) tmp = vv[u] * bcast(myx, u);
        else tmp = __shfl(vv[u / Q] * xu[u / Q], (u % Q) * KL + ff);  // from the block that holds nonzero u
        s1 += tmp;
        q1 += tmp * tmp;
      }
      if (lane < NZ) terms[wave][lane] = (k1 ? myw : 0.0) * myx;     // w_j x_j, the linear term's addends in row order
      if (lane < k16) terms[wave][NZ + lane] = fv ? 0.5 * (s1 * s1 - q1) : 0.0;  // core/Model.h:100, factor order (lanes of block 0)
      if (lane == 0) s_y[wave] = cur.y;
    }
    __syncthreads();
    FMX_T(tA);

    // ---------------------------------------------------------------- S: the scalar chain, in order
    if (wave == 0) {
      double cb0[8], cb1[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) cb0[i] = terms[0][i];
      for (int e = 0; e < G; ++e) {
        if constexpr (KIND == UPD_SGD_L1) { uw += h.lr * h.regw; uv += h.lr * h.regv; }  // SGD_Learner.h:92-97
        // every lane reads the same addresses (LDS broadcast) and runs the same chain: no cross-lane traffic in the chain.
        // The addends arrive eight at a time and in PAIRS of eights (k16: the factor slots are zero-filled to a multiple of 16),
        // each eight requested before the eight in front of it are added, the next example's first eight before this example's
        // multiplier is computed: the additions are the serial floor of the mode, the LDS latency need not be.  Same additions,
        // same order (+0.0 addends change nothing).
        const double* __restrict__ T = terms[e];
        double pred = k0 ? w0 : 0.0;
        auto pair = [&](const double* __restrict__ second, const double* __restrict__ after) {
#pragma unroll
          for (int i = 0; i < 8; ++i) cb1[i] = second[i];
#pragma unroll
          for (int i = 0; i < 8; ++i) pred += cb0[i];
#pragma unroll
          for (int i = 0; i < 8; ++i) cb0[i] = after[i];
#pragma unroll
          for (int i = 0; i < 8; ++i) pred += cb1[i];
        };
        const double* __restrict__ nextT = terms[e + 1 < G ? e + 1 : e];
#pragma unroll
        for (int u = 0; u < NZ; u += 16)  // the linear term's addends; the last `after` is the first factor eight (k = 0: there is none)
          pair(T + u + 8, (u + 16 < NZ || k16 > 0) ? T + u + 16 : nextT);
        for (int f = 0; f < k16; f += 16) pair(T + NZ + f + 8, f + 16 < k16 ? T + NZ + f + 16 : nextT);
        const double mult = seq_grad_mult(h, pred, s_y[e]);
        if (k0) {
          if constexpr (KIND == UPD_TDAP) tdap_coord(mult, w0, h.alpha_w, h.egamma, n0, t_nu, t_delta, t_h, z0);
          else if constexpr (KIND == UPD_FTRL) {
            const double n_old = n0;
            n0 += mult * mult;
            const double delta = (sqrt(n0) - sqrt(n_old)) / h.alpha_w;
            z0 += mult - delta * w0;
          } else w0 -= h.lr * (mult + h.reg0 * w0);
        }
        if constexpr (KIND == UPD_FTRL) w0 = -z0 * h.alpha_w / (h.beta_w + sqrt(n0));
        if constexpr (KIND == UPD_TDAP) w0 = -z0 / t_delta;
        if (lane == 0) { s_mult[e] = mult; s_uw[e] = uw; s_uv[e] = uv; }
      }
    }
    __syncthreads();
    FMX_T(tS);

    // ---------------------------------------------------------------- C: the example's update, from registers
    if (mine) {
      const double mult = s_mult[wave], euw = s_uw[wave], euv = s_uv[wave];
      if (tv && (k1 || KIND == UPD_FTRL)) {
        if constexpr (KIND == UPD_TDAP) {
          const double z = coord_seq<KIND>(h, true, true, myw, myx, mult, euw, stw);
          a.sw[mycol] = z;
        } else {
          coord_seq<KIND>(h, true, k1, myw, myx, mult, euw, stw);
          a.w[mycol] = myw;
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) if (k1) seq_state_ptr<KIND>(a, true, j)[mycol] = stw[j];
      }
#pragma unroll
      for (int j = 0; j < SL; ++j) {
        uint32_t cj;
        double xj;
        if constexpr (Q == 1) { cj = bcast(mycol, j); xj = bcast(myx, j); }
        else { cj = cu[j]; xj = xu[j]; }
        const size_t at = (size_t)cj * kp + ff;
        if (j * Q + fq < len && fv) {
          double th = vv[j];
          double st[NS > 0 ? NS : 1];
#pragma unroll
          for (int n = 0; n < NS; ++n) st[n] = stv[n][j];
          const double grad = s1 * xj - th * xj * xj;
          const double z = coord_seq<KIND>(h, false, true, th, grad, mult, euv, st);
          a.V[at] = th;
          if constexpr (KIND == UPD_TDAP) a.sV[at] = z;
#pragma unroll
          for (int n = 0; n < NS; ++n) seq_state_ptr<KIND>(a, false, n)[at] = st[n];
        }
      }
      if constexpr (KIND == UPD_TDAP) {  // w prox reads z_w by POSITION (TDAP_Learner.h:207); such readers/writers never share a group
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (tv) a.w[mycol] = tdap_prox(a.sw[lane], k1 ? stw[2] : a.t2w[mycol], h.l1w, h.l2w);
      }
    }

    FMX_T(tC);
    // ---------------------------------------------------------------- the next group: members, and whether it must wait
    // for stores still in flight (it must iff one of its candidates shares a feature with an example stored since the
    // last fence; otherwise its gathers overlap this group's stores)
    const int tn = gn + wave;
    const bool in_n = tn < wa.count;
    if (lane == 0) {
      s_cand[wave] = in_n && (wave == 0 || (nxt.conf_t < gn && nxt.conf_g != gn));
      s_need[wave] = in_n && nxt.conf_t >= unfenced;
    }
    __syncthreads();
    int need = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) need |= s_need[i];
    FMX_T(tV);
    if (need) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // own stores done ...
      __syncthreads();                                         // ... and everybody else's
      unfenced = gn;
#ifdef FMX_SEQ_TIMING
      ++nF;
#endif
    }
    FMX_T(tF);
    g = gn;
    cur = nxt;
  }

  if (threadIdx.x == 0) {
    a.scal[SC_W0] = w0; a.scal[SC_Z0] = z0; a.scal[SC_N0] = n0; a.scal[SC_UW] = uw; a.scal[SC_UV] = uv;
    a.scal[SC_T_NU] = t_nu; a.scal[SC_T_DELTA] = t_delta; a.scal[SC_T_H] = t_h;
#ifdef FMX_SEQ_TIMING
    printf("window kernel: %llu groups (%d examples), %llu fences; memtime ticks per group: A %.0f  S %.0f  C %.0f  vote %.0f  fence %.0f\n", nG, wa.count, nF,
           (double)tA / nG, (double)tS / nG, (double)tC / nG, (double)tV / nG, (double)tF / nG);
#endif
  }
#undef FMX_T
}
template <int KIND, int KL, int NZ>
__global__ __launch_bounds__((SeqWin<KIND, KL, NZ>::NW * 64)) void fm_seq_window_k(SeqArgs a, WinArgs wa, Hyper h) {
  seq_window_body<KIND, KL, NZ>(a, wa, h);
}
// the grid form (see fm_seq_pipe_grid_k below): workgroup b is the learner of model b -- the shapes the pipelined kernel does not take (TDAP, the reference's default
// solver; SGD-L1 and FTRL at k > 16)
template <int KIND, int KL, int NZ>
__global__ __launch_bounds__((SeqWin<KIND, KL, NZ>::NW * 64)) void fm_seq_window_grid_k(const SeqArgs* __restrict__ as, WinArgs wa, const Hyper* __restrict__ hs) {
  const SeqArgs a = as[blockIdx.x];
  const Hyper h = hs[blockIdx.x];
  seq_window_body<KIND, KL, NZ>(a, wa, h);
}

// ---- pipelined windowed learner ---------------------------------------------------------------------------------------
// fm_seq_window_k runs A, S and C of a group one after the other: while wave 0 walks the chain (S, about half of a group's time)
// the other waves wait, and while they gather (A, a third) the chain waits.  Here wave 0 does nothing but the chain, waves
// 1..NW-1 each own one example of a group, and the groups are software-pipelined: in step j
//     the chain wave runs        S(g_{j-1});
//     a worker wave runs         C(g_{j-2})  ->  A(g_j)  ->  its vote on joining g_{j+1}  ->  release fence;
// one barrier ends the step.  A worker therefore holds the parameters of TWO examples (g_{j-1}'s wait for their multiplier
// while g_j's are gathered): two register sets, used alternately, and every LDS buffer exists twice (step parity).
// What may be gathered early: A(g_{j+1}) runs in step j+1, when the stores of g_{j-1} (same step) and g_j (one step later)
// have not happened.  Candidate t joins g_{j+1} iff the candidates before it did and conf[t] < start(g_{j-1}): its last
// earlier conflict was stored in step j at the latest and fenced before that step's barrier (C comes first in a step, so
// the fence at the end finds those stores long complete).  If the FIRST candidate fails, g_{j+1} is empty -- a bubble; two
// steps later the pipeline has drained up to it and it passes (conf[t] < t).  The examples, their order, the arithmetic
// and its association are those of fm_seq_window_k; results are bitwise the one-wave kernel's (tests/test_gpu_seq_window.py).
// Not for TDAP (its w prox reads z_w by position, A-6: neighbours may not overlap at all).
template <int KIND, int KL, int NZ>
__device__ __forceinline__ void seq_pipe_body(const SeqArgs& a, const WinArgs& wa, const Hyper& h) {
  static_assert(KIND != UPD_TDAP, "TDAP keeps fm_seq_window_k");
  constexpr int NW = SeqWin<KIND, KL, NZ>::NW, W = NW - 1;
  constexpr int Q = SeqWin<KIND, KL, NZ>::Q;
  constexpr int SL = SeqWin<KIND, KL, NZ>::SL;
  constexpr int WIN_TERMS = SeqWin<KIND, KL, NZ>::TERMS;
  constexpr int NS = SeqState<KIND>::N;
  constexpr int NS1 = NS > 0 ? NS : 1, QN = Q > 1 ? SL : 1;
  __shared__ double terms[2][W][WIN_TERMS];
  __shared__ double s_mult[2][W], s_uw[2][W], s_uv[2][W];
  __shared__ float s_y[2][W];
  __shared__ int s_cand[2][W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ww = wave > 0 ? wave - 1 : 0;  // worker index (the chain wave computes with 0 and discards)
  const bool worker = wave > 0;
  const int k = a.k, kp = a.kp;
  const bool k0 = h.k0 != 0, k1 = h.k1 != 0;
  const int k16 = (k + 15) & ~15;
  double w0 = a.scal[SC_W0], z0 = a.scal[SC_Z0], n0 = a.scal[SC_N0], uw = a.scal[SC_UW], uv = a.scal[SC_UV];
  const int fq = lane / KL, ff = lane % KL;  // this lane's nonzero block and factor (V side)
  const bool fv = ff < k;
  const int fl = fv ? ff : 0;

  struct Meta { int conf_t, len; uint2 en; float y; };
  auto fetch = [&](int start) {  // unconditional loads on a clamped index: every use is guarded by its own range test
    Meta mt;
    const int last = wa.count - 1;
    const int tt = start + ww < last ? start + ww : last;
    mt.conf_t = wa.conf[tt];
    mt.len = wa.ex_len[tt];
    mt.en = wa.packed[(size_t)tt * NZ + (lane & (NZ - 1))];
    mt.y = wa.ex_y[tt];
    return mt;
  };
  auto leading = [&](const int* c) {
    int v[W], G = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = c[i];
    bool run = true;
#pragma unroll
    for (int i = 0; i < W; ++i) { run = run && v[i] != 0; G += run ? 1 : 0; }
    return G;
  };
  // a worker's example: everything C needs later
  struct Held {
    bool mine;
    int len;
    uint32_t mycol;
    double myx, myw, s1;
    double stw[NS1];
    double vv[SL];
    double stv[NS1][SL];
  };  // (the slots' columns and x values are re-read from the owning lanes in C: two shuffles instead of three registers per slot, held twice)

  // ------------------------------------------------------------------ A: gathers and the terms of y_hat (as fm_seq_window_k)
  auto phaseA = [&](Held& R, const Meta& mt, int par) {
    R.len = mt.len;
    const bool tv = lane < R.len;
    R.mycol = tv ? mt.en.x : 0u;
    const uint32_t myxf = tv ? mt.en.y : 0u;      // (the value travels through the shuffles as its float: one word instead of the double's two)
    R.myx = (double)__uint_as_float(myxf);
    double q1 = 0.0;
    double xu[QN];
    R.s1 = 0.0;
    R.myw = a.w[R.mycol];
#pragma unroll
    for (int j = 0; j < NS; ++j) R.stw[j] = seq_state_ptr<KIND>(a, true, j)[R.mycol];
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      uint32_t cj;
      if constexpr (Q == 1) cj = bcast(R.mycol, j);
      else { cj = (uint32_t)__shfl((int)R.mycol, j * Q + fq); xu[j] = (double)__uint_as_float((uint32_t)__shfl((int)myxf, j * Q + fq)); }
      const size_t at = (size_t)cj * kp + fl;
      R.vv[j] = a.V[at];
#pragma unroll
      for (int n = 0; n < NS; ++n) R.stv[n][j] = seq_state_ptr<KIND>(a, false, n)[at];
    }
#pragma unroll
    for (int u = 0; u < NZ; ++u) {  // core/Model.h:83-97
      double tmp;
      if constexpr (Q == 1) tmp = R.vv[u] * bcast(R.myx, u);
      else tmp = __shfl(R.vv[u / Q] * xu[u / Q], (u % Q) * KL + ff);
      R.s1 += tmp;
      q1 += tmp * tmp;
    }
    if (lane < NZ) terms[par][ww][lane] = (k1 ? R.myw : 0.0) * R.myx;
    if (lane < k16) terms[par][ww][NZ + lane] = fv ? 0.5 * (R.s1 * R.s1 - q1) : 0.0;  // core/Model.h:100
    if (lane == 0) s_y[par][ww] = mt.y;
  };

  // ------------------------------------------------------------------ C: the example's update, from registers
  auto phaseC = [&](Held& R, int par) {
    const double mult = s_mult[par][ww], euw = s_uw[par][ww], euv = s_uv[par][ww];
    const bool tv = lane < R.len;
    if (tv && (k1 || KIND == UPD_FTRL)) {
      coord_seq<KIND>(h, true, k1, R.myw, R.myx, mult, euw, R.stw);
      a.w[R.mycol] = R.myw;
#pragma unroll
      for (int j = 0; j < NS; ++j) if (k1) seq_state_ptr<KIND>(a, true, j)[R.mycol] = R.stw[j];
    }
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      uint32_t cj;
      double xj;
      if constexpr (Q == 1) { cj = bcast(R.mycol, j); xj = bcast(R.myx, j); }
      else { cj = (uint32_t)__shfl((int)R.mycol, j * Q + fq); xj = (double)__uint_as_float((uint32_t)__shfl((int)__float_as_uint((float)R.myx), j * Q + fq)); }
      const size_t at = (size_t)cj * kp + ff;
      if (j * Q + fq < R.len && fv) {
        double th = R.vv[j];
        double st[NS1];
#pragma unroll
        for (int n = 0; n < NS; ++n) st[n] = R.stv[n][j];
        const double grad = R.s1 * xj - th * xj * xj;
        coord_seq<KIND>(h, false, true, th, grad, mult, euv, st);
        a.V[at] = th;
#pragma unroll
        for (int n = 0; n < NS; ++n) seq_state_ptr<KIND>(a, false, n)[at] = st[n];
      }
    }
  };

  // ------------------------------------------------------------------ S: the scalar chain of one group, in order (as fm_seq_window_k)
  auto chain = [&](int par, int G) {
    double cb0[8], cb1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cb0[i] = terms[par][0][i];
    for (int e = 0; e < G; ++e) {
      if constexpr (KIND == UPD_SGD_L1) { uw += h.lr * h.regw; uv += h.lr * h.regv; }  // SGD_Learner.h:92-97
      const double* __restrict__ T = terms[par][e];
      double pred = k0 ? w0 : 0.0;
      auto pair = [&](const double* __restrict__ second, const double* __restrict__ after) {
#pragma unroll
        for (int i = 0; i < 8; ++i) cb1[i] = second[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) pred += cb0[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) cb0[i] = after[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) pred += cb1[i];
      };
      const double* __restrict__ nextT = terms[par][e + 1 < G ? e + 1 : e];
#pragma unroll
      for (int u = 0; u < NZ; u += 16) pair(T + u + 8, (u + 16 < NZ || k16 > 0) ? T + u + 16 : nextT);
      for (int f = 0; f < k16; f += 16) pair(T + NZ + f + 8, f + 16 < k16 ? T + NZ + f + 16 : nextT);
      const double mult = seq_grad_mult(h, pred, s_y[par][e]);
      if (k0) {
        if constexpr (KIND == UPD_FTRL) {
          const double n_old = n0;
          n0 += mult * mult;
          const double delta = (sqrt(n0) - sqrt(n_old)) / h.alpha_w;
          z0 += mult - delta * w0;
        } else w0 -= h.lr * (mult + h.reg0 * w0);
      }
      if constexpr (KIND == UPD_FTRL) w0 = -z0 * h.alpha_w / (h.beta_w + sqrt(n0));
      if (lane == 0) { s_mult[par][e] = mult; s_uw[par][e] = uw; s_uv[par][e] = uv; }
    }
  };

  // pipeline state, the same in every wave: starts of g_{j-1} and g_j, sizes of g_{j-2}, g_{j-1}, g_j
  int stS = 0, stA = 0, szC = 0, szS = 0, szA = 0;
  Meta cur = fetch(0);
  if (worker && lane == 0) s_cand[1][ww] = (ww < wa.count) && cur.conf_t < 0;  // g_0: examples without an earlier conflict
  __syncthreads();
  szA = leading(s_cand[1]);

  Held R0, R1;
  R0.mine = false; R1.mine = false;
  // one step; returns true when nothing is left in flight
  auto step = [&](Held& R, int par) {
    const int stN = stA + szA;  // where g_{j+1} starts: its metadata travels behind this step's work
    const Meta nxt = fetch(stN);
    if (!worker) {
      if (szS > 0) chain(par ^ 1, szS);
    } else {
      if (R.mine) phaseC(R, par);   // g_{j-2}: this set, this parity
      R.mine = ww < szA;
      if (R.mine) phaseA(R, cur, par);
      if (lane == 0) s_cand[par][ww] = (stN + ww < wa.count) && nxt.conf_t < stS;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this step's stores (issued first) before the barrier
    }
    __syncthreads();
    const int szN = leading(s_cand[par]);
    szC = szS; stS = stA; szS = szA; stA = stN; szA = szN; cur = nxt;
    return stA >= wa.count && szA == 0 && szS == 0 && szC == 0;
  };
  for (;;) {
    if (step(R0, 0)) break;
    if (step(R1, 1)) break;
  }

  if (threadIdx.x == 0) {
    a.scal[SC_W0] = w0; a.scal[SC_Z0] = z0; a.scal[SC_N0] = n0; a.scal[SC_UW] = uw; a.scal[SC_UV] = uv;
  }
}
template <int KIND, int KL, int NZ>
__global__ __launch_bounds__((SeqWin<KIND, KL, NZ>::NW * 64)) void fm_seq_pipe_k(SeqArgs a, WinArgs wa, Hyper h) {
  seq_pipe_body<KIND, KL, NZ>(a, wa, h);
}
// A GRID of models on ONE visiting order of ONE matrix (a hyper-parameter grid, the folds of fm.select's repeated fm.train calls): workgroup b is the learner of
// model b -- its own parameter tables, optimizer state and hyper-parameters, the examples, their packing and their conflict plan shared.  Every model takes
// exactly the steps its own single-model launch takes (tests: bit for bit); the chip runs as many reference-order learners as it has CUs.
template <int KIND, int KL, int NZ>
__global__ __launch_bounds__((SeqWin<KIND, KL, NZ>::NW * 64)) void fm_seq_pipe_grid_k(const SeqArgs* __restrict__ as, WinArgs wa, const Hyper* __restrict__ hs) {
  const SeqArgs a = as[blockIdx.x];
  const Hyper h = hs[blockIdx.x];
  seq_pipe_body<KIND, KL, NZ>(a, wa, h);
}

// ---- reassociated reference-order learner (cfg.seq_reassociate) -------------------------------------------------------
// The reference's algorithm, its visiting order, one update per example, fp64 -- but the FORWARD's sum is formed as the reference's
// formula read as mathematics, y_hat = w0 + (sum_j w_j x_j + sum_f 0.5 (s_f^2 - q_f)), instead of the 1 + nnz + k ordered additions
// that start at w0 (core/Model.h:77-100).  In that association only w0 carries a recurrence from one example to the next
// (solver/SGD_Learner.h:100-112): the row part r is a fixed tree over the example's own lanes, and the chain is
// `pred = w0 + r; mult; w0 step` -- three dependent operations instead of fifty.  Results differ from the bitwise kernels in the last bits
// of y_hat (and of s_f: its nnz products are summed per lane block, then across the blocks); they are the same from run to run (every
// tree is fixed, and an example never reads a parameter before the earlier examples that touch it have stored theirs).
//
// One workgroup: wave 0 is the chain, waves 1..W own the examples t = w - 1, w - 1 + W, ... of the launch.  No barrier after the start:
// the waves meet through tagged LDS slots (ring of R = a multiple of W, so a slot is always written by the same wave, in order):
//   f_r[t % R] = t + 1     worker -> chain: r (and the label) of example t are in the slot;
//   f_m[t % R] = t + 1     chain -> worker: the multiplier (and the cumulative L1 penalties) of example t are in the slot;
//   f_done[t % R] = t + 1  worker -> workers: example t's stores are complete (released at workgroup scope).
// Example t may gather once every example <= conf[t] (the last earlier one sharing a feature with it, seq_conf_k) is done: the examples
// between finish in any order, so the test covers all of them -- lane s looks at slot s, whose latest tag must have reached the largest
// example <= conf[t] of its residue class (one LDS read per lane, one ballot).  Progress: the chain takes the examples in order; example
// t waits only for examples before it (its conflicts, its owner's previous example), so by induction every wait ends.
// The gradient multiplier of the reassociated chain.  Regression: as seq_grad_mult.  Classification: the reference's -y (1 - 1 / (1 + exp(-y y_hat)))
// (solver/SGD_Learner.h:187-190) evaluated as -y / (1 + exp(a)), a = y y_hat clamped to [-750, 700] (1 + e^700 is finite; beyond the clamps the value is 1 or
// below 1e-304) -- no cancellation (2 ulp of the exact value; within 2.3e-16 ABSOLUTE of the reference's expression, whose own rounding error is of that size).  It is on the one dependent chain of the mode, so its depth is what counts: the exponential's polynomial (the device
// library's degree-11 minimax coefficients) is summed in Estrin's order (depth 4 instead of 11), the reciprocal of 1 + t in (1, 2] is v_rcp_f64 + two Newton steps
// instead of the IEEE division's eleven dependent operations.  NaN stays NaN.
__device__ __forceinline__ double seq_grad_mult_re(const Hyper& h, double y_hat, float y) {
  if (h.task == FMX_TASK_REGRESSION) return seq_grad_mult(h, y_hat, y);
  auto B = [](unsigned long long u) { return __longlong_as_double((long long)u); };
  const double yd = (double)y;
  const double a = yd * y_hat;
  const double x = fmin(fmax(a, -750.0), 700.0);
  const double n = __builtin_rint(x * B(0x3ff71547652b82feull));                       // x / ln 2, to nearest
  double r = __builtin_fma(n, B(0xbfe62e42fefa39efull), x);                            // x - n ln2 (hi, lo)
  r = __builtin_fma(n, B(0xbc7abc9e3b39803full), r);
  const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
  const double p01 = 1.0 + r;
  const double p23 = __builtin_fma(B(0x3fc5555555555511ull), r, B(0x3fe000000000000bull));
  const double p45 = __builtin_fma(B(0x3f81111111122322ull), r, B(0x3fa55555555502a1ull));
  const double p67 = __builtin_fma(B(0x3f2a01a014761f6eull), r, B(0x3f56c16c1852b7b0ull));
  const double p89 = __builtin_fma(B(0x3ec71dee623fde64ull), r, B(0x3efa01997c89e6b0ull));
  const double pab = __builtin_fma(B(0x3e5ade156a5dcb37ull), r, B(0x3e928af3fca7ab0cull));
  const double q0 = __builtin_fma(p23, r2, p01), q1 = __builtin_fma(p67, r2, p45), q2 = __builtin_fma(pab, r2, p89);
  const double t = ldexp(__builtin_fma(q2, r8, __builtin_fma(q1, r4, q0)), (int)n);
  const double d = 1.0 + t;
  double xr = __builtin_amdgcn_rcp(d);
  xr = __builtin_fma(xr, __builtin_fma(-d, xr, 1.0), xr);
  xr = __builtin_fma(xr, __builtin_fma(-d, xr, 1.0), xr);
  return -yd * (a != a ? a : xr);
}

template <int KIND, int KL, int NZ> struct SeqRe {
  static constexpr int Q = 64 / KL, SL = NZ / Q, NS = SeqState<KIND>::N;
  // a worker holds ONE example: SL x (1 + state) doubles of V-side parameters + the slots' columns and values; 16 waves leave 128 VGPRs per lane
  static constexpr int REGS = 2 * SL * (1 + NS) + 3 * SL + 56;
#ifdef FMX_SEQ_RE_NW
  static constexpr int NW = FMX_SEQ_RE_NW;
#else
  // measured (profiles/r06_seq_reassoc.txt): 16 waves where a worker fits 128 VGPRs (SGD-L2, k <= 16, rows of <= 32 entries), 12 (168 VGPRs: three waves per SIMD) for the
  // next size up (SGD-L1 at k <= 16; SGD-L2 at k <= 32 or rows of 33..64 entries: +8..11 % over 8 waves, and over 16 with spills), 8 beyond
  static constexpr int NW = REGS <= 100 ? 16 : (REGS <= 140 ? 12 : 8);
#endif
  static constexpr int W = NW - 1;
  static constexpr int R = (64 / W) * W;   // ring slots, a multiple of W (a slot is always written by the same wave, in order): 60 (15 workers), 55 (11), 63 (7)
};

template <int KIND, int KL, int NZ>
__global__ __launch_bounds__((SeqRe<KIND, KL, NZ>::NW * 64)) void fm_seq_reassoc_k(SeqArgs a, WinArgs wa, Hyper h) {
  // SGD only.  FTRL's chain carries square roots and divisions of its own and its workers' prox arithmetic is what bounds it: with the flag it ran 650 K examples/s
  // against the pipelined kernel's 716 K at k = 16 (profiles/r06_seq_reassoc.txt), so FTRL -- and TDAP, whose w prox reads z_w by position (A-6) -- ignore the flag.
  static_assert(KIND == UPD_SGD_L2 || KIND == UPD_SGD_L1, "the reassociated learner is built for the SGD kinds");
  using S = SeqRe<KIND, KL, NZ>;
  constexpr int W = S::W, R = S::R, Q = S::Q, SL = S::SL, NS = S::NS, NS1 = NS > 0 ? NS : 1;
  __shared__ double s_r[R], s_mult[R], s_uw[R], s_uv[R];
  __shared__ float s_y[R];
  __shared__ int f_r[R], f_m[R], f_done[R];
  __shared__ int s_abort;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = a.k, kp = a.kp, count = wa.count;
  const bool k0 = h.k0 != 0, k1 = h.k1 != 0;
  for (int i = threadIdx.x; i < R; i += blockDim.x) { f_r[i] = 0; f_m[i] = 0; f_done[i] = 0; }
  if (threadIdx.x == 0) s_abort = 0;
  __syncthreads();   // the only barrier: every wave reaches it before the roles part
  // Every wait below is bounded: a wave that has polled ~1e6 times (tens of milliseconds; a legitimate wait is microseconds) raises s_abort, every other wait
  // sees it within 1024 polls, all waves leave, w0 comes back NaN and scal[SC_SEQ_ABORT] is set (fmx_get_params / fmx_sync then return FMX_ERR_HIP) -- a wrong plan
  // (conf[]) or a lost tag fails loudly instead of hanging the device (tests/test_gpu_seq_reassoc.py injects one).
  int spins = 0;
  auto stuck = [&]() {
    if ((++spins & 1023) != 0) return false;
    if (spins > (1 << 20)) __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
  };

  if (wave == 0) {
    // ---------------------------------------------------------------- the chain: examples in order, as many as are ready at once
    double w0 = a.scal[SC_W0], uw = a.scal[SC_UW], uv = a.scal[SC_UV];
    constexpr int B = 16;   // examples looked at per poll
    int e = 0;
    __builtin_amdgcn_s_setprio(3);   // the chain is the critical path of the workgroup: its instructions go first on the SIMD it shares with three workers
#ifdef FMX_SEQ_TIMING
    unsigned long long tIdle = 0, tBusy = 0, nBatch = 0, c0 = __builtin_amdgcn_s_memtime(), c1;
#define FMX_TC(acc) do { c1 = __builtin_amdgcn_s_memtime(); acc += c1 - c0; c0 = c1; } while (0)
#else
#define FMX_TC(acc) do { } while (0)
#endif
    while (e < count) {
      const int te = e + lane;
      const int slot = te % R;
      const bool look = lane < B && te < count;
      const bool ready = look && __hip_atomic_load(&f_r[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == te + 1;
      const unsigned long long mask = __ballot(ready);
      const int n = __builtin_ctzll(~mask);   // the leading ready ones (mask has at most B bits set)
      if (n == 0) {
        if (stuck()) { if (lane == 0) { a.scal[SC_W0] = __longlong_as_double(0x7FF8000000000000ll); a.scal[SC_SEQ_ABORT] = 1.0; } return; }
        __builtin_amdgcn_s_sleep(1);
        FMX_TC(tIdle);
        continue;
      }
      spins = 0;
      const double r = lane < n ? s_r[slot] : 0.0;
      const float y = lane < n ? s_y[slot] : 0.0f;
      auto batch = [&](auto K0) {
      constexpr bool k0 = decltype(K0)::value;   // (a uniform branch outside the loop instead of two selects on w0 inside the chain)
      for (int i = 0; i < n; ++i) {
        if constexpr (KIND == UPD_SGD_L1) { uw += h.lr * h.regw; uv += h.lr * h.regv; }  // SGD_Learner.h:92-97
        const double pred = (k0 ? w0 : 0.0) + bcast(r, i);
        const double mult = seq_grad_mult_re(h, pred, bcast(y, i));
        if (k0) w0 -= h.lr * (mult + h.reg0 * w0);   // SGD_Learner.h:100-112
        // handed over at once (not at the end of the batch: the owners of a batch would otherwise get their multipliers together, post their next sums together,
        // and the chain would sit idle in between -- measured: 43 % of its time)
        // The tag is a RELAXED store behind a compiler fence: a wave's LDS operations execute in program order, so the slot's data is in place when the tag is --
        // a release store would make the chain wait (s_waitcnt lgkmcnt(0)) for its own write before every tag.
        if (lane == i) {
          s_mult[slot] = mult;
          if constexpr (KIND == UPD_SGD_L1) { s_uw[slot] = uw; s_uv[slot] = uv; }
          __atomic_signal_fence(__ATOMIC_SEQ_CST);
          __hip_atomic_store(&f_m[slot], te + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      };
      if (k0) batch(std::true_type{}); else batch(std::false_type{});
      e += n;
#ifdef FMX_SEQ_TIMING
      ++nBatch;
#endif
      FMX_TC(tBusy);
    }
#ifdef FMX_SEQ_TIMING
    if (lane == 0) printf("reassoc chain: %d examples in %llu batches (%.2f per batch); memtime ticks per example: idle %.1f  busy %.1f\n", count, nBatch, (double)count / nBatch,
                          (double)tIdle / count, (double)tBusy / count);
#endif
#undef FMX_TC
    if (lane == 0) { a.scal[SC_W0] = w0; a.scal[SC_UW] = uw; a.scal[SC_UV] = uv; }
    return;
  }

  // ------------------------------------------------------------------ a worker: examples ww, ww + W, ...
  const int ww = wave - 1;
  const int fq = lane / KL, ff = lane % KL;  // this lane's nonzero block and factor (V side)
  const bool fv = ff < k;
  const int fl = fv ? ff : 0;
  struct Meta { int conf_t, len; uint2 en; float y; };
  auto fetch = [&](int t) {  // unconditional loads on a clamped index: every use is guarded by its own range test
    Meta mt;
    const int tt = t < count - 1 ? t : count - 1;
    mt.conf_t = wa.conf[tt];
    mt.len = wa.ex_len[tt];
    mt.en = wa.packed[(size_t)tt * NZ + (lane & (NZ - 1))];
    mt.y = wa.ex_y[tt];
    return mt;
  };
  Meta cur = fetch(ww);
#ifdef FMX_SEQ_TIMING
  unsigned long long tW = 0, tA = 0, tM = 0, tC = 0, tD = 0, nE = 0, c0 = __builtin_amdgcn_s_memtime(), c1;
#define FMX_TW(acc) do { c1 = __builtin_amdgcn_s_memtime(); acc += c1 - c0; c0 = c1; } while (0)
#else
#define FMX_TW(acc) do { } while (0)
#endif
  for (int t = ww; t < count; t += W) {
    const int slot = t % R;
    const Meta nxt = fetch(t + W);   // travels behind this example's work
    // every example <= conf[t] must have stored its parameters
    const int c = cur.conf_t;
    if (c >= 0) {
      const int back = ((c - lane) % R + R) % R;          // lane = slot: the largest example <= c of this residue class is c - back
      const int want = lane < R && c - back >= 0 ? c - back + 1 : 0;
      for (;;) {
        const int got = lane < R ? __hip_atomic_load(&f_done[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
        if (__ballot(got >= want) == ~0ull) break;
        if (stuck()) return;
        __builtin_amdgcn_s_sleep(1);
      }
      spins = 0;
    }
    FMX_TW(tW);
    // ---------------------------------------------------------------- A: gathers, the factor sums, the row part of y_hat
    const int len = cur.len;
    const bool tv = lane < len;
    const uint32_t mycol = tv ? cur.en.x : 0u;
    const uint32_t myxf = tv ? cur.en.y : 0u;          // (the value's float bits: one word through the shuffle below instead of the double's two)
    const double myx = (double)__uint_as_float(myxf);
    double myw = a.w[mycol];
    double stw[NS1];
#pragma unroll
    for (int j = 0; j < NS; ++j) stw[j] = seq_state_ptr<KIND>(a, true, j)[mycol];
    uint32_t cu[SL];
    double xu[SL], vv[SL], stv[NS1][SL];
#pragma unroll
    for (int j = 0; j < SL; ++j) {  // slot j of this lane is nonzero u = j * Q + fq (idle slots: column 0, x = 0)
      if constexpr (Q == 1) { cu[j] = bcast(mycol, j); xu[j] = bcast(myx, j); }
      else { cu[j] = (uint32_t)__shfl((int)mycol, j * Q + fq); xu[j] = (double)__uint_as_float((uint32_t)__shfl((int)myxf, j * Q + fq)); }
      const size_t at = (size_t)cu[j] * kp + fl;
      vv[j] = a.V[at];
#pragma unroll
      for (int n = 0; n < NS; ++n) stv[n][j] = seq_state_ptr<KIND>(a, false, n)[at];
    }
    double s1 = 0.0, q1 = 0.0;
#pragma unroll
    for (int j = 0; j < SL; ++j) {  // this block's nonzeros, in slot order
      const double tmp = vv[j] * xu[j];
      s1 += tmp;
      q1 += tmp * tmp;
    }
    // across the Q blocks (a + b == b + a: the same bits in every block); the steps of x += __shfl_xor(x, o), o = KL .. 32, without the LDS
    if constexpr (KL <= 16) { s1 = xor16_add(s1); q1 = xor16_add(q1); }
    if constexpr (KL <= 32) { s1 = xor32_add(s1); q1 = xor32_add(q1); }
    double part = (lane < NZ ? (k1 ? myw : 0.0) * myx : 0.0) + ((lane < KL && fv) ? 0.5 * (s1 * s1 - q1) : 0.0);
    part = seq_butterfly_allsum(part);
    if (lane == 0) {
      s_r[slot] = part; s_y[slot] = cur.y;
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      __hip_atomic_store(&f_r[slot], t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (LDS order, as for f_m)
    }
    FMX_TW(tA);
    // ---------------------------------------------------------------- the multiplier
    while (__hip_atomic_load(&f_m[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t + 1 || t + 1 == wa.debug_lose) {
      if (stuck()) return;
      __builtin_amdgcn_s_sleep(1);
    }
    spins = 0;
    const double mult = s_mult[slot], euw = KIND == UPD_SGD_L1 ? s_uw[slot] : 0.0, euv = KIND == UPD_SGD_L1 ? s_uv[slot] : 0.0;
    FMX_TW(tM);
    // ---------------------------------------------------------------- C: the example's update, from registers
    if (tv && k1) {
      coord_seq<KIND>(h, true, true, myw, myx, mult, euw, stw);
      a.w[mycol] = myw;
#pragma unroll
      for (int j = 0; j < NS; ++j) seq_state_ptr<KIND>(a, true, j)[mycol] = stw[j];
    }
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const size_t at = (size_t)cu[j] * kp + ff;
      if (j * Q + fq < len && fv) {
        double th = vv[j];
        double st[NS1];
#pragma unroll
        for (int n = 0; n < NS; ++n) st[n] = stv[n][j];
        const double grad = s1 * xu[j] - th * xu[j] * xu[j];
        coord_seq<KIND>(h, false, true, th, grad, mult, euv, st);
        a.V[at] = th;
#pragma unroll
        for (int n = 0; n < NS; ++n) seq_state_ptr<KIND>(a, false, n)[at] = st[n];
      }
    }
    FMX_TW(tC);
    // the stores complete (release), then the tag: whoever waits for this example gathers after it
    if (lane == 0) __hip_atomic_store(&f_done[slot], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    cur = nxt;
#ifdef FMX_SEQ_TIMING
    ++nE;
#endif
    FMX_TW(tD);
  }
#ifdef FMX_SEQ_TIMING
  if (ww == 0 && lane == 0 && nE) printf("reassoc worker 0: %llu examples; memtime ticks per example: conflict wait %.0f  A %.0f  mult wait %.0f  C %.0f  release %.0f\n", nE,
                                         (double)tW / nE, (double)tA / nE, (double)tM / nE, (double)tC / nE, (double)tD / nE);
#endif
#undef FMX_TW
}

// the windowed learner applies when every row is a fast row; FMX_SEQ_WINDOW=0 in the environment keeps the one-wave kernel
// entries per packed row (32 or 64), or 0: the one-wave kernel
static int window_mode(const fmx_engine* e, const fmx_matrix* m) {
  const char* s = getenv("FMX_SEQ_WINDOW");  // read per call: the tests compare both kernels in one process
  const bool off = s && s[0] == '0';
  if (off || !m->rows_sorted || e->k > 64) return 0;
  if (m->max_row_len <= 32) return 32;
  if (m->max_row_len <= 64 && e->k <= 32) return 64;
  return 0;
}

static int ensure_window_workspace(fmx_engine* e, int64_t cap) {
  if (cap <= e->seq_wcap) return FMX_OK;
  FMX_HIP(hipStreamSynchronize(e->stream));
  (void)hipFree(e->seq_packed); (void)hipFree(e->seq_conf); (void)hipFree(e->seq_keys); (void)hipFree(e->seq_sort_tmp);
  e->seq_packed = nullptr; e->seq_conf = nullptr; e->seq_keys = nullptr; e->seq_sort_tmp = nullptr; e->seq_wcap = 0;
  const size_t pairs = (size_t)cap * WIN_NZ_MAX;
  FMX_HIP(hipMalloc(&e->seq_packed, pairs * sizeof(uint2)));
  FMX_HIP(hipMalloc(&e->seq_conf, (size_t)cap * sizeof(int)));
  FMX_HIP(hipMalloc(&e->seq_keys, 4 * pairs * sizeof(uint32_t)));
  size_t tb = 0;
  uint32_t* k = e->seq_keys;
  FMX_HIP(rocprim::radix_sort_pairs(nullptr, tb, k, k + pairs, k + 2 * pairs, k + 3 * pairs, pairs, 0, 32, e->stream));
  FMX_HIP(hipMalloc(&e->seq_sort_tmp, tb ? tb : 16));
  e->seq_sort_tmp_bytes = tb;
  e->seq_wcap = cap;
  return FMX_OK;
}

// Which shapes take the pipelined kernel: measured (profiles/r02_seq_window.txt).  Two register sets fit 256 VGPRs without scratch for
// SGD-L2 at k <= 32 and SGD-L1 at k <= 16 with rows of <= 32 entries; the other shapes spill (llvm's .private_seg_size: 32 B to
// 768 B per lane) and still win while the spill is small next to what the overlap hides: SGD-L2 everywhere (1.2-1.6x),
// SGD-L1 and FTRL at k <= 16 (1.3-1.5x); at k > 16 those two lose or tie and keep fm_seq_window_k.
// FMX_SEQ_WINDOW=1 keeps the unpipelined windowed kernel everywhere, =2 pipelines every non-TDAP shape (tests compare all forms,
// profiles/seq_window_bench.py measures them); read per call.
template <int KIND, int KL, int NZ> struct PipeFits {
  static constexpr bool value = KIND == UPD_SGD_L2 || ((KIND == UPD_SGD_L1 || KIND == UPD_FTRL) && KL == 16);
};
static int pipe_mode() {
  const char* s = getenv("FMX_SEQ_WINDOW");
  return !s ? 1 : (s[0] == '1' ? 0 : (s[0] == '2' ? 2 : 1));  // 0 never, 1 where it fits, 2 always
}

static std::atomic<int> g_lose_next_seq_multiplier{0};
void debug_lose_next_seq_multiplier() { g_lose_next_seq_multiplier.store(1); }

// cfg.seq_reassociate (FMX_SEQ_REASSOC=0/1 in the environment overrides it, read per call: the tests run every case in both forms)
static bool reassoc_mode(const fmx_engine* e) {
  const char* s = getenv("FMX_SEQ_REASSOC");
  return s && s[0] ? s[0] != '0' : e->cfg.seq_reassociate != 0;
}

template <int KIND>
static void launch_window_kind(fmx_engine* e, const SeqArgs& a, const WinArgs& wa, int nz) {
  if constexpr (KIND == UPD_SGD_L2 || KIND == UPD_SGD_L1) {
    if (reassoc_mode(e)) {
      WinArgs war = wa;
      if (wa.count > 8 && g_lose_next_seq_multiplier.exchange(0) > 0) war.debug_lose = 8;   // (test hook: example 7's multiplier is never seen)
#define FMX_RE(KL, NZ) hipLaunchKernelGGL((fm_seq_reassoc_k<KIND, KL, NZ>), dim3(1), dim3(SeqRe<KIND, KL, NZ>::NW * 64), 0, e->stream, a, war, e->hyper)
      if (e->k <= 16) { if (nz == 32) FMX_RE(16, 32); else FMX_RE(16, 64); }
      else if (e->k <= 32) { if (nz == 32) FMX_RE(32, 32); else FMX_RE(32, 64); }
      else FMX_RE(64, 32);
#undef FMX_RE
      return;
    }
  }
  if constexpr (KIND != UPD_TDAP) {
    const int pm = pipe_mode();
#define FMX_PIPE(KL, NZ)                                                                                                                            \
  if (pm == 2 || (pm == 1 && PipeFits<KIND, KL, NZ>::value)) {                                                                                      \
    hipLaunchKernelGGL((fm_seq_pipe_k<KIND, KL, NZ>), dim3(1), dim3(SeqWin<KIND, KL, NZ>::NW * 64), 0, e->stream, a, wa, e->hyper);                   \
    return;                                                                                                                                         \
  }
    if (e->k <= 16) { if (nz == 32) { FMX_PIPE(16, 32) } else { FMX_PIPE(16, 64) } }
    else if (e->k <= 32) { if (nz == 32) { FMX_PIPE(32, 32) } else { FMX_PIPE(32, 64) } }
    else { FMX_PIPE(64, 32) }
#undef FMX_PIPE
  }
#define FMX_WIN(KL, NZ) hipLaunchKernelGGL((fm_seq_window_k<KIND, KL, NZ>), dim3(1), dim3(SeqWin<KIND, KL, NZ>::NW * 64), 0, e->stream, a, wa, e->hyper)
  if (e->k <= 16) { if (nz == 32) FMX_WIN(16, 32); else FMX_WIN(16, 64); }
  else if (e->k <= 32) { if (nz == 32) FMX_WIN(32, 32); else FMX_WIN(32, 64); }
  else FMX_WIN(64, 32);
#undef FMX_WIN
}

int launch_seq_learn(fmx_engine* e, const fmx_matrix* m, const int64_t* d_order, int64_t count) {
  FMX_CHECK(e->k <= 64 * FI, FMX_ERR_INVALID, "sequential mode supports factor.number <= %d", 64 * FI);
  if (count <= 0) return FMX_OK;
  e->als_q_invalidate();   // (this learner writes the fp64 V table)
  if (count > e->seq_cap) {  // grow-only workspace; stream order makes reuse across calls safe
    FMX_HIP(hipStreamSynchronize(e->stream));
    (void)hipFree(e->seq_b); (void)hipFree(e->seq_len); (void)hipFree(e->seq_y);
    e->seq_b = nullptr; e->seq_len = nullptr; e->seq_y = nullptr; e->seq_cap = 0;
    FMX_HIP(hipMalloc(&e->seq_b, (size_t)count * sizeof(int64_t)));
    FMX_HIP(hipMalloc(&e->seq_len, (size_t)count * sizeof(int)));
    FMX_HIP(hipMalloc(&e->seq_y, (size_t)count * sizeof(float)));
    e->seq_cap = count;
  }
  int64_t* ex_b = e->seq_b;
  int* ex_len = e->seq_len;
  float* ex_y = e->seq_y;
  hipLaunchKernelGGL(seq_prepare_k, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, e->stream, d_order, count, m->row_ptr, m->y, ex_b, ex_len, ex_y);
  SeqArgs a{m->row_ptr, m->col, m->val, m->y, d_order, 0, ex_b, ex_len, ex_y, e->dV, e->dw, e->dsV, e->dsw, e->dnV, e->dnw,
            e->dt1V, e->dt1w, e->dt2V, e->dt2w, e->dt3V, e->dt3w, e->scal,
            e->k, e->kp64, m->rows_sorted};
  // bounded launches: a single workgroup walking millions of examples in one dispatch would run for seconds
  const int64_t CHUNK = 1 << 16;
  const int nz = window_mode(e, m);
  const bool windowed = nz > 0;
  if (windowed) FMX_TRY(ensure_window_workspace(e, count < CHUNK ? count : CHUNK));
  for (int64_t off = 0; off < count; off += CHUNK) {
    a.order = d_order + off;
    a.ex_b = ex_b + off; a.ex_len = ex_len + off; a.ex_y = ex_y + off;
    a.count = (count - off < CHUNK) ? count - off : CHUNK;
    if (windowed) {
      // the chunk's examples packed in visiting order, and for each the last earlier example of the chunk sharing a feature
      const int cnt = (int)a.count;
      const size_t pairs = (size_t)cnt * nz;
      uint32_t* keys = e->seq_keys;
      const size_t cap_pairs = (size_t)e->seq_wcap * WIN_NZ_MAX;
      FMX_HIP(hipMemsetAsync(e->seq_conf, 0xFF, (size_t)cnt * sizeof(int), e->stream));  // -1
      hipLaunchKernelGGL(seq_pack_k, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, e->stream, a.ex_b, a.ex_len, cnt, nz, m->col, m->val,
                         e->hyper.kind == UPD_TDAP ? 1 : 0, (uint2*)e->seq_packed, keys, keys + 2 * cap_pairs, e->seq_conf);
      size_t tb = e->seq_sort_tmp_bytes;
      FMX_HIP(rocprim::radix_sort_pairs(e->seq_sort_tmp, tb, keys, keys + cap_pairs, keys + 2 * cap_pairs, keys + 3 * cap_pairs, pairs, 0, 32, e->stream));
      hipLaunchKernelGGL(seq_conf_k, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, e->stream, keys + cap_pairs, keys + 3 * cap_pairs, (int)pairs, e->seq_conf);
      WinArgs wa{(const uint2*)e->seq_packed, a.ex_len, a.ex_y, e->seq_conf, cnt};
      prof_begin(e, FMX_KERNEL_SEQ);
      switch (e->hyper.kind) {
        case UPD_SGD_L2: launch_window_kind<UPD_SGD_L2>(e, a, wa, nz); break;
        case UPD_SGD_L1: launch_window_kind<UPD_SGD_L1>(e, a, wa, nz); break;
        case UPD_TDAP: launch_window_kind<UPD_TDAP>(e, a, wa, nz); break;
        default: launch_window_kind<UPD_FTRL>(e, a, wa, nz); break;
      }
      prof_end(e);
      FMX_HIP(hipGetLastError());
      continue;
    }
    prof_begin(e, FMX_KERNEL_SEQ);
    switch (e->hyper.kind) {
      case UPD_SGD_L2: hipLaunchKernelGGL(fm_seq_learn_k<UPD_SGD_L2>, dim3(1), dim3(64), 0, e->stream, a, e->hyper); break;
      case UPD_SGD_L1: hipLaunchKernelGGL(fm_seq_learn_k<UPD_SGD_L1>, dim3(1), dim3(64), 0, e->stream, a, e->hyper); break;
      case UPD_TDAP: hipLaunchKernelGGL(fm_seq_learn_k<UPD_TDAP>, dim3(1), dim3(64), 0, e->stream, a, e->hyper); break;
      default: hipLaunchKernelGGL(fm_seq_learn_k<UPD_FTRL>, dim3(1), dim3(64), 0, e->stream, a, e->hyper); break;
    }
    prof_end(e);
    FMX_HIP(hipGetLastError());
  }
  return FMX_OK;
}

// n models (engines of one shape: p, k, update kind) trained on the SAME examples in the SAME order, one launch per chunk with one workgroup per model.
// The examples' metadata, packing and conflict plan are made once, in es[0]'s workspace and on es[0]'s stream; the other engines' streams are idle for the
// duration (the caller waits).  Every shape the windowed learners take (fast rows): the pipelined kernel where the single-model launcher picks it, the windowed one elsewhere.
int launch_seq_learn_grid(fmx_engine* const* es, int n, const fmx_matrix* m, const int64_t* d_order, int64_t count) {
  fmx_engine* e = es[0];
  if (count <= 0) return FMX_OK;
  for (int b = 0; b < n; ++b) es[b]->als_q_invalidate();
  const int nz = window_mode(e, m);
  FMX_CHECK(nz > 0, FMX_ERR_INVALID, "grid training needs rows of at most 64 (k <= 32) / 32 entries with ascending columns");
  const int kl = e->k <= 16 ? 16 : (e->k <= 32 ? 32 : 64);
  FMX_CHECK(!(kl == 64 && nz == 64), FMX_ERR_INVALID, "grid training: rows of more than 32 entries need k <= 32");
  if (count > e->seq_cap) {
    FMX_HIP(hipStreamSynchronize(e->stream));
    (void)hipFree(e->seq_b); (void)hipFree(e->seq_len); (void)hipFree(e->seq_y);
    e->seq_b = nullptr; e->seq_len = nullptr; e->seq_y = nullptr; e->seq_cap = 0;
    FMX_HIP(hipMalloc(&e->seq_b, (size_t)count * sizeof(int64_t)));
    FMX_HIP(hipMalloc(&e->seq_len, (size_t)count * sizeof(int)));
    FMX_HIP(hipMalloc(&e->seq_y, (size_t)count * sizeof(float)));
    e->seq_cap = count;
  }
  hipLaunchKernelGGL(seq_prepare_k, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, e->stream, d_order, count, m->row_ptr, m->y, e->seq_b, e->seq_len, e->seq_y);
  const int64_t CHUNK = 1 << 16;
  FMX_TRY(ensure_window_workspace(e, count < CHUNK ? count : CHUNK));
  // the models' argument records: one array per chunk (the examples' pointers move with the chunk), all uploaded before the first launch -- nothing waits between chunks
  const int64_t n_chunks = (count + CHUNK - 1) / CHUNK;
  std::vector<SeqArgs> h_as((size_t)n * (size_t)n_chunks);
  std::vector<Hyper> h_hs((size_t)n);
  SeqArgs* d_as = nullptr; Hyper* d_hs = nullptr;
  FMX_HIP(hipMalloc(&d_as, h_as.size() * sizeof(SeqArgs)));
  if (hipMalloc(&d_hs, (size_t)n * sizeof(Hyper)) != hipSuccess) { (void)hipFree(d_as); set_error("out of device memory"); return FMX_ERR_HIP; }
  for (int64_t c = 0; c < n_chunks; ++c) {
    const int64_t off = c * CHUNK;
    const int64_t cnt = (count - off < CHUNK) ? count - off : CHUNK;
    for (int b = 0; b < n; ++b) {
      fmx_engine* g = es[b];
      h_as[(size_t)c * n + b] = SeqArgs{m->row_ptr, m->col, m->val, m->y, d_order + off, cnt, e->seq_b + off, e->seq_len + off, e->seq_y + off, g->dV, g->dw, g->dsV, g->dsw, g->dnV, g->dnw,
                                        g->dt1V, g->dt1w, g->dt2V, g->dt2w, g->dt3V, g->dt3w, g->scal, g->k, g->kp64, m->rows_sorted};
    }
  }
  for (int b = 0; b < n; ++b) h_hs[(size_t)b] = es[b]->hyper;
  int st = FMX_OK;
  if (hipMemcpy(d_as, h_as.data(), h_as.size() * sizeof(SeqArgs), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_hs, h_hs.data(), (size_t)n * sizeof(Hyper), hipMemcpyHostToDevice) != hipSuccess) { set_error("upload of the grid's arguments failed"); st = FMX_ERR_HIP; }
  for (int64_t off = 0; off < count && st == FMX_OK; off += CHUNK) {
    const int cnt = (int)((count - off < CHUNK) ? count - off : CHUNK);
    const SeqArgs* d_as_chunk = d_as + (size_t)(off / CHUNK) * n;
    const size_t pairs = (size_t)cnt * nz;
    uint32_t* keys = e->seq_keys;
    const size_t cap_pairs = (size_t)e->seq_wcap * WIN_NZ_MAX;
    if (hipMemsetAsync(e->seq_conf, 0xFF, (size_t)cnt * sizeof(int), e->stream) != hipSuccess) { st = FMX_ERR_HIP; break; }
    hipLaunchKernelGGL(seq_pack_k, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, e->stream, e->seq_b + off, e->seq_len + off, cnt, nz, m->col, m->val,
                       e->hyper.kind == UPD_TDAP ? 1 : 0, (uint2*)e->seq_packed, keys,
                       keys + 2 * cap_pairs, e->seq_conf);
    size_t tb = e->seq_sort_tmp_bytes;
    if (rocprim::radix_sort_pairs(e->seq_sort_tmp, tb, keys, keys + cap_pairs, keys + 2 * cap_pairs, keys + 3 * cap_pairs, pairs, 0, 32, e->stream) != hipSuccess) { st = FMX_ERR_HIP; break; }
    hipLaunchKernelGGL(seq_conf_k, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, e->stream, keys + cap_pairs, keys + 3 * cap_pairs, (int)pairs, e->seq_conf);
    const WinArgs wa{(const uint2*)e->seq_packed, e->seq_len + off, e->seq_y + off, e->seq_conf, cnt};
    // the same choice of kernel per shape as the single-model launcher (launch_window_kind): the models' bits must be their single runs'
    const int pm = pipe_mode();
#define FMX_GRID_ONE(KIND, KL, NZ)                                                                                                                              \
  do {                                                                                                                                                          \
    bool piped = false;                                                                                                                                         \
    if constexpr (KIND != UPD_TDAP) {                                                                                                                           \
      if (pm == 2 || (pm == 1 && PipeFits<KIND, KL, NZ>::value)) {                                                                                              \
        hipLaunchKernelGGL((fm_seq_pipe_grid_k<KIND, KL, NZ>), dim3((unsigned)n), dim3(SeqWin<KIND, KL, NZ>::NW * 64), 0, e->stream, d_as_chunk, wa, (const Hyper*)d_hs); \
        piped = true;                                                                                                                                           \
      }                                                                                                                                                         \
    }                                                                                                                                                           \
    if (!piped) hipLaunchKernelGGL((fm_seq_window_grid_k<KIND, KL, NZ>), dim3((unsigned)n), dim3(SeqWin<KIND, KL, NZ>::NW * 64), 0, e->stream, d_as_chunk, wa, (const Hyper*)d_hs); \
  } while (0)
#define FMX_GRID_KIND(KIND)                                                                            \
  do {                                                                                                 \
    if (kl == 16) { if (nz == 32) FMX_GRID_ONE(KIND, 16, 32); else FMX_GRID_ONE(KIND, 16, 64); }       \
    else if (kl == 32) { if (nz == 32) FMX_GRID_ONE(KIND, 32, 32); else FMX_GRID_ONE(KIND, 32, 64); }  \
    else FMX_GRID_ONE(KIND, 64, 32);                                                                   \
  } while (0)
    switch (e->hyper.kind) {
      case UPD_SGD_L2: FMX_GRID_KIND(UPD_SGD_L2); break;
      case UPD_SGD_L1: FMX_GRID_KIND(UPD_SGD_L1); break;
      case UPD_TDAP: FMX_GRID_KIND(UPD_TDAP); break;
      default: FMX_GRID_KIND(UPD_FTRL); break;
    }
#undef FMX_GRID_ONE
#undef FMX_GRID_KIND
    if (hipGetLastError() != hipSuccess) { set_error("grid learner launch failed"); st = FMX_ERR_HIP; }
  }
  if (st == FMX_OK && hipStreamSynchronize(e->stream) != hipSuccess) { set_error("grid learner failed"); st = FMX_ERR_HIP; }
  (void)hipFree(d_as); (void)hipFree(d_hs);
  return st;
}

}  // namespace fmx
