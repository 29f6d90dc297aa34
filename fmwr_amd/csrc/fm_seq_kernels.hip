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
#include "fmx_internal.h"

namespace fmx {

struct SeqArgs {
  const int64_t* row_ptr;
  const uint32_t* col;
  const float* val;
  const float* y;
  const int64_t* order;
  int64_t count;
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

template <int KIND>
__global__ __launch_bounds__(64) void fm_seq_learn_k(SeqArgs a, Hyper h) {
  const int lane = threadIdx.x;
  const int k = a.k, kp = a.kp;
  const bool k0 = h.k0 != 0, k1 = h.k1 != 0;
  double w0 = a.scal[SC_W0], z0 = a.scal[SC_Z0], n0 = a.scal[SC_N0], uw = a.scal[SC_UW], uv = a.scal[SC_UV];
  double t_nu = a.scal[SC_T_NU], t_delta = a.scal[SC_T_DELTA], t_h = a.scal[SC_T_H];

  for (int64_t ex = 0; ex < a.count; ++ex) {
    const int64_t row = a.order[ex];
    const int64_t b = a.row_ptr[row];
    const int len = (int)(a.row_ptr[row + 1] - b);
    const float yv = a.y[row];
    if constexpr (KIND == UPD_SGD_L1) { uw += h.lr * h.regw; uv += h.lr * h.regv; }  // SGD_Learner.h:92-97

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

int launch_seq_learn(fmx_engine* e, const fmx_matrix* m, const int64_t* d_order, int64_t count) {
  FMX_CHECK(e->k <= 64 * FI, FMX_ERR_INVALID, "sequential mode supports factor.number <= %d", 64 * FI);
  SeqArgs a{m->row_ptr, m->col, m->val, m->y, d_order, 0, e->dV, e->dw, e->dsV, e->dsw, e->dnV, e->dnw,
            e->dt1V, e->dt1w, e->dt2V, e->dt2w, e->dt3V, e->dt3w, e->scal,
            e->k, e->kp64, m->rows_sorted};
  // bounded launches: a single wave walking millions of examples in one dispatch would run for seconds
  const int64_t CHUNK = 1 << 16;
  for (int64_t off = 0; off < count; off += CHUNK) {
    a.order = d_order + off;
    a.count = (count - off < CHUNK) ? count - off : CHUNK;
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

}  // namespace fmx
