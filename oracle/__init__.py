"""ctypes front-end of the CPU oracle (oracle/fm_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from fmwr_amd/ (tests/test_no_oracle_in_product.py
enforces that).  Layouts follow the reference: V is factor-major [k][p] float64.

PARITY UNPINNED for the training path in the contract's sense (see the header of fm_oracle.c): the reference holds no
vectors for it and cannot be built here; the restatement is held by the reference-shipped probit tables, the FM identity
of the reference's own test and independent property pins, and agrees with the survey session's Appendix-B literals.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libfm_oracle.so")

CLASSIFICATION, REGRESSION = 10, 20
LL, AUC, ACC, RMSE, MSE, MAE = 0, 111, 222, 333, 444, 555


class Params(C.Structure):
    _fields_ = [
        ("task", C.c_int32), ("k", C.c_int32), ("k0", C.c_int32), ("k1", C.c_int32),
        ("l1_regw", C.c_double), ("l1_regv", C.c_double),
        ("l2_reg0", C.c_double), ("l2_regw", C.c_double), ("l2_regv", C.c_double),
        ("min_target", C.c_double), ("max_target", C.c_double),
        ("learn_rate", C.c_double),
        ("alpha_w", C.c_double), ("beta_w", C.c_double), ("alpha_v", C.c_double), ("beta_v", C.c_double),
        ("random_step", C.c_int32), ("eval_type", C.c_int32),
        ("trace_step", C.c_int64), ("conv_condition", C.c_double),
        ("batch_mean", C.c_int32), ("pad_", C.c_int32), ("gamma", C.c_double),
    ]


class Csr(C.Structure):
    _fields_ = [("n", C.c_int64), ("p", C.c_uint32), ("row_ptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p)]


_OMP_PATH = os.path.join(_HERE, "_build", "libfm_oracle_omp.so")


def build(force=False):
    """Compile oracle/fm_oracle.c -> oracle/_build/libfm_oracle.so and the all-core baselines (gcc)."""
    srcs = [os.path.join(_HERE, "fm_oracle.c"), os.path.join(_HERE, "fm_oracle_omp.c")]
    stale = any(not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s) for s in srcs) for o in (_LIB_PATH, _OMP_PATH))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_omp = None


def lib_omp():
    """The OpenMP baselines (fm_oracle_omp.c): bench.py's all-core CPU numbers only."""
    global _omp
    if _omp is None:
        build()
        _omp = C.CDLL(_OMP_PATH)
        _omp.fmo_omp_sgd_hogwild.restype = C.c_int64
    return _omp


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.fmo_predict.restype = C.c_double
        _lib.fmo_grad_mult.restype = C.c_double
        _lib.fmo_evaluate.restype = C.c_double
        _lib.fmo_sgd_learn.restype = C.c_int64
        _lib.fmo_ftrl_learn.restype = C.c_int64
        _lib.fmo_tdap_learn.restype = C.c_int64
        _lib.fmo_sgd_pass.restype = C.c_int64
        _lib.fmo_visit_order.restype = C.c_int64
        _lib.fmo_random_select.restype = C.c_uint32
    return _lib


def params(task=CLASSIFICATION, k=2, k0=True, k1=True, l1_regw=0.0, l1_regv=0.0, l2_reg0=0.0, l2_regw=0.0,
           l2_regv=0.0, min_target=-1.0, max_target=1.0, learn_rate=0.01, alpha_w=0.1, beta_w=1.0, alpha_v=0.1,
           beta_v=1.0, random_step=1, eval_type=LL, trace_step=-1, conv_condition=1e-4, batch_mean=True, gamma=1e-4):
    return Params(task, k, int(k0), int(k1), l1_regw, l1_regv, l2_reg0, l2_regw, l2_regv, min_target, max_target,
                  learn_rate, alpha_w, beta_w, alpha_v, beta_v, random_step, eval_type, trace_step, conv_condition, int(batch_mean), 0, gamma)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Matrix:
    """CSR holder keeping numpy arrays alive for the C struct."""

    def __init__(self, row_ptr, col, val, p):
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        # one spare element: the reference's Iterator reads col_idx[end]/value[end] (SURVEY A-13)
        self.col = np.ascontiguousarray(np.concatenate([np.asarray(col, dtype=np.uint32), [0]]).astype(np.uint32))
        self.val = np.ascontiguousarray(np.concatenate([np.asarray(val, dtype=np.float32), [0]]).astype(np.float32))
        self.n = len(self.row_ptr) - 1
        self.p = int(p)
        self.nnz = int(self.row_ptr[-1])
        self.c = Csr(self.n, self.p, _ptr(self.row_ptr), _ptr(self.col), _ptr(self.val))

    def transpose(self):
        """(col_ptr int64[p+1], row_idx u32[nnz], val f32[nnz]) -- util/Smatrix.h:155-185 by result."""
        col_ptr = np.zeros(self.p + 1, np.int64)
        row_idx = np.zeros(max(self.nnz, 1), np.uint32)
        val_t = np.zeros(max(self.nnz, 1), np.float32)
        lib().fmo_transpose(C.byref(self.c), _ptr(col_ptr), _ptr(row_idx), _ptr(val_t))
        return col_ptr, row_idx[: self.nnz], val_t[: self.nnz]


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a


def predict(P, X, w0, w, v, row):
    """core/Model.h:75-103; returns (y_hat, m_sum[k], m_sum_sqr[k])."""
    k = max(P.k, 1)
    s = np.zeros(k); q = np.zeros(k)
    w = _f64(w); v = _f64(v)
    yh = lib().fmo_predict(C.byref(P), C.c_uint32(X.p), C.c_double(w0), _ptr(w), _ptr(v), C.byref(X.c),
                           C.c_int64(row), _ptr(s), _ptr(q))
    return yh, s[: P.k], q[: P.k]


def predict_batch(P, X, w0, w, v, prob=False):
    """core/Model.h:106-161 (prob=True: :163-180 logistic; prob="probit": the MCMC/ALS table link)."""
    out = np.zeros(max(X.n, 1))
    w = _f64(w); v = _f64(v)
    fn = lib().fmo_predict_probit if prob == "probit" else lib().fmo_predict_prob if prob else lib().fmo_predict_batch
    fn(C.byref(P), C.c_uint32(X.p), C.c_double(w0), _ptr(w), _ptr(v), C.byref(X.c), _ptr(out))
    return out[: X.n]


def fast_pnorm(x):
    """util/Random.h:95-111, elementwise."""
    f = lib().fmo_fast_pnorm
    f.restype = C.c_double
    return np.array([f(C.c_double(float(t))) for t in np.atleast_1d(x)])


def fast_dpnorm(x):
    """util/Random.h:113-124, elementwise."""
    f = lib().fmo_fast_dpnorm
    f.restype = C.c_double
    return np.array([f(C.c_double(float(t))) for t in np.atleast_1d(x)])


def probit_tables():
    """The regenerated grids (pn_y[2861], dp_y[40001]) of util/RandomData.h / RandomData_.h."""
    pn, dp = np.zeros(2861), np.zeros(40001)
    lib().fmo_probit_tables(_ptr(pn), _ptr(dp))
    return pn, dp


def grad_mult(P, y_hat, y):
    yh = C.c_double(y_hat)
    m = lib().fmo_grad_mult(C.byref(P), C.byref(yh), C.c_float(y))
    return m, yh.value


def evaluate(task, etype, y_hat, y):
    y_hat = _f64(y_hat); y = np.ascontiguousarray(y, np.float32)
    return lib().fmo_evaluate(C.c_int(task), C.c_int(etype), _ptr(y_hat), _ptr(y), C.c_int64(len(y)))


def visit_order(n, random_step, max_iter, seed=None):
    """Row visiting order of SGD_Learner.h:86-88; seed=None keeps libc rand() state as is."""
    if seed is not None:
        lib().fmo_srand(C.c_uint(seed))
    out = np.zeros(max(max_iter, 1), np.int64)
    m = lib().fmo_visit_order(C.c_int64(n), C.c_int(random_step), C.c_int64(max_iter), _ptr(out))
    return out[:m]


def _learn(fn, P, X, y, w0, w, v, max_iter, order, trace_cap):
    y = np.ascontiguousarray(y, np.float32)
    w = _f64(w).copy(); v = _f64(v).copy()
    w0c = C.c_double(w0)
    ti = np.zeros(max(trace_cap, 1), np.int64); tv = np.zeros(max(trace_cap, 1))
    tn = C.c_int64(0); conv = C.c_int32(0)
    if order is not None:
        order = np.ascontiguousarray(order, np.int64)
        max_iter = min(max_iter, len(order))
    it = fn(C.byref(P), C.c_uint32(X.p), C.byref(w0c), _ptr(w), _ptr(v), C.byref(X.c), _ptr(y), C.c_int64(max_iter),
            _ptr(order), _ptr(ti), _ptr(tv), C.c_int64(trace_cap), C.byref(tn), C.byref(conv))
    n = min(tn.value, trace_cap)
    return dict(w0=w0c.value, w=w, v=v, iters=it, trace_iters=ti[:n], trace_vals=tv[:n], convergent=bool(conv.value))


def sgd_learn(P, X, y, w0, w, v, max_iter, order=None, trace_cap=0):
    """solver/SGD_Learner.h:79-178."""
    return _learn(lib().fmo_sgd_learn, P, X, y, w0, w, v, max_iter, order, trace_cap)


def ftrl_learn(P, X, y, w0, w, v, max_iter, order=None, trace_cap=0):
    """solver/FTRL_Learner.h:64-156."""
    return _learn(lib().fmo_ftrl_learn, P, X, y, w0, w, v, max_iter, order, trace_cap)


def tdap_learn(P, X, y, w0, w, v, max_iter, order=None, trace_cap=0):
    """solver/TDAP_Learner.h:79-233."""
    return _learn(lib().fmo_tdap_learn, P, X, y, w0, w, v, max_iter, order, trace_cap)


class SgdMinibatch:
    """Engine mini-batch SGD semantics (fm_oracle.c, 'engine semantics'); state lives here."""

    def __init__(self, P, X, y, w0, w, v):
        self.P, self.X = P, X
        self.y = np.ascontiguousarray(y, np.float32)
        self.w0 = C.c_double(w0); self.w = _f64(w).copy(); self.v = _f64(v).copy()
        self.q_w = np.zeros(max(X.p, 1)); self.q_v = np.zeros(max(P.k, 1) * max(X.p, 1)); self.u = np.zeros(2)

    def step(self, b0, b1):
        lib().fmo_sgd_minibatch_step(C.byref(self.P), C.c_uint32(self.X.p), C.byref(self.w0), _ptr(self.w), _ptr(self.v),
                                     C.byref(self.X.c), _ptr(self.y), C.c_int64(b0), C.c_int64(b1),
                                     _ptr(self.q_w), _ptr(self.q_v), _ptr(self.u))


class FtrlMinibatch:
    def __init__(self, P, X, y, w0, w, v):
        self.P, self.X = P, X
        self.y = np.ascontiguousarray(y, np.float32)
        self.w0 = C.c_double(w0); self.w = _f64(w).copy(); self.v = _f64(v).copy()
        kp = max(P.k, 1) * max(X.p, 1)
        self.zn0 = np.zeros(2); self.z_w = np.zeros(max(X.p, 1)); self.n_w = np.zeros(max(X.p, 1))
        self.z_v = np.zeros(kp); self.n_v = np.zeros(kp)

    def step(self, b0, b1):
        lib().fmo_ftrl_minibatch_step(C.byref(self.P), C.c_uint32(self.X.p), C.byref(self.w0), _ptr(self.w), _ptr(self.v),
                                      C.byref(self.X.c), _ptr(self.y), C.c_int64(b0), C.c_int64(b1),
                                      _ptr(self.zn0), _ptr(self.z_w), _ptr(self.n_w), _ptr(self.z_v), _ptr(self.n_v))


class TdapMinibatch:
    """Mini-batch TDAP (fm_oracle.c fmo_tdap_minibatch_step): state u, nu, delta, h, z per parameter."""

    def __init__(self, P, X, y, w0, w, v):
        self.P, self.X = P, X
        self.y = np.ascontiguousarray(y, np.float32)
        self.w0 = C.c_double(w0); self.w = _f64(w).copy(); self.v = _f64(v).copy()
        p, k = max(X.p, 1), max(P.k, 1)
        self.s0 = np.zeros(5); self.sw = np.zeros(5 * p); self.sv = np.zeros(5 * k * p)

    def step(self, b0, b1):
        lib().fmo_tdap_minibatch_step(C.byref(self.P), C.c_uint32(self.X.p), C.byref(self.w0), _ptr(self.w), _ptr(self.v), C.byref(self.X.c),
                                      _ptr(self.y), C.c_int64(b0), C.c_int64(b1), _ptr(self.s0), _ptr(self.sw), _ptr(self.sv))


def als_update_v(k, X, v, error, alpha=1.0, v_lambda=None, v_mu=None, znorm=None):
    """solver/MCMC_ALS_Learner.h:272-354; znorm ([k][p] standard normals) switches to the MCMC draw; returns (v_new, error_end, v_q_end)."""
    col_ptr, row_idx, val_t = X.transpose()
    v = _f64(v).copy(); err = _f64(error).copy(); vq = np.zeros(max(X.n, 1))
    lam = _f64(v_lambda if v_lambda is not None else np.zeros(k)); mu = _f64(v_mu if v_mu is not None else np.zeros(k))
    row_idx = np.ascontiguousarray(row_idx); val_t = np.ascontiguousarray(val_t)
    lib().fmo_als_update_v(C.c_int(k), C.c_uint32(X.p), _ptr(v), C.c_int64(X.n), _ptr(col_ptr), _ptr(row_idx), _ptr(val_t),
                           _ptr(err), _ptr(vq), C.c_double(alpha), _ptr(lam), _ptr(mu), _ptr(None if znorm is None else _f64(znorm)))
    return v, err, vq


def sgd_pass(P, X, y, w0, w, v):
    """bench.py cpu_baseline: one reference-order serial pass (rows 1..n-1); returns examples done."""
    y = np.ascontiguousarray(y, np.float32)
    w0c = C.c_double(w0)
    return lib().fmo_sgd_pass(C.byref(P), C.c_uint32(X.p), C.byref(w0c), _ptr(w), _ptr(v), C.byref(X.c), _ptr(y))


def omp_threads():
    return lib_omp().fmo_omp_max_threads()


def omp_predict_batch(P, X, w0, w, v, threads):
    """The reference's OpenMP-over-rows forward (core/Model.h:106-161) on `threads` cores."""
    out = np.zeros(max(X.n, 1))
    lib_omp().fmo_omp_predict_batch(C.byref(P), C.c_uint32(X.p), C.c_double(w0), _ptr(_f64(w)), _ptr(_f64(v)), C.byref(X.c), _ptr(out), C.c_int(threads))
    return out[: X.n]


def omp_sgd_hogwild(P, X, y, w0, w, v, threads):
    """One lock-free (Hogwild) pass of the reference's example step over all rows; w, v updated in place."""
    y = np.ascontiguousarray(y, np.float32)
    w0c = C.c_double(w0)
    return lib_omp().fmo_omp_sgd_hogwild(C.byref(P), C.c_uint32(X.p), C.byref(w0c), _ptr(w), _ptr(v), C.byref(X.c), _ptr(y), C.c_int(threads))


def batch_sums(P, X, y, w0, w, v, b0, b1, acc=None):
    """Per-coordinate gradient sums of rows [b0, b1) at parameters (w0, w, v), accumulated into `acc`
    (dict G0,Q0 scalars + Gw,Qw,cw [p], Gv,Qv [k][p]); fm_oracle.c fmo_batch_sums."""
    k, p = max(P.k, 1), max(X.p, 1)
    if acc is None:
        acc = dict(G0=0.0, Q0=0.0, Gw=np.zeros(p), Qw=np.zeros(p), cw=np.zeros(p), Gv=np.zeros(k * p), Qv=np.zeros(k * p))
    y = np.ascontiguousarray(y, np.float32); w = _f64(w); v = _f64(v)
    g0, q0 = C.c_double(), C.c_double()
    lib().fmo_batch_sums(C.byref(P), C.c_uint32(X.p), C.c_double(w0), _ptr(w), _ptr(v), C.byref(X.c), _ptr(y), C.c_int64(b0), C.c_int64(b1),
                         C.byref(g0), C.byref(q0), _ptr(acc["Gw"]), _ptr(acc["Qw"]), _ptr(acc["cw"]), _ptr(acc["Gv"]), _ptr(acc["Qv"]))
    acc["G0"] += g0.value; acc["Q0"] += q0.value
    return acc


def sgd_apply_sums(P, p, state, B, acc):
    """state: dict w0 (c_double), w, v, q_w, q_v, u -- fm_oracle.c fmo_sgd_apply_sums."""
    lib().fmo_sgd_apply_sums(C.byref(P), C.c_uint32(p), C.byref(state["w0"]), _ptr(state["w"]), _ptr(state["v"]), C.c_double(B), C.c_double(acc["G0"]),
                             _ptr(acc["Gw"]), _ptr(acc["cw"]), _ptr(acc["Gv"]), _ptr(state["q_w"]), _ptr(state["q_v"]), _ptr(state["u"]))


def ftrl_apply_sums(P, p, state, B, acc):
    lib().fmo_ftrl_apply_sums(C.byref(P), C.c_uint32(p), C.byref(state["w0"]), _ptr(state["w"]), _ptr(state["v"]), C.c_double(B), C.c_double(acc["G0"]), C.c_double(acc["Q0"]),
                              _ptr(acc["Gw"]), _ptr(acc["Qw"]), _ptr(acc["cw"]), _ptr(acc["Gv"]), _ptr(acc["Qv"]),
                              _ptr(state["zn0"]), _ptr(state["z_w"]), _ptr(state["n_w"]), _ptr(state["z_v"]), _ptr(state["n_v"]))


def scales(n, p, col, val, norm_columns):
    """util/Smatrix.h:98-135: returns (scaled float32 values, mean[p], std[p])."""
    col = np.ascontiguousarray(col, np.uint32); v = np.ascontiguousarray(val, np.float32).copy()
    nc = np.ascontiguousarray(norm_columns, np.int32)
    mean = np.zeros(max(p, 1)); std = np.zeros(max(p, 1))
    lib().fmo_scales(C.c_int64(n), C.c_uint32(p), C.c_int64(len(v)), _ptr(col), _ptr(v), _ptr(nc), C.c_int64(len(nc)), _ptr(mean), _ptr(std))
    return v, mean[:p], std[:p]


def normalize(col, val, mean, std):
    """util/Smatrix.h:137-153."""
    col = np.ascontiguousarray(col, np.uint32); v = np.ascontiguousarray(val, np.float32).copy()
    lib().fmo_normalize(C.c_int64(len(v)), _ptr(col), _ptr(v), _ptr(_f64(mean)), _ptr(_f64(std)))
    return v


def als_learn(P, X, y, w0, w, v, max_iter, with_v=False):
    """solver/MCMC_ALS_Learner.h:91-156 (ALS learner; P.task picks the residual of :520-562): returns (w0, w, v)."""
    col_ptr, row_idx, val_t = X.transpose()
    row_idx = np.ascontiguousarray(row_idx); val_t = np.ascontiguousarray(val_t)
    y = np.ascontiguousarray(y, np.float32)
    w = _f64(w).copy(); v = _f64(v).copy()
    w0c = C.c_double(w0)
    lib().fmo_als_learn(C.byref(P), C.c_uint32(X.p), C.byref(w0c), _ptr(w), _ptr(v), C.byref(X.c), _ptr(col_ptr), _ptr(row_idx), _ptr(val_t),
                        _ptr(y), C.c_int(max_iter), C.c_int(int(with_v)))
    return w0c.value, w, v


def mcmc_draw_shapes(n, p):
    """Shapes of the two standard Gamma variates an MCMC iteration consumes: (alpha_0 + n)/2, (alpha_0 + p + 1)/2 with alpha_0 = 1."""
    return (1.0 + n) / 2.0, (1.0 + p + 1.0) / 2.0


def mcmc_learn(P, X, y, w0, w, v, max_iter, gammas, normals, seed=None):
    """MCMC learner (fm_oracle.c fmo_mcmc_learn): gammas [max_iter][2], normals [max_iter][2 + p] pre-drawn standard
    variates; seed seeds libc rand() for the CLASSIFICATION residual.  Returns (w0, w, v, (alpha, w_lambda, w_mu))."""
    col_ptr, row_idx, val_t = X.transpose()
    row_idx = np.ascontiguousarray(row_idx); val_t = np.ascontiguousarray(val_t)
    y = np.ascontiguousarray(y, np.float32)
    w = _f64(w).copy(); v = _f64(v).copy()
    gammas = np.ascontiguousarray(gammas, np.float64); normals = np.ascontiguousarray(normals, np.float64)
    assert gammas.shape == (max_iter, 2) and normals.shape == (max_iter, 2 + X.p)
    w0c = C.c_double(w0)
    state = np.zeros(3)
    if seed is not None:
        lib().fmo_srand(C.c_uint(seed))
    lib().fmo_mcmc_learn(C.byref(P), C.c_uint32(X.p), C.byref(w0c), _ptr(w), _ptr(v), C.byref(X.c), _ptr(col_ptr), _ptr(row_idx), _ptr(val_t),
                         _ptr(y), C.c_int(max_iter), _ptr(gammas), _ptr(normals), _ptr(state))
    return w0c.value, w, v, tuple(state)


def mcmc_v_hyper(k, p, v, std_gammas, std_normals, v_lambda, v_mu, sample=True):
    """update_v_lambda + update_v_mu (fm_oracle.c fmo_mcmc_v_hyper); returns (v_lambda, v_mu)."""
    lam = _f64(v_lambda).copy(); mu = _f64(v_mu).copy()
    lib().fmo_mcmc_v_hyper(C.c_int(k), C.c_uint32(p), _ptr(_f64(v)), _ptr(_f64(std_gammas)), _ptr(_f64(std_normals)), _ptr(lam), _ptr(mu), C.c_int(int(sample)))
    return lam, mu


def als_learn_traced(P, X, y, w0, w, v, max_iter, with_v=False):
    """als_learn with the tracker on (P.trace_step > 0, P.eval_type): returns (w0, w, v, iters, evals)."""
    col_ptr, row_idx, val_t = X.transpose()
    row_idx = np.ascontiguousarray(row_idx); val_t = np.ascontiguousarray(val_t)
    y = np.ascontiguousarray(y, np.float32)
    w = _f64(w).copy(); v = _f64(v).copy()
    w0c = C.c_double(w0)
    cap = max_iter + 2
    iters = np.zeros(cap, np.int64); vals = np.zeros(cap); n = C.c_int64()
    lib().fmo_als_learn_traced(C.byref(P), C.c_uint32(X.p), C.byref(w0c), _ptr(w), _ptr(v), C.byref(X.c), _ptr(col_ptr), _ptr(row_idx), _ptr(val_t),
                               _ptr(y), C.c_int(max_iter), C.c_int(int(with_v)), _ptr(iters), _ptr(vals), C.c_int64(cap), C.byref(n))
    return w0c.value, w, v, iters[: n.value], vals[: n.value]
