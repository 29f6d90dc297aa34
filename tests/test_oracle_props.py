"""Oracle self-consistency (CPU): FM identity, predict vs predict_batch, mini-batch semantics reduce to the
reference's example step at batch size 1, visiting order quirks."""
import numpy as np

import oracle
from tests import util


def test_fm_identity_bruteforce():
    n, p, k = 40, 30, 4
    rp, col, val = util.random_csr(n, p, 5, seed=2)
    w0, w, v = util.params(p, k, seed=2, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(k=k)
    out = oracle.predict_batch(P, X, w0, w, v.ravel())
    for i in range(n):
        c = col[rp[i]:rp[i + 1]].astype(int); x = val[rp[i]:rp[i + 1]].astype(np.float64)
        brute = w0 + np.dot(w[c], x) + sum(np.dot(v[:, c[a]], v[:, c[b]]) * x[a] * x[b] for a in range(len(c)) for b in range(a + 1, len(c)))
        assert abs(out[i] - brute) < 1e-12
        one, s, q = oracle.predict(P, X, w0, w, v.ravel(), i)
        assert abs(one - out[i]) < 1e-13  # predict vs predict_batch differ only in association
        np.testing.assert_allclose(s, v[:, c] @ x, atol=1e-13)


def test_row0_never_visited_and_iter_counts_examples():
    order = oracle.visit_order(5, 1, 11)
    assert list(order) == [1, 2, 3, 4, 1, 2, 3, 4, 1, 2, 3]  # SURVEY A-2, A-3
    o1 = oracle.visit_order(1000, 4, 50, seed=1)
    o2 = oracle.visit_order(1000, 4, 50, seed=1)
    assert list(o1) == list(o2) and np.all(np.diff(o1)[np.diff(o1) > 0] <= 4) and o1[0] >= 1


def test_minibatch_semantics_reduce_to_reference_step():
    n, p, k = 150, 200, 6
    rp, col, val = util.random_csr(n, p, 8, seed=3)
    y = util.labels(n, 3)
    w0, w, v = util.params(p, k, 3, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    cases = [
        (oracle.params(k=k, l2_regw=1e-3, l2_regv=2e-3, l2_reg0=1e-3, learn_rate=0.05), oracle.sgd_learn, oracle.SgdMinibatch),
        (oracle.params(k=k, l1_regw=1e-3, l1_regv=2e-3, learn_rate=0.05), oracle.sgd_learn, oracle.SgdMinibatch),
        (oracle.params(k=k, task=oracle.REGRESSION, l2_regv=1e-3, learn_rate=0.02, min_target=-0.5, max_target=0.5), oracle.sgd_learn, oracle.SgdMinibatch),
        (oracle.params(k=k, l1_regw=1e-3, l1_regv=1e-3, l2_regw=1e-2, l2_regv=1e-2), oracle.ftrl_learn, oracle.FtrlMinibatch),
    ]
    for P, learn, MB in cases:
        ref = learn(P, X, y, w0, w, v.ravel(), n, order=np.arange(n))
        for mean in (0, 1):  # both batch reductions are the reference step at batch size 1
            P.batch_mean = mean
            mb = MB(P, X, y, w0, w, v.ravel())
            for i in range(n):
                mb.step(i, i + 1)
            assert util.rel_err(mb.v, ref["v"]) < 1e-13 and util.rel_err(mb.w, ref["w"]) < 1e-13 and abs(mb.w0.value - ref["w0"]) < 1e-13


def test_als_update_v_decreases_squared_error():
    """ALS coordinate updates never increase the regularised squared error (lambda = 0 here)."""
    n, p, k = 300, 40, 3
    rp, col, val = util.random_csr(n, p, 6, seed=4, empty_rows=False)
    y = util.labels(n, 4, "regression")
    w0, w, v = util.params(p, k, 4, stdev=0.3, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=k)
    e0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    v1, e1, _ = oracle.als_update_v(k, X, v.ravel(), e0)
    assert np.sum(e1 ** 2) < np.sum(e0 ** 2)
    # the cached residual the sweep maintains equals a fresh forward with the new V
    fresh = oracle.predict_batch(P, X, w0, w, v1) - y
    # (only to ~1e-7: the reference squares x in FLOAT, `val_ * val_`, MCMC_ALS_Learner.h:314,345, and the oracle keeps that)
    np.testing.assert_allclose(e1, fresh, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------
# Independent pins ("what's weak" 1): the reference holds no golden vectors for the solvers, so the oracle's
# gradient and update formulas are additionally checked against quantities that need NO restatement of the reference:
# the loss itself (finite differences of -log sigma(y*y_hat) / 0.5*(y_hat - y)^2 through an O(z^2) pairwise forward that
# never uses the sum-of-squares trick) and the closed-form FTRL-Proximal minimiser.

def _pairwise_forward(w0, w, v, c, x):
    """y_hat = w0 + sum_i w_i x_i + sum_{a<b} <v_a, v_b> x_a x_b, straight from the FM model definition."""
    out = w0 + float(np.dot(w[c], x))
    for a in range(len(c)):
        for b in range(a + 1, len(c)):
            out += float(np.dot(v[:, c[a]], v[:, c[b]])) * x[a] * x[b]
    return out


def _loss(task, y_hat, y):
    if task == oracle.CLASSIFICATION:
        return float(np.log1p(np.exp(-y * y_hat)))  # -log sigma(y * y_hat)
    return 0.5 * (y_hat - y) ** 2


def test_gradient_matches_finite_differences_of_the_loss():
    """mult * x (w), mult * (s_f x - v x^2) (V) and mult (w0) of SGD_Learner.h:102-131 ARE dL/dtheta: central differences
    of the loss through the pairwise forward agree to 1e-7 relative for every coordinate of every row."""
    n, p, k = 25, 18, 5
    rp, col, val = util.random_csr(n, p, 6, seed=11, empty_rows=True)
    w0, w, v = util.params(p, k, seed=11, stdev=0.3, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    h = 1e-6
    for task in (oracle.CLASSIFICATION, oracle.REGRESSION):
        y = util.labels(n, 11, "classification" if task == oracle.CLASSIFICATION else "regression")
        # wide target range: the regression clamp (A-12) is the identity here, so the loss is smooth
        P = oracle.params(task=task, k=k, min_target=-1e9, max_target=1e9)
        for i in range(n):
            c = col[rp[i]:rp[i + 1]].astype(int); x = val[rp[i]:rp[i + 1]].astype(np.float64)
            acc = oracle.batch_sums(P, X, y, w0, w, v.ravel(), i, i + 1)
            f = lambda w0_, w_, v_: _loss(task, _pairwise_forward(w0_, w_, v_, c, x), float(y[i]))
            fd0 = (f(w0 + h, w, v) - f(w0 - h, w, v)) / (2 * h)
            assert abs(fd0 - acc["G0"]) <= 1e-7 * max(1.0, abs(fd0))
            Gv = acc["Gv"].reshape(k, p)
            for j in set(c.tolist()):
                wp, wm = w.copy(), w.copy(); wp[j] += h; wm[j] -= h
                fd = (f(w0, wp, v) - f(w0, wm, v)) / (2 * h)
                assert abs(fd - acc["Gw"][j]) <= 1e-7 * max(1.0, abs(fd)), (task, i, j)
                for fct in range(k):
                    vp, vm = v.copy(), v.copy(); vp[fct, j] += h; vm[fct, j] -= h
                    fd = (f(w0, w, vp) - f(w0, w, vm)) / (2 * h)
                    assert abs(fd - Gv[fct, j]) <= 1e-7 * max(1.0, abs(fd)), (task, i, j, fct)
            # coordinates the row does not hold get no gradient
            untouched = np.setdiff1d(np.arange(p), c)
            assert np.all(acc["Gw"][untouched] == 0.0) and np.all(Gv[:, untouched] == 0.0)


def test_sgd_step_is_gradient_step_plus_lazy_l2():
    """One oracle SGD example step == theta - lr*grad, then theta *= (1 - lr*reg) on the touched coordinates only
    (SGD_Learner.h:106-135), with grad from the finite-difference-checked sums above."""
    n, p, k = 12, 15, 3
    rp, col, val = util.random_csr(n, p, 5, seed=12, empty_rows=False)
    y = util.labels(n, 12)
    w0, w, v = util.params(p, k, seed=12, stdev=0.2, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    lr, rw, rv, r0 = 0.07, 3e-3, 5e-3, 2e-3
    P = oracle.params(k=k, learn_rate=lr, l2_regw=rw, l2_regv=rv, l2_reg0=r0)
    for i in range(1, n):
        acc = oracle.batch_sums(P, X, y, w0, w, v.ravel(), i, i + 1)
        ref = oracle.sgd_learn(P, X, y, w0, w, v.ravel(), 1, order=np.array([i]))
        c = np.unique(col[rp[i]:rp[i + 1]].astype(int))
        w_exp = w.copy(); w_exp[c] = (w[c] - lr * acc["Gw"][c]); w_exp[c] -= lr * rw * w_exp[c]
        v_exp = v.copy(); g = acc["Gv"].reshape(k, p)
        v_exp[:, c] = v[:, c] - lr * g[:, c]; v_exp[:, c] -= lr * rv * v_exp[:, c]
        assert abs(ref["w0"] - (w0 - lr * (acc["G0"] + r0 * w0))) < 1e-15
        np.testing.assert_allclose(ref["w"], w_exp, rtol=0, atol=1e-15)
        np.testing.assert_allclose(ref["v"].reshape(k, p), v_exp, rtol=0, atol=1e-15)


def test_ftrl_prox_is_the_argmin_of_its_objective():
    """FTRL_Learner.h:177-182: theta* = argmin_t  z*t + 0.5*((beta + sqrt(n))/alpha + l2)*t^2 + l1*|t|  -- checked by brute
    force on a grid around the closed form (McMahan et al. 2013, eq. 3), after one oracle FTRL example step."""
    n, p, k = 8, 10, 2
    rp, col, val = util.random_csr(n, p, 4, seed=13, empty_rows=False)
    y = util.labels(n, 13)
    w0, w, v = util.params(p, k, seed=13, stdev=0.4, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    a, b, l1, l2 = 0.3, 0.7, 0.02, 0.05
    P = oracle.params(k=k, alpha_w=a, alpha_v=a, beta_w=b, beta_v=b, l1_regw=l1, l1_regv=l1, l2_regw=l2, l2_regv=l2)
    i = 3
    acc = oracle.batch_sums(P, X, y, w0, w, v.ravel(), i, i + 1)
    ref = oracle.ftrl_learn(P, X, y, w0, w, v.ravel(), 1, order=np.array([i]))
    c = np.unique(col[rp[i]:rp[i + 1]].astype(int))
    g = acc["Gv"].reshape(k, p)
    for j in c:
        for fct in range(k):
            nn = g[fct, j] ** 2                      # n after the first touch (n_old = 0)
            z = g[fct, j] - v[fct, j] * np.sqrt(nn) / a  # z += g - sigma*theta, sigma = (sqrt(n_new) - 0)/alpha
            curv = (b + np.sqrt(nn)) / a + l2
            got = ref["v"].reshape(k, p)[fct, j]
            obj = lambda t: z * t + 0.5 * curv * t * t + l1 * abs(t)
            grid = got + np.linspace(-1e-3, 1e-3, 2001)
            assert obj(got) <= np.min([obj(t) for t in grid]) + 1e-15
            if abs(z) <= l1:
                assert got == 0.0


def test_tdap_minibatch_reduces_to_the_reference_step():
    """Mini-batch TDAP (defined in the oracle) at batch size 1 against the reference's TDAP learner
    (TDAP_Learner.h:79-233).  The shipped w prox reads z_w by the entry's POSITION in the row (A-6); the mini-batch form reads
    the feature's own z, so the two agree exactly where position == column (rows holding columns 0..len-1) and, for any rows,
    when the linear term is off; both reductions."""
    k = 4
    # (a) rows whose entry at position i is column i: the indexing bug is invisible
    rng = np.random.default_rng(5)
    n, p = 120, 12
    lens = rng.integers(2, p + 1, n)
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum(lens)
    col = np.concatenate([np.arange(m) for m in lens]).astype(np.uint32)
    val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = util.labels(n, 5)
    w0, w, v = util.params(p, k, 5, stdev=0.2, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(k=k, l1_regw=1e-3, l1_regv=5e-4, l2_regw=1e-2, l2_regv=1e-2, alpha_w=0.1, alpha_v=0.05, gamma=3e-3)
    ref = oracle.tdap_learn(P, X, y, w0, w, v.ravel(), n, order=np.arange(n))
    for mean in (0, 1):
        P.batch_mean = mean
        mb = oracle.TdapMinibatch(P, X, y, w0, w, v.ravel())
        for i in range(n):
            mb.step(i, i + 1)
        assert util.rel_err(mb.v, ref["v"]) < 1e-13 and util.rel_err(mb.w, ref["w"]) < 1e-13 and abs(mb.w0.value - ref["w0"]) < 1e-13
    # (b) arbitrary rows, linear term off (w is then zeroed on touch by both, V and w0 follow the same formulas)
    n, p = 150, 200
    rp, col, val = util.random_csr(n, p, 8, seed=6)
    y = util.labels(n, 6)
    w0, w, v = util.params(p, k, 6, stdev=0.2, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(k=k, k1=False, l1_regv=5e-4, l2_regv=1e-2, alpha_v=0.05, gamma=3e-3)
    ref = oracle.tdap_learn(P, X, y, w0, w, v.ravel(), n, order=np.arange(n))
    mb = oracle.TdapMinibatch(P, X, y, w0, w, v.ravel())
    for i in range(n):
        mb.step(i, i + 1)
    assert util.rel_err(mb.v, ref["v"]) < 1e-13 and abs(mb.w0.value - ref["w0"]) < 1e-13 and np.array_equal(mb.w, ref["w"])
    # a batch holding a feature c times ages it c times: e^(-gamma c)
    P = oracle.params(k=1, gamma=0.5, batch_mean=False)
    Xd = oracle.Matrix(np.arange(4) * 1, np.zeros(3, np.uint32), np.ones(3, np.float32), 1)
    mb = oracle.TdapMinibatch(P, Xd, np.ones(3, np.float32), 0.0, np.zeros(1), np.zeros(1))
    mb.step(0, 3)
    sigma = np.sqrt(mb.sw[0]) / P.alpha_w
    assert abs(mb.sw[2] - np.exp(-1.5) * sigma) < 1e-15   # delta of w[0] after one step of c = 3 touches
