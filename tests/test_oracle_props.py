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
