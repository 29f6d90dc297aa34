"""GPU parity of the FM forward (phase-1 kernel) against the oracle's Model::predict_batch restatement."""
import numpy as np
import pytest

import oracle
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm():
    from fmwr_amd import engine, _lib
    return engine, _lib


@pytest.mark.parametrize("k", [0, 1, 2, 3, 8, 16, 32, 64, 100])
@pytest.mark.parametrize("mode", ["minibatch", "sequential"])
def test_predict_batch_parity(fm, k, mode):
    engine, L = fm
    n, p = 1500, 400
    rp, col, val = util.random_csr(n, p, 12, seed=100 + k)
    w0, w, v = util.params(p, k, seed=k, fp32=(mode == "minibatch"))
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=k)
    ref = oracle.predict_batch(P, X, w0, w, v.ravel() if k else np.zeros(1))
    e = engine.Engine(p, num_factor=k, task=L.TASK_REGRESSION, mode=L.MODE_MINIBATCH if mode == "minibatch" else L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v if k else None)
    m = engine.Matrix.from_csr(rp, col, val, p)
    out = e.predict(m)
    # fp64 accumulation in row order; only the final cross-factor sum is associated differently
    np.testing.assert_allclose(out, ref, rtol=1e-12, atol=1e-13)
    assert np.array_equal(np.sign(out), np.sign(ref))  # prediction sign is bit-exact


def test_predict_links_and_flags(fm):
    engine, L = fm
    n, p, k = 700, 300, 8
    rp, col, val = util.random_csr(n, p, 9, seed=5)
    w0, w, v = util.params(p, k, seed=3)
    X = oracle.Matrix(rp, col, val, p)
    m = engine.Matrix.from_csr(rp, col, val, p)
    for k0, k1 in [(True, True), (False, True), (True, False), (False, False)]:
        P = oracle.params(task=oracle.CLASSIFICATION, k=k, k0=k0, k1=k1)
        e = engine.Engine(p, num_factor=k, keep_w0=int(k0), keep_w1=int(k1), mode=L.MODE_MINIBATCH, min_target=-0.2, max_target=0.3)
        e.set_params(w0, w, v)
        np.testing.assert_allclose(e.predict(m), oracle.predict_batch(P, X, w0, w, v.ravel()), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(e.predict(m, L.LINK_LOGISTIC), oracle.predict_batch(P, X, w0, w, v.ravel(), prob=True), rtol=1e-12)
        clamped = np.clip(oracle.predict_batch(P, X, w0, w, v.ravel()), -0.2, 0.3)
        np.testing.assert_allclose(e.predict(m, L.LINK_CLAMP), clamped, rtol=1e-12, atol=1e-13)


def test_fm_identity(fm):
    """Analytic check of reference src/test/model.cpp:77-83: sum_{i<j} <v_i,v_j> x_i x_j by brute force."""
    engine, L = fm
    n, p, k = 60, 40, 5
    rp, col, val = util.random_csr(n, p, 6, seed=9)
    w0, w, v = util.params(p, k, seed=4)
    e = engine.Engine(p, num_factor=k, mode=L.MODE_MINIBATCH)
    e.set_params(w0, w, v)
    out = e.predict(engine.Matrix.from_csr(rp, col, val, p))
    for i in range(n):
        c = col[rp[i]:rp[i + 1]].astype(int); x = val[rp[i]:rp[i + 1]].astype(np.float64)
        brute = w0 + np.dot(w[c], x)
        for a in range(len(c)):
            for b in range(a + 1, len(c)):
                brute += np.dot(v[:, c[a]], v[:, c[b]]) * x[a] * x[b]
        assert abs(out[i] - brute) < 1e-10


def test_edge_shapes(fm):
    engine, L = fm
    p, k = 50, 16
    w0, w, v = util.params(p, k, seed=1)
    e = engine.Engine(p, num_factor=k, mode=L.MODE_MINIBATCH)
    e.set_params(w0, w, v)
    # all rows empty -> y_hat == w0
    m = engine.Matrix.from_csr(np.zeros(6, np.int64), np.zeros(0, np.uint32), np.zeros(0, np.float32), p)
    np.testing.assert_array_equal(e.predict(m), np.full(5, w0))
    # zero rows
    m0 = engine.Matrix.from_csr(np.zeros(1, np.int64), np.zeros(0, np.uint32), np.zeros(0, np.float32), p)
    assert e.predict(m0).shape == (0,)
    # one very long row: more entries than one LDS stage (2048) holds
    p2 = 6000
    e2 = engine.Engine(p2, num_factor=k, mode=L.MODE_MINIBATCH)
    w0b, wb, vb = util.params(p2, k, seed=2)
    e2.set_params(w0b, wb, vb)
    rp = np.array([0, 5000, 5001, 5001, 5600], np.int64)
    rng = np.random.default_rng(0)
    col = np.concatenate([np.sort(rng.choice(p2, 5000, replace=False)), [7], np.sort(rng.choice(p2, 599, replace=False))]).astype(np.uint32)
    val = rng.normal(0, 1, len(col)).astype(np.float32)
    ref = oracle.predict_batch(oracle.params(k=k), oracle.Matrix(rp, col, val, p2), w0b, wb, vb.ravel())
    np.testing.assert_allclose(e2.predict(engine.Matrix.from_csr(rp, col, val, p2)), ref, rtol=1e-11, atol=1e-12)


def test_errors(fm):
    engine, L = fm
    with pytest.raises(L.FmxError, match="factor.number"):
        engine.Engine(10, num_factor=1000)
    with pytest.raises(L.FmxError, match="Unknown solver"):
        engine.Engine(10, solver=400)  # BGD: declared in util/Macros.h:19, never implemented
    e = engine.Engine(10, num_factor=2, mode=L.MODE_MINIBATCH)
    m = engine.Matrix.from_csr(np.array([0, 1], np.int64), np.array([3], np.uint32), np.array([1.0], np.float32), 11)
    with pytest.raises(L.FmxError, match="number of input's features is not correct"):
        e.predict(m)
    with pytest.raises(L.FmxError, match="greater then then number of attributes"):
        engine.Matrix.from_csr(np.array([0, 1], np.int64), np.array([30], np.uint32), np.array([1.0], np.float32), 11)
