"""Seeded sweep over shapes the targeted tests do not pin: odd k, tiny / wide feature spaces, ragged and empty rows,
batch and tile sizes that do not divide anything, every solver kind and both reductions, truncated last steps.
Each case: forward, a few mini-batch steps and a short sequential run, all against the oracle."""
import numpy as np
import pytest

import oracle
from tests import util

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([0, 1, 2, 3, 5, 8, 13, 16, 24, 33, 64]))
    p = int(rng.choice([1, 2, 7, 50, 333, 2000, 40_000]))
    n = int(rng.choice([1, 2, 17, 300, 1111]))
    mean_nnz = float(rng.choice([0.5, 2, 9, 40]))
    solver = str(rng.choice(["sgd", "sgd_l1", "ftrl"]))
    task = int(rng.choice([oracle.CLASSIFICATION, oracle.REGRESSION]))
    batch = int(rng.choice([1, 3, 64, 100, 257, 5000]))
    tile = int(rng.choice([0, 1, 7, 50, 128]))
    mean = bool(rng.integers(0, 2))
    k0, k1 = bool(rng.integers(0, 4)), bool(rng.integers(0, 4))
    return dict(k=k, p=p, n=n, mean_nnz=min(mean_nnz, p), solver=solver, task=task, batch=batch, tile=tile, mean=mean, k0=k0, k1=k1)


# The nine seeds of 150..1649 whose fp32-state mini-batch run leaves the 1e-4 bar (profiles/r04_fuzz_more.txt): runs in which the dynamics amplify the fp32
# storage rounding (|V| growing to 5.9 .. 3e80; two of them short and tame-looking: seed 869, 41 steps, |V|max 286: w off by 1.8e-4; seed 940, 10 steps, |V|max 8.9:
# V off by 2.9e-4).  With cfg.state_fp64 = 1 -- the reference's precision, core/Model.h:26-42 -- the same engine follows the fp64 oracle on every one of them.
FP32_AMPLIFYING_SEEDS = [157, 439, 524, 656, 718, 869, 878, 940, 1388]


@pytest.mark.parametrize("seed", FP32_AMPLIFYING_SEEDS)
def test_fp64_state_minibatch_follows_the_oracle_where_fp32_state_does_not(seed):
    """include/fmx.h (state_fp64): the 1e-5-on-V bar against the reference CPU path is GUARANTEED in FMX_MODE_SEQUENTIAL and in the mini-batch mode with fp64
    state; with fp32 state it holds on runs that do not amplify rounding.  These are the amplifying runs the wider fuzz found."""
    from fmwr_amd import _lib as L, engine
    c = _case(seed)
    n, p, k = c["n"], c["p"], c["k"]
    rp, col, val = util.random_csr(n, p, c["mean_nnz"], seed=seed)
    y = util.labels(n, seed, "classification" if c["task"] == oracle.CLASSIFICATION else "regression")
    w0, w, v = util.params(p, k, seed, stdev=0.2, fp32=True)
    reg = dict(l2_regw=1e-3, l2_regv=2e-3, l2_reg0=1e-3)
    if c["solver"] == "sgd_l1":
        reg.update(l1_regw=1e-3, l1_regv=5e-4)
    if c["solver"] == "ftrl":
        reg.update(l1_regw=1e-3, l1_regv=1e-3)
    P = oracle.params(task=c["task"], k=k, k0=c["k0"], k1=c["k1"], learn_rate=0.03, batch_mean=c["mean"], min_target=float(y.min()), max_target=float(y.max()), **reg)
    X = oracle.Matrix(rp, col, val, p)
    vflat = v.ravel() if k else np.zeros(1)
    kw = dict(task=c["task"], solver=L.SOLVER_FTRL if c["solver"] == "ftrl" else L.SOLVER_SGD, num_factor=k, keep_w0=int(c["k0"]), keep_w1=int(c["k1"]),
              l2_w0=P.l2_reg0, l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv, l2_v=P.l2_regv, learn_rate=0.03, min_target=P.min_target, max_target=P.max_target)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    total = min(3 * n + 1, 2500)
    mb = (oracle.FtrlMinibatch if c["solver"] == "ftrl" else oracle.SgdMinibatch)(P, X, y, w0, w, vflat)
    done, step, nb = 0, 0, -(-n // c["batch"])
    while done < total:
        b0 = (step % nb) * c["batch"]
        rows = min(c["batch"], n - b0, total - done)
        mb.step(b0, b0 + rows); done += rows; step += 1
    e = engine.Engine(p, mode=L.MODE_MINIBATCH, batch_rows=c["batch"], tile_rows=c["tile"], batch_reduce=L.REDUCE_MEAN if c["mean"] else L.REDUCE_SUM, state_fp64=1, **kw)
    e.set_params(w0, w, v if k else None)
    assert e.train(m, total) == total
    g0, gw, gv = e.get_params()
    tol = 1e-5   # north_star's bar (measured: 6e-7 on the one chaotic case, 3e-9 .. 1e-14 on the others)
    assert np.all(np.isfinite(gv)) and np.all(np.isfinite(mb.v))
    assert np.max(np.abs(gv - mb.v.reshape(k, p))) < tol * max(np.max(np.abs(mb.v)), 1e-3), c
    assert np.max(np.abs(gw - mb.w)) < tol * max(np.max(np.abs(mb.w)), 1e-3), c
    assert abs(g0 - mb.w0.value) < tol * max(1.0, abs(mb.w0.value)), c


@pytest.mark.parametrize("seed", range(150))
def test_fuzz_against_oracle(seed):
    from fmwr_amd import _lib as L, engine
    c = _case(seed)
    n, p, k = c["n"], c["p"], c["k"]
    rp, col, val = util.random_csr(n, p, c["mean_nnz"], seed=seed)
    y = util.labels(n, seed, "classification" if c["task"] == oracle.CLASSIFICATION else "regression")
    w0, w, v = util.params(p, k, seed, stdev=0.2, fp32=True)
    reg = dict(l2_regw=1e-3, l2_regv=2e-3, l2_reg0=1e-3)
    if c["solver"] == "sgd_l1":
        reg.update(l1_regw=1e-3, l1_regv=5e-4)
    if c["solver"] == "ftrl":
        reg.update(l1_regw=1e-3, l1_regv=1e-3)
    P = oracle.params(task=c["task"], k=k, k0=c["k0"], k1=c["k1"], learn_rate=0.03, batch_mean=c["mean"], min_target=float(y.min()) if n else -1.0,
                      max_target=float(y.max()) if n else 1.0, **reg)
    X = oracle.Matrix(rp, col, val, p)
    vflat = v.ravel() if k else np.zeros(1)
    kw = dict(task=c["task"], solver=L.SOLVER_FTRL if c["solver"] == "ftrl" else L.SOLVER_SGD, num_factor=k, keep_w0=int(c["k0"]), keep_w1=int(c["k1"]),
              l2_w0=P.l2_reg0, l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv, l2_v=P.l2_regv, learn_rate=0.03, min_target=P.min_target, max_target=P.max_target)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    # forward
    e = engine.Engine(p, mode=L.MODE_MINIBATCH, batch_rows=c["batch"], tile_rows=c["tile"], batch_reduce=L.REDUCE_MEAN if c["mean"] else L.REDUCE_SUM, **kw)
    e.set_params(w0, w, v if k else None)
    np.testing.assert_allclose(e.predict(m), oracle.predict_batch(P, X, w0, w, vflat), rtol=1e-11, atol=1e-12)
    # mini-batch steps (wrapping, last one truncated)
    total = min(3 * n + 1, 2500)
    mb = (oracle.FtrlMinibatch if c["solver"] == "ftrl" else oracle.SgdMinibatch)(P, X, y, w0, w, vflat)
    done, step, nb = 0, 0, -(-n // c["batch"])
    while done < total:
        b0 = (step % nb) * c["batch"]
        rows = min(c["batch"], n - b0, total - done)
        mb.step(b0, b0 + rows); done += rows; step += 1
    assert e.train(m, total) == total
    g0, gw, gv = e.get_params()
    # fp32 state against the fp64 oracle: 1e-5 holds on the sparse problems of the targeted tests; here tiny dense feature
    # spaces (p = 1..7: every coordinate moves in every step, dozens of steps) amplify the fp32 rounding, hence 1e-4
    tol = 1e-4
    scale = max(np.max(np.abs(mb.v)) if k else 0.0, 1e-3)
    if k:
        assert np.max(np.abs(gv - mb.v.reshape(k, p))) < tol * scale, c
    assert np.max(np.abs(gw - mb.w)) < tol * max(np.max(np.abs(mb.w)), 1e-3), c
    assert abs(g0 - mb.w0.value) < tol * max(1.0, abs(mb.w0.value)), c
    # sequential
    if n > 1:
        iters = min(2 * n + 3, 400)
        learn = oracle.ftrl_learn if c["solver"] == "ftrl" else oracle.sgd_learn
        ref = learn(P, X, y, w0, w, vflat, iters)
        es = engine.Engine(p, mode=L.MODE_SEQUENTIAL, **kw)
        es.set_params(w0, w, v if k else None)
        assert es.train(m, iters) == iters
        s0, sw, sv = es.get_params()
        if k:
            assert np.max(np.abs(sv - ref["v"].reshape(k, p))) < 1e-10 * max(np.max(np.abs(ref["v"])), 1e-3), c
        assert np.max(np.abs(sw - ref["w"])) < 1e-10 * max(np.max(np.abs(ref["w"])), 1e-3) and abs(s0 - ref["w0"]) < 1e-10, c
