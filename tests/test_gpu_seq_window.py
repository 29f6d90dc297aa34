"""The windowed sequential learner (fm_seq_window_k): groups of consecutive feature-disjoint examples are processed by
one wave each, only the w0 chain runs in order.  Same algorithm, order and association as the one-wave kernel, so the
results must be BITWISE equal to it (FMX_SEQ_WINDOW=0 selects the one-wave kernel), and equal to the oracle at 1e-11."""
import os

import numpy as np
import pytest

import oracle
from tests import util

pytestmark = pytest.mark.gpu

SOLVERS = {
    "sgd_l2": dict(solver="sgd", k=8, l2_regw=1e-3, l2_regv=2e-3, l2_reg0=1e-3, learn_rate=0.05),
    "sgd_l1": dict(solver="sgd", k=4, l1_regw=1e-3, l1_regv=5e-4, learn_rate=0.05),
    "ftrl": dict(solver="ftrl", k=8, l1_regw=1e-3, l1_regv=1e-3, l2_regw=1e-2, l2_regv=1e-2),
    "tdap": dict(solver="tdap", k=6, l1_regw=1e-4, l1_regv=1e-4, l2_regw=1e-2, l2_regv=1e-2, gamma=1e-3, alpha_v=0.05),
    "sgd_k64_nolinear": dict(solver="sgd", k=64, k0=False, k1=False, l2_regv=1e-3, learn_rate=0.02),
    "ftrl_reg": dict(solver="ftrl", task=oracle.REGRESSION, k=16, l1_regw=1e-4, l1_regv=1e-4, l2_regw=1e-4, l2_regv=1e-4),
    # 17 <= k <= 32: two nonzero blocks of 32 lanes
    "sgd_l1_k32": dict(solver="sgd", k=32, l1_regw=1e-3, l1_regv=5e-4, learn_rate=0.03),
    "ftrl_k24": dict(solver="ftrl", k=24, l1_regw=1e-3, l1_regv=1e-3, l2_regw=1e-2, l2_regv=1e-2),
    "tdap_k20": dict(solver="tdap", k=20, l1_regw=1e-4, l1_regv=1e-4, l2_regw=1e-2, l2_regv=1e-2, gamma=1e-3, alpha_v=0.05),
    "ftrl_k40": dict(solver="ftrl", k=40, l1_regw=1e-3, l1_regv=1e-3, l2_regw=1e-2, l2_regv=1e-2),   # four waves per workgroup: three workers in the pipelined form
    "tdap_k40": dict(solver="tdap", k=40, l1_regw=1e-4, l1_regv=1e-4, l2_regw=1e-2, l2_regv=1e-2, gamma=1e-3, alpha_v=0.05),
    # no factors at all, and a single one
    "sgd_k0": dict(solver="sgd", k=0, l2_regw=1e-3, l2_reg0=1e-3, learn_rate=0.05),
    "ftrl_k1": dict(solver="ftrl", k=1, l1_regw=1e-3, l1_regv=1e-3, l2_regw=1e-2, l2_regv=1e-2),
}


def _run(name, c, p, n, nnz, iters, random_step=1, window=True, max_nnz=32):
    from fmwr_amd import engine, _lib as L
    rp, col, val = util.random_csr(n, p, nnz, seed=len(name) + p, empty_rows=True, max_nnz=max_nnz)  # rows of the register-resident path
    task = c.get("task", oracle.CLASSIFICATION)
    y = util.labels(n, 5, "classification" if task == oracle.CLASSIFICATION else "regression")
    kw = {k: v for k, v in c.items() if k != "solver"}
    P = oracle.params(min_target=float(y.min()), max_target=float(y.max()), random_step=random_step, **kw)
    w0, w, v = util.params(p, P.k, 3, fp32=False)
    solver = {"sgd": L.SOLVER_SGD, "ftrl": L.SOLVER_FTRL, "tdap": L.SOLVER_TDAP}[c["solver"]]
    os.environ["FMX_SEQ_WINDOW"] = {True: "1", False: "0"}.get(window, window)  # "2": the pipelined kernel on every non-TDAP shape
    try:
        e = engine.Engine(p, task=P.task, solver=solver, num_factor=P.k, keep_w0=P.k0, keep_w1=P.k1, l2_w0=P.l2_reg0, l1_w1=P.l1_regw,
                          l2_w1=P.l2_regw, l1_v=P.l1_regv, l2_v=P.l2_regv, learn_rate=P.learn_rate, alpha_w=P.alpha_w, alpha_v=P.alpha_v,
                          beta_w=P.beta_w, beta_v=P.beta_v, gamma=P.gamma, random_step=random_step, mode=L.MODE_SEQUENTIAL,
                          min_target=P.min_target, max_target=P.max_target)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        order = oracle.visit_order(n, random_step, iters, seed=7)
        e.train_order(m, order)
        got = e.get_params()
    finally:
        os.environ.pop("FMX_SEQ_WINDOW", None)
    return got, (P, rp, col, val, y, w0, w, v, order)


@pytest.mark.parametrize("name", list(SOLVERS))
@pytest.mark.parametrize("p,nnz", [(40, 6), (3000, 12), (200000, 30), (5000, 32)])
def test_windowed_learner_is_bitwise_the_one_wave_learner(name, p, nnz):
    """p = 40: every example conflicts with its neighbour (groups of one); p = 3000: mixed; p = 200000: groups mostly full;
    (5000, 32): most rows are clipped to exactly the 32 entries the register-resident path holds."""
    c = SOLVERS[name]
    n = 1500
    a, ctx = _run(name, c, p, n, nnz, 2 * n + 11, window=True)
    b, _ = _run(name, c, p, n, nnz, 2 * n + 11, window=False)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    if c["solver"] != "tdap":  # the pipelined kernel (groups overlapped: chain of one group beside the gathers of the next)
        d, _ = _run(name, c, p, n, nnz, 2 * n + 11, window="2")
        assert d[0] == b[0] and np.array_equal(d[1], b[1]) and np.array_equal(d[2], b[2])
    P, rp, col, val, y, w0, w, v, order = ctx
    X = oracle.Matrix(rp, col, val, p)
    learn = {"sgd": oracle.sgd_learn, "ftrl": oracle.ftrl_learn, "tdap": oracle.tdap_learn}[c["solver"]]
    ref = learn(P, X, y, w0, w, v.ravel(), len(order), order=order)
    assert abs(a[0] - ref["w0"]) < 1e-11 and util.rel_err(a[1], ref["w"]) < 1e-11
    if P.k:
        assert util.rel_err(a[2], ref["v"].reshape(P.k, p)) < 1e-11


@pytest.mark.parametrize("name", [n for n, c in SOLVERS.items() if c["k"] <= 32])
def test_rows_of_33_to_64_entries_use_the_wide_layout(name):
    """k <= 32 and rows of up to 64 entries (Criteo's 39 fall here): 64 packed slots per example."""
    c = SOLVERS[name]
    n, p = 1200, 30000
    a, ctx = _run(name, c, p, n, 45, 2 * n + 7, window=True, max_nnz=64)
    b, _ = _run(name, c, p, n, 45, 2 * n + 7, window=False, max_nnz=64)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    if c["solver"] != "tdap":
        d, _ = _run(name, c, p, n, 45, 2 * n + 7, window="2", max_nnz=64)
        assert d[0] == b[0] and np.array_equal(d[1], b[1]) and np.array_equal(d[2], b[2])
    P, rp, col, val, y, w0, w, v, order = ctx
    assert np.diff(rp).max() > 40
    learn = {"sgd": oracle.sgd_learn, "ftrl": oracle.ftrl_learn, "tdap": oracle.tdap_learn}[c["solver"]]
    ref = learn(P, oracle.Matrix(rp, col, val, p), y, w0, w, v.ravel(), len(order), order=order)
    assert abs(a[0] - ref["w0"]) < 1e-11 and util.rel_err(a[1], ref["w"]) < 1e-11
    if P.k:
        assert util.rel_err(a[2], ref["v"].reshape(P.k, p)) < 1e-11


def test_windowed_learner_with_random_strides_and_many_chunks():
    """random_step = 3 (libc rand() strides) over more than one 65 536-example launch."""
    c = SOLVERS["sgd_l2"]
    a, ctx = _run("strides", c, 50000, 60000, 8, 150000, random_step=3, window=True)
    b, _ = _run("strides", c, 50000, 60000, 8, 150000, random_step=3, window=False)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    d, _ = _run("strides", c, 50000, 60000, 8, 150000, random_step=3, window="2")
    assert d[0] == b[0] and np.array_equal(d[1], b[1]) and np.array_equal(d[2], b[2])


def test_rows_longer_than_the_fast_path_keep_the_one_wave_kernel():
    """A matrix with a 40-entry row is outside the windowed kernel's register-resident path: the engine falls back."""
    from fmwr_amd import engine, _lib as L
    n, p, k = 300, 500, 4
    rng = np.random.default_rng(2)
    rows = [np.sort(rng.choice(p, 40 if r == 17 else 5, replace=False)) for r in range(n)]
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows).astype(np.uint32); val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = util.labels(n, 2)
    P = oracle.params(k=k, l2_regv=1e-3, learn_rate=0.05)
    w0, w, v = util.params(p, k, 2, fp32=False)
    order = oracle.visit_order(n, 1, 2 * n)
    ref = oracle.sgd_learn(P, oracle.Matrix(rp, col, val, p), y, w0, w, v.ravel(), len(order), order=order)
    e = engine.Engine(p, num_factor=k, l2_v=1e-3, learn_rate=0.05, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    e.train_order(engine.Matrix.from_csr(rp, col, val, p, y), order)
    g0, gw, gv = e.get_params()
    assert abs(g0 - ref["w0"]) < 1e-11 and util.rel_err(gv, ref["v"].reshape(k, p)) < 1e-11


def _fuzz_case(seed):
    rng = np.random.default_rng(1000 + seed)
    solver = ["sgd", "sgd", "ftrl", "tdap"][seed % 4]
    k = int(rng.choice([0, 1, 2, 5, 8, 16, 17, 31, 32, 33, 64]))
    c = dict(solver=solver, k=k, k0=bool(rng.random() < 0.8), k1=bool(rng.random() < 0.8),
             task=oracle.CLASSIFICATION if rng.random() < 0.6 else oracle.REGRESSION)
    if solver == "sgd":
        c.update(learn_rate=0.03, l2_regw=1e-3, l2_regv=1e-3, l2_reg0=1e-3)
        if seed % 8 == 1:
            c.update(l1_regw=1e-3, l1_regv=5e-4)
    elif solver == "ftrl":
        c.update(l1_regw=1e-3, l1_regv=1e-3, l2_regw=1e-2, l2_regv=1e-2, alpha_v=0.05)
    else:
        c.update(l1_regw=1e-4, l1_regv=1e-4, l2_regw=1e-2, l2_regv=1e-2, gamma=1e-3, alpha_v=0.05)
    p = int(rng.choice([7, 50, 400, 5000, 80000]))
    nnz = int(rng.choice([1, 3, 10, 25, 40]))
    max_nnz = 64 if (nnz > 25 and k <= 32) else 32
    n = int(rng.integers(50, 900))
    return c, p, n, min(nnz, p), max_nnz, int(rng.choice([1, 1, 2, 5]))


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_windowed_equals_one_wave(seed):
    c, p, n, nnz, max_nnz, rstep = _fuzz_case(seed)
    iters = 2 * n + 3
    a, ctx = _run("fuzz%d" % seed, c, p, n, nnz, iters, random_step=rstep, window=True, max_nnz=max_nnz)
    b, _ = _run("fuzz%d" % seed, c, p, n, nnz, iters, random_step=rstep, window=False, max_nnz=max_nnz)
    # (TDAP with keep.w0 off leaves w0 = -0/0 = NaN, in the reference too: TDAP_Learner.h:192)
    same = lambda x, y: np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)
    assert same(a[0], b[0]) and same(a[1], b[1]) and same(a[2], b[2]), (c, p, n, nnz, max_nnz, rstep)
    if c["solver"] != "tdap":
        d, _ = _run("fuzz%d" % seed, c, p, n, nnz, iters, random_step=rstep, window="2", max_nnz=max_nnz)
        assert same(d[0], b[0]) and same(d[1], b[1]) and same(d[2], b[2]), ("pipelined", c, p, n, nnz, max_nnz, rstep)
