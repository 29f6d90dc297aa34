"""GPU parity of the training paths.

SEQUENTIAL mode is the reference's algorithm itself (fp64, reference order): it is held to the reference's own
known answers and to the oracle at ~1e-12; the north-star tolerance (1e-5 relative on V,
prediction sign bit-exact) is far looser.
MINIBATCH mode (fp32 state) is held to the oracle's fp64 restatement of the mini-batch semantics at 1e-5.
"""
import numpy as np
import pytest

import oracle
from tests import kat, util

pytestmark = pytest.mark.gpu

V_RTOL = 1e-5  # BASELINE.json north_star: "within 1e-5 relative on V"


@pytest.fixture(scope="module")
def fm():
    from fmwr_amd import engine, _lib
    return engine, _lib


def _engine(fm, p, P, solver, mode, batch_rows=64, **extra):
    engine, L = fm
    return engine.Engine(p, **extra, batch_reduce=L.REDUCE_MEAN if P.batch_mean else L.REDUCE_SUM, task=P.task, solver=solver, num_factor=P.k, keep_w0=P.k0, keep_w1=P.k1, l2_w0=P.l2_reg0,
                         l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv, l2_v=P.l2_regv, learn_rate=P.learn_rate,
                         alpha_w=P.alpha_w, alpha_v=P.alpha_v, beta_w=P.beta_w, beta_v=P.beta_v, random_step=P.random_step,
                         mode=mode, batch_rows=batch_rows, min_target=P.min_target, max_target=P.max_target)


def test_sequential_sgd_reference_kat(fm):
    engine, L = fm
    P = oracle.params(task=oracle.CLASSIFICATION, k=kat.K, l2_regw=kat.L2_REGW, l2_regv=kat.L2_REGV, learn_rate=0.05)
    e = _engine(fm, kat.P_FEAT, P, L.SOLVER_SGD, L.MODE_SEQUENTIAL)
    e.set_params(0.0, np.zeros(kat.P_FEAT), kat.harness_v0().reshape(kat.K, kat.P_FEAT))
    m = engine.Matrix.from_csr(kat.ROW_PTR, kat.COL, kat.VAL, kat.P_FEAT, kat.Y)
    assert e.train(m, kat.MAX_ITER) == 50
    w0, w, v = e.get_params()
    assert abs(w0 - kat.SGD_W0) < 1e-13
    np.testing.assert_allclose(w, kat.SGD_W, rtol=0, atol=1e-13)
    np.testing.assert_allclose(v[0], kat.SGD_V0, rtol=0, atol=1e-13)
    prob = e.predict(m, L.LINK_LOGISTIC)
    assert abs(oracle.evaluate(oracle.CLASSIFICATION, oracle.LL, prob, kat.Y) - kat.SGD_LL[-1]) < 5e-10


def test_sequential_ftrl_reference_kat(fm):
    engine, L = fm
    P = oracle.params(task=oracle.CLASSIFICATION, k=kat.K, l2_regw=kat.L2_REGW, l2_regv=kat.L2_REGV, l1_regw=0.001, l1_regv=0.001)
    e = _engine(fm, kat.P_FEAT, P, L.SOLVER_FTRL, L.MODE_SEQUENTIAL)
    e.set_params(0.0, np.zeros(kat.P_FEAT), kat.harness_v0().reshape(kat.K, kat.P_FEAT))
    m = engine.Matrix.from_csr(kat.ROW_PTR, kat.COL, kat.VAL, kat.P_FEAT, kat.Y)
    e.train(m, kat.MAX_ITER)
    w0, w, v = e.get_params()
    assert abs(w0 - kat.FTRL_W0) < 1e-13
    assert np.all(v[:, 3] == 0.0)
    prob = e.predict(m, L.LINK_LOGISTIC)
    assert abs(oracle.evaluate(oracle.CLASSIFICATION, oracle.LL, prob, kat.Y) - kat.FTRL_LL[-1]) < 5e-10


CASES = [
    dict(name="sgd_l2_cls", solver="sgd", task=oracle.CLASSIFICATION, k=8, l2_regw=1e-3, l2_regv=2e-3, l2_reg0=1e-3, learn_rate=0.05),
    dict(name="sgd_l2_reg", solver="sgd", task=oracle.REGRESSION, k=16, l2_regw=1e-3, l2_regv=1e-3, learn_rate=0.02),
    dict(name="sgd_l1_cls", solver="sgd", task=oracle.CLASSIFICATION, k=4, l1_regw=1e-3, l1_regv=5e-4, l2_regw=0.5, learn_rate=0.05),
    dict(name="sgd_l1_reg_acts_as_l2", solver="sgd", task=oracle.REGRESSION, k=4, l1_regw=1e-3, l1_regv=5e-4, learn_rate=0.02),
    dict(name="sgd_k3_nolinear", solver="sgd", task=oracle.CLASSIFICATION, k=3, k0=False, k1=False, l2_regv=1e-3, learn_rate=0.05),
    dict(name="sgd_k70", solver="sgd", task=oracle.CLASSIFICATION, k=70, l2_regv=1e-3, learn_rate=0.05),
    dict(name="ftrl_l1l2_cls", solver="ftrl", task=oracle.CLASSIFICATION, k=8, l1_regw=1e-3, l1_regv=1e-3, l2_regw=1e-2, l2_regv=1e-2),
    dict(name="ftrl_reg", solver="ftrl", task=oracle.REGRESSION, k=16, l1_regw=1e-4, l1_regv=1e-4, l2_regw=1e-4, l2_regv=1e-4, alpha_v=0.05),
    dict(name="ftrl_nolinear", solver="ftrl", task=oracle.CLASSIFICATION, k=5, k0=False, k1=False, l1_regv=1e-4),
]


def _problem(c, n=1200, p=300, mean_nnz=10):
    seed = abs(hash(c["name"])) % 1000
    seed = sum(map(ord, c["name"]))
    rp, col, val = util.random_csr(n, p, mean_nnz, seed=seed)
    task = "classification" if c["task"] == oracle.CLASSIFICATION else "regression"
    y = util.labels(n, seed, task)
    kw = {k: v for k, v in c.items() if k not in ("name", "solver")}
    P = oracle.params(min_target=float(y.min()), max_target=float(y.max()), **kw)
    return rp, col, val, y, P, seed


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_sequential_matches_oracle(fm, c):
    engine, L = fm
    rp, col, val, y, P, seed = _problem(c)
    n, p = len(rp) - 1, 300
    w0, w, v = util.params(p, P.k, seed, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    iters = 2 * n + 37  # wraps around the matrix twice: row 0 is never visited (SURVEY A-2)
    learn = oracle.sgd_learn if c["solver"] == "sgd" else oracle.ftrl_learn
    ref = learn(P, X, y, w0, w, v.ravel(), iters)
    e = _engine(fm, p, P, L.SOLVER_SGD if c["solver"] == "sgd" else L.SOLVER_FTRL, L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    assert e.train(m, iters) == ref["iters"] == iters
    g0, gw, gv = e.get_params()
    rv = ref["v"].reshape(P.k, p)
    assert util.rel_err(gv, rv) < 1e-11 and util.rel_err(gw, ref["w"]) < 1e-11 and abs(g0 - ref["w0"]) < 1e-11
    assert util.rel_err(gv, rv) < V_RTOL
    # predictions of the trained model: sign bit-exact against the oracle's
    out = e.predict(m)
    refp = oracle.predict_batch(P, X, ref["w0"], ref["w"], ref["v"])
    assert np.array_equal(np.sign(out), np.sign(refp))


@pytest.mark.parametrize("task", [oracle.CLASSIFICATION, oracle.REGRESSION])
def test_sequential_tdap_matches_oracle(fm, task):
    """TDAP (row f-3), sequential mode: the reference's default solver, shipped indexing bug included."""
    engine, L = fm
    c = dict(name="tdap%d" % task, solver="tdap", task=task, k=6, l1_regw=1e-3, l1_regv=5e-4, l2_regw=1e-2, l2_regv=1e-2, alpha_v=0.05, gamma=3e-4)
    rp, col, val, y, P, seed = _problem(c)
    n, p = len(rp) - 1, 300
    w0, w, v = util.params(p, P.k, seed, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    iters = 2 * n + 11
    ref = oracle.tdap_learn(P, X, y, w0, w, v.ravel(), iters)
    e = engine.Engine(p, task=P.task, solver=L.SOLVER_TDAP, num_factor=P.k, l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv, l2_v=P.l2_regv,
                      alpha_w=P.alpha_w, alpha_v=P.alpha_v, gamma=P.gamma, mode=L.MODE_SEQUENTIAL, min_target=P.min_target, max_target=P.max_target)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    assert e.train(m, iters) == iters
    g0, gw, gv = e.get_params()
    assert util.rel_err(gv, ref["v"].reshape(P.k, p)) < 1e-10 and util.rel_err(gw, ref["w"]) < 1e-10 and abs(g0 - ref["w0"]) < 1e-10


def test_sequential_random_step(fm):
    """random_step > 1: strides come from libc rand() (util/Random.h:20-24), unseeded in the reference."""
    engine, L = fm
    c = dict(name="rs", solver="sgd", task=oracle.CLASSIFICATION, k=4, l2_regv=1e-3, learn_rate=0.05, random_step=3)
    rp, col, val, y, P, seed = _problem(c, n=500)
    p = 300
    w0, w, v = util.params(p, P.k, seed, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    order = oracle.visit_order(500, 3, 700, seed=1)  # glibc default state == srand(1)
    ref = oracle.sgd_learn(P, X, y, w0, w, v.ravel(), 700, order=order)
    e = _engine(fm, p, P, L.SOLVER_SGD, L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    e.train_order(m, order)
    assert util.rel_err(e.get_params()[2], ref["v"].reshape(P.k, p)) < 1e-11
    # and the library's own stride generator reproduces the same list from the same libc state
    oracle.lib().fmo_srand(1)
    e2 = _engine(fm, p, P, L.SOLVER_SGD, L.MODE_SEQUENTIAL)
    e2.set_params(w0, w, v)
    e2.train(m, 700)
    np.testing.assert_array_equal(e2.get_params()[2], e.get_params()[2])


def test_sequential_unsorted_rows_with_duplicates(fm):
    """A row holding the same column twice: the reference updates it twice in sequence (SURVEY A-11)."""
    engine, L = fm
    p, k = 20, 4
    rp = np.array([0, 3, 6, 9], np.int64)
    col = np.array([5, 2, 5, 1, 1, 7, 3, 4, 3], np.uint32)
    val = np.array([1, .5, -1, 2, 1, 1, .5, .25, .5], np.float32)
    y = np.array([1, -1, 1], np.float32)
    w0, w, v = util.params(p, k, 3, fp32=False)
    for solver, learn in (("sgd", oracle.sgd_learn), ("ftrl", oracle.ftrl_learn)):
        P = oracle.params(k=k, l2_regw=1e-2, l2_regv=1e-2, learn_rate=0.1)
        ref = learn(P, oracle.Matrix(rp, col, val, p), y, w0, w, v.ravel(), 40)
        e = _engine(fm, p, P, L.SOLVER_SGD if solver == "sgd" else L.SOLVER_FTRL, L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        e.train(engine.Matrix.from_csr(rp, col, val, p, y), 40)
        g0, gw, gv = e.get_params()
        assert util.rel_err(gv, ref["v"].reshape(k, p)) < 1e-11 and util.rel_err(gw, ref["w"]) < 1e-11


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
@pytest.mark.parametrize("batch", [1, 64, 257])
@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_minibatch_matches_oracle(fm, c, batch, reduce):
    engine, L = fm
    if batch == 1 and c["name"] not in ("sgd_l2_cls", "ftrl_l1l2_cls"):
        pytest.skip("batch 1 covered on two cases")
    n = 300 if batch == 1 else 1200
    rp, col, val, y, P, seed = _problem(c, n=n)
    P.batch_mean = int(reduce == "mean")
    p = 300
    w0, w, v = util.params(p, P.k, seed, fp32=True)
    X = oracle.Matrix(rp, col, val, p)
    mb = (oracle.SgdMinibatch if c["solver"] == "sgd" else oracle.FtrlMinibatch)(P, X, y, w0, w, v.ravel())
    e = _engine(fm, p, P, L.SOLVER_SGD if c["solver"] == "sgd" else L.SOLVER_FTRL, L.MODE_MINIBATCH, batch_rows=batch)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    total = n + n // 2 + 5  # one and a half passes, last batch truncated
    done, step, nb = 0, 0, -(-n // batch)
    while done < total:  # consecutive batches, wrapping at the end of the matrix (fmx.h: fmx_train, MINIBATCH)
        b0 = (step % nb) * batch
        rows = min(batch, n - b0, total - done)
        mb.step(b0, b0 + rows)
        done += rows
        step += 1
    assert e.train(m, total) == total
    g0, gw, gv = e.get_params()
    rv = mb.v.reshape(P.k, p)
    assert util.rel_err(gv, rv) < V_RTOL, util.rel_err(gv, rv)
    assert util.rel_err(gw, mb.w) < V_RTOL or np.max(np.abs(mb.w)) == 0
    assert abs(g0 - mb.w0.value) < V_RTOL * max(1.0, abs(mb.w0.value))


def test_minibatch_batch1_is_the_reference_step(fm):
    """At batch_rows == 1 the mini-batch semantics are the reference's example step (fp32 state => 1e-5, not 1e-12)."""
    engine, L = fm
    c = CASES[0]
    rp, col, val, y, P, seed = _problem(c, n=200)
    p = 300
    w0, w, v = util.params(p, P.k, seed, fp32=True)
    X = oracle.Matrix(rp, col, val, p)
    order = np.arange(0, 200)
    ref = oracle.sgd_learn(P, X, y, w0, w, v.ravel(), 200, order=order)
    e = _engine(fm, p, P, L.SOLVER_SGD, L.MODE_MINIBATCH, batch_rows=1)
    e.set_params(w0, w, v)
    e.train(engine.Matrix.from_csr(rp, col, val, p, y), 200)
    assert util.rel_err(e.get_params()[2], ref["v"].reshape(P.k, p)) < V_RTOL


def test_minibatch_is_bitwise_reproducible(fm):
    engine, L = fm
    c = CASES[0]
    rp, col, val, y, P, seed = _problem(c, n=3000)
    p = 300
    w0, w, v = util.params(p, P.k, seed)
    res = []
    for _ in range(2):
        e = _engine(fm, p, P, L.SOLVER_SGD, L.MODE_MINIBATCH, batch_rows=500)
        e.set_params(w0, w, v)
        e.train(engine.Matrix.from_csr(rp, col, val, p, y), 6000)
        res.append(e.get_params())
    assert res[0][0] == res[1][0]
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])


def test_grad_apply_equals_step(fm):
    """The multi-GPU split (fmx_grad -> [all-reduce] -> fmx_apply) on one GPU equals the fused fmx_step."""
    engine, L = fm
    for c in (CASES[0], CASES[6]):
        rp, col, val, y, P, seed = _problem(c, n=1000)
        p = 300
        w0, w, v = util.params(p, P.k, seed)
        solver = L.SOLVER_SGD if c["solver"] == "sgd" else L.SOLVER_FTRL
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        ea = _engine(fm, p, P, solver, L.MODE_MINIBATCH, batch_rows=250)
        eb = _engine(fm, p, P, solver, L.MODE_MINIBATCH, batch_rows=250)
        ea.set_params(w0, w, v); eb.set_params(w0, w, v)
        ec = _engine(fm, p, P, solver, L.MODE_MINIBATCH, batch_rows=250)
        ec.set_params(w0, w, v)
        for b in range(4):
            ea.step(m, b)
            eb.grad(m, b); eb.apply(250)
            ec.grad(m, b); ec.apply(0)  # row count taken from the exchange buffer's tail
        ea.sync(); eb.sync(); ec.sync()
        pa, pb, pc = ea.get_params(), eb.get_params(), ec.get_params()
        assert util.rel_err(pb[2], pa[2]) < 1e-6 and util.rel_err(pb[1], pa[1]) < 1e-6 and abs(pa[0] - pb[0]) < 1e-6
        assert pb[0] == pc[0] and np.array_equal(pb[1], pc[1]) and np.array_equal(pb[2], pc[2])
        # the tail sits right behind the planes: [sum mult, sum mult^2, rows, 0]
        import ctypes as C
        ptr, nfl = ec.grad_buffer()
        kp = 4
        while kp < P.k:
            kp *= 2
        has_q = c["solver"] == "ftrl" and not P.batch_mean
        assert nfl == p * kp * (2 if has_q else 1) + p * (3 if has_q else 2) + 4


@pytest.mark.parametrize("shape", ["fields", "ragged"])
def test_als_vsweep_matches_oracle(fm, shape):
    """MCMC_ALS_Learner::update_v (ALS branch): exact Gauss-Seidel order through level scheduling."""
    engine, L = fm
    rng = np.random.default_rng(11)
    if shape == "fields":  # one active feature per field and row (one-hot style): as many levels as fields
        n, fields, width, k = 2000, 6, 50, 5
        p = fields * width
        rp = np.arange(n + 1, dtype=np.int64) * fields
        col = (rng.integers(0, width, (n, fields)) + np.arange(fields)[None, :] * width).astype(np.uint32).ravel()
        val = rng.normal(0, 1, n * fields).astype(np.float32)
    else:
        n, p, k = 1500, 120, 4
        rp, col, val = util.random_csr(n, p, 7, seed=12)
    y = util.labels(n, 5, "regression")
    w0, w, v = util.params(p, k, 6, stdev=0.3, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=k)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y  # calculate_error, MCMC_ALS_Learner.h:520-527
    lam = np.linspace(0.0, 0.5, k); mu = np.linspace(-0.1, 0.1, k)
    rv, rerr, _ = oracle.als_update_v(k, X, v.ravel(), err0, alpha=1.3, v_lambda=lam, v_mu=mu)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    gerr = e.als_vsweep(m, err0, alpha=1.3, v_lambda=lam, v_mu=mu)
    gv = e.get_params()[2]
    assert util.rel_err(gv, rv.reshape(k, p)) < 1e-10
    assert util.rel_err(gerr, rerr) < 1e-10


@pytest.mark.parametrize("name", ["sgd_l2_cls", "sgd_l1_cls", "ftrl_l1l2_cls"])
@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_tiled_step_equals_the_batch_semantics(fm, name, reduce):
    """A step cut into tiles (parameters frozen across them, sums accumulated) is the same mini-batch step: the oracle
    knows nothing about tiles.  Also through the multi-GPU split (grad over tiles -> apply)."""
    engine, L = fm
    c = next(x for x in CASES if x["name"] == name)
    n, p, batch = 1500, 300, 600
    rp, col, val, y, P, seed = _problem(c, n=n)
    P.batch_mean = int(reduce == "mean")
    w0, w, v = util.params(p, P.k, seed, fp32=True)
    X = oracle.Matrix(rp, col, val, p)
    mb = (oracle.SgdMinibatch if c["solver"] == "sgd" else oracle.FtrlMinibatch)(P, X, y, w0, w, v.ravel())
    steps = [(0, 600), (600, 1200), (1200, 1500), (0, 600), (600, 900)]  # last one truncated by max_iter
    for b0, b1 in steps:
        mb.step(b0, b1)
    total = sum(b1 - b0 for b0, b1 in steps)
    solver = L.SOLVER_SGD if c["solver"] == "sgd" else L.SOLVER_FTRL
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    rv = mb.v.reshape(P.k, p)
    for tile in (128, 250, 600):
        e = engine.Engine(p, task=P.task, solver=solver, num_factor=P.k, l2_w0=P.l2_reg0, l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv,
                          l2_v=P.l2_regv, learn_rate=P.learn_rate, mode=L.MODE_MINIBATCH, batch_rows=batch, tile_rows=tile,
                          batch_reduce=L.REDUCE_MEAN if P.batch_mean else L.REDUCE_SUM)
        e.set_params(w0, w, v)
        assert e.num_batches(m) == 3
        assert e.train(m, total) == total
        g0, gw, gv = e.get_params()
        assert util.rel_err(gv, rv) < V_RTOL and util.rel_err(gw, mb.w) < V_RTOL and abs(g0 - mb.w0.value) < V_RTOL * max(1.0, abs(mb.w0.value))
        # the grad/apply split over the same tiles
        e2 = engine.Engine(p, task=P.task, solver=solver, num_factor=P.k, l2_w0=P.l2_reg0, l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv,
                           l2_v=P.l2_regv, learn_rate=P.learn_rate, mode=L.MODE_MINIBATCH, batch_rows=batch, tile_rows=tile,
                           batch_reduce=L.REDUCE_MEAN if P.batch_mean else L.REDUCE_SUM)
        e2.set_params(w0, w, v)
        for i, (b0, b1) in enumerate(steps):
            e2.grad(m, i % 3, b1 - b0)
            e2.apply(0)
        e2.sync()
        h0, hw, hv = e2.get_params()
        assert util.rel_err(hv, gv) < 1e-6 and util.rel_err(hw, gw) < 1e-6 and abs(h0 - g0) < 1e-6 * max(1.0, abs(g0))


@pytest.mark.parametrize("with_v", [False, True])
def test_als_training_loop_matches_oracle(fm, with_v):
    """MCMC_ALS_Learner::learn, ALS learner, regression: w0 update + w sweep per iteration (row f-4); as shipped V stays
    untouched (SURVEY A-1), with_v adds the sweep."""
    engine, L = fm
    n, p, k = 1200, 90, 3
    rp, col, val = util.random_csr(n, p, 6, seed=61, empty_rows=True)
    y = util.labels(n, 61, "regression")
    w0, w, v = util.params(p, k, 61, stdev=0.2, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=k, l2_reg0=0.05)
    r0, rw, rv = oracle.als_learn(P, X, y, w0, w, v.ravel(), 4, with_v=with_v)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, l2_w0=0.05, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    e.als_train(m, 4, with_v=with_v)
    g0, gw, gv = e.get_params()
    assert abs(g0 - r0) < 1e-10 and util.rel_err(gw, rw) < 1e-10 and util.rel_err(gv, rv.reshape(k, p)) < 1e-10
    if not with_v:
        assert np.array_equal(gv, v)  # the reference's ALS never moves V
    sse = lambda a, b, c: float(np.sum((oracle.predict_batch(P, X, a, b, np.asarray(c).ravel()) - y) ** 2))
    assert sse(g0, gw, gv) < sse(w0, w, v)


@pytest.mark.parametrize("with_v", [False, True])
def test_als_classification_matches_oracle(fm, with_v):
    """The CLASSIFICATION residual of the ALS learner (MCMC_ALS_Learner.h:545-559: -/+ dnorm/(1-pnorm) through the
    40001-point table) and the probit link of its predictions (fast_pnorm, core/Model.h:166-171)."""
    engine, L = fm
    n, p, k = 1500, 80, 3
    rp, col, val = util.random_csr(n, p, 6, seed=67, empty_rows=True)
    y = util.labels(n, 67)
    w0, w, v = util.params(p, k, 67, stdev=0.3, fp32=False)
    w = w * 5.0  # scores well beyond +-3 and +-5.2: both tables' saturation branches are visited
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.CLASSIFICATION, k=k, l2_reg0=0.02)
    e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_ALS, num_factor=k, l2_w0=0.02, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    raw = oracle.predict_batch(P, X, w0, w, v.ravel())
    assert raw.max() > 5.3 and raw.min() < -5.3
    np.testing.assert_allclose(e.predict(m, L.LINK_PROBIT), oracle.fast_pnorm(raw), rtol=0, atol=1e-15)
    assert abs(e.evaluate(m, L.EVAL_LL) - oracle.evaluate(oracle.CLASSIFICATION, oracle.LL, oracle.fast_pnorm(raw), y)) < 1e-9
    r0, rw, rv = oracle.als_learn(P, X, y, w0, w, v.ravel(), 3, with_v=with_v)
    e.als_train(m, 3, with_v=with_v)
    g0, gw, gv = e.get_params()
    assert abs(g0 - r0) < 1e-10 and util.rel_err(gw, rw) < 1e-10 and util.rel_err(gv, rv.reshape(k, p)) < 1e-10


@pytest.mark.parametrize("name", ["sgd_l2_cls", "sgd_l1_cls", "ftrl_l1l2_cls"])
def test_sparse_tiles_walk_only_occurring_features(fm, name):
    """Far more features than entries per batch (the usual FM regime): phase 2 walks the per-tile list of occurring
    features instead of all p.  Same results as the oracle, and as the dense walk (forced through the grad/apply split)."""
    engine, L = fm
    c = next(x for x in CASES if x["name"] == name)
    n, p, batch = 700, 6000, 100
    rp, col, val, y, P, seed = _problem(c, n=n, p=p, mean_nnz=6)
    w0, w, v = util.params(p, P.k, seed, fp32=True)
    X = oracle.Matrix(rp, col, val, p)
    mb = (oracle.SgdMinibatch if c["solver"] == "sgd" else oracle.FtrlMinibatch)(P, X, y, w0, w, v.ravel())
    for s in range(10):
        b0 = (s % 7) * batch
        mb.step(b0, b0 + batch)
    solver = L.SOLVER_SGD if c["solver"] == "sgd" else L.SOLVER_FTRL
    kw = dict(task=P.task, solver=solver, num_factor=P.k, l2_w0=P.l2_reg0, l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv, l2_v=P.l2_regv,
              learn_rate=P.learn_rate, mode=L.MODE_MINIBATCH, batch_rows=batch)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    e = engine.Engine(p, **kw); e.set_params(w0, w, v)
    e2 = engine.Engine(p, **kw); e2.set_params(w0, w, v)
    for s in range(10):
        e.step(m, s % 7)                   # sparse walk
        e2.grad(m, s % 7); e2.apply(0)     # dense walk through the exchange buffer
    e.sync(); e2.sync()
    g0, gw, gv = e.get_params()
    h0, hw, hv = e2.get_params()
    rv = mb.v.reshape(P.k, p)
    assert util.rel_err(gv, rv) < V_RTOL and util.rel_err(gw, mb.w) < V_RTOL and abs(g0 - mb.w0.value) < V_RTOL * max(1.0, abs(mb.w0.value))
    assert util.rel_err(hv, gv) < 1e-6 and util.rel_err(hw, gw) < 1e-6


def test_mcmc_vsweep_with_caller_drawn_normals(fm):
    """The Gibbs form of update_v (MCMC_ALS_Learner.h:329-331): v ~ N(mean, var) with the standard normals pre-drawn by
    the caller in loop order (BASELINE.json configs[4]: 'MCMC.solver Gibbs sweep over V columns')."""
    engine, L = fm
    rng = np.random.default_rng(71)
    n, fields, width, k = 1800, 5, 40, 4
    p = fields * width
    rp = np.arange(n + 1, dtype=np.int64) * fields
    col = (rng.integers(0, width, (n, fields)) + np.arange(fields)[None, :] * width).astype(np.uint32).ravel()
    val = rng.normal(0, 1, n * fields).astype(np.float32)
    y = util.labels(n, 71, "regression")
    w0, w, v = util.params(p, k, 71, stdev=0.3, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    err0 = oracle.predict_batch(oracle.params(task=oracle.REGRESSION, k=k), X, w0, w, v.ravel()) - y
    z = rng.normal(0, 1, (k, p))
    lam = np.full(k, 2.0); mu = np.linspace(-0.05, 0.05, k)
    rv, rerr, _ = oracle.als_update_v(k, X, v.ravel(), err0, alpha=0.7, v_lambda=lam, v_mu=mu, znorm=z.ravel())
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    gerr = e.als_vsweep(m, err0, alpha=0.7, v_lambda=lam, v_mu=mu, std_normals=z)
    assert util.rel_err(e.get_params()[2], rv.reshape(k, p)) < 1e-10 and util.rel_err(gerr, rerr) < 1e-10
    assert not np.allclose(rv, oracle.als_update_v(k, X, v.ravel(), err0, alpha=0.7, v_lambda=lam, v_mu=mu)[0])  # it really sampled


@pytest.mark.parametrize("name", ["sgd_l2_cls", "sgd_l1_cls", "ftrl_l1l2_cls"])
@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_long_lists_heavy_hitter_features(fm, name, reduce):
    """Skewed data: a few features occur in (almost) every row, so their per-tile lists are far longer than the 64-entry
    threshold and are cut into wave-sized segments (two extra kernels).  Same answers as the oracle, through the fused
    step, tiled steps and the grad/apply split; and bitwise reproducible."""
    engine, L = fm
    c = next(x for x in CASES if x["name"] == name)
    rng = np.random.default_rng(5)
    n, p, batch = 3000, 400, 1500
    rows = []
    for r in range(n):
        hot = [j for j in (0, 1, 7) if rng.random() < (0.95, 0.6, 0.3)[(0, 1, 7).index(j)]]
        cold = rng.choice(np.arange(8, p), 5, replace=False).tolist()
        rows.append(np.sort(np.array(hot + cold)))
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows).astype(np.uint32)
    val = rng.normal(0, 1, len(col)).astype(np.float32)
    task = "classification" if c["task"] == oracle.CLASSIFICATION else "regression"
    y = util.labels(n, 9, task)
    kw = {k: v for k, v in c.items() if k not in ("name", "solver")}
    P = oracle.params(min_target=float(y.min()), max_target=float(y.max()), batch_mean=(reduce == "mean"), **kw)
    w0, w, v = util.params(p, P.k, 9, fp32=True)
    X = oracle.Matrix(rp, col, val, p)
    mb = (oracle.SgdMinibatch if c["solver"] == "sgd" else oracle.FtrlMinibatch)(P, X, y, w0, w, v.ravel())
    for s in range(6):
        b0 = (s % 2) * batch
        mb.step(b0, b0 + batch)
    solver = L.SOLVER_SGD if c["solver"] == "sgd" else L.SOLVER_FTRL
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    rv = mb.v.reshape(P.k, p)
    outs = []
    for tile, split in ((0, False), (0, False), (400, False), (0, True)):  # fused twice (reproducibility), tiled, grad/apply
        e = engine.Engine(p, task=P.task, solver=solver, num_factor=P.k, l2_w0=P.l2_reg0, l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv,
                          l2_v=P.l2_regv, learn_rate=P.learn_rate, mode=L.MODE_MINIBATCH, batch_rows=batch, tile_rows=tile,
                          batch_reduce=L.REDUCE_MEAN if P.batch_mean else L.REDUCE_SUM)
        e.set_params(w0, w, v)
        for s in range(6):
            if split:
                e.grad(m, s % 2); e.apply(0)
            else:
                e.step(m, s % 2)
        e.sync()
        g0, gw, gv = e.get_params()
        assert util.rel_err(gv, rv) < 3e-5 and util.rel_err(gw, mb.w) < 3e-5 and abs(g0 - mb.w0.value) < 3e-5 * max(1.0, abs(mb.w0.value)), (tile, split)
        outs.append((g0, gw, gv))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


def test_long_lists_inside_a_sparse_tile(fm):
    """Both ingest plans at once: far fewer entries than features (occurring-feature walk) AND a heavy hitter (long list)."""
    engine, L = fm
    rng = np.random.default_rng(17)
    n, p, batch, k = 4000, 20000, 2000, 8
    rows = [np.sort(np.concatenate([[3] if rng.random() < 0.9 else [], rng.choice(np.arange(10, p), 3, replace=False)]).astype(np.int64)) for _ in range(n)]
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows).astype(np.uint32)
    val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = util.labels(n, 17)
    P = oracle.params(k=k, l2_regw=1e-3, l2_regv=1e-3, learn_rate=0.05)
    w0, w, v = util.params(p, k, 17, fp32=True)
    mb = oracle.SgdMinibatch(P, oracle.Matrix(rp, col, val, p), y, w0, w, v.ravel())
    e = engine.Engine(p, num_factor=k, l2_w1=1e-3, l2_v=1e-3, learn_rate=0.05, mode=L.MODE_MINIBATCH, batch_rows=batch)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    for s in range(4):
        mb.step((s % 2) * batch, (s % 2 + 1) * batch)
        e.step(m, s % 2)
    e.sync()
    g0, gw, gv = e.get_params()
    assert util.rel_err(gv, mb.v.reshape(k, p)) < V_RTOL and util.rel_err(gw, mb.w) < V_RTOL and abs(g0 - mb.w0.value) < V_RTOL


@pytest.mark.parametrize("mode,solver", [("minibatch", "sgd_l1"), ("minibatch", "ftrl"), ("sequential", "ftrl"), ("sequential", "tdap")])
def test_checkpoint_resumes_bit_identically(fm, tmp_path, mode, solver):
    """fmx_engine_save / _load carry parameters AND optimizer state: a run interrupted by a save/load equals the
    uninterrupted one bit for bit (the reference's fm.update drops the optimizer state)."""
    engine, L = fm
    n, p, k = 900, 120, 5
    rp, col, val = util.random_csr(n, p, 8, seed=81)
    y = util.labels(n, 81)
    w0, w, v = util.params(p, k, 81)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    kw = dict(num_factor=k, learn_rate=0.05, l2_v=1e-3, l1_w1=1e-3, l1_v=1e-3, mode=L.MODE_MINIBATCH if mode == "minibatch" else L.MODE_SEQUENTIAL,
              batch_rows=128, solver={"sgd_l1": L.SOLVER_SGD, "ftrl": L.SOLVER_FTRL, "tdap": L.SOLVER_TDAP}[solver])
    order = np.arange(1, 601)
    def advance(e, part):
        if mode == "minibatch":
            for s in range(4 * part, 4 * part + 4):
                e.step(m, s % 8)
            e.sync()
        else:
            e.train_order(m, order[300 * part:300 * (part + 1)])
    a = engine.Engine(p, **kw); a.set_params(w0, w, v); advance(a, 0); advance(a, 1)
    b = engine.Engine(p, **kw); b.set_params(w0, w, v); advance(b, 0)
    path = tmp_path / "ck.fmx"
    b.save(path)
    c = engine.Engine(p, **kw); c.load(path); advance(c, 1)
    pa, pc = a.get_params(), c.get_params()
    assert pa[0] == pc[0] and np.array_equal(pa[1], pc[1]) and np.array_equal(pa[2], pc[2])
    with pytest.raises(L.FmxError, match="does not match"):
        engine.Engine(p + 1, **kw).load(path)


@pytest.mark.parametrize("wide", [0, 1], ids=["fp32", "fp64"])
@pytest.mark.parametrize("name", ["sgd_l2_cls", "sgd_l1_cls", "ftrl_l1l2_cls"])
@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_chunked_exchange_equals_the_split_step(fm, name, reduce, wide):
    """exchange_chunks > 1 (the pipelined multi-GPU step): forward of the whole step, then gradient sums and update one
    block of features at a time.  Bitwise the same as fmx_grad + fmx_apply, on data with heavy hitters, over several
    tiles per step and a truncated step; and equal to the oracle."""
    engine, L = fm
    c = next(x for x in CASES if x["name"] == name)
    rng = np.random.default_rng(11)
    n, p, batch = 2400, 500, 1200
    rows = []
    for r in range(n):
        hot = [j for j, q in ((2, 0.9), (130, 0.5), (499, 0.4)) if rng.random() < q]  # one heavy hitter in each block
        cold = [j for j in rng.choice(p, 6, replace=False).tolist() if j not in (2, 130, 499)]
        rows.append(np.sort(np.array(hot + cold)))
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows).astype(np.uint32)
    val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = util.labels(n, 11, "classification")
    kw = {k: v for k, v in c.items() if k not in ("name", "solver")}
    P = oracle.params(min_target=float(y.min()), max_target=float(y.max()), batch_mean=(reduce == "mean"), **kw)
    w0, w, v = util.params(p, P.k, 11, fp32=not wide)
    mb = (oracle.SgdMinibatch if c["solver"] == "sgd" else oracle.FtrlMinibatch)(P, oracle.Matrix(rp, col, val, p), y, w0, w, v.ravel())
    steps = [(0, 0), (1, 0), (0, 700), (1, 0)]  # (batch, rows_limit)
    for b, lim in steps:
        mb.step(b * batch, b * batch + (lim or batch))
    solver = L.SOLVER_SGD if c["solver"] == "sgd" else L.SOLVER_FTRL
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    base = dict(task=P.task, solver=solver, num_factor=P.k, l2_w0=P.l2_reg0, l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv, l2_v=P.l2_regv,
                learn_rate=P.learn_rate, mode=L.MODE_MINIBATCH, batch_rows=batch, tile_rows=500, state_fp64=wide,
                batch_reduce=L.REDUCE_MEAN if P.batch_mean else L.REDUCE_SUM)
    ref = engine.Engine(p, **base); ref.set_params(w0, w, v)
    for b, lim in steps:
        ref.grad(m, b, lim); ref.apply(0)
    ref.sync()
    r0, rw, rv = ref.get_params()
    tol = 1e-10 if wide else 3e-5
    assert util.rel_err(rv, mb.v.reshape(P.k, p)) < tol and util.rel_err(rw, mb.w) < tol and abs(r0 - mb.w0.value) < tol * max(1.0, abs(mb.w0.value))
    for chunks in (2, 3, 7):
        e = engine.Engine(p, exchange_chunks=chunks, **base); e.set_params(w0, w, v)
        nc, feats, elems, tail = e.grad_layout()
        assert feats % 64 == 0 and nc == -(-p // feats) and tail == nc * elems and e.grad_buffer()[1] == tail + 4
        for b, lim in steps:
            e.grad_begin(m, b, lim)
            for ch in range(nc):
                e.grad_chunk(m, ch)
            for ch in range(nc):
                e.apply_chunk(ch, 0, ch == nc - 1)
        e.sync()
        g0, gw, gv = e.get_params()
        assert g0 == r0 and np.array_equal(gw, rw) and np.array_equal(gv, rv), chunks
        # the unchunked calls work on the blocked layout as well
        e2 = engine.Engine(p, exchange_chunks=chunks, **base); e2.set_params(w0, w, v)
        for b, lim in steps:
            e2.grad(m, b, lim); e2.apply(0)
        e2.sync()
        h0, hw, hv = e2.get_params()
        assert h0 == r0 and np.array_equal(hw, rw) and np.array_equal(hv, rv), chunks
    with pytest.raises(L.FmxError, match="fmx_grad_begin"):
        engine.Engine(p, exchange_chunks=2, **base).grad_chunk(m, 0)


@pytest.mark.parametrize("task", ["regression", "classification"])
@pytest.mark.parametrize("flags", [(True, True), (False, True), (True, False)], ids=["w0+w", "w only", "w0 only"])
def test_mcmc_learner_matches_oracle(fm, task, flags):
    """MCMC_Learner (do_sample, do_multilevel): alpha ~ Gamma, w0 ~ N, w_lambda ~ Gamma, w_mu ~ N, w_i ~ N(mean, var) with
    the caller's pre-drawn standard variates; CLASSIFICATION subtracts truncated normals drawn from libc rand() row by row
    (same seed => same stream in the oracle and in the engine's host step)."""
    engine, L = fm
    n, p, k = 1100, 60, 3
    k0, k1 = flags
    rp, col, val = util.random_csr(n, p, 6, seed=71, empty_rows=True)
    y = util.labels(n, 71, task)
    w0, w, v = util.params(p, k, 71, stdev=0.2, fp32=False)
    iters = 6
    rng = np.random.default_rng(71)
    a1, a2 = oracle.mcmc_draw_shapes(n, p)
    G = np.stack([rng.gamma(a1, 1.0, iters), rng.gamma(a2, 1.0, iters)], 1)
    Z = rng.normal(0, 1, (iters, 2 + p))
    ot = oracle.REGRESSION if task == "regression" else oracle.CLASSIFICATION
    P = oracle.params(task=ot, k=k, k0=k0, k1=k1, l2_reg0=0.05, min_target=float(y.min()), max_target=float(y.max()))
    X = oracle.Matrix(rp, col, val, p)
    r0, rw, rv, rst = oracle.mcmc_learn(P, X, y, w0, w, v.ravel(), iters, G, Z, seed=123)
    e = engine.Engine(p, task=L.TASK_REGRESSION if task == "regression" else L.TASK_CLASSIFICATION, solver=L.SOLVER_MCMC, num_factor=k,
                      keep_w0=int(k0), keep_w1=int(k1), l2_w0=0.05, mode=L.MODE_SEQUENTIAL, min_target=float(y.min()), max_target=float(y.max()))
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    oracle.lib().fmo_srand(123)  # libc rand() is process-wide: the engine's host step continues from this seed
    st = e.mcmc_train(m, iters, G, Z)
    g0, gw, gv = e.get_params()
    assert abs(g0 - r0) < 1e-9 and util.rel_err(gw, rw) < 1e-9 and np.array_equal(gv, v)
    np.testing.assert_allclose(st, rst, rtol=1e-9)
    if task == "classification":  # predictions of an MCMC model go through the probit table
        prob = e.predict(m, L.LINK_PROBIT)
        np.testing.assert_allclose(prob, oracle.predict_batch(P, X, r0, rw, rv, prob="probit"), rtol=0, atol=1e-9)
        assert abs(e.evaluate(m, L.EVAL_LL) - oracle.evaluate(ot, oracle.LL, prob, y)) < 1e-9
    with pytest.raises(L.FmxError, match="fmx_mcmc_train"):
        e.train(m, 10)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2]: k = 64, FTRL (l1 + l2), ~30 nnz/row -- the configuration's own shape, reduced in rows and
# features to what the oracle walks in seconds (reference: solver/FTRL_Learner.h:64-202).
def _configs2_problem(n=2400, p=4000, seed=64):
    rp, col, val = util.random_csr(n, p, 30, seed=seed, empty_rows=False, max_nnz=64)
    y = util.labels(n, seed)
    P = oracle.params(task=oracle.CLASSIFICATION, k=64, l1_regw=1e-4, l1_regv=1e-4, l2_regw=1e-4, l2_regv=1e-4, alpha_w=0.1, alpha_v=0.1, beta_w=1.0, beta_v=1.0)
    w0, w, v = util.params(p, 64, seed, stdev=0.01, fp32=True)   # V0 ~ N(0, 0.01)
    return rp, col, val, y, P, w0, w, v


@pytest.mark.parametrize("reduce", ["mean", "sum"])
@pytest.mark.parametrize("wide", [0, 1], ids=["fp32", "fp64"])
def test_configs2_ftrl_k64_minibatch_matches_oracle(fm, wide, reduce):
    """Mini-batch FTRL at k = 64 (256-byte fp32 rows, 16 lanes per list) against the oracle's fp64 restatement: 1e-5 on V
    with fp32 state, 1e-11 with the reference's fp64 state; 8 steps of 300 rows, wrapping once."""
    engine, L = fm
    rp, col, val, y, P, w0, w, v = _configs2_problem()
    n, p, B = len(rp) - 1, 4000, 400
    P.batch_mean = int(reduce == "mean")
    mb = oracle.FtrlMinibatch(P, oracle.Matrix(rp, col, val, p), y, w0, w, v.ravel())
    e = engine.Engine(p, task=P.task, solver=L.SOLVER_FTRL, num_factor=64, l1_w1=P.l1_regw, l1_v=P.l1_regv, l2_w1=P.l2_regw, l2_v=P.l2_regv,
                      mode=L.MODE_MINIBATCH, batch_rows=B, state_fp64=wide, batch_reduce=L.REDUCE_MEAN if P.batch_mean else L.REDUCE_SUM)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    nb = n // B
    for s in range(8):
        b0 = (s % nb) * B
        mb.step(b0, b0 + B)
        e.step(m, s % nb)
    e.sync()
    g0, gw, gv = e.get_params()
    tol = 1e-11 if wide else V_RTOL
    rv = mb.v.reshape(64, p)
    assert util.rel_err(gv, rv) < tol and util.rel_err(gw, mb.w) < tol and abs(g0 - mb.w0.value) < tol * max(1.0, abs(mb.w0.value))
    if wide:
        assert np.array_equal(gv == 0.0, rv == 0.0)  # the l1 threshold zeroes the same coordinates
    out = e.predict(m)
    refp = oracle.predict_batch(P, oracle.Matrix(rp, col, val, p), mb.w0.value, mb.w, mb.v)
    big = np.abs(refp) > 1e-4  # sign bit-exact wherever the prediction is not within rounding of zero
    assert np.array_equal(np.sign(out[big]), np.sign(refp[big]))


def test_configs2_ftrl_k64_sequential_matches_oracle(fm):
    """The reference's FTRL learner itself at k = 64 (one lane per factor fills the wave): 1e-11 against the oracle over a
    pass and a half, prediction signs bit-exact."""
    engine, L = fm
    rp, col, val, y, P, w0, w, v = _configs2_problem(n=1500)
    n, p = len(rp) - 1, 4000
    X = oracle.Matrix(rp, col, val, p)
    iters = n + n // 2
    ref = oracle.ftrl_learn(P, X, y, w0, w, v.ravel(), iters)
    e = engine.Engine(p, task=P.task, solver=L.SOLVER_FTRL, num_factor=64, l1_w1=P.l1_regw, l1_v=P.l1_regv, l2_w1=P.l2_regw, l2_v=P.l2_regv, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    assert e.train(m, iters) == iters
    g0, gw, gv = e.get_params()
    assert util.rel_err(gv, ref["v"].reshape(64, p)) < 1e-11 and util.rel_err(gw, ref["w"]) < 1e-11 and abs(g0 - ref["w0"]) < 1e-11
    assert np.array_equal(np.sign(e.predict(m)), np.sign(oracle.predict_batch(P, X, ref["w0"], ref["w"], ref["v"])))


# ---------------------------------------------------------------------------------------------------------------------
# Mini-batch TDAP (the reference's DEFAULT solver, R/fm_solver_control.R:22; solver/TDAP_Learner.h:79-233) -- defined in the
# oracle (fmo_tdap_apply_sums), equal to the reference's example step at batch size 1 where the shipped z_w[position]
# indexing (A-6) is invisible.
def _tdap_problem(task, n=1500, p=260, k=6, seed=91, heavy=False):
    rng = np.random.default_rng(seed)
    rp, col, val = util.random_csr(n, p, 9, seed=seed)
    if heavy:  # two heavy hitters: long lists
        rows = []
        for r in range(n):
            hot = [j for j, q in ((1, 0.9), (40, 0.5)) if rng.random() < q]
            rows.append(np.unique(np.concatenate([hot, col[rp[r]:rp[r + 1]]])).astype(np.uint32))
        rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
        col = np.concatenate(rows); val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = util.labels(n, seed, "classification" if task == oracle.CLASSIFICATION else "regression")
    P = oracle.params(task=task, k=k, l1_regw=1e-3, l1_regv=5e-4, l2_regw=1e-2, l2_regv=1e-2, alpha_w=0.1, alpha_v=0.05, gamma=3e-3,
                      min_target=float(y.min()), max_target=float(y.max()))
    w0, w, v = util.params(p, k, seed, fp32=True)
    return rp, col, val, y, P, w0, w, v


def _tdap_engine(fm, p, P, **kw):
    engine, L = fm
    return engine.Engine(p, task=P.task, solver=L.SOLVER_TDAP, num_factor=P.k, keep_w0=P.k0, keep_w1=P.k1, l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv,
                         l2_v=P.l2_regv, alpha_w=P.alpha_w, alpha_v=P.alpha_v, gamma=P.gamma, min_target=P.min_target, max_target=P.max_target,
                         batch_reduce=L.REDUCE_MEAN if P.batch_mean else L.REDUCE_SUM, **kw)


@pytest.mark.parametrize("wide", [0, 1], ids=["fp32", "fp64"])
@pytest.mark.parametrize("reduce", ["mean", "sum"])
@pytest.mark.parametrize("task", [oracle.CLASSIFICATION, oracle.REGRESSION])
def test_minibatch_tdap_matches_oracle(fm, task, reduce, wide):
    engine, L = fm
    rp, col, val, y, P, w0, w, v = _tdap_problem(task, heavy=True)
    n, p = len(rp) - 1, 260
    P.batch_mean = int(reduce == "mean")
    X = oracle.Matrix(rp, col, val, p)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    tol = 1e-11 if wide else 2e-5
    for B, tile, split in ((257, 0, False), (500, 200, False), (500, 0, True)):   # fused; 3 tiles per step; grad / apply through the buffer
        mb = oracle.TdapMinibatch(P, X, y, w0, w, v.ravel())
        e = _tdap_engine(fm, p, P, mode=L.MODE_MINIBATCH, batch_rows=B, tile_rows=tile, state_fp64=wide)
        e.set_params(w0, w, v)
        nb = -(-n // B)
        for s in range(nb + 2):
            b = s % nb
            mb.step(b * B, min((b + 1) * B, n))
            if split:
                e.grad(m, b); e.apply(0)
            else:
                e.step(m, b)
        e.sync()
        g0, gw, gv = e.get_params()
        assert util.rel_err(gv, mb.v.reshape(P.k, p)) < tol and util.rel_err(gw, mb.w) < tol and abs(g0 - mb.w0.value) < tol * max(1.0, abs(mb.w0.value)), (B, tile, split)


def test_minibatch_tdap_at_batch_one_is_the_reference_learner(fm):
    """fp64 state, one row per step, linear term off (so that the shipped z_w[position] read, A-6, has nothing to read):
    the mini-batch kernels reproduce the reference's TDAP learner (the oracle's sequential restatement) to 1e-12."""
    engine, L = fm
    rp, col, val, y, P, w0, w, v = _tdap_problem(oracle.CLASSIFICATION, n=300)
    P.k1 = 0
    n, p = len(rp) - 1, 260
    ref = oracle.tdap_learn(P, oracle.Matrix(rp, col, val, p), y, w0, w, v.ravel(), n, order=np.arange(n))
    e = _tdap_engine(fm, p, P, mode=L.MODE_MINIBATCH, batch_rows=1, state_fp64=1)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    e.train(m, n)
    g0, gw, gv = e.get_params()
    assert util.rel_err(gv, ref["v"].reshape(P.k, p)) < 1e-12 and abs(g0 - ref["w0"]) < 1e-12 and np.array_equal(gw, ref["w"])


# ---------------------------------------------------------------------------------------------------------------------
# ALS / MCMC sweeps on matrices whose columns do NOT come one per field
def _numpy_grouped_vsweep(k, p, rp, col, val, v, err, group_of, alpha=1.0):
    """The approximate sweep restated: per factor q = X v_f; groups ascending; every feature of a group takes its exact
    coordinate step (MCMC_ALS_Learner.h:303-334) against the (q, e) of the group's start, the corrections (:341-350) are
    merged afterwards.  v: [k][p]."""
    n = len(rp) - 1
    v = v.copy(); e = err.copy()
    rows_of = np.repeat(np.arange(n), np.diff(rp))
    x = val.astype(np.float64); xx = (val * val).astype(np.float64)   # x*x is a float product in the reference (:314)
    for f in range(k):
        q = np.zeros(n); np.add.at(q, rows_of, x * v[f, col])
        for g in range(int(group_of.max()) + 1):
            sel = np.flatnonzero(group_of[col] == g)
            if len(sel) == 0:
                continue
            c = col[sel]; r = rows_of[sel]
            h = x[sel] * q[r] - xx[sel] * v[f, c]
            mean = np.zeros(p); var = np.zeros(p)
            np.add.at(mean, c, h * e[r]); np.add.at(var, c, h * h)
            feats = np.unique(c)
            mean[feats] -= v[f, feats] * var[feats]
            vv = 1.0 / (alpha * var[feats])
            new = -vv * (alpha * mean[feats])
            new = np.where(np.isfinite(vv), new, 0.0)
            diff = np.zeros(p); diff[feats] = v[f, feats] - new
            v[f, feats] = new
            np.subtract.at(q, r, x[sel] * diff[c]); np.subtract.at(e, r, h * diff[c])
    return v, e


def test_als_approximate_grouped_sweep(fm):
    """cfg.als_max_levels: a matrix with i.i.d. columns needs hundreds of levels even at this size; with the cap the sweep
    runs in groups (largest position of a feature in its rows), every group against one snapshot.  Checked against a numpy
    restatement of exactly that; and on one-column-per-field data the grouped sweep IS the exact one."""
    engine, L = fm
    rng = np.random.default_rng(12)
    n, p, z, k = 3000, 500, 8, 3
    cols = np.sort(np.stack([rng.choice(p, z, replace=False) for _ in range(n)]), axis=1)
    rp = np.arange(n + 1, dtype=np.int64) * z
    col = cols.astype(np.uint32).ravel(); val = rng.normal(0, 1, n * z).astype(np.float32)
    y = util.labels(n, 12, "regression")
    w0, w, v = util.params(p, k, 12, stdev=0.3, fp32=False)
    P = oracle.params(task=oracle.REGRESSION, k=k)
    err0 = oracle.predict_batch(P, oracle.Matrix(rp, col, val, p), w0, w, v.ravel()) - y
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    exact = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
    levels, _, approx, _ = exact.als_plan(m)
    assert not approx and levels > 4 * z                      # a deep chain
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=16)
    e.set_params(w0, w, v)
    m2 = engine.Matrix.from_csr(rp, col, val, p, y)
    groups, _, approx, group_of = e.als_plan(m2)
    want = np.zeros(p, np.int64); np.maximum.at(want, col.astype(np.int64), np.tile(np.arange(z), n))
    assert approx and groups == z and np.array_equal(group_of, want)
    gerr = e.als_vsweep(m2, err0)
    rv, rerr = _numpy_grouped_vsweep(k, p, rp, col.astype(np.int64), val, v, err0, group_of.astype(np.int64))
    assert util.rel_err(e.get_params()[2], rv) < 1e-9 and util.rel_err(gerr, rerr) < 1e-9
    assert np.sum(gerr ** 2) < np.sum(err0 ** 2)              # still a descent step on this data
    # field-structured data: groups == exact levels, both forms give the same sweep
    fields, width = 6, 50
    pf = fields * width
    colf = (rng.integers(0, width, (n, fields)) + np.arange(fields)[None, :] * width).astype(np.uint32).ravel()
    rpf = np.arange(n + 1, dtype=np.int64) * fields
    valf = rng.normal(0, 1, n * fields).astype(np.float32)
    w0f, wf, vf = util.params(pf, k, 13, stdev=0.3, fp32=False)
    errf = oracle.predict_batch(P, oracle.Matrix(rpf, colf, valf, pf), w0f, wf, vf.ravel()) - y
    outs = []
    for cap in (0, 3):                                         # exact; approximate (6 levels > 3)
        ee = engine.Engine(pf, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=cap)
        ee.set_params(w0f, wf, vf)
        mm = engine.Matrix.from_csr(rpf, colf, valf, pf, y)
        assert ee.als_plan(mm)[2] == (cap > 0)
        outs.append((ee.als_vsweep(mm, errf), ee.get_params()[2]))
    assert util.rel_err(outs[1][1], outs[0][1]) < 1e-12 and util.rel_err(outs[1][0], outs[0][0]) < 1e-12


def test_als_approximate_sweep_against_a_dense_least_squares_restatement(fm):
    """An independent pin for the grouped (approximate) sweep: nothing entry-wise, no CSC walk -- per factor and group, every feature
    of the group takes the minimiser of its own one-dimensional regularised least-squares problem against the group's residual
    snapshot (np.linalg.lstsq on the stacked system [sqrt(alpha) h; sqrt(lambda)] d = -[sqrt(alpha) e; sqrt(lambda) (v - mu)], with
    h = d y_hat / d v_jf = x_j (q - x_j v_jf) from DENSE matrices), then the snapshot moves by the linearised corrections
    (q += X_G d, e += H d).  The statement of the method (solver/MCMC_ALS_Learner.h:200-268 with one "thread" per feature of a
    group), not a transcription of the kernel."""
    engine, L = fm
    rng = np.random.default_rng(33)
    n, p, z, k = 400, 60, 5, 3
    cols = np.sort(np.stack([rng.choice(p, z, replace=False) for _ in range(n)]), axis=1)
    rp = np.arange(n + 1, dtype=np.int64) * z
    col = cols.astype(np.uint32).ravel(); val = rng.normal(0, 1, n * z).astype(np.float32)
    y = util.labels(n, 33, "regression")
    w0, w, v = util.params(p, k, 33, stdev=0.3, fp32=False)
    alpha, lam, mu = 1.3, np.array([0.5, 0.0, 2.0]), np.array([0.05, -0.1, 0.0])
    Xd = np.zeros((n, p)); Xd[np.repeat(np.arange(n), z), col.astype(np.int64)] = val.astype(np.float64)
    e_res = (w0 + Xd @ w + 0.5 * (((Xd @ v.T) ** 2).sum(1) - ((Xd ** 2) @ (v.T ** 2)).sum(1))) - y    # e = y_hat - y, dense
    eng = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=4)
    eng.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    groups, _, approx, group_of = eng.als_plan(m)
    assert approx and groups == z
    got_err = eng.als_vsweep(m, e_res, alpha=alpha, v_lambda=lam, v_mu=mu)
    got_v = eng.get_params()[2]
    V = v.copy(); e = e_res.copy()
    for f in range(k):
        q = Xd @ V[f]
        for g in range(groups):
            G = np.flatnonzero(group_of == g)
            G = G[(Xd[:, G] != 0).any(0)]                      # features that never occur keep their value (an empty column: no step)
            H = Xd[:, G] * q[:, None] - (Xd[:, G].astype(np.float32) ** 2).astype(np.float64) * V[f, G][None, :]   # x*x is a float product (:314)
            d = np.zeros(len(G))
            for a_, j in enumerate(G):
                A = np.concatenate([np.sqrt(alpha) * H[:, a_], [np.sqrt(lam[f])]])[:, None]
                b = -np.concatenate([np.sqrt(alpha) * e, [np.sqrt(lam[f]) * (V[f, j] - mu[f])]])
                d[a_] = np.linalg.lstsq(A, b, rcond=None)[0][0]
            V[f, G] += d
            q += Xd[:, G] @ d
            e += H @ d
    assert util.rel_err(got_v, V) < 1e-9 and util.rel_err(got_err, e) < 1e-9
    assert np.any(np.abs(got_v - v) > 1e-3)


def test_als_heavy_columns_match_oracle(fm):
    """A feature that occurs in (almost) every row has a column of thousands of entries: it is swept by a whole workgroup
    (als_sweep_k) instead of one wave; exact schedule, against the oracle."""
    engine, L = fm
    rng = np.random.default_rng(14)
    n, p, k = 9000, 120, 3
    rows = [np.unique(np.concatenate([[0] if rng.random() < 0.97 else [], [1] if rng.random() < 0.6 else [], rng.choice(np.arange(2, p), 4, replace=False)])) for _ in range(n)]
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows).astype(np.uint32); val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = util.labels(n, 14, "regression")
    w0, w, v = util.params(p, k, 14, stdev=0.3, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=k)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    rv, rerr, _ = oracle.als_update_v(k, X, v.ravel(), err0)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    gerr = e.als_vsweep(m, err0)
    assert util.rel_err(e.get_params()[2], rv.reshape(k, p)) < 1e-10 and util.rel_err(gerr, rerr) < 1e-10
    # the approximate form on the same matrix: the two heavy features are stepped first, one by one, the rest in position groups
    ea = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=4)
    ea.set_params(w0, w, v)
    ma = engine.Matrix.from_csr(rp, col, val, p, y)
    groups, _, approx, group_of = ea.als_plan(ma)
    assert approx and group_of[0] == 0 and group_of[1] == 1 and group_of[2:].min() >= 2
    aerr = ea.als_vsweep(ma, err0)
    nv, nerr = _numpy_grouped_vsweep(k, p, rp, col.astype(np.int64), val, v, err0, group_of.astype(np.int64))
    assert util.rel_err(ea.get_params()[2], nv) < 1e-9 and util.rel_err(aerr, nerr) < 1e-9 and np.sum(aerr ** 2) < np.sum(err0 ** 2)
    r0, rw, rvv = oracle.als_learn(P, X, y, w0, w, v.ravel(), 2, with_v=True)     # the learner's w sweep takes the same path
    e2 = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
    e2.set_params(w0, w, v)
    e2.als_train(m, 2, with_v=True)
    g0, gw, gv = e2.get_params()
    assert abs(g0 - r0) < 1e-10 and util.rel_err(gw, rw) < 1e-10 and util.rel_err(gv, rvv.reshape(k, p)) < 1e-10


def test_mcmc_v_hyper_priors_and_the_tracker_loop(fm):
    """f-4 completed: (1) update_v_lambda + update_v_mu (MCMC_ALS_Learner.h:448-517; shipped v(f, attr_group[i]) indexing kept)
    against the oracle, MCMC draws and ALS means; one full Gibbs step over V as the commented-out block of update_all would run
    it: hyper-priors, then the sweep with them.  (2) the MCMC chain continued call by call (fmx_mcmc_train_from) equals the
    one-shot chain, which is what the tracker block of learn() (:96-125) needs: evaluate, one iteration, evaluate, ..."""
    engine, L = fm
    n, fields, width, k = 1500, 5, 30, 4
    p = fields * width
    rng = np.random.default_rng(91)
    rp = np.arange(n + 1, dtype=np.int64) * fields
    col = (rng.integers(0, width, (n, fields)) + np.arange(fields)[None, :] * width).astype(np.uint32).ravel()
    val = rng.normal(0, 1, n * fields).astype(np.float32)
    y = util.labels(n, 91, "regression")
    w0, w, v = util.params(p, k, 91, stdev=0.3, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=k, mode=L.MODE_SEQUENTIAL, min_target=float(y.min()), max_target=float(y.max()))
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    lam0 = np.full(k, 0.5); mu0 = np.linspace(-0.1, 0.1, k)
    g = rng.gamma((2.0 + p) / 2.0, 1.0, k); z = rng.normal(0, 1, k)
    for sample in (True, False):
        rl, rm = oracle.mcmc_v_hyper(k, p, v.ravel(), g, z, lam0, mu0, sample=sample)
        gl, gm = e.mcmc_v_hyper(lam0, mu0, g, z) if sample else e.mcmc_v_hyper(lam0, mu0)
        np.testing.assert_allclose(gl, rl, rtol=1e-11); np.testing.assert_allclose(gm, rm, rtol=1e-11, atol=1e-15)
    # a Gibbs step over V: hyper-priors, then the sweep under them
    P = oracle.params(task=oracle.REGRESSION, k=k)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    zn = rng.normal(0, 1, (k, p))
    rl, rm = oracle.mcmc_v_hyper(k, p, v.ravel(), g, z, lam0, mu0)
    rv, rerr, _ = oracle.als_update_v(k, X, v.ravel(), err0, alpha=0.8, v_lambda=rl, v_mu=rm, znorm=zn.ravel())
    gl, gm = e.mcmc_v_hyper(lam0, mu0, g, z)
    gerr = e.als_vsweep(m, err0, alpha=0.8, v_lambda=gl, v_mu=gm, std_normals=zn)
    assert util.rel_err(e.get_params()[2], rv.reshape(k, p)) < 1e-10 and util.rel_err(gerr, rerr) < 1e-10
    # (2) the chain in pieces, with an evaluation before every iteration (Tracker: metric of the model at the START of the iteration)
    iters = 5
    a1, a2 = oracle.mcmc_draw_shapes(n, p)
    G = np.stack([rng.gamma(a1, 1.0, iters), rng.gamma(a2, 1.0, iters)], 1)
    Z = rng.normal(0, 1, (iters, 2 + p))
    one = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=k, mode=L.MODE_SEQUENTIAL, min_target=float(y.min()), max_target=float(y.max()))
    one.set_params(w0, w, v)
    st_one = one.mcmc_train(m, iters, G, Z)
    step = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=k, mode=L.MODE_SEQUENTIAL, min_target=float(y.min()), max_target=float(y.max()))
    step.set_params(w0, w, v)
    trace = [step.evaluate(m, L.EVAL_RMSE)]
    st = step.mcmc_train(m, 1, G[:1], Z[:1])
    for it in range(1, iters):
        trace.append(step.evaluate(m, L.EVAL_RMSE))
        st = step.mcmc_train_from(m, 1, G[it:it + 1], Z[it:it + 1], st)
    a, b = one.get_params(), step.get_params()
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and st == st_one
    r0, rw, rvv, rst = oracle.mcmc_learn(P, X, y, w0, w, v.ravel(), iters, G, Z)
    np.testing.assert_allclose(st, rst, rtol=1e-9)
    # the first record is the start model's clamped RMSE (:103-111), from the oracle's forward
    yh = np.clip(oracle.predict_batch(P, X, w0, w, v.ravel()), y.min(), y.max())
    assert abs(trace[0] - oracle.evaluate(oracle.REGRESSION, oracle.RMSE, yh, y)) < 1e-9 and len(trace) == iters


@pytest.mark.parametrize("law", ["iid", "ragged", "fields", "heavy"])
def test_als_levels_by_frontier_walk_equal_the_relaxation_and_a_host_restatement(fm, law):
    """The level of a feature (its place in the exact ALS schedule) = the longest chain of row predecessors.  The frontier
    walk (default), the round-1 relaxation (FMX_ALS_LEVELS=relax) and a plain host loop must agree on every feature,
    including features that never occur (level 0), empty rows and rows of different lengths."""
    import os
    engine, L = fm
    rng = np.random.default_rng({"iid": 1, "ragged": 2, "fields": 3, "heavy": 4}[law])
    n, p = 4000, 700
    if law == "heavy":       # columns of more than 16 384 entries are walked by the whole grid (level_heavy_k): features 3 and 40 sit in every row
        n = 18000
        cols = [np.unique(np.concatenate([[3, 40], rng.choice(p - 50, 5, replace=False)])) for _ in range(n)]
    elif law == "fields":
        fields, width = 7, 100
        cols = [np.sort(rng.integers(0, width, fields) + np.arange(fields) * width) for _ in range(n)]
    else:
        lens = np.full(n, 9) if law == "iid" else rng.integers(0, 14, n)
        cols = [np.sort(rng.choice(p - 50, int(q), replace=False)) for q in lens]      # the last 50 features never occur
    rp = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int64)
    col = np.concatenate(cols).astype(np.uint32)
    val = rng.normal(0, 1, len(col)).astype(np.float32)
    want = np.zeros(p, np.int64)
    order = np.argsort(col, kind="stable")                 # features ascending: a feature's predecessors are all smaller
    row_of = np.repeat(np.arange(n), np.diff(rp))
    run = np.full(n, -1, np.int64)                         # level of the row's latest placed entry
    for j in np.unique(col):
        rows = row_of[order[np.searchsorted(col[order], j, "left"):np.searchsorted(col[order], j, "right")]]
        want[j] = run[rows].max() + 1
        run[rows] = want[j]
    got = {}
    for form in ("frontier", "relax"):
        os.environ["FMX_ALS_LEVELS"] = form
        try:
            m = engine.Matrix.from_csr(rp, col, val, p, np.zeros(n, np.float32))
            e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=2, mode=L.MODE_SEQUENTIAL)
            levels, _, approx, lof = e.als_plan(m)
            got[form] = (levels, approx, lof.astype(np.int64))
            e.close(); m.close()
        finally:
            os.environ.pop("FMX_ALS_LEVELS", None)
    assert not got["frontier"][1] and not got["relax"][1]
    assert np.array_equal(got["frontier"][2], want) and np.array_equal(got["relax"][2], want)
    assert got["frontier"][0] == got["relax"][0] == want.max() + 1


@pytest.mark.parametrize("wide", [0, 1], ids=["fp32", "fp64"])
@pytest.mark.parametrize("one_hot", [False, True], ids=["values", "one-hot"])
@pytest.mark.parametrize("reduce", ["mean", "sum"])
@pytest.mark.parametrize("name", ["sgd_l2_cls", "sgd_l1_cls", "ftrl_l1l2_cls"])
def test_sparse_tiles_with_lists_of_a_few_entries(fm, name, reduce, one_hot, wide):
    """The Criteo regime in small: fewer entries per step than features (sparse directory), but the features that occur are
    drawn from a small pool, so their lists hold 2..16 entries on average -- the lean list-by-list form of phase 2 WITH the
    list's first entry inline in the directory (its S row gathered beside the V row) -- plus one feature in almost every row
    (a long list inside the sparse tile) and a ragged last step.  fp32 state against the fp64 oracle at 1e-5, fp64 state at 1e-11."""
    engine, L = fm
    c = next(x for x in CASES if x["name"] == name)
    rng = np.random.default_rng(sum(map(ord, name)) + int(one_hot))
    n, p, batch, z = 7000, 50000, 3000, 12                       # 36 000 entries per step < 50 000 features
    pool = np.sort(rng.choice(np.arange(50, p), 5000, replace=False))
    rows = []
    for _ in range(n):
        own = rng.choice(pool, z - 1, replace=False)
        rows.append(np.sort(np.concatenate([[7] if rng.random() < 0.95 else [pool[0]], own])).astype(np.int64))
        rows[-1] = np.unique(rows[-1])
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows).astype(np.uint32)
    val = np.ones(len(col), np.float32) if one_hot else rng.normal(0, 1, len(col)).astype(np.float32)
    seed = 31
    y = util.labels(n, seed, "classification")
    kw = {k: v for k, v in c.items() if k not in ("name", "solver")}
    P = oracle.params(min_target=float(y.min()), max_target=float(y.max()), **kw)
    P.batch_mean = int(reduce == "mean")
    if reduce == "sum":
        P.learn_rate = P.learn_rate / 40.0                       # feature 7 sums ~2 850 gradients per step
    w0, w, v = util.params(p, P.k, seed, fp32=True)
    X = oracle.Matrix(rp, col, val, p)
    mb = (oracle.SgdMinibatch if c["solver"] == "sgd" else oracle.FtrlMinibatch)(P, X, y, w0, w, v.ravel())
    e = _engine(fm, p, P, L.SOLVER_SGD if c["solver"] == "sgd" else L.SOLVER_FTRL, L.MODE_MINIBATCH, batch_rows=batch, state_fp64=wide)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    tol = 1e-11 if wide else V_RTOL
    info = e.compact_info(m)
    assert info[2], "the tiles of this matrix must be sparse (fewer entries than features)"
    lists = e.compact_count(m, 0)
    assert 2 * lists <= batch * z < 16 * lists, (lists, batch * z)   # the regime this test is about
    nb = -(-n // batch)
    for s in range(nb + 2):
        b = s % nb
        mb.step(b * batch, min((b + 1) * batch, n))
        e.step(m, b)
    e.sync()
    g0, gw, gv = e.get_params()
    rv = mb.v.reshape(P.k, p)
    assert util.rel_err(gv, rv) < tol, util.rel_err(gv, rv)
    assert util.rel_err(gw, mb.w) < tol and abs(g0 - mb.w0.value) < tol * max(1.0, abs(mb.w0.value))
    touched = np.zeros(p, bool); touched[col] = True
    assert np.array_equal(gv[:, ~touched], v[:, ~touched].astype(np.float32).astype(np.float64))   # features that never occur are not written


def test_als_very_long_columns_are_split_over_workgroups(fm):
    """Columns of more than 65 536 entries (two features that sit in almost every one of 80 000 rows) are cut into segments swept by
    one workgroup each (als_vh_*_k): V sweep and the full ALS learner (w0, w and V sweeps) against the oracle, and against the
    one-workgroup-per-column form (FMX_ALS_SPLIT=0)."""
    import os
    engine, L = fm
    rng = np.random.default_rng(15)
    n, p, k = 80000, 60, 2
    a = rng.random(n) < 0.97
    b = rng.random(n) < 0.9
    others = rng.integers(2, p, (n, 3))
    rows = [np.unique(np.concatenate([[0] if a[i] else [], [1] if b[i] else [], others[i]])) for i in range(n)]
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows).astype(np.uint32); val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = util.labels(n, 15, "regression")
    w0, w, v = util.params(p, k, 15, stdev=0.3, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=k)
    assert np.bincount(col, minlength=p)[:2].min() > 65536
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    rv, rerr, _ = oracle.als_update_v(k, X, v.ravel(), err0)
    r0, rw, rvv = oracle.als_learn(P, X, y, w0, w, v.ravel(), 2, with_v=True)
    got = {}
    for split in ("1", "0"):
        os.environ["FMX_ALS_SPLIT"] = split
        try:
            m = engine.Matrix.from_csr(rp, col, val, p, y)
            e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
            e.set_params(w0, w, v)
            gerr = e.als_vsweep(m, err0)
            gv = e.get_params()[2]
            e2 = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
            e2.set_params(w0, w, v)
            e2.als_train(m, 2, with_v=True)
            got[split] = (gv, gerr, e2.get_params())
        finally:
            os.environ.pop("FMX_ALS_SPLIT", None)
    for split in ("1", "0"):
        gv, gerr, (g0, gw, gvv) = got[split]
        assert util.rel_err(gv, rv.reshape(k, p)) < 1e-10 and util.rel_err(gerr, rerr) < 1e-10, split
        assert abs(g0 - r0) < 1e-10 and util.rel_err(gw, rw) < 1e-10 and util.rel_err(gvv, rvv.reshape(k, p)) < 1e-10, split


_SCHEDULE_CHILD = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
P, Z, K, N, B = 300_000, 12, int(sys.argv[1]), 3 * 70_000 + 999, 70_000
m = engine.Matrix.synthetic_iid(N, P, Z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (K, P)).astype(np.float32).astype(np.float64)
e = engine.Engine(P, num_factor=K, solver=L.SOLVER_SGD, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
e.set_params(0.0, None, v0)
nb = e.num_batches(m)
for s in range(30):          # more wide launches than the 14 the tuner times
    e.step(m, s % nb)
e.sync()
w0, w, v = e.get_params()
yh = e.predict(m, L.LINK_LOGISTIC)
print("HASH", hashlib.sha256(v.tobytes() + w.tobytes() + np.float64(w0).tobytes() + yh.tobytes()).hexdigest(), e.rows_tune()[0])
'''


@pytest.mark.parametrize("k", [16, 64])
def test_phase_1_schedules_do_not_change_a_bit(k):
    """Phase 1 issues a row's gathers in one of two schedules -- one entry's requests at a time, or four entries in flight per lane group
    (fm_batch_kernels.hip: RowsTune; the engine times both on its first wide launches and keeps the faster) -- and both add the row's
    terms in the same order.  Separate processes with FMX_ROWS_SERIAL pinned either way, and one left to the engine's own choice:
    identical parameters and predictions after 30 steps of 70 000 rows."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = {}
    for name, env in (("chosen", {}), ("serial", {"FMX_ROWS_SERIAL": "1"}), ("four", {"FMX_ROWS_SERIAL": "0"})):
        r = subprocess.run([sys.executable, "-c", _SCHEDULE_CHILD, str(k)], env=dict(os.environ, **env), cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("HASH")][0].split()
        seen[name] = line[1]
        if name == "chosen":
            assert int(line[2]) in (0, 1)      # decided after 14 timed launches
    assert len(set(seen.values())) == 1, seen
