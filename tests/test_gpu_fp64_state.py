"""MINIBATCH mode with cfg.state_fp64: the same kernels as the fp32 path instantiated on the sequential mode's fp64
tables.  With fp64 state the engine must reproduce the oracle's mini-batch restatement to rounding error (1e-11), not
just to the north-star 1e-5: this pins the LOGIC of the shared kernel templates (tiles, exchange buffer, long lists,
sparse walk, every update kind) far below the noise floor of the fp32 state.
"""
import numpy as np
import pytest

import oracle
from tests import util
from tests.test_gpu_train import CASES, _problem

pytestmark = pytest.mark.gpu

TIGHT = 1e-11


@pytest.fixture(scope="module")
def fm():
    from fmwr_amd import engine, _lib
    return engine, _lib


def _kw(L, P, solver, batch, **extra):
    kw = dict(task=P.task, solver=L.SOLVER_SGD if solver == "sgd" else L.SOLVER_FTRL, num_factor=P.k, keep_w0=P.k0, keep_w1=P.k1,
              l2_w0=P.l2_reg0, l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv, l2_v=P.l2_regv, learn_rate=P.learn_rate,
              alpha_w=P.alpha_w, alpha_v=P.alpha_v, beta_w=P.beta_w, beta_v=P.beta_v, mode=L.MODE_MINIBATCH, batch_rows=batch,
              min_target=P.min_target, max_target=P.max_target, batch_reduce=L.REDUCE_MEAN if P.batch_mean else L.REDUCE_SUM, state_fp64=1)
    kw.update(extra)
    return kw


def _close(e, mb, k, p, tol=TIGHT):
    g0, gw, gv = e.get_params()
    rv = mb.v.reshape(k, p)
    ev, ew, e0 = util.rel_err(gv, rv), (util.rel_err(gw, mb.w) if np.max(np.abs(mb.w)) > 0 else float(np.max(np.abs(gw)))), abs(g0 - mb.w0.value)
    assert ev < tol and ew < tol and e0 < tol * max(1.0, abs(mb.w0.value)), (ev, ew, e0)
    return g0, gw, gv


def _oracle_mb(c, P, X, y, w0, w, v):
    return (oracle.SgdMinibatch if c["solver"] == "sgd" else oracle.FtrlMinibatch)(P, X, y, w0, w, v.ravel())


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
@pytest.mark.parametrize("batch", [1, 64, 257])
@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_fp64_state_matches_oracle_to_rounding(fm, c, batch, reduce):
    engine, L = fm
    if batch == 1 and c["name"] not in ("sgd_l2_cls", "sgd_l1_cls", "ftrl_l1l2_cls"):
        pytest.skip("batch 1 covered on three cases")
    n, p = (300 if batch == 1 else 1200), 300
    rp, col, val, y, P, seed = _problem(c, n=n)
    P.batch_mean = int(reduce == "mean")
    w0, w, v = util.params(p, P.k, seed, fp32=False)  # full fp64 start values: the state keeps them
    mb = _oracle_mb(c, P, oracle.Matrix(rp, col, val, p), y, w0, w, v)
    e = engine.Engine(p, **_kw(L, P, c["solver"], batch))
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    total = n + n // 2 + 5
    done, step, nb = 0, 0, -(-n // batch)
    while done < total:
        b0 = (step % nb) * batch
        rows = min(batch, n - b0, total - done)
        mb.step(b0, b0 + rows)
        done += rows
        step += 1
    assert e.train(m, total) == total
    g0, gw, gv = _close(e, mb, P.k, p)
    # the forward on the trained fp64 tables
    refp = oracle.predict_batch(P, oracle.Matrix(rp, col, val, p), g0, gw, gv.ravel())
    np.testing.assert_allclose(e.predict(m), refp, rtol=1e-12, atol=1e-12)


def test_fp64_state_batch1_is_the_reference_learner(fm):
    """batch_rows == 1 with fp64 state IS SGD_Learner::learn on the given order: equal to the sequential oracle at 1e-12."""
    engine, L = fm
    for name in ("sgd_l2_cls", "sgd_l1_cls", "ftrl_l1l2_cls"):
        c = next(x for x in CASES if x["name"] == name)
        rp, col, val, y, P, seed = _problem(c, n=200)
        p = 300
        w0, w, v = util.params(p, P.k, seed, fp32=False)
        X = oracle.Matrix(rp, col, val, p)
        learn = oracle.sgd_learn if c["solver"] == "sgd" else oracle.ftrl_learn
        ref = learn(P, X, y, w0, w, v.ravel(), 200, order=np.arange(200))
        for reduce in (L.REDUCE_MEAN, L.REDUCE_SUM):
            e = engine.Engine(p, **_kw(L, P, c["solver"], 1, batch_reduce=reduce))
            e.set_params(w0, w, v)
            e.train(engine.Matrix.from_csr(rp, col, val, p, y), 200)
            g0, gw, gv = e.get_params()
            assert util.rel_err(gv, ref["v"].reshape(P.k, p)) < 1e-12 and util.rel_err(gw, ref["w"]) < 1e-12 and abs(g0 - ref["w0"]) < 1e-12, name


@pytest.mark.parametrize("name", ["sgd_l2_cls", "sgd_l1_cls", "ftrl_l1l2_cls"])
@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_fp64_state_tiles_split_and_long_lists(fm, name, reduce):
    """Heavy-hitter features (long-list kernels) through the fused step, tiled steps and the grad/apply split."""
    engine, L = fm
    c = next(x for x in CASES if x["name"] == name)
    rng = np.random.default_rng(5)
    n, p, batch = 3000, 400, 1500
    rows = []
    for r in range(n):
        hot = [j for j, q in ((0, 0.95), (1, 0.6), (7, 0.3)) if rng.random() < q]
        rows.append(np.sort(np.array(hot + rng.choice(np.arange(8, p), 5, replace=False).tolist())))
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows).astype(np.uint32)
    val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = util.labels(n, 9, "classification")
    kw = {k: v for k, v in c.items() if k not in ("name", "solver")}
    P = oracle.params(min_target=float(y.min()), max_target=float(y.max()), batch_mean=(reduce == "mean"), **kw)
    w0, w, v = util.params(p, P.k, 9, fp32=False)
    mb = _oracle_mb(c, P, oracle.Matrix(rp, col, val, p), y, w0, w, v)
    for s in range(6):
        mb.step((s % 2) * batch, (s % 2 + 1) * batch)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    outs = []
    for tile, split in ((0, False), (0, False), (400, False), (0, True), (400, True)):
        e = engine.Engine(p, **_kw(L, P, c["solver"], batch, tile_rows=tile))
        e.set_params(w0, w, v)
        for s in range(6):
            if split:
                e.grad(m, s % 2); e.apply(0)
            else:
                e.step(m, s % 2)
        e.sync()
        outs.append(_close(e, mb, P.k, p, 1e-10))  # sums of ~1400 terms in a different (fixed) association
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert e.grad_elem_bytes() == 8
    has_q = c["solver"] == "ftrl" and reduce == "sum"
    kp = 2
    while kp < P.k:
        kp *= 2
    assert e.grad_buffer()[1] == p * kp * (2 if has_q else 1) + p * (3 if has_q else 2) + 4


@pytest.mark.parametrize("name", ["sgd_l2_cls", "sgd_l1_cls", "ftrl_l1l2_cls"])
def test_fp64_state_sparse_tiles(fm, name):
    engine, L = fm
    c = next(x for x in CASES if x["name"] == name)
    n, p, batch = 700, 6000, 100
    rp, col, val, y, P, seed = _problem(c, n=n, p=p, mean_nnz=6)
    w0, w, v = util.params(p, P.k, seed, fp32=False)
    mb = _oracle_mb(c, P, oracle.Matrix(rp, col, val, p), y, w0, w, v)
    for s in range(10):
        mb.step((s % 7) * batch, (s % 7 + 1) * batch)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    e = engine.Engine(p, **_kw(L, P, c["solver"], batch)); e.set_params(w0, w, v)
    for s in range(10):
        e.step(m, s % 7)
    e.sync()
    _close(e, mb, P.k, p)


def test_fp64_state_odd_k_and_wide_k(fm):
    """k = 3 (padded to 4 doubles, 2 lanes), k = 70 (128 doubles, 64 lanes: a whole wave per row) and k = 0."""
    engine, L = fm
    for k in (3, 70, 0, 1):
        c = dict(name="k%d" % k, solver="sgd", task=oracle.CLASSIFICATION, k=k, l2_regw=1e-3, l2_regv=1e-3, learn_rate=0.05)
        rp, col, val, y, P, seed = _problem(c, n=600)
        p = 300
        w0, w, v = util.params(p, k, seed, fp32=False)
        mb = oracle.SgdMinibatch(P, oracle.Matrix(rp, col, val, p), y, w0, w, v.ravel())
        for s in range(5):
            mb.step((s % 3) * 200, (s % 3 + 1) * 200)
        e = engine.Engine(p, **_kw(L, P, "sgd", 200)); e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        for s in range(5):
            e.step(m, s % 3)
        e.sync()
        g0, gw, gv = e.get_params()
        assert abs(g0 - mb.w0.value) < TIGHT and util.rel_err(gw, mb.w) < TIGHT, k
        if k:
            assert util.rel_err(gv, mb.v.reshape(k, p)) < TIGHT, k


def test_fp64_state_checkpoint_and_shape_guard(fm, tmp_path):
    engine, L = fm
    n, p, k = 900, 120, 16  # k = 16 pads to 16 in both table types: only the header flag tells the checkpoints apart
    rp, col, val = util.random_csr(n, p, 8, seed=81)
    y = util.labels(n, 81)
    w0, w, v = util.params(p, k, 81, fp32=False)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    kw = dict(num_factor=k, l1_w1=1e-3, l1_v=1e-3, mode=L.MODE_MINIBATCH, batch_rows=128, solver=L.SOLVER_FTRL, state_fp64=1)
    def advance(e, part):
        for s in range(4 * part, 4 * part + 4):
            e.step(m, s % 8)
        e.sync()
    a = engine.Engine(p, **kw); a.set_params(w0, w, v); advance(a, 0); advance(a, 1)
    b = engine.Engine(p, **kw); b.set_params(w0, w, v); advance(b, 0)
    path = tmp_path / "ck64.fmx"
    b.save(path)
    c = engine.Engine(p, **kw); c.load(path); advance(c, 1)
    pa, pc = a.get_params(), c.get_params()
    assert pa[0] == pc[0] and np.array_equal(pa[1], pc[1]) and np.array_equal(pa[2], pc[2])
    with pytest.raises(L.FmxError, match="does not match"):
        engine.Engine(p, **dict(kw, state_fp64=0)).load(path)
    # get_params returns the stored doubles unchanged
    f = engine.Engine(p, **kw); f.set_params(w0, w, v)
    q = f.get_params()
    assert q[0] == w0 and np.array_equal(q[1], w) and np.array_equal(q[2], v)


def test_fp32_state_tracks_fp64_state(fm):
    """The default fp32 state against the fp64 state on the same run: the whole difference is storage rounding."""
    engine, L = fm
    c = CASES[0]
    rp, col, val, y, P, seed = _problem(c, n=4000, p=500, mean_nnz=12)
    p = 500
    w0, w, v = util.params(p, P.k, seed, fp32=True)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    out = []
    for wide in (0, 1):
        e = engine.Engine(p, **_kw(L, P, "sgd", 512, state_fp64=wide)); e.set_params(w0, w, v)
        e.train(m, 12000)
        out.append(e.get_params())
    assert util.rel_err(out[0][2], out[1][2]) < 1e-5 and util.rel_err(out[0][1], out[1][1]) < 1e-5

