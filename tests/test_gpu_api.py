"""The reference's user-facing surface (fm.train / fm.update / predict.FM mirrored in fmwr_amd.api) on the GPU,
including BASELINE.json configs[0]: MovieLens-100K-shaped, k=8, SGD (shape-matched synthetic: the data set itself is
not available offline)."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from tests import util

pytestmark = pytest.mark.gpu


def _movielens_shaped(n=100_000, users=943, items=1682, seed=0):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, users, n); i = rng.integers(0, items, n) + users
    rating = rng.integers(1, 6, n).astype(np.float64)
    X = sp.csr_matrix((np.ones(2 * n), np.stack([u, i], 1).ravel(), np.arange(0, 2 * n + 1, 2)), shape=(n, users + items))
    return X, rating


@pytest.mark.parametrize("mode", ["sequential", "sequential_bitwise"])
@pytest.mark.parametrize("task", ["REGRESSION", "CLASSIFICATION"])
def test_config0_movielens_shaped_sgd(task, mode):
    """configs[0] at full size through the mirror of the R API, in both reference-order forms: "sequential" (the default: SGD's forward sum reassociated,
    cfg.seq_reassociate) and "sequential_bitwise" (the reference's association).  p = 2625 and two entries per row: nearly every example shares a feature with
    one of its neighbours, so the reassociated learner's waves hand the parameters to each other through the done tags for the whole run."""
    import fmwr_amd as fm
    X, rating = _movielens_shaped()
    y = rating if task == "REGRESSION" else (rating >= 4).astype(np.float64)  # {0,1} labels -> {-1,+1} (R/fm_train.R:112-122)
    data = fm.fm_matrix(X, y)
    ctl = [fm.model_control(task, **{"factor.number": 8, "L2.w1": 1e-3, "L2.v": 1e-3, "v.init_stdev": 0.05}),
           fm.solver_control(max_iter=100_000, solver=fm.SGD_solver(learn_rate=0.02))]
    fit = fm.fm_train(data, normalize=False, control=ctl, seed=42, mode=mode)
    # oracle run from the same V0
    k, p, n = 8, X.shape[1], X.shape[0]
    v0 = np.random.default_rng(42).normal(0.0, 0.05, (k, p))
    yy = y if task == "REGRESSION" else np.where(y < 1, -1.0, 1.0)
    P = oracle.params(task=oracle.REGRESSION if task == "REGRESSION" else oracle.CLASSIFICATION, k=k, l2_regw=1e-3, l2_regv=1e-3,
                      learn_rate=0.02, min_target=float(yy.min()), max_target=float(yy.max()))
    Xo = oracle.Matrix(X.indptr, X.indices, X.data, p)
    ref = oracle.sgd_learn(P, Xo, yy.astype(np.float32), 0.0, np.zeros(p), v0.ravel(), 100_000)
    rv = ref["v"].reshape(k, p)
    assert np.max(np.abs(fit["Model"]["v"] - rv)) <= 1e-5 * np.max(np.abs(rv))       # north_star: 1e-5 relative on V
    assert np.max(np.abs(fit["Model"]["v"] - rv)) <= (1e-10 if mode == "sequential" else 1e-13) * np.max(np.abs(rv))
    assert abs(fit["Model"]["w0"] - ref["w0"]) < 1e-10
    pred = fm.predict(fit, data, normalize=False)
    raw = oracle.predict_batch(P, Xo, ref["w0"], ref["w"], ref["v"])
    want = np.clip(raw, yy.min(), yy.max()) if task == "REGRESSION" else 1.0 / (1.0 + np.exp(-raw))
    np.testing.assert_allclose(pred, want, rtol=1e-9, atol=1e-12)
    if task == "CLASSIFICATION":
        assert np.array_equal(pred >= 0.5, raw >= 0)  # prediction sign bit-exact
    assert fit["Scales"]["target.range"] == (float(yy.min()), float(yy.max()))


def test_update_warm_start_and_errors():
    import fmwr_amd as fm
    X, rating = _movielens_shaped(n=5000, users=50, items=80, seed=3)
    data = fm.fm_matrix(X, rating)
    ctl = [fm.model_control("REGRESSION", **{"factor.number": 4, "L2.v": 1e-3}), fm.solver_control(max_iter=4000, solver=fm.FTRL_solver(alpha_v=0.05))]
    fit = fm.fm_train(data, normalize=False, control=ctl, seed=1)
    fit2 = fm.fm_update(fit, data, normalize=False, max_iter=3000)
    # oracle: two learn() calls, parameters carried over, z/n reset
    p, k = X.shape[1], 4
    v0 = np.random.default_rng(1).normal(0.0, 0.01, (k, p))
    P = oracle.params(task=oracle.REGRESSION, k=k, l2_regv=1e-3, alpha_v=0.05, min_target=1.0, max_target=5.0)
    Xo = oracle.Matrix(X.indptr, X.indices, X.data, p)
    r1 = oracle.ftrl_learn(P, Xo, rating.astype(np.float32), 0.0, np.zeros(p), v0.ravel(), 4000)
    r2 = oracle.ftrl_learn(P, Xo, rating.astype(np.float32), r1["w0"], r1["w"], r1["v"], 3000)
    assert np.max(np.abs(fit2["Model"]["v"] - r2["v"].reshape(k, p))) <= 1e-10 * np.max(np.abs(r2["v"]))
    # error behaviour mirrored from R/fm_train.R, R/fm_predict.R
    with pytest.raises(ValueError, match="there are no labels in data"):
        fm.fm_train(fm.fm_matrix(X))
    with pytest.raises(ValueError, match="target should have two levels"):
        fm.fm_train(fm.fm_matrix(X, rating), control=[fm.model_control("CLASSIFICATION")])
    with pytest.raises(ValueError, match="newdata is null"):
        fm.predict(fit)
    with pytest.raises(ValueError, match="not the same"):
        fm.fm_update(fit, fm.fm_matrix(X[:, :100], rating), normalize=False)
    with pytest.warns(UserWarning, match="maximum number of iteratorions"):
        assert fm.solver_control(solver=fm.MCMC_solver())["max_iter"] == 100
    assert fm.solver_control()["solver"]["solver"] == "TDAP"  # the reference's default (R/fm_solver_control.R:22)


def test_minibatch_mode_through_api_learns():
    """Mini-batch engine behind the same API: the loss goes down on a learnable problem."""
    import fmwr_amd as fm
    rng = np.random.default_rng(5)
    n, p, k = 20000, 200, 4
    X = sp.random(n, p, density=0.05, format="csr", random_state=5, data_rvs=lambda s: rng.normal(0, 1, s))
    vt = rng.normal(0, 0.3, (k, p)); wt = rng.normal(0, 0.3, p)
    Xd = X.toarray()
    score = Xd @ wt + 0.5 * (((Xd @ vt.T) ** 2).sum(1) - ((Xd ** 2) @ (vt.T ** 2)).sum(1))
    y = (score > 0).astype(np.float64)
    data = fm.fm_matrix(X, y)
    ctl = [fm.model_control("CLASSIFICATION", **{"factor.number": k, "v.init_stdev": 0.1}), fm.solver_control(max_iter=20 * n, solver=fm.SGD_solver(learn_rate=0.05))]
    fit = fm.fm_train(data, normalize=False, control=ctl, seed=0, mode="minibatch", batch_rows=256)
    acc = np.mean((fm.predict(fit, data, normalize=False) >= 0.5) == (y > 0))
    assert acc > 0.85, acc
    # the same run on fp64 state: the same model up to the fp32 storage rounding
    fit64 = fm.fm_train(data, normalize=False, control=ctl, seed=0, mode="minibatch_fp64", batch_rows=256)
    v32, v64 = np.asarray(fit["Model"]["v"]), np.asarray(fit64["Model"]["v"])
    assert np.max(np.abs(v32 - v64)) < 1e-4 * np.max(np.abs(v64))
    upd = fm.fm_update(fit64, data, normalize=False, max_iter=n)  # the engine choice travels with the fit
    assert upd["engine"]["mode"] == "minibatch_fp64"


def test_track_and_select_through_the_api():
    """track.control(step_size > 0) -> Trace; fm.track on new data; fm.select picks the best snapshot (R/fm_select.R)."""
    import fmwr_amd as fm
    rng = np.random.default_rng(8)
    n, p, k = 4000, 60, 3
    X = sp.random(n, p, density=0.15, format="csr", random_state=8, data_rvs=lambda s: rng.normal(0, 1, s))
    vt = rng.normal(0, 0.4, (k, p)); Xd = X.toarray()
    score = 0.5 * (((Xd @ vt.T) ** 2).sum(1) - ((Xd ** 2) @ (vt.T ** 2)).sum(1)) + Xd @ rng.normal(0, 0.5, p)
    y = (score > np.median(score)).astype(np.float64)
    train, test = fm.fm_matrix(X[:3000], y[:3000]), fm.fm_matrix(X[3000:], y[3000:])
    ctl = [fm.model_control("CLASSIFICATION", **{"factor.number": k, "v.init_stdev": 0.1}),
           fm.solver_control(max_iter=9000, solver=fm.SGD_solver(learn_rate=0.02)), fm.track_control(step_size=1000, evaluate_metric="LL", convergence=0.0)]
    fit = fm.fm_train(train, normalize=False, control=ctl, seed=4)
    assert list(fit["Trace"]["trace"][0]) == [0, 1000, 2000, 3000, 4000, 5000, 6000, 7000, 8000, 8999]
    assert len(fit["Trace"]["trace"]) == 11 and len(fit["Trace"]["evaluation.train"]) == 10
    ll = fit["Trace"]["evaluation.train"]
    assert ll[-1] > ll[0]  # log-likelihood improves
    tr = fm.fm_track(fit, newdata=test, evaluate_metric="LL")
    assert tr["trace.train"] is not None and len(tr["trace.test"]) == 10
    acc = fm.fm_track(fit, data=train, newdata=test, evaluate_metric="ACC")
    # the oracle's metric on the last snapshot
    snap = fit["Trace"]["trace"][-1]
    P = oracle.params(k=k)
    Xo = oracle.Matrix(X[3000:].indptr, X[3000:].indices, X[3000:].data, p)
    prob = oracle.predict_batch(P, Xo, snap["w0"], snap["w"], snap["v"].ravel(), prob=True)
    yt = np.where(y[3000:] < 1, -1.0, 1.0).astype(np.float32)
    assert abs(acc["trace.test"][-1] - oracle.evaluate(oracle.CLASSIFICATION, oracle.ACC, prob, yt)) < 1e-12
    assert abs(tr["trace.test"][-1] - oracle.evaluate(oracle.CLASSIFICATION, oracle.LL, prob, yt)) < 1e-9
    best = fm.fm_select(fit, trace=tr)
    bi = int(np.argmax(tr["trace.test"]))
    assert np.array_equal(best["Model"]["v"], fit["Trace"]["trace"][bi + 1]["v"])
    with pytest.raises(ValueError, match="evaluate.metric is error"):
        fm.fm_track(fit, newdata=test, evaluate_metric="RMSE")
    with pytest.raises(ValueError, match="trace is missing"):
        fm.fm_select(fit)


def test_normalize_true_scales_columns_like_the_reference():
    """fm.train(normalize=TRUE): SMatrix::scales on the stored entries (util/Smatrix.h:98-135), Scales returned, predict
    applies them (SMatrix::normalize); all against the oracle's restatement."""
    import fmwr_amd as fm
    rng = np.random.default_rng(12)
    n, p, k = 1500, 40, 3
    X = sp.random(n, p, density=0.2, format="csr", random_state=12, data_rvs=lambda s: rng.normal(2.0, 3.0, s))
    X.sort_indices()
    y = rng.normal(0, 1, n)
    data = fm.fm_matrix(X, y)
    ctl = [fm.model_control("REGRESSION", **{"factor.number": k, "L2.v": 1e-3}), fm.solver_control(max_iter=2500, solver=fm.SGD_solver(learn_rate=0.01))]
    fit = fm.fm_train(data, normalize=True, control=ctl, seed=6)
    sval, mean, std = oracle.scales(n, p, X.indices, X.data, np.arange(p))
    np.testing.assert_array_equal(fit["Scales"]["mean"], mean)
    np.testing.assert_array_equal(fit["Scales"]["std"], std)
    v0 = np.random.default_rng(6).normal(0.0, 0.01, (k, p))
    P = oracle.params(task=oracle.REGRESSION, k=k, l2_regv=1e-3, learn_rate=0.01, min_target=float(y.min()), max_target=float(y.max()))  # FM.cpp:89-90: min/max of the R (double) labels
    Xo = oracle.Matrix(X.indptr, X.indices, sval, p)
    ref = oracle.sgd_learn(P, Xo, y.astype(np.float32), 0.0, np.zeros(p), v0.ravel(), 2500)
    assert np.max(np.abs(fit["Model"]["v"] - ref["v"].reshape(k, p))) <= 1e-10 * np.max(np.abs(ref["v"]))
    # predict on new data with the model's Scales
    Xn = sp.random(300, p, density=0.2, format="csr", random_state=13, data_rvs=lambda s: rng.normal(2.0, 3.0, s)); Xn.sort_indices()
    pred = fm.predict(fit, fm.fm_matrix(Xn))
    nval = oracle.normalize(Xn.indices, Xn.data, mean, std)
    raw = oracle.predict_batch(P, oracle.Matrix(Xn.indptr, Xn.indices, nval, p), ref["w0"], ref["w"], ref["v"])
    np.testing.assert_allclose(pred, np.clip(raw, P.min_target, P.max_target), rtol=1e-9, atol=1e-11)
    # a subset of columns, and fm.update following the model's settings
    fit2 = fm.fm_train(data, normalize=[3, 7, 8], control=ctl, seed=6)
    _, mean2, std2 = oracle.scales(n, p, X.indices, X.data, [3, 7, 8])
    np.testing.assert_array_equal(fit2["Scales"]["mean"], mean2)
    fit3 = fm.fm_update(fit2, data, max_iter=500)
    np.testing.assert_array_equal(fit3["Scales"]["std"], std2)
    with pytest.raises(ValueError, match="different from those in previously saved model"):
        fm.fm_update(fit2, data, normalize=[1, 2])


def test_als_solver_through_the_api():
    import fmwr_amd as fm
    rng = np.random.default_rng(15)
    n, p, k = 2000, 50, 2
    X = sp.random(n, p, density=0.1, format="csr", random_state=15, data_rvs=lambda s: rng.normal(0, 1, s)); X.sort_indices()
    y = X @ rng.normal(0, 1, p) + rng.normal(0, 0.1, n)
    data = fm.fm_matrix(X, y)
    with pytest.warns(UserWarning, match="maximum number of iteratorions"):
        sc = fm.solver_control(max_iter=500, solver=fm.ALS_solver())
    assert sc["max_iter"] == 100
    fit = fm.fm_train(data, normalize=False, control=[fm.model_control("REGRESSION", **{"factor.number": k}), fm.solver_control(max_iter=5, solver=fm.ALS_solver())], seed=2)
    v0 = np.random.default_rng(2).normal(0.0, 0.01, (k, p))
    P = oracle.params(task=oracle.REGRESSION, k=k)
    r0, rw, rv = oracle.als_learn(P, oracle.Matrix(X.indptr, X.indices, X.data, p), y.astype(np.float32), 0.0, np.zeros(p), v0.ravel(), 5)
    assert abs(fit["Model"]["w0"] - r0) < 1e-10 and np.max(np.abs(fit["Model"]["w"] - rw)) < 1e-9
    pred = fm.predict(fit, data, normalize=False)
    assert np.mean((pred - y) ** 2) < 0.1 * np.var(y)  # the linear part is recovered


def test_als_classification_through_the_api_uses_the_probit_tables():
    """ALS on CLASSIFICATION: residual -/+ dnorm/(1-pnorm) through the table (MCMC_ALS_Learner.h:545-559), predictions
    through fast_pnorm (Model::predict_prob, core/Model.h:166-171)."""
    import fmwr_amd as fm
    rng = np.random.default_rng(23)
    n, p, k = 3000, 60, 2
    X = sp.random(n, p, density=0.1, format="csr", random_state=23, data_rvs=lambda s: rng.normal(0, 1, s)); X.sort_indices()
    y = np.where(X @ rng.normal(0, 1, p) + 0.2 > 0, 1.0, 0.0)  # 0/1 labels: the API maps them to -1/+1 (R/fm_train.R:60-69)
    data = fm.fm_matrix(X, y)
    ctl = [fm.model_control("CLASSIFICATION", **{"factor.number": k}), fm.solver_control(max_iter=8, solver=fm.ALS_solver())]
    fit = fm.fm_train(data, normalize=False, control=ctl, seed=3)
    v0 = np.random.default_rng(3).normal(0.0, 0.01, (k, p))
    P = oracle.params(task=oracle.CLASSIFICATION, k=k)
    Xo = oracle.Matrix(X.indptr, X.indices, X.data, p)
    r0, rw, rv = oracle.als_learn(P, Xo, np.where(y > 0, 1.0, -1.0).astype(np.float32), 0.0, np.zeros(p), v0.ravel(), 8)
    assert abs(fit["Model"]["w0"] - r0) < 1e-10 and np.max(np.abs(fit["Model"]["w"] - rw)) < 1e-9
    prob = fm.predict(fit, data, normalize=False)
    ref = oracle.predict_batch(P, Xo, r0, rw, rv, prob="probit")
    np.testing.assert_allclose(prob, ref, rtol=0, atol=1e-9)
    assert np.mean((prob >= 0.5) == (y > 0)) > 0.93


def test_mcmc_solver_through_the_api():
    """MCMC.solver: the chain's variates come from the generator seeded by fm_train(seed=) in the reference's call order;
    the same variates fed to the oracle's restatement give the same chain."""
    import fmwr_amd as fm
    from fmwr_amd.api import _mcmc_draws
    rng = np.random.default_rng(31)
    n, p, k = 2500, 40, 2
    X = sp.random(n, p, density=0.15, format="csr", random_state=31, data_rvs=lambda s: rng.normal(0, 1, s)); X.sort_indices()
    wt = rng.normal(0, 1, p)
    y = X @ wt + 0.5 + rng.normal(0, 0.2, n)
    data = fm.fm_matrix(X, y)
    ctl = [fm.model_control("REGRESSION", **{"factor.number": k}), fm.solver_control(max_iter=25, solver=fm.MCMC_solver())]
    fit = fm.fm_train(data, normalize=False, control=ctl, seed=9)
    g = np.random.default_rng(9)
    v0 = g.normal(0.0, 0.01, (k, p))
    gam, z = _mcmc_draws(g, n, p, 25, True, True)
    P = oracle.params(task=oracle.REGRESSION, k=k, min_target=float(y.min()), max_target=float(y.max()))
    r0, rw, rv, st = oracle.mcmc_learn(P, oracle.Matrix(X.indptr, X.indices, X.data, p), y.astype(np.float32), 0.0, np.zeros(p), v0.ravel(), 25, gam, z)
    assert abs(fit["Model"]["w0"] - r0) < 1e-9 and np.max(np.abs(fit["Model"]["w"] - rw)) < 1e-9
    assert np.max(np.abs(fit["Model"]["w"] - wt)) < 0.1 and abs(fit["Model"]["w0"] - 0.5) < 0.1  # the posterior sits on the truth
    pred = fm.predict(fit, data, normalize=False)
    assert np.mean((pred - y) ** 2) < 0.1 * np.var(y)
    # with track.control the loop's tracker block runs (MCMC_ALS_Learner::learn :96-125): same chain, plus the RMSE of the model
    # at the START of iterations 0, 10, 20 and of the last one, with their snapshots; fm.select can pick from them
    ctl_t = ctl + [fm.track_control(step_size=10, evaluate_metric="RMSE")]
    fit_t = fm.fm_train(data, normalize=False, control=ctl_t, seed=9)
    assert fit_t["Model"]["w0"] == fit["Model"]["w0"] and np.array_equal(fit_t["Model"]["w"], fit["Model"]["w"])
    tr = fit_t["Trace"]
    assert list(tr["trace"][0]) == [0, 10, 20, 24] and len(tr["trace"]) == 5
    ev = tr["evaluation.train"]
    assert ev[0] > ev[1] > ev[-1] * 0.5 and ev[-1] < 0.5 * ev[0]          # the chain moves to the data
    assert tr["trace"][1]["w0"] == 0.0 and np.all(tr["trace"][1]["w"] == 0.0)   # record 0 is the initial model
    r10, rw10, _, _ = oracle.mcmc_learn(P, oracle.Matrix(X.indptr, X.indices, X.data, p), y.astype(np.float32), 0.0, np.zeros(p), v0.ravel(), 10, gam[:10], z[:10])
    assert abs(tr["trace"][2]["w0"] - r10) < 1e-9 and np.max(np.abs(tr["trace"][2]["w"] - rw10)) < 1e-9  # the snapshot before iteration 10


def test_c_abi_argument_checks_for_the_newer_entry_points():
    """Misuse of the step / exchange / MCMC entry points is refused with a message, never a fault."""
    from fmwr_amd import engine, _lib as L
    import ctypes as C
    rng = np.random.default_rng(1)
    n, p = 300, 40
    X = sp.random(n, p, density=0.2, format="csr", random_state=1, data_rvs=lambda s: rng.normal(0, 1, s)); X.sort_indices()
    y = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    m = engine.Matrix.from_csr(X.indptr.astype(np.int64), X.indices.astype(np.uint32), X.data.astype(np.float32), p, y.astype(np.float32))
    seq = engine.Engine(p, num_factor=2, mode=L.MODE_SEQUENTIAL)
    mb = engine.Engine(p, num_factor=2, mode=L.MODE_MINIBATCH, batch_rows=100, exchange_chunks=2)
    with pytest.raises(L.FmxError, match="MINIBATCH"):
        seq.grad_begin(m, 0)
    with pytest.raises(L.FmxError, match="MINIBATCH"):
        seq.grad_layout()
    with pytest.raises(L.FmxError, match="out of range"):
        mb.grad_begin(m, 99)
    mb.grad_begin(m, 0)
    with pytest.raises(L.FmxError, match="out of range"):
        mb.grad_chunk(m, 5)
    with pytest.raises(L.FmxError, match="out of range"):
        mb.apply_chunk(-1, 0, True)
    with pytest.raises(L.FmxError, match="FMX_MODE_SEQUENTIAL"):
        mb.als_train(m, 1)
    mc = engine.Engine(p, num_factor=2, solver=L.SOLVER_MCMC, task=L.TASK_REGRESSION, mode=L.MODE_SEQUENTIAL)
    with pytest.raises(ValueError, match="std_gammas"):
        mc.mcmc_train(m, 2, np.ones((2, 2)), np.zeros((2, 3)))
    assert L.lib().fmx_mcmc_train(mc.h, m.h, C.c_int32(2), None, None, None) == L.ERR_INVALID
    with pytest.raises(L.FmxError, match="unknown link"):
        seq.predict(m, 7)
    with pytest.raises(L.FmxError, match="Unknown solver"):
        engine.Engine(p, solver=400)
    engine.Engine(p, solver=L.SOLVER_TDAP, mode=L.MODE_MINIBATCH).close()   # mini-batch TDAP exists since round 2
    # compact exchange: only for steps of one sparse tile; streamed training: mini-batch engines, one tile per step
    with pytest.raises(L.FmxError, match="sparse"):
        mb.grad_compact(m, 0)                       # 300 x 40 with 20 % density: more entries than features per tile
    with pytest.raises(L.FmxError, match="MINIBATCH"):
        seq.compact_info(m)
    with pytest.raises(L.FmxError, match="MINIBATCH"):
        seq.train_stream(1000, nnz_per_row=3)
    with pytest.raises(L.FmxError, match="nnz_per_row"):
        mb.train_stream(1000, nnz_per_row=0)
    with pytest.raises(L.FmxError, match="features is not correct"):
        mb.train_stream(1000, seed=1, fields=(2, [5, 7], 1.0))   # 2 + 12 features, the engine has 40
    with pytest.raises(L.FmxError, match="out of range"):
        mb.get_rows([p])


def test_failed_plan_build_leaves_no_half_built_cache(monkeypatch):
    """a failure while the per-tile plans are being built (out of memory is the expected one) must not leave a
    cache that the next call takes for valid.  The builder commits to the matrix only at the end; a forced failure is
    followed by a clean retry with the same results as an undisturbed run."""
    from fmwr_amd import _lib as L, engine
    n, p, k = 600, 90, 4
    rp, col, val = util.random_csr(n, p, 6, seed=5)
    y = util.labels(n, 5)
    w0, w, v = util.params(p, k, 5)
    kw = dict(num_factor=k, learn_rate=0.05, l2_v=1e-3, mode=L.MODE_MINIBATCH, batch_rows=128)
    ref = engine.Engine(p, **kw); ref.set_params(w0, w, v)
    mref = engine.Matrix.from_csr(rp, col, val, p, y)
    for s in range(4):
        ref.step(mref, s)
    ref.sync()
    e = engine.Engine(p, **kw); e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    L.check(L.lib().fmx_debug_fail_next_plan_build())
    with pytest.raises(L.FmxError, match="plan build failed"):
        e.step(m, 0)
    with pytest.raises(L.FmxError):      # nothing was cached: asking for a step of a plan-less matrix fails again only if the build does
        L.check(L.lib().fmx_debug_fail_next_plan_build())
        e.num_batches(m)
    for s in range(4):                   # the flag cleared itself: the retry builds everything and trains
        e.step(m, s)
    e.sync()
    a, b = ref.get_params(), e.get_params()
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_host_hand_over_in_pieces():
    """fmx_matrix_from_rlist / _from_csr / fmx_set_params / fmx_get_params copy the caller's arrays as they are, in pinned pieces, and narrow / check /
    prefix-sum them on the device (fm_ingest.hip: ingest_host_arrays, params_to_device): arrays of several pieces -- 20 M entries, a 1 M x 16 model --
    arrive exactly (values narrowed to f32, columns, offsets, labels), a bad column or row size far into the arrays is reported at its position, and
    the parameters make the round trip bit for bit (fp64 state) or as their f32 images."""
    from fmwr_amd import engine, _lib as L
    rng = np.random.default_rng(7)
    n, z, p = 1_000_000, 20, 300_000
    col = np.sort(rng.integers(0, p // z, (n, z)) + (np.arange(z) * (p // z))[None, :], axis=1).astype(np.int32).ravel()
    value = rng.normal(0, 1, n * z)
    sizes = np.full(n, z, np.int32)
    labels = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    m = engine.Matrix.from_rlist(value, col, sizes, p, labels)
    rp, c2, v2, y2 = m.export()
    assert np.array_equal(rp, np.arange(n + 1, dtype=np.int64) * z) and np.array_equal(c2, col.astype(np.uint32))
    assert np.array_equal(v2, value.astype(np.float32)) and np.array_equal(y2, labels.astype(np.float32))
    m2 = engine.Matrix.from_csr(rp, c2, v2, p, y2)           # the CSR entry takes the same path
    r2 = m2.export()
    assert np.array_equal(r2[0], rp) and np.array_equal(r2[1], c2) and np.array_equal(r2[2], v2) and np.array_equal(r2[3], y2)
    bad = col.copy(); bad[17_000_003] = p                      # far into the third piece
    with pytest.raises(L.FmxError, match="out of range at 17000003"):
        engine.Matrix.from_rlist(value, bad, sizes, p, labels)
    bad_sizes = sizes.copy(); bad_sizes[900_001] = -1
    with pytest.raises(L.FmxError, match="negative row_size at row 900001"):
        engine.Matrix.from_rlist(value, col, bad_sizes, p, labels)
    short = sizes.copy(); short[5] = z - 1
    with pytest.raises(L.FmxError, match="row_size is not correct"):
        engine.Matrix.from_rlist(value, col, short, p, labels)
    rp_bad = rp.copy(); rp_bad[700_000] = rp_bad[700_001] + 1
    with pytest.raises(L.FmxError, match="row_ptr decreases at row 700000"):
        engine.Matrix.from_csr(rp_bad, c2, v2, p, y2)
    # the model: 1 M x 16 doubles = 128 MB, three pieces of whole rows
    P, K = 1_000_000, 16
    w = rng.normal(0, 1, P); v = rng.normal(0, 1, (K, P))
    for wide, layout in ((1, "0"), (0, "0"), (0, "1")):
        import os
        os.environ["FMX_W_IN_ROW"] = layout
        try:
            e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH, state_fp64=wide)
        finally:
            del os.environ["FMX_W_IN_ROW"]
        e.set_params(0.25, w, v)
        w0, gw, gv = e.get_params()
        if wide:
            assert w0 == 0.25 and np.array_equal(gw, w) and np.array_equal(gv, v)
        else:
            assert w0 == 0.25 and np.array_equal(gw, w.astype(np.float32).astype(np.float64)) and np.array_equal(gv, v.astype(np.float32).astype(np.float64))
        e.set_params(0.0, None, None)
        w0, gw, gv = e.get_params()
        assert w0 == 0.0 and not gw.any() and not gv.any()
        e.close()


def test_gather_probes_answer_and_refuse():
    """The measurement aids behind bench.py's `ceiling_frac` (fm_measure.hip; nothing on the product path): the uniformly random probe and the one driven
    by a matrix's own column ids return a rate; a skewed matrix's rows are served faster than uniformly random ones from the same table (its heads stay
    on-die), which is why the skewed workload is priced against the matrix-driven probe; bad geometry is refused."""
    from fmwr_amd import engine, _lib as L
    vocab = engine.CRITEO_VOCAB
    p = 13 + sum(vocab)
    m = engine.Matrix.synthetic_fields(100_000, 13, vocab, 3.0, 5)
    skew = engine.measure_gather_matrix(m, 0, 100_000, p, 128, in_flight=4, reps=5)
    flat = engine.measure_gather(p * 128, 128, n_groups=100_000, per_group=40, in_flight=4, reps=5)
    assert skew > 0 and flat > 0 and skew > 1.2 * flat, (skew, flat)
    with pytest.raises(L.FmxError, match="row_bytes"):
        engine.measure_gather_matrix(m, 0, 1000, p, 48)
    with pytest.raises(L.FmxError, match="geometry"):
        engine.measure_gather_matrix(m, 0, 200_000, p, 128)
    with pytest.raises(L.FmxError, match="geometry"):
        engine.measure_gather(1 << 20, 64, n_groups=1000, per_group=30)    # per_group must be a multiple of 8


def test_a_source_keeps_its_engine_alive_and_the_engine_closes_its_sources():
    """fmx_source holds a raw fmx_engine* (fmx_source_close waits on that engine's stream): the Python Source keeps the Engine alive, matrices it
    hands out keep the Source, and Engine.close() closes open sources first (`Engine(...).source(...)`, shutdown order)."""
    import gc
    from fmwr_amd import _lib as L, engine
    p = 3_000
    kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=4, mode=L.MODE_MINIBATCH, batch_rows=512)
    src = engine.Engine(p, **kw).source(2048, nnz_per_row=6, seed=3)     # the engine has no other reference
    gc.collect()
    m = src.next()
    assert m is not None and m.n == 512 and m._source is src
    assert src.close() >= 0.0
    e = engine.Engine(p, **kw)
    s2 = e.source(1024, nnz_per_row=6, seed=4)
    assert s2.next().n == 512
    e.close()                      # closes s2 while the engine still exists
    assert s2.h is None
    with pytest.raises(ValueError):
        s2.next()
    assert s2.close() == 0.0       # idempotent


def test_the_ragged_generator_and_training_on_its_rows_against_the_oracle():
    """fmx_matrix_synthetic_ragged (SURVEY 8(d)'s variant: row lengths Poisson(30) clipped to [1, 64]): lengths in range with the clipped law's mean, rows
    strictly ascending, shard independent; and the hot path on such rows -- forward and three mini-batch steps -- equals the oracle on the exported rows
    (the kernels walk a row in padded rounds of a fixed lane group: ragged lengths exercise every padding count)."""
    from fmwr_amd import _lib as L, engine
    n, p, k = 6_000, 5_000, 16
    m = engine.Matrix.synthetic_ragged(n, p, 30.0, seed=9)
    rp, col, val, y = m.export()
    lens = np.diff(rp)
    assert lens.min() >= 1 and lens.max() <= 64 and abs(lens.mean() - 30.0) < 0.5 and 4.5 < lens.std() < 6.5
    assert len(set(lens.tolist())) > 25
    for i in range(0, n, 97):
        c = col[rp[i]:rp[i + 1]]
        assert np.all(np.diff(c.astype(np.int64)) > 0) and c[-1] < p
    assert np.all(val == 1.0) and set(np.unique(y).tolist()) <= {-1.0, 1.0}
    tail = engine.Matrix.synthetic_ragged(1_000, p, 30.0, seed=9, row_offset=5_000)     # rows 5000.. of the same stream
    rp2, col2, _, y2 = tail.export()
    assert np.array_equal(np.diff(rp2), lens[5_000:]) and np.array_equal(col2, col[rp[5_000]:]) and np.array_equal(y2, y[5_000:])
    with pytest.raises(L.FmxError):
        engine.Matrix.synthetic_ragged(10, p, 30.0, seed=1, min_nnz=5, max_nnz=80)
    w0, w, v = util.params(p, k, 9)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.CLASSIFICATION, k=k, l2_regw=1e-4, l2_regv=1e-4, learn_rate=0.05)
    e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, l2_w1=1e-4, l2_v=1e-4, learn_rate=0.05, mode=L.MODE_MINIBATCH, batch_rows=2_000)
    e.set_params(w0, w, v)
    np.testing.assert_allclose(e.predict(m), oracle.predict_batch(P, X, w0, w, v.ravel()), rtol=0, atol=1e-5)
    e.train(m, n)
    mb = oracle.SgdMinibatch(P, X, y, w0, w, v.ravel())
    for b in range(0, n, 2_000):
        mb.step(b, b + 2_000)
    gw0, gw, gv = e.get_params()
    assert util.rel_err(gv, mb.v.reshape(k, p)) < 1e-5 and util.rel_err(gw, mb.w) < 1e-5 and abs(gw0 - mb.w0.value) < 1e-5


@pytest.mark.parametrize("gen", ["stratified", "iid", "ragged"])
def test_the_value_variant_redraws_real_values_and_the_kernels_read_them(gen):
    """fmx_matrix_synthetic_values ("val = 1.0f (variant: U(0,1))"): every stored value lands in (0, 1), exact in fp32 (23-bit draws + a half), keyed by
    the global row (a shard draws what the whole matrix would), columns and labels untouched; the matrix stops being one-hot, so forward, mini-batch steps and
    the sequential learner run the value-reading kernels -- checked against the oracle on the exported rows (util/Smatrix.h:44-61: real float values)."""
    from fmwr_amd import _lib as L, engine
    n, p, k, z = 6_000, 5_000, 16, 30
    make = {"stratified": lambda nn, off: engine.Matrix.synthetic(nn, p, z, 13, row_offset=off),
            "iid": lambda nn, off: engine.Matrix.synthetic_iid(nn, p, z, 13, row_offset=off),
            "ragged": lambda nn, off: engine.Matrix.synthetic_ragged(nn, p, float(z), 13, row_offset=off)}[gen]
    m = make(n, 0)
    rp0, col0, val0, y0 = m.export()
    assert np.all(val0 == 1.0)
    m.synthetic_values(77)
    rp, col, val, y = m.export()
    assert np.array_equal(rp, rp0) and np.array_equal(col, col0) and np.array_equal(y, y0)
    assert val.min() > 0.0 and val.max() < 1.0 and abs(val.mean() - 0.5) < 0.01 and abs(val.std() - 12 ** -0.5) < 0.01
    assert np.all(val.astype(np.float64) * 8388608.0 - 0.5 == np.floor(val.astype(np.float64) * 8388608.0))   # 23-bit draws, centred: never 0, never 1
    assert len(np.unique(val)) > 0.95 * len(val)
    tail = make(1_000, 5_000).synthetic_values(77, row_offset=5_000)                  # rows 5000.. of the same stream
    assert np.array_equal(tail.export()[2], val[rp[5_000]:])
    again = make(n, 0).synthetic_values(78)
    assert not np.array_equal(again.export()[2], val)
    w0, w, v = util.params(p, k, 9)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.CLASSIFICATION, k=k, l2_regw=1e-4, l2_regv=1e-4, learn_rate=0.05)
    kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, l2_w1=1e-4, l2_v=1e-4, learn_rate=0.05)
    e = engine.Engine(p, mode=L.MODE_MINIBATCH, batch_rows=2_000, **kw)
    e.set_params(w0, w, v)
    np.testing.assert_allclose(e.predict(m), oracle.predict_batch(P, X, w0, w, v.ravel()), rtol=0, atol=1e-5)
    e.train(m, n)
    mb = oracle.SgdMinibatch(P, X, y, w0, w, v.ravel())
    for b in range(0, n, 2_000):
        mb.step(b, b + 2_000)
    gw0, gw, gv = e.get_params()
    assert util.rel_err(gv, mb.v.reshape(k, p)) < 1e-5 and util.rel_err(gw, mb.w) < 1e-5 and abs(gw0 - mb.w0.value) < 1e-5
    es = engine.Engine(p, mode=L.MODE_SEQUENTIAL, **kw)
    es.set_params(w0, w, v)
    es.train(m, 3_000)
    ref = oracle.sgd_learn(P, X, y, w0, w, v.ravel(), 3_000)
    assert util.rel_err(es.get_params()[2], ref["v"].reshape(k, p)) < 1e-10
    with pytest.raises(L.FmxError):
        L.check(L.lib().fmx_matrix_synthetic_values(None, 1, 0))


def test_lane_groups_pulling_rows_changes_no_bit(monkeypatch):
    """Phase 1 on rows of differing lengths, opt-in form (FMX_ROWS_PULL=1, fm_rows_forward_dyn_k): the lane groups of a workgroup pull rows from a counter
    instead of owning one row each.  A row is still walked by one lane group in row order and stored under its own index, the w0 partial sums keep their
    granularity: the static kernel gives the same bits -- forward, S rows and w0 (through three training steps), fp32 and fp64 state, k = 8 (two lanes
    per row), 16 and 64, values and one-hot rows, empty rows included.  (Measured slower than the static kernel: kept as the record)"""
    from fmwr_amd import _lib as L, engine
    n, p = 140_000, 4_000          # >= 512 wide workgroups per launch: the 256-thread forms
    for k, wide, values in ((16, 0, "normal"), (8, 0, "ones"), (16, 1, "normal"), (64, 0, "normal")):
        rp, col, val = util.random_csr(n, p, 12, seed=31 + k, empty_rows=True, values=values)
        y = util.labels(n, 31)
        w0, w, v = util.params(p, k, 31)
        out = []
        monkeypatch.delenv("FMX_ROWS_FLAT", raising=False)     # (the flat form associates differently: tests/test_gpu_flat_rows.py)
        for flag in ("1", "0"):
            monkeypatch.setenv("FMX_ROWS_PULL", flag)
            m = engine.Matrix.from_csr(rp, col, val, p, y)
            e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, l2_w1=1e-4, l2_v=1e-4, learn_rate=0.05, mode=L.MODE_MINIBATCH, batch_rows=n // 2,
                              state_fp64=wide)
            e.set_params(w0, w, v)
            pred = e.predict(m)
            e.train(m, n + n // 2)
            out.append((pred, e.get_params()))
        assert np.array_equal(out[0][0], out[1][0]), (k, wide)
        assert out[0][1][0] == out[1][1][0] and np.array_equal(out[0][1][1], out[1][1][1]) and np.array_equal(out[0][1][2], out[1][1][2]), (k, wide)


@pytest.mark.parametrize("p,n,z", [(3_000, 30_000, 9), (1_000_000, 300_000, 30), (20_000_000, 120_000, 12), (200, 5_000, 6)])
def test_the_hand_written_pair_sort_builds_the_library_sorts_plan(monkeypatch, p, n, z):
    """Tiles without a field layout (ragged rows, i.i.d. columns) are sorted by (column, row) by the library's onesweep sort; FMX_PAIR_SORT=hand takes the hand-written
    LSD sort of the per-field plans run as ONE field [0, p) instead (fm_ingest.hip: plan_build; measured 1.9 x slower on tile-sized sorts, hence opt-in).  Both are
    stable, so the plans are the same and training on them gives the same bits: 12-bit ids (two passes), 20-bit (three), 25-bit (four), 8-bit (one pass, straight
    into the plan); dense and sparse tiles; ragged rows; one-hot rows and rows with real values (sorted as (column, position), rows and values gathered afterwards)."""
    from fmwr_amd import _lib as L, engine
    k = 8
    out = []
    for flag in ("hand", "rocprim"):
        monkeypatch.setenv("FMX_PAIR_SORT", flag)
        m = engine.Matrix.synthetic_ragged(n, p, float(z), seed=5, min_nnz=1, max_nnz=min(64, p))
        e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_FTRL, num_factor=k, l1_w1=1e-4, l2_v=1e-3, mode=L.MODE_MINIBATCH, batch_rows=max(1000, n // 3))
        e.init_normal(3, 0.0, 0.05)
        assert e.train(m, n) == n
        ids = np.arange(0, p, max(1, p // 5000), dtype=np.uint32)
        out.append(e.get_rows(ids))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.any(out[0][1] != 0.0)
    if p > 100_000:
        return
    # real values: (column, position) sorted by hand, rows and values gathered afterwards -- against the library's sort of a 64-bit payload
    rp, col, val = util.random_csr(n, p, z, seed=6, empty_rows=True)
    y = util.labels(n, 6)
    out = []
    for flag in ("hand", "rocprim"):
        monkeypatch.setenv("FMX_PAIR_SORT", flag)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, l2_v=1e-3, learn_rate=0.05, mode=L.MODE_MINIBATCH, batch_rows=max(1000, n // 3))
        e.init_normal(3, 0.0, 0.05)
        assert e.train(m, n) == n
        out.append(e.get_params())
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


def test_a_dgcmatrix_is_transposed_on_the_device():
    """fmx_matrix_from_dgc: the slots of a dgCMatrix (x, 0-based row indices, column pointers) handed over as they lie in R; R/fm_matrix.R:26-33 transposes on the
    host (Matrix::t) and passes the row-major list.  The device transposition gives the matrix fmx_matrix_from_rlist builds from the transposed list -- offsets,
    columns (ascending inside a row), float32 values, labels -- and the same predictions bit for bit; empty rows and empty columns, one-hot values (found on the
    device), a field layout (found on the device), a column whose rows are not ascending (not a valid dgCMatrix, still transposed: the sort is stable), and the
    refusals: row index out of range, decreasing pointers, a pointer total other than nnz."""
    import scipy.sparse as sp
    from fmwr_amd import _lib as L, engine
    rng = np.random.default_rng(5)
    nrow, ncol, k = 30_000, 1_500, 8
    dense_rows = sp.random(nrow, ncol, density=0.004, format="csr", random_state=7, dtype=np.float64)
    dense_rows.data = rng.normal(0, 1, dense_rows.nnz)
    dense_rows[17] = 0; dense_rows[:, 40] = 0; dense_rows.eliminate_zeros()          # an empty row, an empty column
    csr = dense_rows.tocsr(); csr.sort_indices()
    csc = csr.tocsc(); csc.sort_indices()
    y = util.labels(nrow, 5).astype(np.float64)
    a = engine.Matrix.from_dgc(csc.data, csc.indices, csc.indptr, nrow, ncol, labels=y)
    b = engine.Matrix.from_rlist(csr.data, csr.indices, np.diff(csr.indptr), ncol, labels=y)
    ea, eb = a.export(), b.export()
    for u, v in zip(ea, eb):
        assert np.array_equal(u, v)
    assert (a.n, a.p, a.nnz) == (nrow, ncol, csr.nnz)
    w0, w, v = util.params(ncol, k, 5)
    e = engine.Engine(ncol, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.05, mode=L.MODE_MINIBATCH, batch_rows=8_192)
    e.set_params(w0, w, v)
    assert np.array_equal(e.predict(a), e.predict(b))
    e.train(a, nrow)
    pa = e.get_params()
    e.set_params(w0, w, v)
    e.train(b, nrow)
    pb = e.get_params()
    assert pa[0] == pb[0] and np.array_equal(pa[1], pb[1]) and np.array_equal(pa[2], pb[2])
    # one-hot rows with a field layout: three factor columns of a data frame, one dummy column of each per row
    f = np.stack([rng.integers(0, 50, 4_000), 50 + rng.integers(0, 200, 4_000), 250 + rng.integers(0, 30, 4_000)], axis=1)
    oh = sp.csr_matrix((np.ones(12_000), f.ravel(), np.arange(0, 12_001, 3)), shape=(4_000, 280)).tocsc(); oh.sort_indices()
    c = engine.Matrix.from_dgc(oh.data, oh.indices, oh.indptr, 4_000, 280)
    rp, col, val, _ = c.export()
    assert np.array_equal(np.diff(rp), np.full(4_000, 3)) and np.array_equal(col.reshape(-1, 3), f) and np.all(val == 1.0)
    # rows of a column out of order: the transposition does not depend on it
    x2, i2 = csc.data.copy(), csc.indices.copy()
    j = int(np.argmax(np.diff(csc.indptr) >= 3)); s0 = csc.indptr[j]
    x2[s0], x2[s0 + 2] = x2[s0 + 2], x2[s0]; i2[s0], i2[s0 + 2] = i2[s0 + 2], i2[s0]
    d = engine.Matrix.from_dgc(x2, i2, csc.indptr, nrow, ncol)
    for u, v in zip(d.export()[:3], ea[:3]):
        assert np.array_equal(u, v)
    # refusals
    bad_i = csc.indices.copy(); bad_i[5] = nrow
    with pytest.raises(L.FmxError, match="out of range"):
        engine.Matrix.from_dgc(csc.data, bad_i, csc.indptr, nrow, ncol)
    bad_p = csc.indptr.copy(); bad_p[10] = bad_p[9] - 1 if bad_p[9] > 0 else bad_p[11] + 1
    with pytest.raises(L.FmxError, match="decrease"):
        engine.Matrix.from_dgc(csc.data, csc.indices, bad_p, nrow, ncol)
    with pytest.raises(L.FmxError, match="stored entries"):
        engine.Matrix.from_dgc(csc.data[:-1], csc.indices[:-1], csc.indptr, nrow, ncol)
    empty = engine.Matrix.from_dgc(np.zeros(0), np.zeros(0, np.int32), np.zeros(ncol + 1, np.int32), 10, ncol)
    assert (empty.n, empty.nnz) == (10, 0)
    # the R-facing surface: fm.matrix on a column-major sparse matrix keeps its slots, training gives the model of the row-major hand-over
    import fmwr_amd as fm
    small, ys = csr[:3_000], np.where(y[:3_000] > 0, 1.0, 0.0)
    ctl = [fm.model_control("CLASSIFICATION", **{"factor.number": 4}), fm.solver_control(max_iter=2_000, solver=fm.SGD_solver(learn_rate=0.02))]
    da, db = fm.fm_matrix(small.tocsc(), ys), fm.fm_matrix(small.tocsr(), ys)
    assert "col_ptr" in da.features and "row_size" in db.features
    fa = fm.fm_train(da, normalize=False, control=ctl, seed=3, mode="sequential")
    fb = fm.fm_train(db, normalize=False, control=ctl, seed=3, mode="sequential")
    assert np.array_equal(fa["Model"]["v"], fb["Model"]["v"]) and fa["Model"]["w0"] == fb["Model"]["w0"]
    assert np.array_equal(fm.predict(fa, da, normalize=False), fm.predict(fb, db, normalize=False))
