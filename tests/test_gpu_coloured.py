"""cfg.als_max_levels = -1, the COLOURED order of the ALS / Gibbs sweeps: every coordinate step exact, the features visited in (colour, index) order of a proper
colouring of the "share a row" graph instead of the reference's index order (on i.i.d. columns the reference's order is a chain of thousands of dependent levels).
The claim is checked literally: relabel the features in the engine's order, and the ORACLE -- the reference's index-order sweep -- on the relabelled matrix must give
the engine's numbers (1e-10)."""
import numpy as np
import pytest

import oracle
from tests import util

pytestmark = pytest.mark.gpu

K, Z = 8, 30


def _relabel(rp, col, val, rank):
    """the CSR with feature j renamed rank[j], rows re-sorted by the new names"""
    n = len(rp) - 1
    nc = rank[col].astype(np.int64)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    order = np.lexsort((nc, rows))
    return rp.copy(), nc[order].astype(np.uint32), val[order]


@pytest.mark.parametrize("gibbs,values", [(False, "ones"), (True, "normal")])
def test_coloured_sweep_is_the_reference_sweep_on_the_relabelled_matrix(gibbs, values):
    from fmwr_amd import _lib as L, engine
    n, p = 12_000, 4_000
    m0 = engine.Matrix.synthetic_iid(n, p, Z, 91, law=L.COLUMNS_UNIFORM)
    rp, col, val, _ = m0.export(); m0.close()
    if values == "normal":
        val = np.random.default_rng(4).normal(0, 1, len(val)).astype(np.float32)
    y = util.labels(n, 91, "regression")
    w0, w, v = util.params(p, K, 61, stdev=0.1, fp32=False)
    lam = np.linspace(10.0, 20.0, K) if gibbs else np.linspace(0.1, 0.5, K); mu = np.linspace(-0.05, 0.05, K)
    z = np.random.default_rng(15).normal(0, 1, (K, p)) if gibbs else None
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL, als_max_levels=-1)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    levels, largest, _, level_of = e.als_plan(m)
    level_of = level_of.copy()
    assert e.als_plan_kind(m) == 2
    # a proper colouring: no row holds two features of one level; far fewer levels than the reference's order needs
    lv = level_of[col].reshape(n, Z)
    assert all(len(np.unique(r)) == Z for r in lv[:: max(1, n // 500)]) and np.all(np.sort(lv, axis=1)[:, 1:] != np.sort(lv, axis=1)[:, :-1])
    e_exact = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
    m_exact = engine.Matrix.from_csr(rp, col, val, p, y)
    assert levels < e_exact.als_plan(m_exact)[0] / 3
    e_exact.close(); m_exact.close()
    # the engine's order: (colour, index)
    order = np.lexsort((np.arange(p), level_of))
    rank = np.empty(p, np.int64); rank[order] = np.arange(p)
    rp2, col2, val2 = _relabel(rp, col, val, rank)
    X = oracle.Matrix(rp, col, val, p); X2 = oracle.Matrix(rp2, col2, val2, p)
    P = oracle.params(task=oracle.REGRESSION, k=K)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    v2 = v[:, order]                                        # row f of V in the new names
    z2 = z[:, order] if gibbs else None
    rv2, rerr, _ = oracle.als_update_v(K, X2, np.ascontiguousarray(v2).ravel(), err0, alpha=1.1, v_lambda=lam, v_mu=mu, znorm=np.ascontiguousarray(z2).ravel() if gibbs else None)
    rv = np.empty_like(v); rv[:, order] = rv2.reshape(K, p)
    gerr = e.als_vsweep(m, err0, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
    assert util.rel_err(e.get_params()[2], rv) < 1e-10 and util.rel_err(gerr, rerr) < 1e-10
    # the plan does not depend on timing: a second matrix, a second engine, the same colours
    e3 = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL, als_max_levels=-1)
    m3 = engine.Matrix.from_csr(rp, col, val, p, y)
    assert np.array_equal(e3.als_plan(m3)[3], level_of)
    for x in (e, e3): x.close()
    for x in (m, m3): x.close()


def test_coloured_order_leaves_one_column_per_field_data_alone():
    """30 fields = 30 colours at most: the plan is as wide as the exact one (which the block form then takes)."""
    from fmwr_amd import _lib as L, engine
    n, p = 8_000, 3_000
    m = engine.Matrix.synthetic(n, p, Z, 5)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=4, mode=L.MODE_SEQUENTIAL, als_max_levels=-1)
    levels, _, _, level_of = e.als_plan(m)
    assert levels <= 2 * Z and e.als_plan_kind(m) == 2
    e.close(); m.close()


def test_coloured_sweep_on_zipf_columns_with_heavy_heads():
    """Zipf(1.05) columns: the heads' columns (more than 16 384 entries) take a colour of their own each, first, in index order; the light features colour around
    them.  Against the oracle on the relabelled matrix, as above."""
    from fmwr_amd import _lib as L, engine
    n, p, k = 150_000, 20_000, 4
    m0 = engine.Matrix.synthetic_iid(n, p, Z, 93, law=L.COLUMNS_ZIPF, zipf_s=1.05)
    rp, col, val, _ = m0.export(); m0.close()
    counts = np.bincount(col, minlength=p)
    assert (counts > 16_384).sum() >= 2                      # there ARE heavy heads
    y = util.labels(n, 93, "regression")
    w0, w, v = util.params(p, k, 67, stdev=0.05, fp32=False)
    lam = np.linspace(0.5, 1.0, k); mu = np.zeros(k)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=-1)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    levels, _, _, level_of = e.als_plan(m)
    level_of = level_of.copy()
    assert e.als_plan_kind(m) == 2
    heavy = np.flatnonzero(counts > 16_384)
    assert np.array_equal(level_of[heavy], np.arange(len(heavy)))   # one colour each, first, in index order
    lv = level_of[col].reshape(n, Z)
    srt = np.sort(lv, axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1])                 # proper
    order = np.lexsort((np.arange(p), level_of))
    rank = np.empty(p, np.int64); rank[order] = np.arange(p)
    rp2, col2, val2 = _relabel(rp, col, val, rank)
    X = oracle.Matrix(rp, col, val, p); X2 = oracle.Matrix(rp2, col2, val2, p)
    P = oracle.params(task=oracle.REGRESSION, k=k)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    rv2, rerr, _ = oracle.als_update_v(k, X2, np.ascontiguousarray(v[:, order]).ravel(), err0, alpha=1.0, v_lambda=lam, v_mu=mu)
    rv = np.empty_like(v); rv[:, order] = rv2.reshape(k, p)
    gerr = e.als_vsweep(m, err0, alpha=1.0, v_lambda=lam, v_mu=mu)
    assert util.rel_err(e.get_params()[2], rv) < 1e-10 and util.rel_err(gerr, rerr) < 1e-10
    e.close(); m.close()


def _numpy_sweep(rp, col, val, p, V, err, alpha, lam, mu, z, coords):
    """The coordinate steps of MCMC_ALS_Learner::update_v (:283-351) taken one after the other in the order `coords` gives them -- a plain
    restatement (one vectorised step per coordinate) that keeps e and EVERY factor's q = X v_f current.  Pinned below against the oracle in the oracle's own order."""
    n = len(rp) - 1
    K = V.shape[0]
    V = V.copy(); e = err.copy()
    rows_of = np.repeat(np.arange(n), np.diff(rp))
    order = np.argsort(col, kind="stable")
    cp = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=p))])
    crow, cval = rows_of[order], val[order].astype(np.float32)
    Q = np.zeros((n, K))
    for f in range(K):
        np.add.at(Q[:, f], rows_of, val.astype(np.float64) * V[f, col])
    for j, f in coords:
        r = crow[cp[j]:cp[j + 1]]; x = cval[cp[j]:cp[j + 1]]
        xx = (x * x).astype(np.float64); xd = x.astype(np.float64)
        old = V[f, j]
        h = xd * Q[r, f] - xx * old
        mean = float(np.sum(h * e[r])); var = float(np.sum(h * h))
        mean -= old * var
        with np.errstate(divide="ignore", invalid="ignore"):
            var = 1.0 / (lam[f] + alpha * var)
            mean = -var * (alpha * mean - mu[f] * lam[f])
            nv = 0.0 if not np.isfinite(var) else (mean + np.sqrt(var) * z[f, j] if z is not None else mean)
        if not np.isfinite(nv):
            continue
        V[f, j] = nv
        diff = old - nv
        Q[r, f] -= xd * diff
        e[r] -= h * diff
    return V, e


@pytest.mark.parametrize("gibbs,values", [(False, "ones"), (True, "normal")])
def test_feature_major_coloured_sweep(gibbs, values):
    """cfg.als_max_levels = -2: the coloured order with ALL factors of a feature stepped while its rows' state (e and every q_f) is on the chip -- the coordinates in
    (colour, feature, factor) order.  Checked against the coordinate-by-coordinate restatement above, which is itself pinned against the oracle in the oracle's order."""
    from fmwr_amd import _lib as L, engine
    n, p, k = 6_000, 1_500, 6
    m0 = engine.Matrix.synthetic_iid(n, p, 20, 95, law=L.COLUMNS_UNIFORM)
    rp, col, val, _ = m0.export(); m0.close()
    if values == "normal":
        val = np.random.default_rng(6).normal(0, 1, len(val)).astype(np.float32)
    y = util.labels(n, 95, "regression")
    w0, w, v = util.params(p, k, 71, stdev=0.1, fp32=False)
    lam = np.linspace(10.0, 20.0, k) if gibbs else np.linspace(0.1, 0.5, k); mu = np.linspace(-0.05, 0.05, k)
    z = np.random.default_rng(17).normal(0, 1, (k, p)) if gibbs else None
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=k)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    # the restatement in the oracle's order (factor outer, feature inner) IS the oracle
    rv, rerr, _ = oracle.als_update_v(k, X, v.ravel(), err0, alpha=1.1, v_lambda=lam, v_mu=mu, znorm=z.ravel() if gibbs else None)
    nv, ne = _numpy_sweep(rp, col, val, p, v, err0, 1.1, lam, mu, z, [(j, f) for f in range(k) for j in range(p)])
    assert util.rel_err(nv, rv.reshape(k, p)) < 1e-11 and util.rel_err(ne, rerr) < 1e-11
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=-2)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    _, _, _, level_of = e.als_plan(m)
    level_of = level_of.copy()
    assert e.als_plan_kind(m) == 3
    order = np.lexsort((np.arange(p), level_of))
    fv, fe = _numpy_sweep(rp, col, val, p, v, err0, 1.1, lam, mu, z, [(int(j), f) for j in order for f in range(k)])
    gerr = e.als_vsweep(m, err0, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
    assert util.rel_err(e.get_params()[2], fv) < 1e-10 and util.rel_err(gerr, fe) < 1e-10
    # a second sweep from there, and reproducibility
    gerr2 = e.als_vsweep(m, gerr, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
    fv2, fe2 = _numpy_sweep(rp, col, val, p, fv, fe, 1.1, lam, mu, z, [(int(j), f) for j in order for f in range(k)])
    assert util.rel_err(e.get_params()[2], fv2) < 1e-9 and util.rel_err(gerr2, fe2) < 1e-9
    e2 = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=-2)
    e2.set_params(w0, w, v)
    m2 = engine.Matrix.from_csr(rp, col, val, p, y)
    g1 = e2.als_vsweep(m2, err0, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
    g2 = e2.als_vsweep(m2, g1, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
    assert np.array_equal(g1, gerr) and np.array_equal(g2, gerr2) and np.array_equal(e2.get_params()[2], e.get_params()[2])   # bit for bit, run to run
    for x_ in (e, e2): x_.close()
    for x_ in (m, m2): x_.close()


@pytest.mark.parametrize("n,p,k,rows", [(12_000, 600, 16, "384 < rows <= 512: the 256-thread register kernel"), (12_000, 300, 16, "rows > 512: the LDS-resident kernel"),
                                        (9_000, 2_000, 8, "lists of ~90 rows, k = kp = 8: one wave, two rows per lane"), (9_000, 700, 11, "~260 rows, k < kp = 16: one wave, six rows per lane"),
                                        (9_000, 1_000, 16, "~180 rows: one wave, four rows per lane")])
def test_feature_major_kernel_forms(n, p, k, rows, monkeypatch):
    """The feature-major level runs as one WAVE per feature where a list fits its registers (kp 8 or 16, up to 384 rows: als_level_allf_wave_k), as a 256-thread
    workgroup with the lines in registers up to 512 rows (als_level_allf_reg_k) and with the lines in LDS beyond (als_level_allf_k).  The two workgroup forms share the
    thread -> row map and the order of every sum: bit for bit equal.  The one-wave form adds in its own order: equal to rounding, and to the restatement."""
    from fmwr_amd import _lib as L, engine
    m0 = engine.Matrix.synthetic_iid(n, p, 20, 314, law=L.COLUMNS_UNIFORM)
    rp, col, val, _ = m0.export(); m0.close()
    val = np.random.default_rng(8).uniform(0.2, 1.0, len(val)).astype(np.float32)
    y = util.labels(n, 314, "regression")
    w0, w, v = util.params(p, k, 72, stdev=0.1, fp32=False)
    lam = np.linspace(10.0, 20.0, k); mu = np.linspace(-0.05, 0.05, k)
    z = np.random.default_rng(18).normal(0, 1, (k, p))
    X = oracle.Matrix(rp, col, val, p)
    err0 = oracle.predict_batch(oracle.params(task=oracle.REGRESSION, k=k), X, w0, w, v.ravel()) - y
    out = {}
    for form in ("0", "0", "1", "2"):
        monkeypatch.setenv("FMX_ALS_ALLF_FORM", form)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=-2)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        _, _, _, level_of = e.als_plan(m)
        level_of = level_of.copy()
        assert e.als_plan_kind(m) == 3
        g1 = e.als_vsweep(m, err0, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
        g2 = e.als_vsweep(m, g1, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
        res = (g1, g2, e.get_params()[2].copy())
        if form in out:
            for a, b in zip(out[form], res):
                assert np.array_equal(a, b)            # run to run: bit for bit
        out[form] = res
        e.close(); m.close()
    for a, b in zip(out["1"], out["2"]):
        assert np.array_equal(a, b)
    for a, b in zip(out["0"], out["1"]):
        assert util.rel_err(a, b) < 1e-10
    order = np.lexsort((np.arange(p), level_of))
    fv, fe = _numpy_sweep(rp, col, val, p, v, err0, 1.1, lam, mu, z, [(int(j), f) for j in order for f in range(k)])
    assert util.rel_err(out["0"][0], fe) < 1e-10


def test_field_structured_data_takes_the_exact_levels_as_colours():
    """One column per field and row: the exact schedule's levels (as many as fields) ARE a proper colouring, in the reference's own feature order -- a coloured plan
    takes them instead of colouring speculatively (which spreads such data over hundreds of classes), so cfg.als_max_levels = -1 gives the reference's numbers there,
    bit for bit the exact plan's."""
    from fmwr_amd import _lib as L, engine
    n, p, z, k = 20_000, 3_000, 10, 5
    res = []
    for cap in (0, -1):
        m = engine.Matrix.synthetic(n, p, z, 77)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=cap)
        e.init_normal(5, 0.0, 0.1)
        levels, largest, _, level_of = e.als_plan(m)
        assert levels == z and e.als_plan_kind(m) == (2 if cap else 0)
        err0 = np.random.default_rng(3).normal(0, 1, n)
        g = e.als_vsweep(m, err0, alpha=1.0, v_lambda=np.full(k, 0.5))
        res.append((level_of.copy(), g, e.get_params()[2].copy()))
        e.close(); m.close()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("k,p,n,note", [(1, 800, 4_000, "kp = 2: the LDS-resident kernel"), (3, 800, 4_000, "kp = 4"), (20, 900, 5_000, "kp = 32: sixteen lanes per row"),
                                        (4, 40, 30_000, "lists of ~7 500 rows: NOT feature-major -- the sweep nests factor outer, as -1"),
                                        (20, 103, 8_000, "kp = 32 and lists of ~800 rows: NOT feature-major -- their lines would need 230 KB of LDS (577 rows fit at kp = 32)")])
def test_feature_major_edges(k, p, n, note):
    """Factor counts outside the register kernels' (kp 8 / 16), columns that never occur and empty rows; and a plan whose lists are too long for the feature-major
    form: fmx_als_plan_info says which nesting a -2 plan takes (3 feature-major, 2 factor outer) and the sweep agrees with the restatement in THAT order."""
    from fmwr_amd import _lib as L, engine
    rng = np.random.default_rng(k * 1000 + p)
    z = 10
    rows = []
    for r in range(n):
        if r % 97 == 0:
            rows.append(np.zeros(0, np.int64)); continue                       # an empty row
        rows.append(np.sort(rng.choice(np.arange(3, p), size=min(z, p - 3), replace=False)))   # columns 0..2 never occur
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate(rows).astype(np.uint32)
    val = rng.uniform(0.2, 1.0, len(col)).astype(np.float32)
    y = util.labels(n, 5, "regression")
    w0, w, v = util.params(p, k, 73, stdev=0.1, fp32=False)
    lam = np.linspace(10.0, 20.0, k); mu = np.linspace(-0.05, 0.05, k)
    zz = rng.normal(0, 1, (k, p))
    X = oracle.Matrix(rp, col, val, p)
    err0 = oracle.predict_batch(oracle.params(task=oracle.REGRESSION, k=k), X, w0, w, v.ravel()) - y
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=-2)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    _, _, _, level_of = e.als_plan(m)
    level_of = level_of.copy()
    kind = e.als_plan_kind(m)
    assert kind == (2 if "NOT" in note else 3)
    order = np.lexsort((np.arange(p), level_of))
    coords = [(int(j), f) for j in order for f in range(k)] if kind == 3 else [(int(j), f) for f in range(k) for j in order]
    fv, fe = _numpy_sweep(rp, col, val, p, v, err0, 1.1, lam, mu, zz, coords)
    g = e.als_vsweep(m, err0, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=zz)
    assert util.rel_err(g, fe) < 1e-10 and util.rel_err(e.get_params()[2], fv) < 1e-10
    e.close(); m.close()


@pytest.mark.parametrize("k", [8, 6])   # (6: k < kp = 8 -- e rides in the lines' spare slot during the V sweep)
def test_learners_on_a_feature_major_plan(k):
    """The learners' loops on a cfg.als_max_levels = -2 plan.  The w sweep goes in (colour, index) order: without a V sweep the ALS learner IS the oracle's learner on
    the relabelled matrix.  With the V sweep nested feature-major there is no oracle run to compare with -- exact coordinate steps of the least-squares
    objective never raise it, in any order, and the run is bit for bit reproducible; the MCMC learner's draws stay finite and reproducible."""
    from fmwr_amd import _lib as L, engine
    n, p = 5_000, 1_200
    m0 = engine.Matrix.synthetic_iid(n, p, Z, 41, law=L.COLUMNS_UNIFORM)
    rp, col, val, _ = m0.export(); m0.close()
    val = np.random.default_rng(2).uniform(0.3, 1.0, len(val)).astype(np.float32)
    y = util.labels(n, 41, "regression")
    w0, w, v = util.params(p, k, 19, stdev=0.1, fp32=False)
    def engine_(solver):
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=solver, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=-2)
        e.set_params(w0, w, v)
        return e
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    e = engine_(L.SOLVER_ALS)
    _, _, _, level_of = e.als_plan(m)
    level_of = level_of.copy()
    assert e.als_plan_kind(m) == 3
    # (a) w0 + w sweeps only: the oracle's learner on the relabelled matrix
    order = np.lexsort((np.arange(p), level_of))
    rank = np.empty(p, np.int64); rank[order] = np.arange(p)
    rp2, col2, val2 = _relabel(rp, col, val, rank)
    P = oracle.params(task=oracle.REGRESSION, k=k)
    ow0, ow, ov = oracle.als_learn(P, oracle.Matrix(rp2, col2, val2, p), y, w0, w[order], np.ascontiguousarray(v[:, order]).ravel(), 3, with_v=False)
    e.als_train(m, 3, with_v=False)
    g0, gw, gv = e.get_params()
    assert abs(g0 - ow0) < 1e-10 and util.rel_err(gw[order], ow) < 1e-10 and np.array_equal(gv, v)
    e.close()
    # (b) with the V sweep: the objective never rises, the run repeats bit for bit
    X = oracle.Matrix(rp, col, val, p)
    def objective(par):
        a0, aw, av = par
        r = oracle.predict_batch(P, X, a0, aw, av.ravel()) - y
        return float(r @ r)          # (the ALS learner steps with alpha = 1 and every lambda 0: fmo_als_learn_traced)
    runs = []
    for _ in range(2):
        e = engine_(L.SOLVER_ALS)
        obj = [objective((w0, w, v))]
        for it in range(3):
            e.als_train(m, 1, with_v=True)
            obj.append(objective(e.get_params()))
        assert all(b <= a * (1 + 1e-12) for a, b in zip(obj, obj[1:])) and obj[-1] < 0.9 * obj[0], obj
        runs.append(e.get_params()); e.close()
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    # (c) the MCMC learner on the same plan
    g = np.random.default_rng(10)
    gam = g.gamma((1 + n) / 2, 1.0, (3, 2)); nor = g.normal(0, 1, (3, 2 + p))
    runs = []
    for _ in range(2):
        e = engine_(L.SOLVER_MCMC)
        e.mcmc_train(m, 3, gam, nor)
        runs.append(e.get_params()); e.close()
    assert np.isfinite(runs[0][2]).all() and np.isfinite(runs[0][1]).all()
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    m.close()


@pytest.mark.parametrize("seed", range(12))
def test_feature_major_fuzz(seed):
    """Random shapes through the feature-major sweep: k 1 ... 20, 1 ... 24 entries per row, lists of a few to ~900 rows, real or unit values, ALS or Gibbs,
    a handful of empty rows -- whatever kernel form each level takes, against the restatement in the order the plan reports."""
    from fmwr_amd import _lib as L, engine
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.integers(1, 21)); z = int(rng.integers(1, 25))
    p = int(rng.integers(max(z + 1, 30), 1500)); n = int(rng.integers(200, 9000))
    gibbs = bool(rng.integers(0, 2)); unit = bool(rng.integers(0, 2))
    lens = np.where(rng.random(n) < 0.02, 0, z)
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum(lens)
    col = np.concatenate([np.sort(rng.choice(p, size=l, replace=False)) for l in lens if l > 0] or [np.zeros(0, np.int64)]).astype(np.uint32)
    val = np.ones(len(col), np.float32) if unit else rng.uniform(-1.0, 1.0, len(col)).astype(np.float32)
    y = util.labels(n, seed, "regression")
    w0, w, v = util.params(p, k, seed + 5, stdev=0.1, fp32=False)
    lam = np.linspace(10.0, 20.0, k) if gibbs else np.linspace(0.1, 0.9, k); mu = np.linspace(-0.05, 0.05, k)
    zz = rng.normal(0, 1, (k, p)) if gibbs else None
    X = oracle.Matrix(rp, col, val, p)
    err0 = oracle.predict_batch(oracle.params(task=oracle.REGRESSION, k=k), X, w0, w, v.ravel()) - y
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=-2)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    _, _, _, level_of = e.als_plan(m)
    level_of = level_of.copy()
    kind = e.als_plan_kind(m)
    assert kind in (2, 3)
    order = np.lexsort((np.arange(p), level_of))
    coords = [(int(j), f) for j in order for f in range(k)] if kind == 3 else [(int(j), f) for f in range(k) for j in order]
    fv, fe = _numpy_sweep(rp, col, val, p, v, err0, 0.9, lam, mu, zz, coords)
    g = e.als_vsweep(m, err0, alpha=0.9, v_lambda=lam, v_mu=mu, std_normals=zz)
    assert util.rel_err(g, fe) < 1e-9 and util.rel_err(e.get_params()[2], fv) < 1e-9, (k, z, p, n, gibbs, unit, kind)
    e.close(); m.close()


@pytest.mark.parametrize("z,law", [(20, "uniform"), (40, "uniform"), (20, "zipf")])
def test_row_wise_collision_test_gives_the_same_colouring(z, law, monkeypatch):
    """Crowded rounds of the colouring find their collisions row by row (colour_resolve_rows_k: 32 or 64 lanes per row), sparse ones by the features' walk: the same
    losers, so the same plan whichever runs (FMX_COLOUR_ROWS=0: the walk everywhere)."""
    from fmwr_amd import _lib as L, engine
    n, p = 40_000, 6_000
    plans = []
    for rows in ("1", "0"):
        monkeypatch.setenv("FMX_COLOUR_ROWS", rows)
        m = engine.Matrix.synthetic_iid(n, p, z, 12, law=L.COLUMNS_ZIPF if law == "zipf" else L.COLUMNS_UNIFORM, zipf_s=1.05)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=4, mode=L.MODE_SEQUENTIAL, als_max_levels=-1)
        levels, largest, _, level_of = e.als_plan(m)
        assert e.als_plan_kind(m) == 2 and levels > 130          # (deeper than the exact walk's cut-off: the colouring ran)
        plans.append((levels, largest, level_of.copy()))
        e.close(); m.close()
    assert plans[0][0] == plans[1][0] and plans[0][1] == plans[1][1] and np.array_equal(plans[0][2], plans[1][2])


@pytest.mark.parametrize("k,n,p", [(11, 9_000, 700), (5, 9_000, 1_500), (13, 12_000, 300), (3, 6_000, 600)])
def test_e_in_the_lines_spare_slot(k, n, p, monkeypatch):
    """k < kp: for the length of a feature-major sweep e rides in the spare last slot of the rows' lines (one line per row and level instead of a line and a pair);
    FMX_ALS_ALLF_EIL=0 keeps it in the pair table as at k = kp.  Only where e is stored differs: the same bits (wave, 256-thread and LDS kernels; kp 4, 8, 16)."""
    from fmwr_amd import _lib as L, engine
    m0 = engine.Matrix.synthetic_iid(n, p, 20, 55, law=L.COLUMNS_UNIFORM)
    rp, col, val, _ = m0.export(); m0.close()
    val = np.random.default_rng(9).uniform(0.2, 1.0, len(val)).astype(np.float32)
    y = util.labels(n, 55, "regression")
    w0, w, v = util.params(p, k, 75, stdev=0.1, fp32=False)
    lam = np.linspace(10.0, 20.0, k)
    z = np.random.default_rng(19).normal(0, 1, (k, p))
    err0 = np.random.default_rng(20).normal(0, 1, n)
    out = []
    for eil in ("1", "0"):
        monkeypatch.setenv("FMX_ALS_ALLF_EIL", eil)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=-2)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        e.als_plan(m)
        assert e.als_plan_kind(m) == 3
        g1 = e.als_vsweep(m, err0, alpha=1.1, v_lambda=lam, std_normals=z)
        g2 = e.als_vsweep(m, g1, alpha=1.1, v_lambda=lam, std_normals=z)
        out.append((g1, g2, e.get_params()[2].copy()))
        e.close(); m.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("k", [16, 12])
def test_feature_major_carried_q(k):
    """fmx_als_carry_q on a feature-major plan: the sweep corrects every q_f as it goes, so the table it leaves is X v_f of the new V and the next sweep skips its forward
    pass -- as long as V is bit for bit what the sweep left (fingerprint) and the matrix's values are the same.  Agrees with the rebuilt form to rounding; a
    fmx_set_params or redrawn values in between are noticed."""
    from fmwr_amd import _lib as L, engine
    n, p = 9_000, 900
    w0, w, v = util.params(p, k, 76, stdev=0.1, fp32=False)
    lam = np.linspace(0.2, 0.8, k)
    err0 = np.random.default_rng(4).normal(0, 1, n)
    res = {}
    for carry in (True, False):
        m = engine.Matrix.synthetic_iid(n, p, 20, 78, law=L.COLUMNS_UNIFORM).synthetic_values(3)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=-2)
        e.set_params(w0, w, v)
        e.als_carry_q(carry)
        assert e.als_plan(m)[0] > 0 and e.als_plan_kind(m) == 3
        g = err0
        out = []
        for it in range(4):
            g = e.als_vsweep(m, g, alpha=1.0, v_lambda=lam); out.append((g.copy(), e.get_params()[2].copy()))
        e.set_params(w0, w, v * 0.5)                               # V replaced behind the table's back
        g = e.als_vsweep(m, err0, alpha=1.0, v_lambda=lam); out.append((g.copy(), e.get_params()[2].copy()))
        g = e.als_vsweep(m, g, alpha=1.0, v_lambda=lam); out.append((g.copy(), e.get_params()[2].copy()))
        m.synthetic_values(8)                                      # the matrix's values redrawn: q = X v is another table
        g = e.als_vsweep(m, g, alpha=1.0, v_lambda=lam); out.append((g.copy(), e.get_params()[2].copy()))
        vn = e.get_params()[2].copy(); vn[0] = -vn[0]; vn[1] = -vn[1]   # two factor columns negated: an even number of sign bits (a linear fingerprint is blind to it)
        e.set_params(w0, w, vn)
        g = e.als_vsweep(m, g, alpha=1.0, v_lambda=lam); out.append((g.copy(), e.get_params()[2].copy()))
        ids = np.arange(0, p, 7, dtype=np.uint32)                  # ... and rows rewritten through fmx_set_rows
        rw, rv_ = e.get_rows(ids)
        e.set_rows(ids, rw, -rv_)
        g = e.als_vsweep(m, g, alpha=1.0, v_lambda=lam); out.append((g.copy(), e.get_params()[2].copy()))
        res[carry] = out
        e.close(); m.close()
    for (ga, va), (gb, vb) in zip(res[True], res[False]):
        assert util.rel_err(ga, gb) < 1e-10 and util.rel_err(va, vb) < 1e-10
