"""BASELINE.json configs[4]: "Synthetic 10M x 1M, k=16, MCMC.solver Gibbs sweep over V columns" -- the V sweep of
MCMC_ALS_Learner::update_v (solver/MCMC_ALS_Learner.h:272-354) at the config's own shape.

  * parity vs the oracle at k = 16, 30 entries per row, on the engine's own generators (one column per stratum: what bench.py
    --solver als trains on; i.i.d. uniform sorted columns: SURVEY 8(d)'s law), ALS and Gibbs forms, 1e-10.  k = 16 selects
    kp64 = 16: the padded layout and the stride of the all-factor q table that no k <= 6 case touches;
  * at the full 10 M x 1 M size, properties that need no CPU pass: the stratified matrix plans into exactly 30 levels, the
    residual's sum of squares falls strictly over two sweeps, two runs agree bit for bit, and columns cut into segments over many
    workgroups give the sums of the uncut form;
  * the device-resident entry (fmx_vsweep_device) equals the host-pointer one.
"""
import ctypes as C

import numpy as np
import pytest

import oracle
from tests import util

pytestmark = pytest.mark.gpu

K, Z = 16, 30


def _problem(engine, L, law, n, p, seed, values):
    if law == "stratified":
        m = engine.Matrix.synthetic(n, p, Z, seed)
    else:
        m = engine.Matrix.synthetic_iid(n, p, Z, seed, law=L.COLUMNS_UNIFORM)
    rp, col, val, _ = m.export()
    assert np.all(np.diff(rp) == Z)
    if values == "normal":   # SURVEY 8(d)'s variant: real values instead of the one-hot 1.0
        val = np.random.default_rng(seed).normal(0, 1, len(val)).astype(np.float32)
    y = util.labels(n, seed, "regression")
    m.close()
    return rp, col, val, y


@pytest.mark.parametrize("values", ["ones", "normal"])
@pytest.mark.parametrize("law", ["stratified", "iid"])
def test_configs4_als_vsweep_k16_matches_oracle(law, values):
    from fmwr_amd import _lib as L, engine
    n, p = 20_000, 6_000
    rp, col, val, y = _problem(engine, L, law, n, p, 41, values)
    w0, w, v = util.params(p, K, 17, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=K)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    lam = np.linspace(0.0, 0.4, K); mu = np.linspace(-0.05, 0.05, K)
    rv, rerr, _ = oracle.als_update_v(K, X, v.ravel(), err0, alpha=1.2, v_lambda=lam, v_mu=mu)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    levels, _, approx, _ = e.als_plan(m)
    assert not approx and (levels == Z if law == "stratified" else levels > Z)
    gerr = e.als_vsweep(m, err0, alpha=1.2, v_lambda=lam, v_mu=mu)
    gv = e.get_params()[2]
    assert util.rel_err(gv, rv.reshape(K, p)) < 1e-10
    assert util.rel_err(gerr, rerr) < 1e-10
    assert np.sum(gerr ** 2) < np.sum(err0 ** 2)


@pytest.mark.parametrize("law", ["stratified", "iid"])
def test_configs4_gibbs_vsweep_k16_matches_oracle(law):
    """The MCMC form (do_sample, :329-331) with the caller's standard normals in the reference's (f, j) draw order."""
    from fmwr_amd import _lib as L, engine
    n, p = 20_000, 6_000
    rp, col, val, y = _problem(engine, L, law, n, p, 43, "ones")
    w0, w, v = util.params(p, K, 19, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=K)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    lam = np.full(K, 2.0); mu = np.linspace(-0.02, 0.02, K)
    z = np.random.default_rng(5).normal(0, 1, (K, p))
    rv, rerr, _ = oracle.als_update_v(K, X, v.ravel(), err0, alpha=0.8, v_lambda=lam, v_mu=mu, znorm=z.ravel())
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=K, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    gerr = e.als_vsweep(m, err0, alpha=0.8, v_lambda=lam, v_mu=mu, std_normals=z)
    gv = e.get_params()[2]
    assert util.rel_err(gv, rv.reshape(K, p)) < 1e-10
    assert util.rel_err(gerr, rerr) < 1e-10


@pytest.mark.parametrize("tile_rows,n", [(4096, 20_000), (2048, 16_384), (8192, 16_385)])
@pytest.mark.parametrize("values,gibbs", [("ones", False), ("normal", True), ("normal", False), ("ones", True)])
def test_configs4_level_order_sweep_matches_oracle_and_the_three_pass_form(monkeypatch, values, gibbs, tile_rows, n):
    """The level-order form of the V sweep (fm_als_tiled.hip: the (q, e) pairs kept in the list order of the level that consumes them next -- a streaming
    sums + step kernel and a correct-and-permute kernel per level) on one-column-per-field data, forced onto a small matrix: several tiles with a short last
    one, a matrix of whole tiles, and a last tile of ONE row; one-hot and real values, ALS and Gibbs forms.  Against the oracle (1e-10), against the three-pass
    tiled form (FMX_ALS_ORDER=0) and against itself (bitwise, two runs)."""
    from fmwr_amd import _lib as L, engine
    monkeypatch.setenv("FMX_ALS_TILED", "1")
    monkeypatch.setenv("FMX_ALS_TILE_ROWS", str(tile_rows))
    p = 6_000
    rp, col, val, y = _problem(engine, L, "stratified", n, p, 59, values)
    w0, w, v = util.params(p, K, 31, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=K)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    lam = np.linspace(0.1, 0.5, K); mu = np.linspace(-0.05, 0.05, K)
    z = np.random.default_rng(9).normal(0, 1, (K, p)) if gibbs else None
    rv, rerr, _ = oracle.als_update_v(K, X, v.ravel(), err0, alpha=1.1, v_lambda=lam, v_mu=mu, znorm=z.ravel() if gibbs else None)
    res = {}
    for order in ("1", "1", "0"):
        monkeypatch.setenv("FMX_ALS_ORDER", order)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        assert e.als_tiled(m)[0] == e.als_plan(m)[0] == Z
        assert e.als_level_order(m) == (order == "1")
        gerr = e.als_vsweep(m, err0, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
        gv = e.get_params()[2]
        assert util.rel_err(gv, rv.reshape(K, p)) < 1e-10 and util.rel_err(gerr, rerr) < 1e-10
        # a second sweep from there (the pairs re-enter in row order)
        gerr2 = e.als_vsweep(m, gerr, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
        res.setdefault(order, []).append((gv, gerr, e.get_params()[2], gerr2))
        e.close(); m.close()
    a, b = res["1"]
    assert all(np.array_equal(x, y_) for x, y_ in zip(a, b))                       # bitwise run to run
    for x, y_ in zip(a, res["0"][0]):
        assert util.rel_err(x, y_) < 1e-10                                          # the three-pass form: the same sweeps, sums associated differently


@pytest.mark.parametrize("block_rows,n,p", [(512, 20_000, 6_000), (1024, 16_384, 30_000), (0, 16_385, 6_000), (2048, 12_000, 150_000), (4096, 9_000, 600_000)])
@pytest.mark.parametrize("values,gibbs", [("ones", False), ("normal", True), ("normal", False), ("ones", True)])
def test_configs4_block_form_matches_oracle_and_the_other_forms(monkeypatch, values, gibbs, block_rows, n, p):
    """The BLOCK form of the level-order V sweep (fm_als_blocks.hip: the level's array feature-block-major, one kernel per level -- a block's pairs stream into LDS,
    its lists are summed, stepped and corrected there, and leave as runs for the next level's blocks), forced onto small matrices: many blocks per level, the
    default capacity, lists of ~100, ~17, ~2.4 and ~0.5 rows (64, 16, 4 and 1 lanes per list; features WITHOUT rows still take their step), a last block of any
    size; one-hot and real values, ALS and Gibbs.  Against the oracle (1e-10), against the tile form (FMX_ALS_ORDER=1) and the three-pass form (0), and against
    itself (bitwise, two runs)."""
    from fmwr_amd import _lib as L, engine
    monkeypatch.setenv("FMX_ALS_TILED", "1")
    monkeypatch.setenv("FMX_ALS_TILE_ROWS", "4096")
    if block_rows:
        monkeypatch.setenv("FMX_ALS_BLOCK_ROWS", str(block_rows))
    rp, col, val, y = _problem(engine, L, "stratified", n, p, 67, values)
    w0, w, v = util.params(p, K, 37, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=K)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    # Gibbs draws on lists of a few rows under a weak prior are chaotic: a last-bit difference of one list's sums grows to 1e-7 .. O(1) over the 480 levels (V goes
    # from 0.1 to ~1.5) in EVERY form that does not happen to add in the oracle's own order (profiles/probes/block_form_debug.py, gpurun record
    # profiles/r05_block_form_gibbs_conditioning.txt: lambda 0.1-0.5: three-pass form 2.9e-7 / 0.35, block form 2.8e-7; lambda 10-20: all forms 2e-14).  The
    # sampled cases therefore take a prior under which the sweep contracts; the ALS cases keep the weak one.
    tol = 1e-10
    lam = np.linspace(10.0, 20.0, K) if gibbs else np.linspace(0.1, 0.5, K); mu = np.linspace(-0.05, 0.05, K)
    z = np.random.default_rng(11).normal(0, 1, (K, p)) if gibbs else None
    rv, rerr, _ = oracle.als_update_v(K, X, v.ravel(), err0, alpha=1.1, v_lambda=lam, v_mu=mu, znorm=z.ravel() if gibbs else None)
    res = {}
    for order in ("2", "2", "1", "0"):
        monkeypatch.setenv("FMX_ALS_ORDER", order)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        assert e.als_tiled(m)[0] == e.als_plan(m)[0] == Z
        assert e.als_level_order_form(m) == int(order)
        gerr = e.als_vsweep(m, err0, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
        gv = e.get_params()[2]
        assert util.rel_err(gv, rv.reshape(K, p)) < tol and util.rel_err(gerr, rerr) < tol
        gerr2 = e.als_vsweep(m, gerr, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)   # a second sweep from there (the pairs re-enter in row order)
        res.setdefault(order, []).append((gv, gerr, e.get_params()[2], gerr2))
        e.close(); m.close()
    a, b = res["2"]
    assert all(np.array_equal(x, y_) for x, y_ in zip(a, b))                       # bitwise run to run
    for other in ("1", "0"):
        for x, y_ in zip(a, res[other][0]):
            assert util.rel_err(x, y_) < tol                                        # the same sweeps, sums associated differently


def test_configs4_block_form_pipelined_and_plain_kernels_agree_bit_for_bit(monkeypatch):
    """The block form's level kernel exists twice: one workgroup per block (als_block_level_k: what a factor's first level, which takes its q in, always runs) and
    resident workgroups that prefetch their next block (als_block_level_pipe_k: every other level).  Same slots, same lane groups, same order: the same bits
    (FMX_ALS_BLOCK_PIPE=0 sends every level through the first)."""
    from fmwr_amd import _lib as L, engine
    monkeypatch.setenv("FMX_ALS_TILED", "1")
    monkeypatch.setenv("FMX_ALS_TILE_ROWS", "4096")
    monkeypatch.setenv("FMX_ALS_BLOCK_ROWS", "1024")
    n, p = 30_000, 6_000
    rp, col, val, y = _problem(engine, L, "stratified", n, p, 71, "normal")
    w0, w, v = util.params(p, K, 41, stdev=0.1, fp32=False)
    err0 = np.random.default_rng(3).normal(0, 1, n)
    z = np.random.default_rng(13).normal(0, 1, (K, p))
    out = {}
    for pipe in ("1", "0"):
        monkeypatch.setenv("FMX_ALS_BLOCK_PIPE", pipe)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=K, mode=L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        assert e.als_level_order_form(m) == 2
        gerr = e.als_vsweep(m, err0, alpha=1.0, v_lambda=np.full(K, 5.0), v_mu=np.zeros(K), std_normals=z)
        out[pipe] = (gerr, e.get_params()[2])
        e.close(); m.close()
    assert np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1], out["0"][1])


@pytest.mark.parametrize("values,gibbs", [("ones", False), ("normal", True)])
def test_configs4_carried_q_sweeps_match_the_oracle_and_notice_a_changed_v(monkeypatch, values, gibbs):
    """fmx_als_carry_q: the block form writes every factor's final q back into the q table, and the next sweep on the same plan skips the forward pass that
    rebuilds it -- if V is bit for bit what the sweep left.  Four sweeps in a row against four oracle sweeps (1e-10 each); then V is replaced through
    set_params (the fingerprint differs: the table must be rebuilt) and a fifth sweep is compared with the oracle from there."""
    from fmwr_amd import _lib as L, engine
    monkeypatch.setenv("FMX_ALS_TILED", "1")
    monkeypatch.setenv("FMX_ALS_TILE_ROWS", "4096")
    monkeypatch.setenv("FMX_ALS_BLOCK_ROWS", "1024")
    n, p = 20_000, 6_000
    rp, col, val, y = _problem(engine, L, "stratified", n, p, 73, values)
    w0, w, v = util.params(p, K, 43, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=K)
    err = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    lam = np.linspace(10.0, 20.0, K) if gibbs else np.linspace(0.1, 0.5, K); mu = np.linspace(-0.05, 0.05, K)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    e.als_carry_q(True)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    assert e.als_level_order_form(m) == 2
    rv, gerr = v.ravel().copy(), err.copy()
    for it in range(4):
        z = np.random.default_rng(100 + it).normal(0, 1, (K, p)) if gibbs else None
        rv, rerr, _ = oracle.als_update_v(K, X, rv, gerr if it == 0 else rerr_prev, alpha=1.1, v_lambda=lam, v_mu=mu, znorm=z.ravel() if gibbs else None)
        gerr = e.als_vsweep(m, gerr, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
        assert util.rel_err(e.get_params()[2], rv.reshape(K, p)) < 1e-10 and util.rel_err(gerr, rerr) < 1e-10, it
        rerr_prev = rerr
    # V replaced from outside: the carried table is stale and must not be used
    _, _, v2 = util.params(p, K, 47, stdev=0.1, fp32=False)
    e.set_params(w0, w, v2)
    err2 = oracle.predict_batch(P, X, w0, w, v2.ravel()) - y
    z = np.random.default_rng(200).normal(0, 1, (K, p)) if gibbs else None
    rv2, rerr2, _ = oracle.als_update_v(K, X, v2.ravel(), err2, alpha=1.1, v_lambda=lam, v_mu=mu, znorm=z.ravel() if gibbs else None)
    gerr2 = e.als_vsweep(m, err2, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
    assert util.rel_err(e.get_params()[2], rv2.reshape(K, p)) < 1e-10 and util.rel_err(gerr2, rerr2) < 1e-10
    # ... and a change a LINEAR fingerprint cannot see (two factor columns negated flip an even number of sign bits; round 5's sum of bits x (2 i + 1) was
    # unchanged by that): set_params drops the carried table explicitly, whatever V looks like
    v3 = e.get_params()[2].copy()
    v3[0] = -v3[0]; v3[1] = -v3[1]
    e.set_params(w0, w, v3)
    err3 = oracle.predict_batch(P, X, w0, w, v3.ravel()) - y
    rv3, rerr3, _ = oracle.als_update_v(K, X, v3.ravel(), err3, alpha=1.1, v_lambda=lam, v_mu=mu, znorm=z.ravel() if gibbs else None)
    gerr3 = e.als_vsweep(m, err3, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
    assert util.rel_err(e.get_params()[2], rv3.reshape(K, p)) < 1e-10 and util.rel_err(gerr3, rerr3) < 1e-10
    e.close(); m.close()


@pytest.mark.parametrize("k", [1, 5])
def test_configs4_block_form_with_odd_factor_counts(monkeypatch, k):
    """k = 1 and k = 5 (kp64 = 2 and 6: the padded layout of the fp64 tables, the stride of the q table): the block form against the oracle."""
    from fmwr_amd import _lib as L, engine
    monkeypatch.setenv("FMX_ALS_TILED", "1")
    monkeypatch.setenv("FMX_ALS_TILE_ROWS", "4096")
    monkeypatch.setenv("FMX_ALS_BLOCK_ROWS", "2048")
    n, p = 15_000, 6_000
    rp, col, val, y = _problem(engine, L, "stratified", n, p, 79, "normal")
    w0, w, v = util.params(p, k, 53, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=k)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    lam = np.linspace(0.1, 0.5, k); mu = np.linspace(-0.05, 0.05, k)
    rv, rerr, _ = oracle.als_update_v(k, X, v.ravel(), err0, alpha=1.1, v_lambda=lam, v_mu=mu)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    assert e.als_level_order_form(m) == 2
    gerr = e.als_vsweep(m, err0, alpha=1.1, v_lambda=lam, v_mu=mu)
    assert util.rel_err(e.get_params()[2], rv.reshape(k, p)) < 1e-10 and util.rel_err(gerr, rerr) < 1e-10
    e.close(); m.close()


def test_configs4_level_order_needs_a_complete_plan(monkeypatch):
    """Rows that lack a level (i.i.d. columns: many narrow levels) or levels that keep the column-walking kernels leave the plan incomplete: the V sweep
    then takes the three-pass form level by level, as before."""
    from fmwr_amd import _lib as L, engine
    monkeypatch.setenv("FMX_ALS_TILED", "1")
    monkeypatch.setenv("FMX_ALS_TILE_ROWS", "4096")
    n, p = 8_000, 3_000
    rp, col, val, y = _problem(engine, L, "iid", n, p, 61, "ones")
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=4, mode=L.MODE_SEQUENTIAL)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    assert e.als_tiled(m)[0] > 0 and not e.als_level_order(m)
    # one-column-per-field rows with one row cut short: a row lacks the last level
    rp2, col2, val2, y2 = _problem(engine, L, "stratified", n, p, 61, "ones")
    rp3 = rp2.copy(); rp3[-1] -= 1
    m2 = engine.Matrix.from_csr(rp3, col2[:-1], val2[:-1], p, y2)
    assert not e.als_level_order(m2)
    m3 = engine.Matrix.from_csr(rp2, col2, val2, p, y2)
    assert e.als_level_order(m3)


@pytest.mark.parametrize("lg", [1, 4])
@pytest.mark.parametrize("law,values,gibbs", [("stratified", "ones", False), ("stratified", "normal", True), ("iid", "normal", False), ("iid", "ones", True)])
def test_configs4_row_tiled_sweep_matches_oracle(monkeypatch, law, values, gibbs, lg):
    """The row-tiled form of the wide levels (fm_als_tiled.hip: per-tile sums against a slice of the (q, e) pairs, the coordinate steps, a row-major
    correction pass), forced onto a small matrix with 4 096-row tiles (five tiles, the last one short): every level of the stratified matrix is a
    field; the i.i.d. matrix has many narrow levels and rows that hold nothing at most of them.  ALS and Gibbs forms, one and four lanes per list."""
    from fmwr_amd import _lib as L, engine
    monkeypatch.setenv("FMX_ALS_TILED", "1")
    monkeypatch.setenv("FMX_ALS_ORDER", "0")   # (this test is about the three-pass form; the level-order form has its own)
    monkeypatch.setenv("FMX_ALS_TILE_ROWS", "4096")
    monkeypatch.setenv("FMX_ALS_TILE_LG", str(lg))
    n, p = 20_000, 6_000
    rp, col, val, y = _problem(engine, L, law, n, p, 53, values)
    w0, w, v = util.params(p, K, 31, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=K)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    lam = np.linspace(0.1, 0.5, K); mu = np.linspace(-0.05, 0.05, K)
    z = np.random.default_rng(9).normal(0, 1, (K, p)) if gibbs else None
    rv, rerr, _ = oracle.als_update_v(K, X, v.ravel(), err0, alpha=1.1, v_lambda=lam, v_mu=mu, znorm=z.ravel() if gibbs else None)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    levels = e.als_plan(m)[0]
    tiled, tile_rows, n_tiles = e.als_tiled(m)
    assert tile_rows == 4096 and n_tiles == 5
    assert tiled == levels if law == "stratified" else 0 < tiled <= levels
    gerr = e.als_vsweep(m, err0, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
    assert util.rel_err(e.get_params()[2], rv.reshape(K, p)) < 1e-10
    assert util.rel_err(gerr, rerr) < 1e-10


@pytest.mark.parametrize("values", ["normal", "ones"])
def test_configs4_row_tiled_learners_equal_the_column_walking_form(monkeypatch, values):
    """The ALS learner's loop (w0, w sweep, V sweep) and the MCMC learner's (w0, w sweep with draws) through the block form (w sweep AND V sweep: one kernel per
    level, fm_als_blocks.hip; several blocks per level), the tile form of the V sweep (w sweep: three passes), the three-pass tiled form and the column-walking
    kernels: the same sweeps, sums associated differently.  Real values and one-hot rows (a one-hot plan keeps no copy of the values: the ALS learner's forward
    pass, which runs on the block form's permuted CSR and leaves q for the V sweep, must not read them)."""
    from fmwr_amd import _lib as L, engine
    n, p, k = 12_000, 3_000, 8
    rp, col, val, y = _problem(engine, L, "stratified", n, p, 57, values)
    w0, w, v = util.params(p, k, 37, stdev=0.1, fp32=False)
    g = np.random.default_rng(10)
    gam = g.gamma((1 + n) / 2, 1.0, (3, 2)); nor = g.normal(0, 1, (3, 2 + p))
    out = {}
    monkeypatch.setenv("FMX_ALS_BLOCK_ROWS", "1024")
    for mode, order, form in (("1", "2", 2), ("1", "1", 1), ("1", "0", 0), ("0", "2", 0)):
        monkeypatch.setenv("FMX_ALS_TILED", mode)
        monkeypatch.setenv("FMX_ALS_ORDER", order)
        monkeypatch.setenv("FMX_ALS_TILE_ROWS", "2048")
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        assert (e.als_tiled(m)[0] > 0) == (mode == "1")
        assert e.als_level_order_form(m) == form
        e.als_train(m, 3, with_v=True)
        e2 = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=k, mode=L.MODE_SEQUENTIAL)
        e2.set_params(w0, w, v)
        e2.mcmc_train(m, 3, gam, nor)
        out[(mode, order)] = (e.get_params(), e2.get_params())
    ref = out[("0", "2")]
    for key in (("1", "2"), ("1", "1"), ("1", "0")):
        for a, b in zip(out[key], ref):
            assert abs(a[0] - b[0]) < 1e-10 * max(1.0, abs(b[0])), key
            assert util.rel_err(a[1], b[1]) < 1e-10 and util.rel_err(a[2], b[2]) < 1e-10, key


def test_configs4_device_resident_sweep_equals_the_host_pointer_one():
    from fmwr_amd import _lib as L, engine
    n, p = 20_000, 6_000
    rp, col, val, y = _problem(engine, L, "stratified", n, p, 47, "ones")
    w0, w, v = util.params(p, K, 23, stdev=0.1, fp32=False)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    z = np.random.default_rng(6).normal(0, 1, (K, p))
    lam = np.full(K, 1.0)
    out = {}
    for form in ("host", "device"):
        for gibbs in (False, True):
            e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=K, mode=L.MODE_SEQUENTIAL)
            e.set_params(w0, w, v)
            err0 = e.predict(m) - y
            if form == "host":
                err = e.als_vsweep(m, err0, alpha=1.0, v_lambda=lam, std_normals=z if gibbs else None)
            else:
                d_err = util.DevBuf.from_numpy(err0)
                d_z = util.DevBuf.from_numpy(z)
                e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=lam, dev_std_normals=d_z.ptr.value if gibbs else None)
                err = d_err.numpy()
            out[(form, gibbs)] = (err, e.get_params()[2])
    for gibbs in (False, True):
        assert np.array_equal(out[("host", gibbs)][0], out[("device", gibbs)][0])
        assert np.array_equal(out[("host", gibbs)][1], out[("device", gibbs)][1])


# ---- the config's own size ---------------------------------------------------------------------------------------------------
N, P, SEED = 10_000_000, 1_000_000, 20240001


def _sumsq(buf):
    a = buf.numpy()
    return float(np.dot(a, a))


def _labels(L, m, n):
    """the generator's labels through fmx_matrix_export, in slabs (no host pass over the matrix's entries)"""
    lab = np.zeros(n, np.float64)
    step = 2_000_000
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        yy = np.zeros(r1 - r0, np.float32)
        L.check(L.lib().fmx_matrix_export(m.h, C.c_int64(r0), C.c_int64(r1), None, None, None, yy.ctypes.data_as(C.c_void_p)))
        lab[r0:r1] = yy
    return lab


def test_configs4_full_size_levels_descent_and_reproducibility():
    """10 M x 1 M, k = 16, the stratified generator bench.py trains on: 30 levels (one per stratum), the residual falls strictly
    over two ALS sweeps, and a second engine reproduces V and the residual bit for bit."""
    from fmwr_amd import _lib as L, engine
    m = engine.Matrix.synthetic(N, P, Z, SEED)
    res = []
    for run in range(2):
        e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
        e.init_normal(SEED, 0.0, 0.1)
        if run == 0:
            levels, largest, approx, lof = e.als_plan(m)
            assert levels == Z and not approx
            # one level per stratum: feature j sits in stratum j * Z // P (the generator's bounds are (i * P) // Z)
            bounds = (np.arange(Z + 1, dtype=np.int64) * P) // Z
            want = np.searchsorted(bounds, np.arange(P), side="right") - 1
            assert np.array_equal(lof, want.astype(np.int32))
            assert largest == int(np.max(np.diff(bounds)))
        d_err = util.DevBuf(N)
        L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(N), d_err.ptr, C.c_int(L.LINK_NONE)))
        e.sync()
        err0 = d_err.numpy() - _labels(L, m, N)     # calculate_error, REGRESSION: e = y_hat - y (:520-527)
        d_err.upload(err0)
        s0 = float(np.dot(err0, err0))
        e.vsweep_device(m, d_err.ptr.value, alpha=1.0)
        s1 = _sumsq(d_err)
        e.vsweep_device(m, d_err.ptr.value, alpha=1.0)
        s2 = _sumsq(d_err)
        assert s2 < s1 < s0, (s0, s1, s2)
        ids = np.arange(0, P, 997, dtype=np.uint32)
        res.append((d_err.numpy(), e.get_rows(ids)[1], (s0, s1, s2)))
        e.close(); d_err.free()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    m.close()


def test_configs4_full_size_iid_columns_in_the_reference_order(monkeypatch):
    """10 M x 1 M, k = 16, SURVEY 8(d)'s i.i.d. columns under the exact schedule (cfg.als_max_levels = 0): a chain of ~19 400 dependent levels of at most ~110 features.
    One ALS sweep (310 000 level steps) through the record-ordered form (one launch per factor, the default), again, through the counter form (FMX_ALS_PERSIST=counter) and
    through one launch per level (FMX_ALS_PERSIST=0) from the same start: the whole residual and V's sampled rows bit for bit, and the residual falls."""
    from fmwr_amd import _lib as L, engine
    m = engine.Matrix.synthetic_iid(N, P, Z, SEED, law=L.COLUMNS_UNIFORM)
    err0 = np.random.default_rng(7).normal(0, 1, N)
    res = []
    for persist in ("1", "1", "0", "counter"):
        monkeypatch.setenv("FMX_ALS_PERSIST", persist)
        e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL, als_max_levels=0)
        e.init_normal(SEED, 0.0, 0.01)
        levels, largest, approx, _ = e.als_plan(m)
        assert not approx and levels > 10_000 and largest <= 256
        d_err = util.DevBuf(N)
        d_err.upload(err0)
        e.vsweep_device(m, d_err.ptr.value, alpha=1.0)
        res.append((d_err.numpy(), e.get_rows(np.arange(0, P, 97, dtype=np.uint32))[1]))
        e.close(); d_err.free()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[0][0], res[2][0]) and np.array_equal(res[0][1], res[2][1])
    assert np.array_equal(res[0][0], res[3][0]) and np.array_equal(res[0][1], res[3][1])
    assert float(np.dot(res[0][0], res[0][0])) < 0.5 * float(np.dot(err0, err0))
    m.close()


def test_configs4_full_size_level_order_and_row_tiled_forms_equal_the_column_walking_form(monkeypatch):
    """10 M x 1 M, k = 16.  The default: a complete plan whose lists all fit a block, so the V sweep takes the BLOCK form of the level-order sweep (one kernel
    per level, fm_als_blocks.hip); FMX_ALS_ORDER=1: its tile form (two kernels per level, 153 tiles of 65 536 rows); FMX_ALS_ORDER=0: the three-pass row-tiled
    form (77 tiles of 131 072 rows); FMX_ALS_TILED=0: the column-walking kernels.  One Gibbs sweep each from the same start: V (sampled rows) and the residual
    agree to 1e-10; the block form twice: bit for bit."""
    from fmwr_amd import _lib as L, engine
    out = {}
    for name, env in (("order", {}), ("order_again", {}), ("order_tiles", {"FMX_ALS_ORDER": "1"}), ("three_pass", {"FMX_ALS_ORDER": "0"}), ("columns", {"FMX_ALS_TILED": "0"})):
        monkeypatch.delenv("FMX_ALS_ORDER", raising=False); monkeypatch.delenv("FMX_ALS_TILED", raising=False)
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        m = engine.Matrix.synthetic(N, P, Z, SEED)
        e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=K, mode=L.MODE_SEQUENTIAL)
        e.init_normal(SEED, 0.0, 0.1)
        want = {"order": (Z, 65536, 153), "order_again": (Z, 65536, 153), "order_tiles": (Z, 65536, 153), "three_pass": (Z, 131072, 77), "columns": (0, 0, 0)}[name]
        assert e.als_tiled(m) == want
        assert e.als_level_order_form(m) == {"order": 2, "order_again": 2, "order_tiles": 1, "three_pass": 0, "columns": 0}[name]
        d_err = util.DevBuf(N)
        L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(N), d_err.ptr, C.c_int(L.LINK_NONE)))
        e.sync()
        d_z = util.DevBuf.from_numpy(np.random.default_rng(12).normal(0, 1, (K, P)))
        e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=np.full(K, 1.0), dev_std_normals=d_z.ptr.value)
        out[name] = (d_err.numpy(), e.get_rows(np.arange(0, P, 499, dtype=np.uint32))[1])
        e.close(); d_err.free(); d_z.free(); m.close()
    assert np.array_equal(out["order"][0], out["order_again"][0]) and np.array_equal(out["order"][1], out["order_again"][1])
    for name in ("order", "order_tiles", "three_pass"):
        assert util.rel_err(out[name][0], out["columns"][0]) < 1e-10 and util.rel_err(out[name][1], out["columns"][1]) < 1e-10


def test_configs4_full_size_block_form_with_real_values_equals_the_tile_form(monkeypatch):
    """10 M x 1 M, k = 16, the stored values redrawn U(0, 1) (SURVEY 8(d)'s value variant): the block form then works on 6 144-pair blocks with the entry values in
    LDS beside the pairs.  One ALS sweep through it and through the tile form (FMX_ALS_ORDER=1) from the same start: V (sampled rows) and the residual to 1e-10."""
    from fmwr_amd import _lib as L, engine
    out = {}
    for name, env in (("blocks", {}), ("tiles", {"FMX_ALS_ORDER": "1"})):
        monkeypatch.delenv("FMX_ALS_ORDER", raising=False)
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        m = engine.Matrix.synthetic(N, P, Z, SEED).synthetic_values(SEED + 1)
        e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
        e.init_normal(SEED, 0.0, 0.1)
        assert e.als_level_order_form(m) == (2 if name == "blocks" else 1)
        d_err = util.DevBuf(N)
        L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(N), d_err.ptr, C.c_int(L.LINK_NONE)))
        e.sync()
        e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=np.full(K, 1.0))
        out[name] = (d_err.numpy(), e.get_rows(np.arange(0, P, 499, dtype=np.uint32))[1])
        e.close(); d_err.free(); m.close()
    assert util.rel_err(out["blocks"][0], out["tiles"][0]) < 1e-10 and util.rel_err(out["blocks"][1], out["tiles"][1]) < 1e-10


def test_configs4_full_size_split_columns_equal_the_unsplit_form(monkeypatch):
    """10 M rows x 1 M features in 30 one-hot fields of which four are tiny (3 to 40 values: columns of 0.25 M to 3.3 M entries,
    cut into segments over many workgroups, fm_als_kernels.hip als_vh_*): the same levels, and the sums of the uncut form to
    1e-10 (the partial sums associate differently)."""
    from fmwr_amd import _lib as L, engine
    small = [3, 7, 16, 40]
    big = (P - sum(small)) // 26
    vocab = [big] * 25 + [P - sum(small) - 25 * big] + small
    assert sum(vocab) == P and len(vocab) == Z
    m = engine.Matrix.synthetic_fields(N, 0, vocab, 1.0, SEED)
    out = []
    for split in ("1", "0"):
        monkeypatch.setenv("FMX_ALS_SPLIT", split)
        mm = m if split == "1" else engine.Matrix.synthetic_fields(N, 0, vocab, 1.0, SEED)   # the plan is cached per matrix
        e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
        e.init_normal(SEED, 0.0, 0.1)
        levels, _, approx, _ = e.als_plan(mm)
        assert levels == Z and not approx
        d_err = util.DevBuf(N)
        L.check(L.lib().fmx_predict_device(e.h, mm.h, C.c_int64(0), C.c_int64(N), d_err.ptr, C.c_int(L.LINK_NONE)))
        e.sync()
        s0 = _sumsq(d_err)                       # residual against y = 0: the sweep drives y_hat towards 0
        e.vsweep_device(mm, d_err.ptr.value, alpha=1.0)
        s1 = _sumsq(d_err)
        assert s1 < s0
        ids = np.concatenate([np.arange(0, P, 1009), np.arange(P - sum(small), P)]).astype(np.uint32)
        out.append((d_err.numpy(), e.get_rows(ids)[1]))
        e.close(); d_err.free()
        if mm is not m:
            mm.close()
    assert util.rel_err(out[0][0], out[1][0]) < 1e-10 and util.rel_err(out[0][1], out[1][1]) < 1e-10
    m.close()


def test_an_approximate_sweep_that_raises_the_residual_falls_back_to_the_exact_schedule():
    """cfg.als_max_levels asks for the grouped (approximate) form; five features that always occur together and land in one group
    step against the same snapshot and overshoot fivefold -- the residual's sum of squares goes UP.  The engine notices, puts V and the
    residual back, marks the matrix exact-only and runs the sweep through the level schedule: the caller gets the exact sweep's result
    (the approximate form used to return 1e22 silently on Zipf columns)."""
    from fmwr_amd import _lib as L, engine
    rng = np.random.default_rng(3)
    S = [1000, 1001, 1002, 1003, 1004]
    rows = [np.array(S, np.uint32) for _ in range(1500)]
    for s in S:                                                    # "lift" rows: every feature of S takes position 4 somewhere -> one group
        rows += [np.array([10, 11, 12, 13, s], np.uint32) for _ in range(40)]
    order = rng.permutation(len(rows))
    rows = [rows[i] for i in order]
    n, p, k = len(rows), 1100, 4
    rp = np.arange(n + 1, dtype=np.int64) * 5
    col = np.concatenate(rows); val = np.ones(len(col), np.float32)
    y = util.labels(n, 3, "regression")
    w0, w, v = util.params(p, k, 3, stdev=0.3, fp32=False)
    P = oracle.params(task=oracle.REGRESSION, k=k)
    X = oracle.Matrix(rp, col, val, p)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    rv, rerr, _ = oracle.als_update_v(k, X, v.ravel(), err0)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=2)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    groups, _, approx, group_of = e.als_plan(m)
    assert approx and len(set(group_of[S])) == 1                   # the approximate plan, S in one group
    gerr = e.als_vsweep(m, err0)
    assert util.rel_err(e.get_params()[2], rv.reshape(k, p)) < 1e-10 and util.rel_err(gerr, rerr) < 1e-10
    assert np.sum(gerr ** 2) < np.sum(err0 ** 2)
    assert not e.als_plan(m)[2]                                    # exact from now on


def test_deep_plans_replayed_as_a_graph_equal_the_eager_launches(monkeypatch):
    """i.i.d. columns: hundreds of levels of a few features.  The level launches of one factor are captured once as a HIP graph and
    replayed per factor (fm_als_kernels.hip: sweep_graph); FMX_ALS_GRAPH=0 launches them eagerly.  Same kernels, same order: same bits,
    ALS and Gibbs forms, two sweeps (the second replays the first's graph)."""
    from fmwr_amd import _lib as L, engine
    n, p = 20_000, 6_000
    rp, col, val, y = _problem(engine, L, "iid", n, p, 51, "normal")
    w0, w, v = util.params(p, K, 29, stdev=0.1, fp32=False)
    z = np.random.default_rng(8).normal(0, 1, (K, p))
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("FMX_ALS_GRAPH", flag)
        for gibbs in (False, True):
            e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=K, mode=L.MODE_SEQUENTIAL)
            e.set_params(w0, w, v)
            m = engine.Matrix.from_csr(rp, col, val, p, y)
            err = e.predict(m) - y
            lam = np.full(K, 1.5)
            err = e.als_vsweep(m, err, alpha=1.1, v_lambda=lam, std_normals=z if gibbs else None)
            err = e.als_vsweep(m, err, alpha=0.9, v_lambda=lam, std_normals=z if gibbs else None)
            out[(flag, gibbs)] = (err, e.get_params()[2])
    for gibbs in (False, True):
        assert np.array_equal(out[("1", gibbs)][0], out[("0", gibbs)][0]) and np.array_equal(out[("1", gibbs)][1], out[("0", gibbs)][1])


@pytest.mark.parametrize("values,gibbs", [("ones", False), ("normal", True), ("normal", False), ("ones", True)])
def test_configs4_deep_exact_plan_in_one_launch_per_factor(monkeypatch, values, gibbs):
    """SURVEY 8(d)'s i.i.d. columns in the reference's own index order: a chain of dependent levels (20 K x 6 K, 30 per row: more than 1 000).  The persistent forms --
    als_exact_flow_k (the default: one launch per factor, every row's record tagged with the number of features that have corrected it, a step takes a record once the
    tag equals its entry's rank in the row) and als_exact_persist_k (FMX_ALS_PERSIST=counter: the levels ordered by a counter of completed features) -- must give bit
    for bit what one launch per level gives (FMX_ALS_PERSIST=0), the default twice, and the oracle's numbers to 1e-10."""
    from fmwr_amd import _lib as L, engine
    n, p = 20_000, 6_000
    rp, col, val, y = _problem(engine, L, "iid", n, p, 47, values)
    w0, w, v = util.params(p, K, 23, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=K)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    lam = np.linspace(0.2, 0.6, K); mu = np.linspace(-0.05, 0.05, K)
    z = np.random.default_rng(11).normal(0, 1, (K, p)) if gibbs else None
    rv, rerr, _ = oracle.als_update_v(K, X, v.ravel(), err0, alpha=1.1, v_lambda=lam, v_mu=mu, znorm=z.ravel() if gibbs else None)
    res = []
    for persist in ("1", "1", "0", "counter"):
        monkeypatch.setenv("FMX_ALS_PERSIST", persist)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        levels, largest, approx, _ = e.als_plan(m)
        assert not approx and levels > 1_000 and largest <= 256   # (narrow levels: the shape the persistent form takes)
        gerr = e.als_vsweep(m, err0, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
        gerr = e.als_vsweep(m, gerr, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z) if not gibbs else gerr   # a second sweep on the first one's state (ALS)
        res.append((e.get_params()[2], gerr))
        e.close(); m.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[0][0], res[2][0]) and np.array_equal(res[0][1], res[2][1])
    assert np.array_equal(res[0][0], res[3][0]) and np.array_equal(res[0][1], res[3][1])
    if gibbs:
        assert util.rel_err(res[0][0], rv.reshape(K, p)) < 1e-10 and util.rel_err(res[0][1], rerr) < 1e-10
    else:
        rv2, rerr2, _ = oracle.als_update_v(K, X, rv, rerr, alpha=1.1, v_lambda=lam, v_mu=mu)
        assert util.rel_err(res[0][0], rv2.reshape(K, p)) < 1e-10 and util.rel_err(res[0][1], rerr2) < 1e-10


def test_configs4_deep_exact_plan_learner_w_sweep_in_one_launch(monkeypatch):
    """The ALS learner on the same chain-shaped plan: its w sweep (update_w, :208-256) takes the persistent form too; iterations with and without the V sweep, bit
    for bit the one-launch-per-level form."""
    from fmwr_amd import _lib as L, engine
    n, p = 20_000, 6_000
    rp, col, val, y = _problem(engine, L, "iid", n, p, 53, "normal")
    w0, w, v = util.params(p, K, 29, stdev=0.05, fp32=False)
    res = []
    for persist in ("1", "0", "counter"):
        monkeypatch.setenv("FMX_ALS_PERSIST", persist)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL, l2_w1=0.1, l2_v=0.1)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        e.als_train(m, 2, with_v=True)
        res.append(e.get_params())
        e.close(); m.close()
    for other in res[1:]:
        assert res[0][0] == other[0] and np.array_equal(res[0][1], other[1]) and np.array_equal(res[0][2], other[2])
    assert not np.array_equal(res[0][1], w)


def test_configs4_deep_exact_plan_with_columns_longer_than_the_kept_slots(monkeypatch):
    """Columns of ~600 entries (more than the 512 a wave keeps in registers: the persistent kernel's tail loop reads and writes its pairs past the L1 too) on a
    chain-shaped plan: bit for bit the one-launch-per-level form, the oracle to 1e-10."""
    from fmwr_amd import _lib as L, engine
    k, n, p, z = 4, 60_000, 3_000, 30
    m0 = engine.Matrix.synthetic_iid(n, p, z, 61, law=L.COLUMNS_UNIFORM)
    rp, col, val, _ = m0.export(); m0.close()
    val = np.random.default_rng(3).uniform(0.3, 1.0, len(val)).astype(np.float32)
    y = util.labels(n, 61, "regression")
    assert np.bincount(col, minlength=p).max() > 8 * 64
    w0, w, v = util.params(p, k, 37, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    err0 = oracle.predict_batch(oracle.params(task=oracle.REGRESSION, k=k), X, w0, w, v.ravel()) - y
    lam = np.linspace(0.5, 1.0, k)
    rv, rerr, _ = oracle.als_update_v(k, X, v.ravel(), err0, alpha=1.0, v_lambda=lam)
    res = []
    for persist in ("1", "0", "counter"):
        monkeypatch.setenv("FMX_ALS_PERSIST", persist)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        levels, largest, approx, _ = e.als_plan(m)
        assert not approx and levels > 1_000 and largest <= 256
        gerr = e.als_vsweep(m, err0, alpha=1.0, v_lambda=lam)
        res.append((e.get_params()[2], gerr))
        e.close(); m.close()
    for other in res[1:]:
        assert np.array_equal(res[0][0], other[0]) and np.array_equal(res[0][1], other[1])
    assert util.rel_err(res[0][0], rv.reshape(k, p)) < 1e-10 and util.rel_err(res[0][1], rerr) < 1e-10


@pytest.mark.parametrize("waves", ["1", "7", "2048"])
def test_configs4_record_ordered_sweep_with_any_number_of_waves(monkeypatch, waves):
    """als_exact_flow_k deals the plan's positions over however many one-wave workgroups it is given (FMX_ALS_FLOW_WAVES; never more than the device holds at once):
    one wave alone, seven, as many as fit -- bit for bit one launch per level."""
    from fmwr_amd import _lib as L, engine
    k, n, p = 3, 20_000, 6_000
    rp, col, val, y = _problem(engine, L, "iid", n, p, 79, "normal")
    w0, w, v = util.params(p, k, 53, stdev=0.1, fp32=False)
    err0 = np.random.default_rng(13).normal(0, 1, n)
    res = []
    for persist in ("1", "0"):
        monkeypatch.setenv("FMX_ALS_PERSIST", persist)
        monkeypatch.setenv("FMX_ALS_FLOW_WAVES", waves)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        gerr = e.als_vsweep(m, err0, alpha=1.0, v_lambda=np.full(k, 0.8))
        res.append((e.get_params()[2], gerr))
        e.close(); m.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_configs4_deep_exact_plan_steps_that_keep_their_old_value(monkeypatch):
    """CHECK_PARAM (MCMC_ALS_Learner.h:336): a step whose new value is not a number keeps the old one and skips its corrections.  In the record-ordered form the rows'
    tags must still move on (the next feature of the row waits for them): a residual with one NaN in it makes ~30 such steps per factor; all three forms agree bit
    for bit (NaNs in the same places) and none hangs."""
    from fmwr_amd import _lib as L, engine
    k, n, p = 4, 20_000, 6_000
    rp, col, val, y = _problem(engine, L, "iid", n, p, 71, "normal")
    w0, w, v = util.params(p, k, 43, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    err0 = oracle.predict_batch(oracle.params(task=oracle.REGRESSION, k=k), X, w0, w, v.ravel()) - y
    err0[1234] = np.nan
    lam = np.linspace(0.5, 1.0, k)
    res = []
    for persist in ("1", "0", "counter"):
        monkeypatch.setenv("FMX_ALS_PERSIST", persist)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        gerr = e.als_vsweep(m, err0, alpha=1.0, v_lambda=lam)
        res.append((e.get_params()[2], gerr))
        e.close(); m.close()
    for other in res[1:]:
        assert np.array_equal(res[0][0], other[0], equal_nan=True) and np.array_equal(res[0][1], other[1], equal_nan=True)
    kept = col[rp[1234]:rp[1235]]
    assert np.array_equal(res[0][0][:, kept], v[:, kept]) and np.isnan(res[0][1][1234]) and np.isfinite(np.delete(res[0][1], 1234)).all()


def test_configs4_rows_too_long_for_the_records_tags(monkeypatch):
    """A row of 70 000 entries: its entries' ranks do not fit the 16 bits als_rank keeps, the record-ordered form declines and the counter form runs the plan (a chain
    of 70 000 levels through that row); bit for bit the one-launch-per-level form."""
    from fmwr_amd import _lib as L, engine
    k, n, p = 2, 500, 70_000
    rng = np.random.default_rng(5)
    rows = [np.arange(p, dtype=np.uint32)] + [np.sort(rng.choice(p, 20, replace=False)).astype(np.uint32) for _ in range(n - 1)]
    rp = np.zeros(n + 1, dtype=np.int64); rp[1:] = np.cumsum([len(r) for r in rows])
    col = np.concatenate(rows); val = rng.uniform(0.5, 1.0, len(col)).astype(np.float32)
    y = util.labels(n, 73, "regression")
    w0, w, v = util.params(p, k, 47, stdev=0.1, fp32=False)
    err0 = rng.normal(0, 1, n)
    res = []
    for persist in ("1", "0"):
        monkeypatch.setenv("FMX_ALS_PERSIST", persist)
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        levels, largest, approx, _ = e.als_plan(m)
        assert not approx and levels == p
        gerr = e.als_vsweep(m, err0, alpha=1.0, v_lambda=np.full(k, 0.7))
        res.append((e.get_params()[2], gerr))
        e.close(); m.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("form", ["1", "counter"])
def test_configs4_persistent_sweep_that_gives_up_is_reported(monkeypatch, form):
    """Fault injection (fmx_debug_stall_next_persistent_sweep: the plan's first feature never hands its rows on -- their tags stay where they were / it is never counted):
    every bounded wait gives up and the sweep fails with FMX_ERR_HIP instead of hanging or returning a half-swept model as if it were whole; the next sweep on the same
    engine is right again."""
    monkeypatch.setenv("FMX_ALS_PERSIST", form)
    from fmwr_amd import _lib as L, engine
    k, n, p = 4, 20_000, 6_000
    rp, col, val, y = _problem(engine, L, "iid", n, p, 67, "ones")
    w0, w, v = util.params(p, k, 41, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    err0 = oracle.predict_batch(oracle.params(task=oracle.REGRESSION, k=k), X, w0, w, v.ravel()) - y
    lam = np.linspace(0.5, 1.0, k)
    rv, rerr, _ = oracle.als_update_v(k, X, v.ravel(), err0, alpha=1.0, v_lambda=lam)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    levels, largest, approx, _ = e.als_plan(m)
    assert levels > 1_000 and largest <= 256
    L.check(L.lib().fmx_debug_stall_next_persistent_sweep())
    with pytest.raises(Exception, match="gave up waiting"):
        e.als_vsweep(m, err0, alpha=1.0, v_lambda=lam)
    e.set_params(w0, w, v)
    gerr = e.als_vsweep(m, err0, alpha=1.0, v_lambda=lam)
    assert util.rel_err(e.get_params()[2], rv.reshape(k, p)) < 1e-10 and util.rel_err(gerr, rerr) < 1e-10
