"""The reassociated reference-order learner (cfg.seq_reassociate, fm_seq_reassoc_k): the reference's algorithm, visiting order and precision, the
forward's sum formed as w0 + (row part) so that only w0 chains one example to the next.  Not the oracle's bits: held to the oracle at 1e-10 on V
(north_star's bar is 1e-5) with exact prediction signs on every case the bitwise kernels are held to, and to itself bit for bit from run to run --
including the regimes where nearly every example conflicts with its neighbours (the waves then meet through the done tags)."""
import os

import numpy as np
import pytest

import oracle
from tests import util
from tests import test_gpu_seq_window as sw
from tests import test_gpu_train as tt
from tests.golden import make_golden as mg

pytestmark = pytest.mark.gpu

TOL = 1e-10
LEARN = {"sgd": oracle.sgd_learn, "ftrl": oracle.ftrl_learn, "tdap": oracle.tdap_learn}


def _run_re(*a, **kw):
    os.environ["FMX_SEQ_REASSOC"] = "1"
    try:
        return sw._run(*a, **kw)
    finally:
        os.environ.pop("FMX_SEQ_REASSOC", None)


def _same(a, b):
    return all(np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True) for x, y in zip(a, b))


def _close(got, ref, P, p):
    assert abs(got[0] - ref["w0"]) <= TOL * max(1.0, abs(ref["w0"]))
    assert util.rel_err(got[1], ref["w"]) < TOL
    if P.k:
        assert util.rel_err(got[2], ref["v"].reshape(P.k, p)) < TOL


@pytest.mark.parametrize("c", tt.CASES, ids=[c["name"] for c in tt.CASES])
def test_every_sequential_parity_case_through_the_config_flag(c):
    """tests/test_gpu_train.py::test_sequential_matches_oracle with cfg.seq_reassociate = 1 (k = 70 is outside the windowed learners: the flag is ignored there)"""
    from fmwr_amd import engine, _lib as L
    fm = (engine, L)
    rp, col, val, y, P, seed = tt._problem(c)
    n, p = len(rp) - 1, 300
    w0, w, v = util.params(p, P.k, seed, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    iters = 2 * n + 37
    ref = LEARN[c["solver"]](P, X, y, w0, w, v.ravel(), iters)
    e = tt._engine(fm, p, P, L.SOLVER_SGD if c["solver"] == "sgd" else L.SOLVER_FTRL, L.MODE_SEQUENTIAL, seq_reassociate=1)
    e.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    assert e.train(m, iters) == iters
    got = e.get_params()
    _close(got, ref, P, p)
    out = e.predict(m)
    refp = oracle.predict_batch(P, X, ref["w0"], ref["w"], ref["v"])
    assert np.array_equal(np.sign(out), np.sign(refp))


@pytest.mark.parametrize("name", [n for n, c in mg.CASES.items() if c["solver"] != "tdap"])
def test_golden_fixtures(name):
    from fmwr_amd import _lib as L, engine
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_v1.npz"))
    c = mg.CASES[name]
    _, _, _, y, _, _, _, Pm = mg.problem(name, c)
    e = engine.Engine(mg.P, task=Pm.task, solver={"sgd": L.SOLVER_SGD, "ftrl": L.SOLVER_FTRL}[c["solver"]], num_factor=Pm.k, gamma=Pm.gamma,
                      l2_w0=Pm.l2_reg0, l1_w1=Pm.l1_regw, l2_w1=Pm.l2_regw, l1_v=Pm.l1_regv, l2_v=Pm.l2_regv, learn_rate=Pm.learn_rate,
                      alpha_w=Pm.alpha_w, alpha_v=Pm.alpha_v, beta_w=Pm.beta_w, beta_v=Pm.beta_v, mode=L.MODE_SEQUENTIAL,
                      min_target=Pm.min_target, max_target=Pm.max_target, seq_reassociate=1)
    e.set_params(float(g[f"{name}/w0_in"]), g[f"{name}/w_in"], g[f"{name}/v_in"])
    m = engine.Matrix.from_csr(g[f"{name}/row_ptr"], g[f"{name}/col"], g[f"{name}/val"], mg.P, g[f"{name}/y"])
    e.train_order(m, g[f"{name}/order"])
    w0, w, v = e.get_params()
    assert np.max(np.abs(v - g[f"{name}/v"])) < TOL * np.max(np.abs(g[f"{name}/v"]))
    assert np.max(np.abs(w - g[f"{name}/w"])) < TOL * max(np.max(np.abs(g[f"{name}/w"])), 1e-300)
    assert np.array_equal(np.sign(e.predict(m)), np.sign(g[f"{name}/pred"]))


@pytest.mark.parametrize("name", [n for n, c in sw.SOLVERS.items() if c["solver"] != "tdap"])
@pytest.mark.parametrize("p,nnz", [(40, 6), (3000, 12), (200000, 30), (5000, 32)])
def test_conflict_regimes_against_the_oracle_and_run_to_run(name, p, nnz):
    """p = 40: every example conflicts with its neighbour; p = 3000: mixed; p = 200000: almost none; (5000, 32): rows of exactly 32 entries."""
    c = sw.SOLVERS[name]
    n = 1500
    a, ctx = _run_re(name, c, p, n, nnz, 2 * n + 11)
    b, _ = _run_re(name, c, p, n, nnz, 2 * n + 11)
    assert _same(a, b)
    P, rp, col, val, y, w0, w, v, order = ctx
    ref = LEARN[c["solver"]](P, oracle.Matrix(rp, col, val, p), y, w0, w, v.ravel(), len(order), order=order)
    _close(a, ref, P, p)


@pytest.mark.parametrize("name", [n for n, c in sw.SOLVERS.items() if c["k"] <= 32 and c["solver"] != "tdap"])
def test_rows_of_33_to_64_entries(name):
    c = sw.SOLVERS[name]
    n, p = 1200, 30000
    a, ctx = _run_re(name, c, p, n, 45, 2 * n + 7, max_nnz=64)
    b, _ = _run_re(name, c, p, n, 45, 2 * n + 7, max_nnz=64)
    assert _same(a, b)
    P, rp, col, val, y, w0, w, v, order = ctx
    assert np.diff(rp).max() > 40
    ref = LEARN[c["solver"]](P, oracle.Matrix(rp, col, val, p), y, w0, w, v.ravel(), len(order), order=order)
    _close(a, ref, P, p)


def test_random_strides_over_many_launches_and_the_kernel_is_the_reassociated_one():
    """random_step = 3 over three 65 536-example launches: within 1e-10 of the bitwise kernel, the same bits twice -- and NOT the bitwise kernel's bits
    (150 000 examples whose forward sums are associated differently: if every bit agreed the flag would not have been honoured)"""
    c = sw.SOLVERS["sgd_l2"]
    a, _ = _run_re("strides", c, 50000, 60000, 8, 150000, random_step=3)
    b, _ = _run_re("strides", c, 50000, 60000, 8, 150000, random_step=3)
    d, _ = sw._run("strides", c, 50000, 60000, 8, 150000, random_step=3, window=True)
    assert _same(a, b)
    assert abs(a[0] - d[0]) < TOL and util.rel_err(a[1], d[1]) < TOL and util.rel_err(a[2], d[2]) < TOL
    assert not np.array_equal(a[2], d[2])


def test_tdap_ignores_the_flag():
    c = sw.SOLVERS["tdap"]
    a, _ = _run_re("tdap", c, 3000, 1500, 12, 3011)
    d, _ = sw._run("tdap", c, 3000, 1500, 12, 3011, window=True)
    assert _same(a, d)


@pytest.mark.parametrize("seed", [s for s in range(48) if s % 4 != 3])
def test_fuzz_against_the_one_wave_kernel(seed):
    c, p, n, nnz, max_nnz, rstep = sw._fuzz_case(seed)
    iters = 2 * n + 3
    a, _ = _run_re("fuzz%d" % seed, c, p, n, nnz, iters, random_step=rstep, max_nnz=max_nnz)
    a2, _ = _run_re("fuzz%d" % seed, c, p, n, nnz, iters, random_step=rstep, max_nnz=max_nnz)
    b, _ = sw._run("fuzz%d" % seed, c, p, n, nnz, iters, random_step=rstep, window=False, max_nnz=max_nnz)
    assert _same(a, a2), (c, p, n, nnz, max_nnz, rstep)
    scale = lambda x: max(float(np.max(np.abs(x))) if np.size(x) else 0.0, 1e-300)
    assert abs(a[0] - b[0]) <= TOL * max(1.0, abs(b[0])), (c, p, n, nnz, max_nnz, rstep)
    assert np.max(np.abs(a[1] - b[1]), initial=0.0) <= TOL * scale(b[1]) and np.max(np.abs(a[2] - b[2]), initial=0.0) <= TOL * scale(b[2]), (c, p, n, nnz, max_nnz, rstep)


@pytest.mark.parametrize("count", [1, 2, 7, 14, 15, 16, 29, 31, 59, 60, 61, 121])
def test_launches_shorter_than_the_ring(count):
    """Fewer examples than worker waves (15), than ring slots (60), one more and one less: waves without an example leave at once, the chain stops at `count`."""
    c = sw.SOLVERS["sgd_l2"]
    a, ctx = _run_re("tiny", c, 3000, 400, 12, count)
    b, _ = sw._run("tiny", c, 3000, 400, 12, count, window=False)
    assert abs(a[0] - b[0]) < TOL and util.rel_err(a[1], b[1]) < TOL and util.rel_err(a[2], b[2]) < TOL
    P, rp, col, val, y, w0, w, v, order = ctx
    assert len(order) == count


def test_a_wait_that_gives_up_is_reported_not_returned_as_nan():
    """Fault injection (fmx_debug_lose_next_seq_multiplier: a worker never sees its example's multiplier): the bounded waits end the launch, and fmx_sync / fmx_get_params
    say so (FMX_ERR_HIP) instead of handing out a NaN; fmx_set_params clears the state and the same engine trains correctly again."""
    from fmwr_amd import engine, _lib as L
    c = sw.SOLVERS["sgd_l2"]
    a, ctx = _run_re("giveup", c, 3000, 600, 12, 500)     # the healthy run, for reference
    P, rp, col, val, y, w0, w, v, order = ctx
    e = engine.Engine(3000, task=P.task, solver=L.SOLVER_SGD, num_factor=P.k, l2_w0=P.l2_reg0, l2_w1=P.l2_regw, l2_v=P.l2_regv, learn_rate=P.learn_rate,
                      mode=L.MODE_SEQUENTIAL, seq_reassociate=1, min_target=P.min_target, max_target=P.max_target)
    m = engine.Matrix.from_csr(rp, col, val, 3000, y)
    e.set_params(w0, w, v)
    L.check(L.lib().fmx_debug_lose_next_seq_multiplier())
    e.train_order(m, order)
    with pytest.raises(Exception, match="gave up waiting"):
        e.get_params()
    with pytest.raises(Exception, match="gave up waiting"):
        e.sync()
    e.set_params(w0, w, v)
    e.train_order(m, order)
    got = e.get_params()
    assert got[0] == a[0] and np.array_equal(got[1], a[1]) and np.array_equal(got[2], a[2])
