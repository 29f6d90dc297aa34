"""Phase 1 on rows of differing lengths: the flat form (fm_rows_forward_flat_k, fm_batch_kernels.hip).

The entries of a block of rows are one stream cut evenly over the lane groups; a row's sums are then the sum of its pieces, not one sequential
sum (core/Model.h:83-97 is the loop both restate).  So the flat form is held to the oracle's bars, not to the static kernel's bits; what it promises
on top is that a row's bits depend on the MATRIX alone (its block of rows is cut as a whole whatever launch reaches it).  Opt-in (FMX_ROWS_FLAT=1): it is
measured slower than one lane group per row (profiles/r04_ragged_forms.txt) and kept, with these tests, as the record of the form north_star names.
"""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

N, P = 140_000, 4_000   # >= 512 wide workgroups per launch: the 256-thread forms


def _engine(L, engine, p, k, wide=0, solver=None, **kw):
    solver = L.SOLVER_SGD if solver is None else solver
    reg = dict(l1_w1=1e-4, l2_v=1e-3) if solver == L.SOLVER_FTRL else dict(l2_w1=1e-4, l2_v=1e-4, learn_rate=0.05)
    return engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=solver, num_factor=k, mode=L.MODE_MINIBATCH, state_fp64=wide, **reg, **kw)


def _predict_rows(L, e, m, r0, r1):
    buf = util.DevBuf(r1 - r0)
    L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(r0), C.c_int64(r1), buf.ptr, C.c_int(L.LINK_NONE)))
    e.sync()
    out = buf.numpy()
    buf.free()
    return out


def test_the_form_is_pinned_by_the_environment_alone(monkeypatch):
    """rows_flat (fmx_internal.h): the flat form is opt-in (measured slower, profiles/r04_ragged_forms.txt) -- FMX_ROWS_FLAT=1 on rows of differing lengths,
    never on rows of one length; read through fmx_matrix_rows_form (0 static, 1 flat, 2 pulled)."""
    from fmwr_amd import engine
    monkeypatch.delenv("FMX_ROWS_FLAT", raising=False)
    monkeypatch.delenv("FMX_ROWS_PULL", raising=False)
    fixed = engine.Matrix.synthetic_iid(5_000, 1_000, 12, seed=1)
    ragged = engine.Matrix.synthetic_ragged(5_000, 1_000, 30.0, seed=1)
    assert (fixed.rows_form(), ragged.rows_form()) == (0, 0)
    monkeypatch.setenv("FMX_ROWS_FLAT", "1")
    assert (fixed.rows_form(), ragged.rows_form()) == (0, 1)
    monkeypatch.setenv("FMX_ROWS_PULL", "1")
    assert ragged.rows_form() == 2


@pytest.mark.parametrize("k,wide,values,solver", [(16, 0, "normal", "sgd"), (8, 0, "ones", "sgd"), (16, 1, "normal", "sgd"), (64, 0, "normal", "ftrl"), (3, 0, "normal", "sgd"),
                                                  (128, 0, "ones", "sgd")])
def test_flat_and_static_agree_to_rounding(monkeypatch, k, wide, values, solver):
    """The same matrix through both forms: predictions to 1e-12 of their scale (fp64 sums, associated differently), three training steps to the state's own
    rounding (fp32 tables: 2e-6 relative; fp64 tables: 1e-11) -- empty rows, one-entry rows, rows crossing several lane groups' ranges (k = 64 / 128: 16 / 8
    groups per block), values and one-hot."""
    from fmwr_amd import _lib as L, engine
    rp, col, val = util.random_csr(N, P, 12, seed=31 + k, empty_rows=True, values=values)
    lens = np.diff(rp)
    assert lens.min() == 0 and lens.max() >= 24
    y = util.labels(N, 31)
    w0, w, v = util.params(P, k, 31)
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("FMX_ROWS_FLAT", flag)
        m = engine.Matrix.from_csr(rp, col, val, P, y)
        assert m.rows_form() == int(flag)
        e = _engine(L, engine, P, k, wide, L.SOLVER_FTRL if solver == "ftrl" else L.SOLVER_SGD, batch_rows=N // 2)
        e.set_params(w0, w, v)
        pred = e.predict(m)
        e.train(m, N + N // 2)
        out.append((pred, e.get_params()))
    scale = np.max(np.abs(out[1][0]))
    assert np.max(np.abs(out[0][0] - out[1][0])) <= 1e-12 * scale
    assert not np.array_equal(out[0][0], out[1][0])          # (the forms DO associate differently: this test would not notice a flag that is ignored otherwise)
    tol = 1e-11 if wide else 2e-6
    assert abs(out[0][1][0] - out[1][1][0]) <= tol * max(1.0, abs(out[1][1][0]))
    assert util.rel_err(out[0][1][1], out[1][1][1]) <= tol and util.rel_err(out[0][1][2], out[1][1][2]) <= tol


@pytest.mark.parametrize("wide", [0, 1])
def test_flat_form_against_the_oracle(monkeypatch, wide):
    """SURVEY 8(d)'s ragged law through the flat form against the CPU restatement: forward (core/Model.h:75-101) and three mini-batch SGD steps
    (solver/SGD_Learner.h:44-204 summed per step), at the bars of the static kernel's own parity tests."""
    from fmwr_amd import _lib as L, engine
    import oracle
    monkeypatch.setenv("FMX_ROWS_FLAT", "1")
    n, p, k, B = 120_000, 5_000, 16, 40_000
    m = engine.Matrix.synthetic_ragged(n, p, 30.0, seed=11)
    assert m.rows_form() == 1
    rp, col, val, y = m.export()
    w0, w, v = util.params(p, k, 11)
    X = oracle.Matrix(rp, col, val, p)
    Pm = oracle.params(task=oracle.CLASSIFICATION, k=k, l2_regw=1e-4, l2_regv=1e-4, learn_rate=0.05)
    e = _engine(L, engine, p, k, wide, batch_rows=B)
    e.set_params(w0, w, v)
    np.testing.assert_allclose(e.predict(m), oracle.predict_batch(Pm, X, w0, w, v.ravel()), rtol=0, atol=1e-11 if wide else 1e-5)
    assert e.train(m, n) == n
    mb = oracle.SgdMinibatch(Pm, X, y, w0, w, v.ravel())
    for b in range(0, n, B):
        mb.step(b, b + B)
    gw0, gw, gv = e.get_params()
    tol = 1e-10 if wide else 1e-5
    assert util.rel_err(gv, mb.v.reshape(k, p)) < tol and util.rel_err(gw, mb.w) < tol and abs(gw0 - mb.w0.value) < tol


def test_a_rows_bits_depend_on_the_matrix_alone(monkeypatch):
    """The flat form cuts the MATRIX's blocks of rows, each as a whole: a launch that starts and ends inside blocks (rows 1 037 .. 131 101) gives every row the
    bits the whole-matrix launch gives it; a training step that starts inside a block (batch_rows = 70 001, second step) is reproducible."""
    from fmwr_amd import _lib as L, engine
    monkeypatch.setenv("FMX_ROWS_FLAT", "1")
    k = 16
    rp, col, val = util.random_csr(N, P, 12, seed=77, empty_rows=True)
    y = util.labels(N, 77)
    w0, w, v = util.params(P, k, 77)
    m = engine.Matrix.from_csr(rp, col, val, P, y)
    e = _engine(L, engine, P, k, batch_rows=N)
    e.set_params(w0, w, v)
    full = e.predict(m)
    for r0, r1 in ((1_037, 131_101), (64 * 500, N), (63, N - 1), (0, 70_001)):
        assert np.array_equal(_predict_rows(L, e, m, r0, r1), full[r0:r1]), (r0, r1)
    # second step of the 70 001-row batches starts at row 70 001 (inside a block)
    e3 = _engine(L, engine, P, k, batch_rows=70_001)
    e3.set_params(w0, w, v)
    e3.step(m, 1)
    p1 = e3.get_params()
    e4 = _engine(L, engine, P, k, batch_rows=70_001)
    e4.set_params(w0, w, v)
    e4.step(m, 1)
    p2 = e4.get_params()
    assert p1[0] == p2[0] and np.array_equal(p1[2], p2[2]) and np.any(p1[2] != v)


def test_rows_longer_than_a_stage_and_a_lane_groups_range(monkeypatch):
    """Rows of thousands of entries among short ones: a row then spans several stage chunks (2 048 entries) and the ranges of many lane groups, a chunk can lie
    wholly inside one row, and its pieces are added up over chunks -- flat = static to rounding (forward and one step), k = 16 and k = 64, rows of no entries beside
    the long ones."""
    from fmwr_amd import _lib as L, engine
    rng = np.random.default_rng(3)
    n, p = 36_000, 6_000
    lens = rng.poisson(25, n)
    lens[rng.integers(0, n, 400)] = 0
    heavy = rng.integers(0, n, 60)
    lens[heavy] = rng.integers(2_100, 5_900, 60)
    lens[heavy[:5] + 1 - (heavy[:5] == n - 1)] = 0          # an empty row right behind a long one
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum(lens)
    col = np.zeros(int(rp[-1]), np.uint32)
    for r in range(n):
        col[rp[r]:rp[r + 1]] = np.sort(rng.choice(p, int(lens[r]), replace=False))
    val = rng.normal(0, 0.05, len(col)).astype(np.float32)
    y = util.labels(n, 3)
    for k in (16, 64):
        w0, w, v = util.params(p, k, 3, stdev=0.02)
        out = []
        for flag in ("1", "0"):
            monkeypatch.setenv("FMX_ROWS_FLAT", flag)
            m = engine.Matrix.from_csr(rp, col, val, p, y)
            e = _engine(L, engine, p, k, batch_rows=n)
            e.set_params(w0, w, v)
            pred = e.predict(m)
            e.step(m, 0)
            out.append((pred, e.get_params()))
        scale = np.max(np.abs(out[1][0]))
        assert np.max(np.abs(out[0][0] - out[1][0])) <= 1e-12 * scale and not np.array_equal(out[0][0], out[1][0])
        assert util.rel_err(out[0][1][2], out[1][1][2]) <= 2e-6 and util.rel_err(out[0][1][1], out[1][1][1]) <= 2e-6
