"""The "w in the row" layout of the fp32 mini-batch tables (fmx_internal.h: w_in_row; chosen by itself from 3 M features up, forced
here with FMX_W_IN_ROW): V rows lie 2 * kp floats apart and a feature's linear weight sits in slot kp of its own row, so that out
of the caches a nonzero costs one memory request instead of two.  Only addresses change: every result must be BITWISE the one of
the separate-table layout -- training through every phase-2 form (dense and sparse tiles, long lists, tiles of a step, the
grad / apply split, the chunked and the compact exchange, N replicas behind one handle), prediction, row access, device init,
checkpoints."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

SOLVERS = {
    "sgd": dict(l2_w1=1e-3, l2_v=1e-3, learn_rate=0.05),
    "sgd_l1": dict(l1_w1=1e-3, l1_v=1e-3, learn_rate=0.05),
    "ftrl": dict(l1_w1=1e-3, l1_v=1e-4, l2_w1=1e-2, l2_v=1e-2),
    "tdap": dict(l1_v=1e-4, l2_w1=1e-2, l2_v=1e-2),
}


def _engine(engine, L, p, solver, k, **kw):
    sid = {"sgd": L.SOLVER_SGD, "sgd_l1": L.SOLVER_SGD, "ftrl": L.SOLVER_FTRL, "tdap": L.SOLVER_TDAP}[solver]
    return engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=sid, num_factor=k, mode=L.MODE_MINIBATCH, **SOLVERS[solver], **kw)


def _both(monkeypatch, fn):
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("FMX_W_IN_ROW", flag)
        out.append(fn())
    return out


def _same(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("k", [3, 8, 16])
@pytest.mark.parametrize("solver", ["sgd", "sgd_l1", "ftrl", "tdap"])
@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_dense_tiles_steps_of_several_tiles_and_the_split(monkeypatch, solver, k, reduce):
    from fmwr_amd import _lib as L, engine
    n, p = 3000, 400
    rp, col, val = util.random_csr(n, p, 9, seed=31)
    y = util.labels(n, 31)
    w0, w, v = util.params(p, k, 31)
    red = L.REDUCE_MEAN if reduce == "mean" else L.REDUCE_SUM

    def run():
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        res = []
        e = _engine(engine, L, p, solver, k, batch_rows=700, tile_rows=256, batch_reduce=red)     # three tiles per step, ragged last step
        e.set_params(w0, w, v)
        e.train(m, 2 * n + 150)
        res.append(e.get_params()); res.append(e.predict(m, L.LINK_LOGISTIC))
        e2 = _engine(engine, L, p, solver, k, batch_rows=700, tile_rows=256, batch_reduce=red, exchange_chunks=3)
        e2.set_params(w0, w, v)
        for s in range(5):
            e2.grad_begin(m, s % e2.num_batches(m))
            for c in range(e2.grad_layout()[0]):
                e2.grad_chunk(m, c)
            for c in range(e2.grad_layout()[0]):
                e2.apply_chunk(c, 0, c == e2.grad_layout()[0] - 1)
        e2.sync()
        res.append(e2.get_params())
        return res
    a, b = _both(monkeypatch, run)
    assert _same(a[0], b[0]) and np.array_equal(a[1], b[1]) and _same(a[2], b[2])
    assert np.any(a[0][2] != v) and np.any(a[0][1] != w)


@pytest.mark.parametrize("k", [8, 16])
@pytest.mark.parametrize("solver", ["sgd", "sgd_l1", "ftrl", "tdap"])
def test_sparse_tiles_long_lists_compact_exchange_and_replicas(monkeypatch, solver, k):
    from fmwr_amd import _lib as L, engine
    rng = np.random.default_rng(5)
    n, p, z, B = 6000, 200_000, 10, 1500
    rows = []
    for r in range(n):
        hot = [j for j, q in ((3, 0.9), (70_000, 0.4)) if rng.random() < q]
        rows.append(np.unique(np.concatenate([hot, rng.integers(0, 3000, 3), rng.integers(3000, p, z - 3)])).astype(np.uint32))
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows); val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = util.labels(n, 5)
    v0 = np.random.default_rng(2).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
    w_init = np.random.default_rng(3).normal(0, 0.05, p).astype(np.float32).astype(np.float64)

    def run():
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        e = _engine(engine, L, p, solver, k, batch_rows=B)          # fused steps over sparse tiles (the lean list-by-list form)
        e.set_params(0.1, w_init, v0)
        e.train(m, n + 700)
        out = [e.get_params()]
        g = _engine(engine, L, p, solver, k, batch_rows=B // 2, n_gpus=2, gpus_share_device=1)   # records exchanged between replicas
        g.set_params(0.1, w_init, v0)
        g.train(m, n + 700)
        out.append(g.get_params())
        ids = np.array([3, 70_000, 5, p - 1], np.uint32)
        out.append(e.get_rows(ids))
        return out
    a, b = _both(monkeypatch, run)
    assert _same(a[0], b[0]) and _same(a[1], b[1])
    assert np.array_equal(a[2][0], b[2][0]) and np.array_equal(a[2][1], b[2][1])
    assert a[0][1][3] != w_init[3] and np.any(a[0][2] != v0)


def test_row_access_device_init_and_checkpoint(monkeypatch, tmp_path):
    from fmwr_amd import _lib as L, engine
    n, p, k = 2000, 900, 16
    rp, col, val = util.random_csr(n, p, 7, seed=8)
    y = util.labels(n, 8)

    def run():
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        e = _engine(engine, L, p, "ftrl", k, batch_rows=500)
        e.init_normal(99, 0.0, 0.1)
        ids = np.array([0, 17, 899], np.uint32)
        e.set_rows(ids, w=np.array([0.5, -0.25, 2.0]))           # w alone: the V part of the rows stays
        e.set_rows(ids[:1], v=np.arange(k, dtype=np.float64).reshape(k, 1) / 64)
        e.train(m, 1500)
        path = tmp_path / "ck.fmx"
        e.save(path)
        f = _engine(engine, L, p, "ftrl", k, batch_rows=500)
        f.load(path)
        e.train(m, 1000); f.train(m, 1000)
        a, b = e.get_params(), f.get_params()
        assert _same(a, b)
        return a, e.get_rows(ids)
    a, b = _both(monkeypatch, run)
    assert _same(a[0], b[0]) and np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1])


def test_checkpoints_are_one_format_whatever_the_device_layout(monkeypatch, tmp_path):
    """The w-in-row layout is a tuning choice (p, k, FMX_W_IN_ROW at engine creation) and stays out of the file: both layouts write the SAME bytes
    (V[p][kp] then w[p], then the optimizer tables), either loads the other's file and continues bit for bit, and a round-3 file of a w-in-row
    engine (header word reserved[1] = 1: rows of 2 kp floats with w inside, no w array) is still read by both."""
    import struct
    from fmwr_amd import _lib as L, engine
    n, p, k = 2000, 900, 16
    rp, col, val = util.random_csr(n, p, 7, seed=8)
    y = util.labels(n, 8)
    files, finals = {}, {}
    for flag in ("0", "1"):
        monkeypatch.setenv("FMX_W_IN_ROW", flag)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        e = _engine(engine, L, p, "ftrl", k, batch_rows=500)
        assert e.w_in_row() == (flag == "1")
        e.init_normal(99, 0.0, 0.1)
        e.set_rows(np.array([0, 17, 899], np.uint32), w=np.array([0.5, -0.25, 2.0]))
        e.train(m, 1500)
        path = tmp_path / f"ck{flag}.fmx"
        e.save(path)
        files[flag] = open(path, "rb").read()
        e.train(m, 1000)
        finals[flag] = e.get_params()
    assert files["0"] == files["1"]
    assert _same(finals["0"], finals["1"])
    # a round-3 style file: header reserved[1] = 1, first table [p][2 kp] with w in slot kp, no w table
    raw = files["0"]
    hdr, scal = raw[:64], raw[64:64 + 12 * 8]
    kp = 16
    body = raw[64 + 12 * 8:]
    V = np.frombuffer(body[: p * kp * 4], np.float32).reshape(p, kp)
    w = np.frombuffer(body[p * kp * 4: p * kp * 4 + p * 4], np.float32)
    rest = body[p * kp * 4 + p * 4:]
    wide = np.zeros((p, 2 * kp), np.float32); wide[:, :kp] = V; wide[:, kp] = w
    words = list(struct.unpack("<4sIQiiiiI7I", hdr))
    assert words[8] == 0 and words[9] == 0          # fp32 state, canonical layout
    words[9] = 1
    legacy = tmp_path / "legacy.fmx"
    open(legacy, "wb").write(struct.pack("<4sIQiiiiI7I", *words) + scal + wide.tobytes() + rest)
    for src in ("ck0.fmx", "ck1.fmx", "legacy.fmx"):
        for flag in ("0", "1"):
            monkeypatch.setenv("FMX_W_IN_ROW", flag)
            m = engine.Matrix.from_csr(rp, col, val, p, y)
            f = _engine(engine, L, p, "ftrl", k, batch_rows=500)
            f.load(tmp_path / src)
            f.train(m, 1000)
            assert _same(f.get_params(), finals["0"]), (src, flag)
    # a file of another shape is refused, and the message names what differs
    monkeypatch.setenv("FMX_W_IN_ROW", "0")
    g = _engine(engine, L, p, "ftrl", 8, batch_rows=500)
    with pytest.raises(L.FmxError, match="does not match"):
        g.load(tmp_path / "ck0.fmx")


def test_rows_wider_than_64_bytes_keep_the_separate_tables(monkeypatch):
    """k = 32: a V row already fills its 128-byte line; the flag changes nothing."""
    from fmwr_amd import _lib as L, engine
    n, p, k = 1500, 300, 32
    rp, col, val = util.random_csr(n, p, 8, seed=4)
    y = util.labels(n, 4)
    w0, w, v = util.params(p, k, 4)

    def run():
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        e = _engine(engine, L, p, "sgd", k, batch_rows=400)
        e.set_params(w0, w, v)
        e.train(m, 2000)
        return e.get_params()
    a, b = _both(monkeypatch, run)
    assert _same(a, b)
