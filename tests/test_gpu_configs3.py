"""BASELINE.json configs[3] at its real feature shape: p = 33 M features, k = 32, 39 nnz/row (13 always-present "dense"
features + 26 skewed categorical ones), mini-batch SGD.  The oracle cannot hold k x p doubles, and does not have to: a
step only reads and writes the rows of the features that occur, so it runs on the problem restricted to those features
(ids remapped order-preservingly, which keeps every row's entry order and every feature's list order), starting from
the very V0 rows the engine holds (fmx_init_normal draws them on the device; fmx_get_rows reads them back)."""
import numpy as np
import pytest

import oracle
from tests import util

pytestmark = pytest.mark.gpu

P_FULL, K, Z = 33_000_000, 32, 39


def _criteo_shaped(n, seed):
    rng = np.random.default_rng(seed)
    dense = np.tile(np.arange(13, dtype=np.int64), (n, 1))
    cat = 13 + np.floor((P_FULL - 13) * rng.random((n, 40)) ** 2).astype(np.int64)   # skewed towards small ids: collisions between rows
    rows = []
    for r in range(n):
        u = np.unique(cat[r])
        u = np.sort(rng.choice(u, 26, replace=False))
        rows.append(np.concatenate([dense[r], u]))
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows).astype(np.uint32)
    val = np.where(col < 13, rng.random(len(col)), 1.0).astype(np.float32)   # dense features carry a value, categorical ones are one-hot
    y = np.where(rng.random(n) < 0.5, -1.0, 1.0).astype(np.float32)
    return rp, col, val, y


@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_configs3_shape_minibatch_matches_oracle_on_the_touched_features(reduce):
    from fmwr_amd import _lib as L, engine
    n, B = 6144, 2048
    rp, col, val, y = _criteo_shaped(n, 33)
    lr = 5e-5 if reduce == "sum" else 0.05   # SUM: the 13 always-present features occur 2048 times per step (lr * c must stay small)
    e = engine.Engine(P_FULL, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=K, learn_rate=lr, l2_w1=1e-4, l2_v=1e-4,
                      mode=L.MODE_MINIBATCH, batch_rows=B, batch_reduce=L.REDUCE_MEAN if reduce == "mean" else L.REDUCE_SUM)
    e.init_normal(20240001, 0.0, 0.01)
    touched = np.unique(col)
    assert 13 < len(touched) < n * Z and touched[-1] > 30_000_000
    w_t, v_t = e.get_rows(touched)
    assert np.all(w_t == 0.0) and abs(np.std(v_t) - 0.01) < 2e-4 and abs(np.mean(v_t)) < 1e-4   # N(0, 0.01) as asked
    rng = np.random.default_rng(1)
    other = np.setdiff1d(rng.integers(0, P_FULL, 5000).astype(np.uint32), touched)
    w_o, v_o = e.get_rows(other)
    m = engine.Matrix.from_csr(rp, col, val, P_FULL, y)
    rec_elems, cap, usable = e.compact_info(m)
    assert usable and rec_elems == K + 4 and cap <= B * Z     # sparse single-tile steps: no p-sized array per tile
    # the oracle on the restricted problem
    pc = len(touched)
    colc = np.searchsorted(touched, col).astype(np.uint32)
    P = oracle.params(task=oracle.CLASSIFICATION, k=K, l2_regw=1e-4, l2_regv=1e-4, learn_rate=lr, batch_mean=(reduce == "mean"))
    mb = oracle.SgdMinibatch(P, oracle.Matrix(rp, colc, val, pc), y, 0.0, w_t, v_t.ravel())
    for s in range(5):
        b = s % (n // B)
        mb.step(b * B, (b + 1) * B)
        e.step(m, b)
    e.sync()
    g_w, g_v = e.get_rows(touched)
    assert util.rel_err(g_v, mb.v.reshape(K, pc)) < 1e-5 and util.rel_err(g_w, mb.w) < 1e-5
    w0 = e.get_rows(touched[:1])  # (w0 travels with get_params only; read it through a 1-row predict instead)
    one = engine.Matrix.from_csr(np.array([0, 0], np.int64), np.zeros(0, np.uint32), np.zeros(0, np.float32), P_FULL)
    assert abs(e.predict(one)[0] - mb.w0.value) < 1e-6 * max(1.0, abs(mb.w0.value))
    # features that never occurred keep their rows bit for bit (lazy regularisation, SURVEY A-10)
    w_o2, v_o2 = e.get_rows(other)
    assert np.array_equal(w_o, w_o2) and np.array_equal(v_o, v_o2)
    # forward at this shape against the oracle on the restricted problem
    out = e.predict(m)
    ref = oracle.predict_batch(P, oracle.Matrix(rp, colc, val, pc), mb.w0.value, mb.w, mb.v)
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=2e-5)
    assert np.array_equal(np.sign(out[np.abs(ref) > 1e-4]), np.sign(ref[np.abs(ref) > 1e-4]))


@pytest.mark.parametrize("fused", ["1", "0"])
def test_streamed_training_equals_resident_training(monkeypatch, fused):
    """fmx_train_stream (rows generated step by step, each step's inverted index built two steps ahead of the step that trains on
    it, nothing kept) against the same rows trained from a resident matrix: bit for bit, for the uniform and for the
    Criteo-shaped generator, sparse and dense tiles, a ragged last step.  fused: a Criteo-shaped step's generator also writes what the
    plan builder's split pass would make of the rows (synth_fields_split_k; FMX_STREAM_FUSED=0: generate, then split)."""
    from fmwr_amd import _lib as L, engine
    monkeypatch.setenv("FMX_STREAM_FUSED", fused)
    vocab = [50_000, 20_000, 3_000, 400, 30, 4]
    cases = [dict(p=200_000, z=12, fields=None, B=3000),          # sparse tiles: 36 000 entries against 200 000 features
             dict(p=4_000, z=12, fields=None, B=3000),            # dense tiles
             dict(p=3 + sum(vocab), z=9, fields=(3, vocab, 2.5), B=2500),   # heavy hitters: dense features and the small fields
             dict(p=13 + sum(engine.CRITEO_VOCAB[8:]) + 60_000 + 9_000, z=13 + 20, fields=(13, [60_000, 9_000] + engine.CRITEO_VOCAB[8:], 3.0), B=4099)]   # 13 dense + 20 fields, steps that are no multiple of 64 rows
    for c in cases:
        n = 4 * c["B"] + 777
        kw = dict(num_factor=8, learn_rate=0.05, l2_w1=1e-3, l2_v=1e-3, mode=L.MODE_MINIBATCH, batch_rows=c["B"])
        v0 = np.random.default_rng(2).normal(0, 0.05, (8, c["p"])).astype(np.float32).astype(np.float64)
        a = engine.Engine(c["p"], **kw); a.set_params(0.0, None, v0)
        if c["fields"] is None:
            m = engine.Matrix.synthetic(n, c["p"], c["z"], 77, row_offset=1000)
            done, _ = (lambda e: e.train_stream(n, nnz_per_row=c["z"], seed=77, row_offset=1000))(a)
        else:
            m = engine.Matrix.synthetic_fields(n, c["fields"][0], c["fields"][1], c["fields"][2], 77, row_offset=1000)
            done, _ = a.train_stream(n, seed=77, row_offset=1000, fields=c["fields"])
        assert done == n and m.p == c["p"]
        b = engine.Engine(c["p"], **kw); b.set_params(0.0, None, v0)
        assert b.train(m, n) == n
        pa, pb = a.get_params(), b.get_params()
        assert pa[0] == pb[0] and np.array_equal(pa[1], pb[1]) and np.array_equal(pa[2], pb[2])
        assert np.any(pa[2] != v0)


def test_field_structured_plan_equals_the_general_sort(monkeypatch):
    """Criteo-shaped rows (dense columns first, then one one-hot entry per field): the plan builder writes the dense columns' lists
    directly and sorts only the one-hot part as (column, row) pairs (fm_ingest.hip: fields_split_k); FMX_FIELDS_SPLIT=0 sends the
    whole tile through the general (column, row, value) sort.  Same plan: the same training bit for bit -- resident tiles (several per
    matrix, a ragged last one) and streamed ones, and from a CSR upload of the same rows (which the builder does not recognise as
    field-structured: the general path again)."""
    from fmwr_amd import _lib as L, engine
    vocab = [40_000, 9_000, 700, 40, 5, 3]
    n, B, k = 4 * 2048 + 500, 2048, 8
    p = 4 + sum(vocab)
    kw = dict(num_factor=k, learn_rate=0.05, l2_w1=1e-3, l2_v=1e-3, mode=L.MODE_MINIBATCH, batch_rows=B)
    v0 = np.random.default_rng(4).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("FMX_FIELDS_SPLIT", flag)
        m = engine.Matrix.synthetic_fields(n, 4, vocab, 2.5, 31)
        e = engine.Engine(p, **kw); e.set_params(0.0, None, v0)
        assert e.train(m, n + 3 * B) == n + 3 * B
        s = engine.Engine(p, **kw); s.set_params(0.0, None, v0)
        assert s.train_stream(n, seed=31, fields=(4, vocab, 2.5))[0] == n
        res[flag] = (e.get_params(), s.get_params())
    monkeypatch.setenv("FMX_FIELDS_SPLIT", "1")
    rp, col, val, y = engine.Matrix.synthetic_fields(n, 4, vocab, 2.5, 31).export()
    u = engine.Engine(p, **kw); u.set_params(0.0, None, v0)
    assert u.train(engine.Matrix.from_csr(rp, col, val, p, y), n + 3 * B) == n + 3 * B
    same = lambda a, b: a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert same(res["1"][0], res["0"][0]) and same(res["1"][1], res["0"][1]) and same(res["1"][0], u.get_params())
    assert np.any(res["1"][0][2] != v0)


def test_per_field_sort_equals_the_pair_sort(monkeypatch):
    """Where the fields' id ranges are known (the field generator: resident matrices and streamed sources) the one-hot part is sorted field
    by field on ceil(log2 vocab) bits (fm_ingest.hip: field_sort) instead of as one array of 25-bit column ids; FMX_FIELD_SORT=0 keeps the
    pair sort.  Same plan, hence the same training bit for bit: fields of one to four passes (vocabularies of 1 .. 20 000 000),
    several 8192-entry blocks per field with a ragged last one, tiles shorter than a block, uniform and skewed ids."""
    from fmwr_amd import _lib as L, engine
    same = lambda a, b: a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for vocab, skew, B, n, d in (([5_000_000, 300_000, 513, 512, 600, 17, 2, 1], 3.0, 5 * 4096 + 100, 2 * (5 * 4096 + 100) + 900, 2),
                                 ([70_000, 5_000, 3], 1.0, 8192, 8192 + 4096, 2),
                                 ([40_000, 9_000, 700, 40, 5, 3], 2.5, 1000, 3500, 2),
                                 ([200_000, 9_000, 700, 40, 5, 3], 2.0, 5000, 12_000, 0),      # no dense part: one-hot rows (sparse tiles)
                                 ([20_000_000, 70_000, 3], 1.0, 9000, 20_000, 1)):             # 25 bits: four passes (the buffers change roles three times)
        k = 4
        p = d + sum(vocab)
        kw = dict(num_factor=k, learn_rate=0.05, l2_w1=1e-3, l2_v=1e-3, mode=L.MODE_MINIBATCH, batch_rows=B)
        v0 = np.random.default_rng(4).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
        res = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("FMX_FIELD_SORT", flag)
            m = engine.Matrix.synthetic_fields(n, d, vocab, skew, 31)
            e = engine.Engine(p, **kw); e.set_params(0.0, None, v0)
            assert e.train(m, n + B) == n + B
            s = engine.Engine(p, **kw); s.set_params(0.0, None, v0)
            assert s.train_stream(n, seed=31, fields=(d, vocab, skew))[0] == n
            res[flag] = (e.get_params(), s.get_params())
            e.close(); s.close(); m.close()
        assert same(res["1"][0], res["0"][0]) and same(res["1"][1], res["0"][1])
        assert np.any(res["1"][0][2] != v0)
    # the uniform generator: entry i of a row is a column of stratum i -- its one-hot rows are field-structured too (dense and sparse tiles)
    for p, z, B, n in ((3_000, 12, 5000, 12_345), (400_000, 30, 6000, 20_000)):
        k = 4
        kw = dict(num_factor=k, learn_rate=0.05, l2_w1=1e-3, l2_v=1e-3, mode=L.MODE_MINIBATCH, batch_rows=B)
        v0 = np.random.default_rng(4).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
        res = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("FMX_FIELD_SORT", flag)
            m = engine.Matrix.synthetic(n, p, z, 31)
            e = engine.Engine(p, **kw); e.set_params(0.0, None, v0)
            assert e.train(m, n + B) == n + B
            s = engine.Engine(p, **kw); s.set_params(0.0, None, v0)
            assert s.train_stream(n, nnz_per_row=z, seed=31)[0] == n
            res[flag] = (e.get_params(), s.get_params())
        assert same(res["1"][0], res["0"][0]) and same(res["1"][1], res["0"][1])
        assert np.any(res["1"][0][2] != v0)


def test_a_caller_can_vouch_for_the_field_layout(monkeypatch):
    """fmx_matrix_set_fields: rows uploaded as CSR (what fm.matrix hands over for a one-hot encoded data frame) with the field layout named by the
    caller take the per-field plan builder like the generator's own matrices -- same training bit for bit as the general sort -- after a check on the
    device; a layout the rows do not have is refused and leaves the matrix as it was."""
    from fmwr_amd import _lib as L, engine
    same = lambda a, b: a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for d, vocab in ((3, [50_000, 6_000, 300, 12, 2]), (0, [90_000, 5_000, 40])):
        n, B, k = 9000, 2500, 4
        p = d + sum(vocab)
        base = d + np.concatenate([[0], np.cumsum(vocab)]).astype(np.int64)
        rp, col, val, y = engine.Matrix.synthetic_fields(n, d, vocab, 2.0, 9).export()
        kw = dict(num_factor=k, learn_rate=0.05, l2_w1=1e-3, l2_v=1e-3, mode=L.MODE_MINIBATCH, batch_rows=B)
        v0 = np.random.default_rng(4).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
        res = []
        for fields in (False, True):
            m = engine.Matrix.from_csr(rp, col, val, p, y)
            if fields:
                m.set_fields(d, base)
            e = engine.Engine(p, **kw); e.set_params(0.0, None, v0)
            assert e.train(m, n + B) == n + B
            res.append(e.get_params())
        assert same(res[0], res[1]) and np.any(res[0][2] != v0)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        wrong2 = base.copy(); wrong2[1] -= 1000                  # field 0 too narrow: its larger ids are out of range
        with pytest.raises(L.FmxError, match="do not have this layout"):
            m.set_fields(d, wrong2)
        with pytest.raises(L.FmxError, match="field_base must start"):
            m.set_fields(d + 1, base)
        col2 = col.copy(); col2[d + 1] = col2[d]                 # a row with two ids of the same field (and unsorted): refused
        m2 = engine.Matrix.from_csr(rp, col2, val, p, y)
        with pytest.raises(L.FmxError, match="do not have this layout"):
            m2.set_fields(d, base)
        e = engine.Engine(p, **kw); e.set_params(0.0, None, v0)   # the refused matrix still trains (general path)
        assert e.train(m, B) == B


def test_the_field_layout_of_uploaded_rows_is_found(monkeypatch):
    """Rows handed over as CSR with one length and disjoint, ascending column ranges per entry position -- user id / item id pairs (BASELINE.json
    configs[0]'s shape), a one-hot encoded data frame with numeric columns first -- are recognised as field-structured on the device (fm_ingest.hip:
    detect_fields) and planned field by field, without a hint.  Same training bit for bit as with the pair sort (FMX_FIELD_SORT=0) and the general sort
    (FMX_FIELDS_SPLIT=0), for ids that leave gaps in their ranges; rows whose positions overlap keep the general path and train all the same."""
    from fmwr_amd import _lib as L, engine
    rng = np.random.default_rng(3)
    same = lambda a, b: a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    n, B, k = 20_000, 6000, 4
    cases = {}
    # (a) MovieLens-shaped: a user id in [0, 943), an item id in [943, 2625) -- the ids that occur start at 5 and 1000
    u = rng.integers(5, 943, n); it = rng.integers(1000, 2625, n)
    cases["user_item"] = (2625, np.stack([u, it], 1).astype(np.uint32), np.ones((n, 2), np.float32))
    # (b) two numeric columns with values, then three factors
    f1 = 2 + rng.integers(0, 50_000, n); f2 = 60_000 + rng.integers(0, 300, n); f3 = 70_000 + rng.integers(0, 3, n)
    cols = np.stack([np.zeros(n), np.ones(n), f1, f2, f3], 1).astype(np.uint32)
    vals = np.concatenate([rng.normal(0, 1, (n, 2)), np.ones((n, 3))], 1).astype(np.float32)
    cases["frame"] = (70_003, cols, vals)
    # (c) positions that overlap: three sorted draws from one range -- not fields
    tri = np.sort(np.stack([rng.choice(5000, 3, replace=False) for _ in range(n)]), 1).astype(np.uint32)
    cases["overlap"] = (5000, tri, np.ones((n, 3), np.float32))
    y = np.where(rng.random(n) < 0.5, -1.0, 1.0).astype(np.float32)
    for name, (p, cols, vals) in cases.items():
        z = cols.shape[1]
        rp = np.arange(n + 1, dtype=np.int64) * z
        kw = dict(num_factor=k, learn_rate=0.05, l2_w1=1e-3, l2_v=1e-3, mode=L.MODE_MINIBATCH, batch_rows=B)
        v0 = np.random.default_rng(4).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
        res = {}
        for env in ({}, {"FMX_FIELD_SORT": "0"}, {"FMX_FIELDS_SPLIT": "0", "FMX_FIELD_SORT": "0"}):
            for key in ("FMX_FIELD_SORT", "FMX_FIELDS_SPLIT"):
                monkeypatch.delenv(key, raising=False)
            for key, value in env.items():
                monkeypatch.setenv(key, value)
            m = engine.Matrix.from_csr(rp, cols.ravel(), vals.ravel(), p, y)
            e = engine.Engine(p, **kw); e.set_params(0.0, None, v0)
            assert e.train(m, n + B) == n + B
            res[tuple(env)] = e.get_params()
        first = res[()]
        assert all(same(first, r) for r in res.values()), name
        assert np.any(first[2] != v0)


def test_per_field_sort_at_the_full_configs3_shape(monkeypatch):
    """The same comparison at BASELINE.json configs[3]'s own step: 33 M features in 13 dense + 26 categorical fields of 3 .. 9.9 M values,
    k = 32, steps of 262 144 rows (one sparse tile of 10.2 M entries, 64 blocks per field, three sort passes for the large fields).  Three
    streamed steps with the per-field sort and with the pair sort: the rows of the features that occur are the same bit for bit."""
    from fmwr_amd import _lib as L, engine
    vocab, B = engine.CRITEO_VOCAB, 262_144
    p = 13 + sum(vocab)
    first = engine.Matrix.synthetic_fields(40_000, 13, vocab, 3.0, 77, row_offset=0)
    touched = np.unique(first.export()[1])
    first.close()
    assert len(touched) > 100_000 and touched[-1] > 30_000_000
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("FMX_FIELD_SORT", flag)
        e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=32, learn_rate=0.05, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
        e.init_normal(20240001, 0.0, 0.01)
        assert e.train_stream(3 * B, seed=77, row_offset=0, fields=(13, vocab, 3.0))[0] == 3 * B
        got[flag] = e.get_rows(touched)
        e.close()
    assert np.array_equal(got["1"][0], got["0"][0]) and np.array_equal(got["1"][1], got["0"][1])
    assert np.any(got["1"][0] != 0.0)


def test_fields_generator_shape():
    from fmwr_amd import engine
    vocab = [1000, 50, 7, 2]
    m = engine.Matrix.synthetic_fields(20_000, 3, vocab, 3.0, 5)
    rp, col, val, y = m.export()
    z = 3 + len(vocab)
    assert m.p == 3 + sum(vocab) and np.array_equal(rp, np.arange(20_001) * z)
    c = col.reshape(-1, z).astype(np.int64)
    assert np.all(np.diff(c, axis=1) > 0) and np.array_equal(c[:, :3], np.tile(np.arange(3), (20_000, 1)))
    base = 3 + np.concatenate([[0], np.cumsum(vocab)[:-1]])
    for f, v in enumerate(vocab):
        ids = c[:, 3 + f] - base[f]
        assert ids.min() >= 0 and ids.max() < v
    ids0 = c[:, 3] - base[0]
    assert np.mean(ids0 < 10) > 0.15          # skew 3: P(id < 1 % of the field) = 0.01^(1/3) = 0.215
    v = val.reshape(-1, z)
    assert np.all(v[:, 3:] == 1.0) and 0.45 < v[:, :3].mean() < 0.55 and set(np.unique(y)) == {-1.0, 1.0}
    part = engine.Matrix.synthetic_fields(500, 3, vocab, 3.0, 5, row_offset=19_500).export()   # shard independent
    assert np.array_equal(part[1], col[19_500 * z:]) and np.array_equal(part[3], y[19_500:])
