"""Properties at BASELINE.json's full size (10M x 1M, 30 nnz/row, k=16) that need no CPU pass over the whole matrix,
plus the synthetic generator against an independent numpy Philox4x32-10."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

N, P, Z, K, SEED, B = 10_000_000, 1_000_000, 30, 16, 20240001, 262_144


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(x, np.uint64) for x in (c0, c1, c2, c3))
    k0 = np.uint64(k0); k1 = np.uint64(k1)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M
        n1 = p1 & M
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M
        n3 = p0 & M
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    return c0, c1, c2, c3


def numpy_synthetic(n, p, z, seed, row_offset):
    g = np.arange(row_offset, row_offset + n, dtype=np.uint64)
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    col = np.zeros((n, z), np.uint32)
    for blk in range((z + 3) // 4):
        words = philox4x32_10(g & np.uint64(0xFFFFFFFF), g >> np.uint64(32), np.full(n, blk, np.uint64), np.zeros(n, np.uint64), k0, k1)
        for j in range(4):
            i = blk * 4 + j
            if i >= z:
                break
            lo, hi = (i * p) // z, ((i + 1) * p) // z
            col[:, i] = lo + ((words[j] * np.uint64(hi - lo)) >> np.uint64(32)).astype(np.uint32)
    lab = philox4x32_10(g & np.uint64(0xFFFFFFFF), g >> np.uint64(32), np.full(n, 0xFFFFFFFF, np.uint64), np.zeros(n, np.uint64), k0, k1)[0]
    y = np.where(lab & np.uint64(1), 1.0, -1.0).astype(np.float32)
    return col, y


@pytest.fixture(scope="module")
def fm():
    from fmwr_amd import engine, _lib
    return engine, _lib


@pytest.fixture(scope="module")
def big(fm):
    engine, L = fm
    return engine.Matrix.synthetic(N, P, Z, SEED)


@pytest.fixture(scope="module")
def big_iid(fm):
    """the headline's own matrix (bench.py's default: SURVEY 8(d)'s primary column law, fmx_matrix_synthetic_iid)"""
    engine, L = fm
    return engine.Matrix.synthetic_iid(N, P, Z, SEED, law=L.COLUMNS_UNIFORM)


def test_generator_matches_numpy_philox(fm):
    engine, L = fm
    for (n, p, z, off) in [(5000, 1000, 7, 0), (3000, 1_000_000, 30, 9_999_000), (100, 64, 64, 2 ** 33)]:
        m = engine.Matrix.synthetic(n, p, z, SEED, row_offset=off)
        rp, col, val, y = m.export()
        ecol, ey = numpy_synthetic(n, p, z, SEED, off)
        np.testing.assert_array_equal(rp, np.arange(n + 1) * z)
        np.testing.assert_array_equal(col.reshape(n, z), ecol)
        np.testing.assert_array_equal(y, ey)
        assert np.all(val == 1.0) and np.all(np.diff(col.reshape(n, z).astype(np.int64), axis=1) > 0)
    # shard independence: rows [a, b) of the stream are the same whoever generates them
    whole = engine.Matrix.synthetic(4000, 5000, 9, 5).export()
    part = engine.Matrix.synthetic(1000, 5000, 9, 5, row_offset=3000).export()
    np.testing.assert_array_equal(whole[1][3000 * 9:], part[1])
    np.testing.assert_array_equal(whole[3][3000:], part[3])


def test_fields_generator_matches_numpy_philox(fm):
    """The Criteo-shaped generator (fm_ingest.hip: synth_fields_k) against the same independent numpy Philox: entry i of row g is word i mod 4 of
    the block keyed (g, i / 4, 0xF1E1D5); a dense feature's value is the word / 2^32 in fp32, a categorical id floor(vocab * u^skew) clamped into
    the field, its value 1."""
    engine, L = fm
    vocab, d, skew, seed = [5_000_000, 70_000, 900, 17, 3, 1], 3, 3.0, 77
    for n, off in ((4000, 0), (1000, 2 ** 33 + 5)):
        m = engine.Matrix.synthetic_fields(n, d, vocab, skew, seed, row_offset=off)
        rp, col, val, y = m.export()
        z = d + len(vocab)
        g = np.arange(off, off + n, dtype=np.uint64)
        k0, k1 = seed & 0xFFFFFFFF, seed >> 32
        base = d + np.concatenate([[0], np.cumsum(vocab)[:-1]])
        ecol = np.zeros((n, z), np.uint32); eval_ = np.ones((n, z), np.float32)
        for blk in range((z + 3) // 4):
            words = philox4x32_10(g & np.uint64(0xFFFFFFFF), g >> np.uint64(32), np.full(n, blk, np.uint64), np.full(n, 0xF1E1D5, np.uint64), k0, k1)
            for j in range(4):
                i = blk * 4 + j
                if i >= z:
                    break
                u = words[j].astype(np.float64) / 4294967296.0
                if i < d:
                    ecol[:, i] = i; eval_[:, i] = u.astype(np.float32)
                else:
                    f = i - d
                    ecol[:, i] = base[f] + np.minimum(np.floor(u * u * u * vocab[f]).astype(np.int64), vocab[f] - 1)
        np.testing.assert_array_equal(col.reshape(n, z), ecol)
        np.testing.assert_array_equal(val.reshape(n, z), eval_)
        np.testing.assert_array_equal(rp, np.arange(n + 1) * z)


def test_full_size_forward_closed_form(fm, big):
    """Every row holds exactly Z distinct features with x = 1: with w = c and every V row = a,
    y_hat = w0 + Z*c + 0.5*(Z*Z - Z)*sum(a^2) for all 10M rows."""
    engine, L = fm
    e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH)
    a = (np.arange(1, K + 1) / 64.0)
    e.set_params(0.25, np.full(P, 0.5), np.repeat(a[:, None], P, axis=1))
    out = e.predict(big)
    want = 0.25 + Z * 0.5 + 0.5 * (Z * Z - Z) * np.sum(a * a)
    assert out.shape == (N,)
    np.testing.assert_allclose(out, want, rtol=1e-13)
    # geometry independence: a slab predicted on its own is bitwise the same
    import ctypes as C
    sub = engine.Matrix.synthetic(50_000, P, Z, SEED, row_offset=7_000_000)
    np.testing.assert_array_equal(e.predict(sub), out[7_000_000:7_050_000])


def test_full_size_first_step_against_host_counts(fm, big):
    """From w = 0, V = 0 every example has y_hat = 0, mult = -y/2.  One SUM-mode step therefore gives
    w0 = lr/2 * sum(y) and w_j = lr/2 * sum of the labels of the batch rows holding j: checks the w0 reduction and the
    per-batch CSC at full size against a host bincount on the exported batch."""
    engine, L = fm
    lr = 0.125
    e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH, batch_rows=B, batch_reduce=L.REDUCE_SUM, learn_rate=lr)
    nb = e.num_batches(big)
    assert nb == -(-N // B)
    for batch in (0, nb - 1):  # a full batch and the ragged last one
        e.set_params(0.0, None, None)
        e.step(big, batch)
        e.sync()
        w0, w, v = e.get_params()
        r0 = batch * B
        rp, col, val, y = big.export(r0, min(r0 + B, N))
        assert w0 == lr * 0.5 * float(np.sum(y.astype(np.float64)))
        want = lr * 0.5 * np.bincount(col, weights=np.repeat(y.astype(np.float64), Z), minlength=P)
        np.testing.assert_allclose(w, want, rtol=0, atol=1e-6)
        assert np.all(v == 0.0)  # s_f = 0 and v = 0: the pairwise gradient vanishes


def test_full_size_training_is_reproducible_and_lr0_is_identity(fm, big):
    engine, L = fm
    v0 = np.random.default_rng(3).normal(0, 0.01, (K, P)).astype(np.float32).astype(np.float64)
    res = []
    for _ in range(2):
        e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH, batch_rows=B, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4)
        e.set_params(0.0, None, v0)
        for b in range(6):
            e.step(big, b)
        e.sync()
        res.append(e.get_params())
    assert res[0][0] == res[1][0]
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])
    assert np.all(np.isfinite(res[0][2])) and np.any(res[0][2] != v0)
    e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH, batch_rows=B, learn_rate=0.0)
    e.set_params(0.5, None, v0)
    e.step(big, 1)
    e.sync()
    w0, w, v = e.get_params()
    assert w0 == 0.5 and np.all(w == 0) and np.array_equal(v, v0)


def test_full_size_iid_headline_matrix(fm, big_iid):
    """The 10 M x 1 M matrix the headline is quoted on (i.i.d. uniform columns, sorted inside the row, repeats bumped: Z distinct columns per row) -- the closed-form
    forward, the first SUM-mode step against host counts (a full batch and the ragged last one), bitwise reproducibility of six steps, lr = 0 as the identity, a
    slab generated on its own equal to the same rows of the whole (the full-size properties ran on the stratified generator only)."""
    engine, L = fm
    rp, col, val, y = big_iid.export(0, 200_000)
    c = col.reshape(-1, Z).astype(np.int64)
    assert np.all(np.diff(rp) == Z) and np.all(val == 1.0) and np.all(np.diff(c, axis=1) > 0) and c.max() < P
    # columns are NOT stratified here: position i of a row ranges over (almost) all of [0, p)
    assert c[:, 0].max() > P // 4 and c[:, Z - 1].min() < 3 * P // 4
    e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH)
    a = (np.arange(1, K + 1) / 64.0)
    e.set_params(0.25, np.full(P, 0.5), np.repeat(a[:, None], P, axis=1))
    out = e.predict(big_iid)
    np.testing.assert_allclose(out, 0.25 + Z * 0.5 + 0.5 * (Z * Z - Z) * np.sum(a * a), rtol=1e-13)
    sub = engine.Matrix.synthetic_iid(50_000, P, Z, SEED, law=L.COLUMNS_UNIFORM, row_offset=7_000_000)
    np.testing.assert_array_equal(sub.export()[1], big_iid.export(7_000_000, 7_050_000)[1])
    np.testing.assert_array_equal(e.predict(sub), out[7_000_000:7_050_000])
    e.close(); sub.close()
    lr = 0.125
    e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH, batch_rows=B, batch_reduce=L.REDUCE_SUM, learn_rate=lr)
    nb = e.num_batches(big_iid)
    for batch in (0, nb - 1):
        e.set_params(0.0, None, None)
        e.step(big_iid, batch)
        e.sync()
        w0, w, v = e.get_params()
        r0 = batch * B
        _, bcol, _, by = big_iid.export(r0, min(r0 + B, N))
        assert w0 == lr * 0.5 * float(np.sum(by.astype(np.float64)))
        np.testing.assert_allclose(w, lr * 0.5 * np.bincount(bcol, weights=np.repeat(by.astype(np.float64), Z), minlength=P), rtol=0, atol=1e-6)
        assert np.all(v == 0.0)
    e.close()
    v0 = np.random.default_rng(3).normal(0, 0.01, (K, P)).astype(np.float32).astype(np.float64)
    res = []
    for _ in range(2):
        e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH, batch_rows=B, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4)
        e.set_params(0.0, None, v0)
        for b in range(6):
            e.step(big_iid, b)
        e.sync()
        res.append(e.get_params())
        e.close()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.all(np.isfinite(res[0][2])) and np.any(res[0][2] != v0)
    e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH, batch_rows=B, learn_rate=0.0)
    e.set_params(0.5, None, v0)
    e.step(big_iid, 1)
    e.sync()
    w0, w, v = e.get_params()
    assert w0 == 0.5 and np.all(w == 0) and np.array_equal(v, v0)
    e.close()


def test_full_size_iid_reference_order_learners_agree(fm, big_iid):
    """The reference-order SGD learner over 150 000 examples of the headline matrix (random strides 1..3, three launches): the reassociated form (cfg.seq_reassociate)
    within 1e-10 of the bitwise windowed kernel on V, w and w0, and bit for bit itself on a second run."""
    import oracle
    engine, L = fm
    order = oracle.visit_order(N, 3, 150_000, seed=11)
    v0 = np.random.default_rng(6).normal(0, 0.01, (K, P))
    out = []
    for re in (1, 1, 0):
        e = engine.Engine(P, solver=L.SOLVER_SGD, num_factor=K, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_SEQUENTIAL, seq_reassociate=re)
        e.set_params(0.0, None, v0)
        e.train_order(big_iid, order)
        out.append(e.get_params())
        e.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert abs(out[0][0] - out[2][0]) < 1e-10 and util.rel_err(out[0][2], out[2][2]) < 1e-10 and util.rel_err(out[0][1], out[2][1]) < 1e-10
    assert np.any(out[0][2] != v0)


def test_full_size_tiling_and_exchange_forms_agree(fm, big):
    """At full size: a 1 048 576-row step cut into 4 or 2 tiles, run fused, through the grad/apply split, and through the
    pipelined split (8 blocks) -- the same step.  Splits are bitwise equal to each other; tile counts and the fused form
    differ only in the association of the tile sums (fp32 exchange buffer)."""
    engine, L = fm
    v0 = np.random.default_rng(5).normal(0, 0.01, (K, P)).astype(np.float32).astype(np.float64)
    kw = dict(num_factor=K, mode=L.MODE_MINIBATCH, batch_rows=1 << 20, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4)
    def run(form, **extra):
        e = engine.Engine(P, **dict(kw, **extra)); e.set_params(0.0, None, v0)
        for b in (0, 3):
            if form == "fused":
                e.step(big, b)
            elif form == "split":
                e.grad(big, b); e.apply(0)
            else:
                e.grad_begin(big, b)
                nc = e.grad_layout()[0]
                for c in range(nc):
                    e.grad_chunk(big, c)
                for c in range(nc):
                    e.apply_chunk(c, 0, c == nc - 1)
        e.sync()
        return e.get_params()
    fused4 = run("fused")
    fused2 = run("fused", tile_rows=1 << 19)
    split = run("split")
    chunked = run("chunked", exchange_chunks=8)
    split8 = run("split", exchange_chunks=8)
    assert chunked[0] == split8[0] and np.array_equal(chunked[1], split8[1]) and np.array_equal(chunked[2], split8[2])
    scale = np.max(np.abs(fused4[2]))
    for other in (fused2, split, chunked):
        assert np.max(np.abs(other[2] - fused4[2])) < 2e-6 * scale
        assert np.max(np.abs(other[1] - fused4[1])) < 2e-6 * max(np.max(np.abs(fused4[1])), 1e-6)
        assert abs(other[0] - fused4[0]) < 1e-9


def test_full_size_sequential_window_equals_one_wave(fm, big):
    """The reference's algorithm over 150 000 examples of the full-size matrix (random strides 1..3): the windowed kernel
    (groups of ~8 feature-disjoint examples) and the strictly serial one-wave kernel give bitwise the same model."""
    import os
    engine, L = fm
    import oracle
    order = oracle.visit_order(N, 3, 150_000, seed=11)
    v0 = np.random.default_rng(6).normal(0, 0.01, (K, P))
    out = []
    for win in ("1", "0"):
        os.environ["FMX_SEQ_WINDOW"] = win
        try:
            e = engine.Engine(P, solver=L.SOLVER_FTRL, num_factor=K, l1_w1=1e-4, l1_v=1e-4, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_SEQUENTIAL)
            e.set_params(0.0, None, v0)
            e.train_order(big, order)
            out.append(e.get_params())
        finally:
            os.environ.pop("FMX_SEQ_WINDOW", None)
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert np.any(out[0][2] != v0)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2] at FULL size: 10M x 1M, k = 64, FTRL (l1 + l2).  Properties that need no CPU pass over the matrix.
K2 = 64
FTRL2 = dict(alpha_w=0.1, alpha_v=0.1, beta_w=1.0, beta_v=1.0, l1_w1=1e-4, l1_v=1e-4, l2_w1=1e-4, l2_v=1e-4)


def test_configs2_full_size_first_ftrl_step_against_host_counts(fm, big):
    """From w = 0, V = 0, z = n = 0 every example has y_hat = 0 and multiplier -y/2, so one SUM-mode FTRL step has a closed
    form per feature (FTRL_Learner.h:88-98, :172-183 with the batch sums): G_j = -1/2 * sum of the labels of the batch rows
    holding j, n_j = c_j / 4, z_j = G_j, w_j = prox(z_j, n_j); w0 likewise from all rows; V stays 0 (s_f = 0, v = 0).
    Checked on a full and on the ragged last batch against a host bincount of the exported rows."""
    engine, L = fm
    e = engine.Engine(P, solver=L.SOLVER_FTRL, num_factor=K2, mode=L.MODE_MINIBATCH, batch_rows=B, batch_reduce=L.REDUCE_SUM, **FTRL2)
    nb = e.num_batches(big)
    a, b_, l1, l2 = FTRL2["alpha_w"], FTRL2["beta_w"], FTRL2["l1_w1"], FTRL2["l2_w1"]
    for batch in (1, nb - 1):
        e.set_params(0.0, None, None)
        e.step(big, batch)
        e.sync()
        w0, w, v = e.get_params()
        r0 = batch * B
        rp, col, val, y = big.export(r0, min(r0 + B, N))
        yy = y.astype(np.float64)
        G0, n0 = -0.5 * yy.sum(), 0.25 * len(yy)
        assert abs(w0 - (-G0 * a / (b_ + np.sqrt(n0)))) < 1e-15
        G = -0.5 * np.bincount(col, weights=np.repeat(yy, Z), minlength=P)
        c = np.bincount(col, minlength=P).astype(np.float64)
        want = np.where(np.abs(G) <= l1, 0.0, -(G - np.sign(G) * l1) / ((b_ + np.sqrt(0.25 * c)) / a + l2))
        np.testing.assert_allclose(w, want, rtol=0, atol=2e-7)   # fp32 state
        assert np.all(w[c == 0] == 0.0) and np.all(v == 0.0)


def test_configs2_full_size_l1_threshold_zeroes_exactly_the_touched_rows(fm, big):
    """With an l1 weight no |z| can exceed, the prox sets every coordinate it touches to exactly 0 and leaves the rest alone
    (FTRL_Learner.h:177,194): after one step V is 0 on precisely the features that occur in the batch -- the per-tile
    inverted index at full size, k = 64, against the exported rows."""
    engine, L = fm
    v0 = np.random.default_rng(8).normal(0, 0.01, (K2, P)).astype(np.float32).astype(np.float64)
    e = engine.Engine(P, solver=L.SOLVER_FTRL, num_factor=K2, mode=L.MODE_MINIBATCH, batch_rows=65_536, **dict(FTRL2, l1_v=1e9, l1_w1=1e9))
    e.set_params(0.0, None, v0)
    e.step(big, 5)
    e.sync()
    _, w, v = e.get_params()
    rp, col, val, y = big.export(5 * 65_536, 6 * 65_536)
    touched = np.zeros(P, bool); touched[col] = True
    assert 0.5 < touched.mean() < 0.95
    assert np.all(v[:, touched] == 0.0) and np.array_equal(v[:, ~touched], v0[:, ~touched])


def test_configs2_full_size_ftrl_is_reproducible_and_forms_agree(fm, big):
    """k = 64 FTRL, 524 288-row steps (the engine's tile size at this width): two runs bit for bit; the fused step, the
    grad/apply split and 2 tiles per step agree up to the fp32 rounding of the sums between tiles."""
    engine, L = fm
    v0 = np.random.default_rng(9).normal(0, 0.01, (K2, P)).astype(np.float32).astype(np.float64)
    def run(form, **extra):
        e = engine.Engine(P, solver=L.SOLVER_FTRL, num_factor=K2, mode=L.MODE_MINIBATCH, batch_rows=524_288, **dict(FTRL2, **extra))
        e.set_params(0.0, None, v0)
        for b in (0, 2, 4):
            if form == "fused":
                e.step(big, b)
            else:
                e.grad(big, b); e.apply(0)
        e.sync()
        return e.get_params()
    a, b_, c, d = run("fused"), run("fused"), run("split"), run("fused", tile_rows=262_144)
    assert a[0] == b_[0] and np.array_equal(a[1], b_[1]) and np.array_equal(a[2], b_[2])
    assert np.all(np.isfinite(a[2])) and np.any(a[2] != v0)
    scale = np.max(np.abs(a[2]))
    for other in (c, d):
        assert np.max(np.abs(other[2] - a[2])) < 5e-6 * scale and abs(other[0] - a[0]) < 1e-8


def test_iid_generators_rows_are_strictly_ascending_and_follow_their_law(fm):
    """SURVEY 8(d)'s i.i.d. column laws (fmx_matrix_synthetic_iid): rows strictly ascending and in range, labels as the other
    generators', shard independent; uniform columns are flat over [0, p), Zipf(1.05) columns pile up on the first ids."""
    engine, L = fm
    n, p, z = 20_000, 50_000, 30
    for law, s_exp in ((L.COLUMNS_UNIFORM, 1.05), (L.COLUMNS_ZIPF, 1.05)):
        m = engine.Matrix.synthetic_iid(n, p, z, 9, law, s_exp)
        rp, col, val, y = m.export()
        c = col.reshape(n, z).astype(np.int64)
        assert np.array_equal(rp, np.arange(n + 1) * z) and c.min() >= 0 and c.max() < p
        assert np.all(np.diff(c, axis=1) > 0) and np.all(val == 1.0) and set(np.unique(y)) == {-1.0, 1.0}
        head = np.mean(c < p // 100)
        assert (0.005 < head < 0.02) if law == L.COLUMNS_UNIFORM else head > 0.4
        part = engine.Matrix.synthetic_iid(1000, p, z, 9, law, s_exp, row_offset=n - 1000).export()
        assert np.array_equal(part[1], col[(n - 1000) * z:]) and np.array_equal(part[3], y[n - 1000:])


@pytest.mark.parametrize("solver", ["sgd", "ftrl"])
def test_zipf_columns_train_like_the_oracle(fm, solver):
    """SURVEY 8(d)'s conflict-stress variant at test size: Zipf(1.05) columns (a few features occur in most rows).  Mini-batch steps: the head features' lists
    are long lists (segments of 1 024 entries on the side stream beside the short-list kernel); sequential learner: nearly every example conflicts with its
    predecessor, so the windowed / pipelined kernels run groups of one.  Both against the oracle on the exported rows."""
    import oracle
    engine, L = fm
    n, p, z, k = 12_000, 30_000, 30, 16
    m = engine.Matrix.synthetic_iid(n, p, z, 21, L.COLUMNS_ZIPF, 1.05)
    rp, col, val, y = m.export()
    counts = np.bincount(col, minlength=p)
    assert counts.max() > 4000 and np.sum(counts > 64) > 50          # heavy hitters: long lists in every 4 000-row step
    w0, w, v = util.params(p, k, 5, stdev=0.05, fp32=True)
    X = oracle.Matrix(rp, col, val, p)
    ftrl = solver == "ftrl"
    P = oracle.params(task=oracle.CLASSIFICATION, k=k, l2_regw=1e-4, l2_regv=1e-4, learn_rate=0.01, **(dict(l1_regw=1e-4, l1_regv=1e-4) if ftrl else {}))
    kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_FTRL if ftrl else L.SOLVER_SGD, num_factor=k, l2_w1=1e-4, l2_v=1e-4, learn_rate=0.01,
              l1_w1=1e-4 if ftrl else 0.0, l1_v=1e-4 if ftrl else 0.0)
    e = engine.Engine(p, mode=L.MODE_MINIBATCH, batch_rows=4_000, **kw)
    e.set_params(w0, w, v)
    np.testing.assert_allclose(e.predict(m), oracle.predict_batch(P, X, w0, w, v.ravel()), rtol=0, atol=1e-5)
    e.train(m, n)
    mb = (oracle.FtrlMinibatch if ftrl else oracle.SgdMinibatch)(P, X, y, w0, w, v.ravel())
    for b in range(0, n, 4_000):
        mb.step(b, b + 4_000)
    g0, gw, gv = e.get_params()
    assert util.rel_err(gv, mb.v.reshape(k, p)) < 1e-5 and util.rel_err(gw, mb.w) < 1e-5 and abs(g0 - mb.w0.value) < 1e-5
    es = engine.Engine(p, mode=L.MODE_SEQUENTIAL, **kw)
    es.set_params(w0, w, v)
    es.train(m, 3_000)
    ref = (oracle.ftrl_learn if ftrl else oracle.sgd_learn)(P, X, y, w0, w, v.ravel(), 3_000)
    s0, sw, sv = es.get_params()
    assert util.rel_err(sv, ref["v"].reshape(k, p)) < 1e-10 and util.rel_err(sw, ref["w"]) < 1e-10 and abs(s0 - ref["w0"]) < 1e-10
    assert np.array_equal(np.sign(es.predict(m)), np.sign(oracle.predict_batch(P, X, ref["w0"], ref["w"], ref["v"])))
