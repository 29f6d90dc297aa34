"""A RESIDENT matrix with more than 2^32 stored entries.  The reference's SMatrix counts entries in `uint` (util/Smatrix.h:10-17:
size, row_idx) and cannot hold one; here row offsets are int64 and every per-tile structure is tile-relative.  150 M rows x 30 = 4.5e9 entries
(CSR 38 GB + per-tile plans 20 GB of the 288 GB): the rows past entry 2^32 (row 143 165 577 on) are the ones every check looks at.

  * forward: closed form on a 7 M-row slab straddling the boundary, and a slab generated on its own (row_offset) bit for bit;
  * the first SUM-mode SGD step of a batch past the boundary against a host bincount of the exported rows (fmx_matrix_export past 2^32 too);
  * the per-tile plan of the whole matrix + one full pass (573 steps), twice: the same bits;
  * the whole-matrix CSC of the ALS sweeps is REFUSED above 2^32 entries (FMX_ERR_INVALID, the limit stated) rather than built unverified.
"""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

N, P, Z, K, SEED, B = 150_000_000, 1_000_000, 30, 16, 20240001, 262_144
EDGE = (1 << 32) // Z          # the row holding entry 2^32


@pytest.fixture(scope="module")
def huge():
    from fmwr_amd import engine
    m = engine.Matrix.synthetic(N, P, Z, SEED)
    assert m.nnz == N * Z > (1 << 32)
    yield m
    m.close()


def _predict_slab(L, e, m, r0, r1):
    buf = util.DevBuf(r1 - r0)
    L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(r0), C.c_int64(r1), buf.ptr, C.c_int(L.LINK_NONE)))
    e.sync()
    out = buf.numpy()
    buf.free()
    return out


def test_forward_past_the_2p32_boundary(huge):
    from fmwr_amd import _lib as L, engine
    e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH)
    a = np.arange(1, K + 1) / 64.0
    e.set_params(0.25, np.full(P, 0.5), np.repeat(a[:, None], P, axis=1))
    want = 0.25 + Z * 0.5 + 0.5 * (Z * Z - Z) * np.sum(a * a)
    out = _predict_slab(L, e, huge, EDGE - 3_000_000, EDGE + 4_000_000)
    np.testing.assert_allclose(out, want, rtol=1e-13)
    np.testing.assert_allclose(_predict_slab(L, e, huge, N - 1_000_000, N), want, rtol=1e-13)
    # with random parameters every row's prediction depends on its own columns: rows past the boundary against the same rows generated on their own
    e.init_normal(7, 0.0, 0.1)
    r0 = EDGE + 1_234_567
    sub = engine.Matrix.synthetic(50_000, P, Z, SEED, row_offset=r0)
    np.testing.assert_array_equal(_predict_slab(L, e, huge, r0, r0 + 50_000), e.predict(sub))
    # and the export of those rows is the generator's stream
    rp, col, val, y = huge.export(r0, r0 + 50_000)
    rp2, col2, val2, y2 = sub.export()
    assert np.array_equal(rp, rp2) and np.array_equal(col, col2) and np.array_equal(y, y2)


def test_first_step_of_a_batch_past_the_boundary_against_host_counts(huge):
    from fmwr_amd import _lib as L, engine
    lr = 0.125
    e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH, batch_rows=B, batch_reduce=L.REDUCE_SUM, learn_rate=lr)
    nb = e.num_batches(huge)
    assert nb == -(-N // B)
    for batch in (EDGE // B, EDGE // B + 14, nb - 1):     # the batch that straddles entry 2^32, one past it, the ragged last one
        e.set_params(0.0, None, None)
        e.step(huge, batch)
        e.sync()
        w0, w, v = e.get_params()
        r0 = batch * B
        rp, col, val, y = huge.export(r0, min(r0 + B, N))
        assert w0 == lr * 0.5 * float(np.sum(y.astype(np.float64)))
        want = lr * 0.5 * np.bincount(col, weights=np.repeat(y.astype(np.float64), Z), minlength=P)
        np.testing.assert_allclose(w, want, rtol=0, atol=1e-6)
        assert np.all(v == 0.0)


def test_one_full_pass_twice_gives_the_same_bits(huge):
    from fmwr_amd import _lib as L, engine
    ids = np.arange(0, P, 997, dtype=np.uint32)
    res = []
    for _ in range(2):
        e = engine.Engine(P, num_factor=K, mode=L.MODE_MINIBATCH, batch_rows=B, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4)
        e.init_normal(SEED, 0.0, 0.01)
        done = e.train(huge, N)
        assert done == N
        w0 = e.get_rows(ids)
        res.append((w0[0], w0[1]))
        e.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.all(np.isfinite(res[0][1])) and np.any(res[0][0] != 0.0)


def test_the_whole_matrix_csc_is_refused_above_2p32_entries(huge):
    """The ALS / MCMC sweeps need the CSC of the WHOLE matrix (one sort over all entries, 32-bit positions inside a column list): stated limit."""
    from fmwr_amd import _lib as L, engine
    e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=4, mode=L.MODE_SEQUENTIAL)
    with pytest.raises(L.FmxError, match="2\\^32"):
        e.als_plan(huge)
