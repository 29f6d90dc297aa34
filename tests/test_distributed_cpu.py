"""world_size-2 rehearsal (gloo, CPU) of the data-parallel driver: fmwr_amd.distributed.DataParallel + shard_rows +
GradLayout are the product code under test; the per-rank compute is stood in for by the oracle (no GPU here), packed
into the same exchange-buffer layout the HIP kernels use.  The result must equal the single-process mini-batch oracle
on the union of the ranks' batches."""
import contextlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from tests import util

N, P_FEAT, K, B_LOCAL, STEPS = 900, 120, 4, 100, 6


def _problem():
    rp, col, val = util.random_csr(N, P_FEAT, 8, seed=21)
    y = util.labels(N, 21)
    w0, w, v = util.params(P_FEAT, K, 21, fp32=False)
    return rp, col, val, y, w0, w, v


def _params(solver):
    if solver == "sgd":
        return oracle.params(k=K, l2_regw=1e-3, l2_regv=2e-3, l2_reg0=1e-3, learn_rate=0.05, batch_mean=True)
    return oracle.params(k=K, l1_regw=1e-3, l1_regv=1e-3, l2_regw=1e-2, l2_regv=1e-2, batch_mean=False)


def _state(solver, w0, w, v):
    import ctypes as C
    st = dict(w0=C.c_double(w0), w=w.copy(), v=v.ravel().copy())
    if solver == "sgd":
        st.update(q_w=np.zeros(P_FEAT), q_v=np.zeros(K * P_FEAT), u=np.zeros(2))
    else:
        st.update(zn0=np.zeros(2), z_w=np.zeros(P_FEAT), n_w=np.zeros(P_FEAT), z_v=np.zeros(K * P_FEAT), n_v=np.zeros(K * P_FEAT))
    return st


class OracleStepper:
    """CPU stand-in with the engine stepper's interface and buffer layout (fp32 buffer, like the GPU's)."""

    def __init__(self, solver, rows, problem, chunks=1):
        from fmwr_amd.distributed import GradLayout
        rp, col, val, y, w0, w, v = problem
        self.solver, self.r0, self.r1 = solver, rows[0], rows[1]
        self.P = _params(solver)
        self.X = oracle.Matrix(rp, col, val, P_FEAT)
        self.y = y
        self.st = _state(solver, w0, w, v)
        self.lay = GradLayout(P_FEAT, K, has_q=(solver == "ftrl"), chunks=chunks)  # this test runs FTRL with SUM, SGD with MEAN
        self.n_chunks = self.lay.n_chunks
        self.buf = torch.zeros(self.lay.size, dtype=torch.float32)

    def _sums(self, batch):
        b0 = self.r0 + batch * B_LOCAL
        b1 = min(b0 + B_LOCAL, self.r1)
        return oracle.batch_sums(self.P, self.X, self.y, self.st["w0"].value, self.st["w"], self.st["v"], b0, b1), b1 - b0

    def _pack(self, acc, c):
        L, b = self.lay, self.buf.numpy()
        (f0, f1), (e0, _) = L.features(c), L.block(c)
        n = f1 - f0
        gv = acc["Gv"].reshape(K, P_FEAT).T  # [p][kp] feature-major, as the kernels store it
        b[e0 + L.gv:e0 + L.gv + n * K] = gv[f0:f1].ravel()
        b[e0 + L.gw:e0 + L.gw + n] = acc["Gw"][f0:f1]; b[e0 + L.cnt:e0 + L.cnt + n] = acc["cw"][f0:f1]
        if L.has_q:
            b[e0 + L.qv:e0 + L.qv + n * K] = acc["Qv"].reshape(K, P_FEAT).T[f0:f1].ravel(); b[e0 + L.qw:e0 + L.qw + n] = acc["Qw"][f0:f1]

    def _unpack(self, acc, c):
        L, b = self.lay, self.buf.numpy().astype(np.float64)
        (f0, f1), (e0, _) = L.features(c), L.block(c)
        n = f1 - f0
        acc["Gv"][f0:f1] = b[e0 + L.gv:e0 + L.gv + n * K].reshape(n, K)
        acc["Gw"][f0:f1] = b[e0 + L.gw:e0 + L.gw + n]; acc["cw"][f0:f1] = b[e0 + L.cnt:e0 + L.cnt + n]
        if L.has_q:
            acc["Qv"][f0:f1] = b[e0 + L.qv:e0 + L.qv + n * K].reshape(n, K); acc["Qw"][f0:f1] = b[e0 + L.qw:e0 + L.qw + n]

    def _finish(self, acc):
        b = self.buf.numpy().astype(np.float64)
        t = self.lay.tail
        full = dict(G0=b[t], Q0=b[t + 1], Gw=acc["Gw"], cw=acc["cw"], Qw=acc["Qw"], Gv=acc["Gv"].T.ravel().copy(), Qv=acc["Qv"].T.ravel().copy())
        (oracle.sgd_apply_sums if self.solver == "sgd" else oracle.ftrl_apply_sums)(self.P, P_FEAT, self.st, b[t + 2] * 4096.0 + b[t + 3], full)

    @staticmethod
    def _empty():
        return dict(Gv=np.zeros((P_FEAT, K)), Qv=np.zeros((P_FEAT, K)), Gw=np.zeros(P_FEAT), Qw=np.zeros(P_FEAT), cw=np.zeros(P_FEAT))

    # unchunked interface
    def grad(self, batch, rows_limit=0):
        self.grad_begin(batch, rows_limit)
        for c in range(self.n_chunks):
            self.grad_chunk(c)

    def buffer(self):
        return self.buf

    def comm_context(self):
        return contextlib.nullcontext()

    def apply(self):
        for c in range(self.n_chunks):
            self.apply_chunk(c, c == self.n_chunks - 1)

    # chunked interface
    def grad_begin(self, batch, rows_limit=0):
        self.acc, rows = self._sums(batch)
        t = self.lay.tail
        self.buf.numpy()[t:t + 4] = [self.acc["G0"], self.acc["Q0"], rows // 4096, rows % 4096]  # the row count travels in two parts
        self.reduced = self._empty()

    def grad_chunk(self, c):
        self._pack(self.acc, c)

    def tail(self):
        return self.buf[self.lay.tail:self.lay.tail + 4]

    def chunk(self, c):
        e0, e1 = self.lay.block(c)
        return self.buf[e0:e1]

    def apply_chunk(self, c, last):
        self._unpack(self.reduced, c)
        if last:
            self._finish(self.reduced)  # the oracle applies all coordinates at once; coordinates are independent


    # compact interface: one record per occurring feature, G[K] | (Q[K]) | Gw | Qw | cnt | id (bit pattern), ascending ids
    device = torch.device("cpu")

    def compact_usable(self):
        self.rec_elems = K * (2 if self.lay.has_q else 1) + 4
        return True

    def compact_counts(self):
        nb = -(-(self.r1 - self.r0) // B_LOCAL)
        return [int(np.count_nonzero(self._sums(b)[0]["cw"])) for b in range(nb)]

    def compact_reserve(self, cap):
        self.rec = torch.zeros(cap, self.rec_elems, dtype=torch.float32)
        self.ctail = torch.zeros(4, dtype=torch.float32)

    def grad_compact(self, batch, rows_limit=0):
        acc, rows = self._sums(batch)
        ids = np.flatnonzero(acc["cw"]).astype(np.uint32)
        if getattr(self, "world", 1) > 1:
            ids = ids[np.argsort(ids % self.world, kind="stable")]   # owner-major records
        r = self.rec.numpy()
        q = K if self.lay.has_q else 0
        r[:len(ids), :K] = acc["Gv"].reshape(K, P_FEAT).T[ids]
        if q:
            r[:len(ids), K:2 * K] = acc["Qv"].reshape(K, P_FEAT).T[ids]
        r[:len(ids), K + q] = acc["Gw"][ids]
        r[:len(ids), K + q + 1] = acc["Qw"][ids] if q else 0.0
        r[:len(ids), K + q + 2] = acc["cw"][ids]
        r[:len(ids), K + q + 3] = ids.view(np.float32)
        self.ctail.numpy()[:] = [acc["G0"], acc["Q0"], rows // 4096, rows % 4096]

    def compact_tail(self):
        return self.ctail

    def compact_send(self, n):
        return self.rec[:n]

    def compact_recv(self, world, n):
        self._recv = torch.zeros(world * n, self.rec_elems, dtype=torch.float32)
        return self._recv

    def apply_compact(self, recv, counts, stride):
        """merge by feature id; a feature's parts are added in rank order, in the buffer's element type (fp32), exactly as an
        all-reduce(sum) of the dense buffer adds two ranks"""
        r = recv.numpy()
        q = K if self.lay.has_q else 0
        red = {key: val.astype(np.float32) for key, val in self._empty().items()}
        for part, cnt in enumerate(counts):
            blk = r[part * stride:part * stride + int(cnt)]
            ids = np.ascontiguousarray(blk[:, K + q + 3]).view(np.uint32)
            red["Gv"][ids] = red["Gv"][ids] + blk[:, :K]
            if q:
                red["Qv"][ids] = red["Qv"][ids] + blk[:, K:2 * K]
                red["Qw"][ids] = red["Qw"][ids] + blk[:, K + q + 1]
            red["Gw"][ids] = red["Gw"][ids] + blk[:, K + q]
            red["cw"][ids] = red["cw"][ids] + blk[:, K + q + 2]
        red = {key: val.astype(np.float64) for key, val in red.items()}
        b = self.ctail.numpy().astype(np.float64)
        full = dict(G0=b[0], Q0=b[1], Gw=red["Gw"], cw=red["cw"], Qw=red["Qw"], Gv=red["Gv"].T.ravel().copy(), Qv=red["Qv"].T.ravel().copy())
        (oracle.sgd_apply_sums if self.solver == "sgd" else oracle.ftrl_apply_sums)(self.P, P_FEAT, self.st, b[2] * 4096.0 + b[3], full)


    # owner-sharded exchange (include/fmx.h): feature j belongs to rank j mod N; records and ids travel in owner-major order (a stable
    # partition of the ascending ids), rows come back in the STATE's type (fp64 here, like the oracle's state)
    m = object()   # "a resident matrix": the per-step counts are known up front

    def num_batches(self):
        return -(-(self.r1 - self.r0) // B_LOCAL)

    def owner_configure(self, world, rank):
        self.world, self.rank = world, rank

    def owner_usable(self):
        return self.compact_usable()

    def _owner_order(self, batch):
        ids = np.flatnonzero(self._sums(batch)[0]["cw"]).astype(np.uint32)
        order = np.argsort(ids % self.world, kind="stable")
        return ids, order

    def owner_counts(self, batch):
        ids, _ = self._owner_order(batch)
        return np.bincount(ids % self.world, minlength=self.world).astype(np.int64)

    def owner_ids(self, batch, n):
        ids, order = self._owner_order(batch)
        assert n == len(ids)
        return torch.from_numpy(ids[order].astype(np.int32))

    def state_dtype(self):
        return torch.float64

    def ids_buffer(self, n):
        return torch.zeros(n, dtype=torch.int32)

    def rows_buffer(self, which, n):
        return torch.zeros(n, K + 4, dtype=torch.float64)

    def rows_pack(self, ids, out):
        j = ids.numpy().astype(np.int64)
        o = out.numpy()
        o[:, :K] = self.st["v"].reshape(K, P_FEAT)[:, j].T
        o[:, K] = self.st["w"][j]
        o[:, K + 1:] = 0.0

    def rows_unpack(self, ids, rows):
        j = ids.numpy().astype(np.int64)
        r = rows.numpy()
        self.st["v"].reshape(K, P_FEAT)[:, j] = r[:, :K].T
        self.st["w"][j] = r[:, K]

    def records_view(self, n):
        return self.rec[:n]

    def records_buffer(self, n):
        return torch.zeros(n, self.rec_elems, dtype=torch.float32)

    def apply_parts(self, recv, counts, starts):
        """the owner's merge: parts at explicit positions, added in rank order in the buffer's element type"""
        r = recv.numpy()
        q = K if self.lay.has_q else 0
        red = {key: val.astype(np.float32) for key, val in self._empty().items()}
        for cnt, st0 in zip(counts, starts):
            blk = r[int(st0):int(st0) + int(cnt)]
            ids = np.ascontiguousarray(blk[:, K + q + 3]).view(np.uint32)
            assert np.all(ids % self.world == self.rank) and np.all(np.diff(ids.astype(np.int64)) > 0)
            red["Gv"][ids] = red["Gv"][ids] + blk[:, :K]
            if q:
                red["Qv"][ids] = red["Qv"][ids] + blk[:, K:2 * K]
                red["Qw"][ids] = red["Qw"][ids] + blk[:, K + q + 1]
            red["Gw"][ids] = red["Gw"][ids] + blk[:, K + q]
            red["cw"][ids] = red["cw"][ids] + blk[:, K + q + 2]
        red = {key: val.astype(np.float64) for key, val in red.items()}
        b = self.ctail.numpy().astype(np.float64)
        full = dict(G0=b[0], Q0=b[1], Gw=red["Gw"], cw=red["cw"], Qw=red["Qw"], Gv=red["Gv"].T.ravel().copy(), Qv=red["Qv"].T.ravel().copy())
        (oracle.sgd_apply_sums if self.solver == "sgd" else oracle.ftrl_apply_sums)(self.P, P_FEAT, self.st, b[2] * 4096.0 + b[3], full)


def _worker(rank, world, port, solver, out_dir, chunks, exchange="dense"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fmwr_amd.distributed import DataParallel, shard_rows
    stepper = OracleStepper(solver, shard_rows(N, rank, world), _problem(), chunks)
    if exchange == "owner":
        stepper.compact_usable(); stepper.compact_reserve(P_FEAT)
    dp = DataParallel(stepper, exchange=exchange)
    assert dp.exchange == exchange
    for s in range(STEPS):
        dp.step(s % (4 if world == 2 else 3))
    if exchange == "owner":
        assert len(dp.bytes_sent) == STEPS and all(b > 0 for b in dp.bytes_sent)
        mine_before = {key: np.array(stepper.st[key]).copy() for key in ("w", "v")}
        dp.pull_all(P_FEAT)   # refresh the copies of the features other ranks own
        own = np.arange(rank, P_FEAT, world)
        assert np.array_equal(mine_before["w"][own], stepper.st["w"][own])   # the owner's rows are never overwritten by anyone else's
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), w0=stepper.st["w0"].value, w=stepper.st["w"], v=stepper.st["v"])
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _expected(solver, world):
    from fmwr_amd.distributed import shard_rows
    rp, col, val, y, w0, w, v = _problem()
    P = _params(solver)
    X = oracle.Matrix(rp, col, val, P_FEAT)
    st = _state(solver, w0, w, v)
    for s in range(STEPS):
        acc, rows = None, 0
        for r in range(world):
            r0, r1 = shard_rows(N, r, world)
            b0 = r0 + (s % (4 if world == 2 else 3)) * B_LOCAL; b1 = min(b0 + B_LOCAL, r1)
            acc = oracle.batch_sums(P, X, y, st["w0"].value, st["w"], st["v"], b0, b1, acc)
            rows += b1 - b0
        if solver == "sgd":
            oracle.sgd_apply_sums(P, P_FEAT, st, float(rows), acc)
        else:
            oracle.ftrl_apply_sums(P, P_FEAT, st, float(rows), acc)
    return st


def _run(solver, tmp_path, chunks=1, exchange="dense", world=2):
    mp.spawn(_worker, args=(world, _free_port(), solver, str(tmp_path), chunks, exchange), nprocs=world, join=True)
    exp = _expected(solver, world)
    got = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    # replicas are bit-identical ...
    for r in range(1, world):
        for key in ("w0", "w", "v"):
            np.testing.assert_array_equal(got[0][key], got[r][key])
    # ... and equal the single-process result up to the fp32 rounding of the exchanged sums
    assert util.rel_err(got[0]["v"], exp["v"]) < 1e-5
    assert util.rel_err(got[0]["w"], exp["w"]) < 1e-5
    assert abs(float(got[0]["w0"]) - exp["w0"].value) < 1e-6


def test_data_parallel_sgd_world2(tmp_path):
    _run("sgd", tmp_path)


def test_data_parallel_ftrl_world2(tmp_path):
    _run("ftrl", tmp_path)


def test_data_parallel_pipelined_exchange_world2(tmp_path):
    """exchange_chunks > 1: tail first, then one asynchronous all-reduce per block of features (2 blocks of 64 features)."""
    _run("sgd", tmp_path, chunks=2)
    _run("ftrl", tmp_path, chunks=2)


@pytest.mark.parametrize("solver", ["sgd", "ftrl"])
def test_compact_exchange_is_bitwise_the_dense_all_reduce_world2(tmp_path, solver):
    """The touched-row exchange (all-gather of one record per occurring feature, merged in rank order) against the dense
    all-reduce of the (k + 2) * p buffer: the same parameters bit for bit on both replicas, for SGD (MEAN) and FTRL (SUM,
    which also exchanges the sums of squares)."""
    d = tmp_path / "dense"; c = tmp_path / "compact"
    d.mkdir(); c.mkdir()
    _run(solver, d)
    _run(solver, c, exchange="compact")
    for r in range(2):
        a, b = np.load(d / f"rank{r}.npz"), np.load(c / f"rank{r}.npz")
        for key in ("w0", "w", "v"):
            np.testing.assert_array_equal(a[key], b[key])


@pytest.mark.parametrize("solver", ["sgd", "ftrl"])
def test_owner_sharded_exchange_is_bitwise_the_dense_all_reduce_world2(tmp_path, solver):
    """records routed to the rank that owns the feature (id mod N), the owner adds the parts in rank order and
    updates its slice, current rows are pulled by whoever reads them next.  With two ranks the additions are those of the dense
    all-reduce: the same parameters bit for bit (every rank, after a final pull of the rows it does not own)."""
    d = tmp_path / "dense"; o = tmp_path / "owner"
    d.mkdir(); o.mkdir()
    _run(solver, d)
    _run(solver, o, exchange="owner")
    for r in range(2):
        a, b = np.load(d / f"rank{r}.npz"), np.load(o / f"rank{r}.npz")
        for key in ("w0", "w", "v"):
            np.testing.assert_array_equal(a[key], b[key])


@pytest.mark.parametrize("solver", ["sgd", "ftrl"])
def test_owner_sharded_exchange_world3_equals_the_all_gather_form(tmp_path, solver):
    """Three ranks: a feature's parts are added in rank order by its owner, exactly as every replica adds them after the all-gather
    of records (bitwise).  The dense all-reduce of three ranks adds in the ring's order, which no exchange of records can
    reproduce bit for bit (fp32 addition does not associate): against it the bar is the fp32 rounding of the sums."""
    c = tmp_path / "compact"; o = tmp_path / "owner"; d = tmp_path / "dense"
    c.mkdir(); o.mkdir(); d.mkdir()
    _run(solver, c, exchange="compact", world=3)
    _run(solver, o, exchange="owner", world=3)
    _run(solver, d, world=3)
    for r in range(3):
        a, b, dd = np.load(c / f"rank{r}.npz"), np.load(o / f"rank{r}.npz"), np.load(d / f"rank{r}.npz")
        for key in ("w0", "w", "v"):
            np.testing.assert_array_equal(a[key], b[key])
        assert util.rel_err(b["v"], dd["v"]) < 1e-6 and util.rel_err(b["w"], dd["w"]) < 1e-6


def test_grad_layout_blocks():
    from fmwr_amd.distributed import GradLayout
    one = GradLayout(1_000_000, 16)
    assert (one.n_chunks, one.F, one.size) == (1, 1_000_000, 1_000_000 * 18 + 4)
    L = GradLayout(1_000_000, 16, has_q=True, chunks=8)
    assert L.F % 64 == 0 and L.n_chunks == 8 and L.F * 8 >= 1_000_000 > L.F * 7
    assert L.block_elems == L.F * (2 * 16 + 3) and L.tail == 8 * L.block_elems and L.size == L.tail + 4
    assert L.features(7) == (7 * L.F, 1_000_000)
    assert GradLayout(10, 4).F == 12  # one block, feature count rounded up to a multiple of 4: planes stay 16-byte aligned


def test_shard_rows_partition():
    from fmwr_amd.distributed import shard_rows
    for n in (0, 1, 7, 10_000_000):
        for world in (1, 2, 3, 8):
            cuts = [shard_rows(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no torch.distributed.run around it: the parent only spawns the launcher as a child (no torch import, no HIP call in
    the parent) and relays the ranks' return code.  Here there is no GPU, so both ranks must stop with bench.py's own message, and the code must come back."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=root)
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr, r.stderr[-1500:]
    assert "launch N > 1 through" not in r.stderr
