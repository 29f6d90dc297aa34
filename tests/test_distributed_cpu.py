"""world_size-2 rehearsal (gloo, CPU) of the data-parallel driver: fmwr_amd.distributed.DataParallel + shard_rows +
GradLayout are the product code under test; the per-rank compute is stood in for by the oracle (no GPU here), packed
into the same exchange-buffer layout the HIP kernels use.  The result must equal the single-process mini-batch oracle
on the union of the ranks' batches."""
import contextlib
import os
import socket

import numpy as np
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

    def __init__(self, solver, rows, problem):
        from fmwr_amd.distributed import GradLayout
        rp, col, val, y, w0, w, v = problem
        self.solver, self.r0, self.r1 = solver, rows[0], rows[1]
        self.P = _params(solver)
        self.X = oracle.Matrix(rp, col, val, P_FEAT)
        self.y = y
        self.st = _state(solver, w0, w, v)
        self.lay = GradLayout(P_FEAT, K, has_q=(solver == "ftrl"))  # this test runs FTRL with SUM, SGD with MEAN
        self.buf = torch.zeros(self.lay.size, dtype=torch.float32)

    def grad(self, batch, rows_limit=0):
        b0 = self.r0 + batch * B_LOCAL
        b1 = min(b0 + B_LOCAL, self.r1)
        acc = oracle.batch_sums(self.P, self.X, self.y, self.st["w0"].value, self.st["w"], self.st["v"], b0, b1)
        L, b = self.lay, self.buf.numpy()
        b[L.gv:L.gw] = acc["Gv"].reshape(K, P_FEAT).T.ravel()   # [p][kp] feature-major, as the kernels store it
        b[L.gw:L.cnt] = acc["Gw"]; b[L.cnt:L.qv] = acc["cw"]
        if L.has_q:
            b[L.qv:L.qw] = acc["Qv"].reshape(K, P_FEAT).T.ravel(); b[L.qw:L.tail] = acc["Qw"]
        b[L.tail:L.tail + 4] = [acc["G0"], acc["Q0"], b1 - b0, 0.0]

    def buffer(self):
        return self.buf

    def comm_context(self):
        return contextlib.nullcontext()

    def apply(self):
        L, b = self.lay, self.buf.numpy().astype(np.float64)
        acc = dict(G0=b[L.tail], Q0=b[L.tail + 1], Gw=b[L.gw:L.cnt].copy(), cw=b[L.cnt:L.qv].copy(),
                   Gv=b[L.gv:L.gw].reshape(P_FEAT, K).T.ravel().copy())
        acc["Qw"] = b[L.qw:L.tail].copy() if L.has_q else np.zeros(P_FEAT)
        acc["Qv"] = b[L.qv:L.qw].reshape(P_FEAT, K).T.ravel().copy() if L.has_q else np.zeros(K * P_FEAT)
        if self.solver == "sgd":
            oracle.sgd_apply_sums(self.P, P_FEAT, self.st, b[L.tail + 2], acc)
        else:
            oracle.ftrl_apply_sums(self.P, P_FEAT, self.st, b[L.tail + 2], acc)


def _worker(rank, world, port, solver, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fmwr_amd.distributed import DataParallel, shard_rows
    stepper = OracleStepper(solver, shard_rows(N, rank, world), _problem())
    dp = DataParallel(stepper)
    for s in range(STEPS):
        dp.step(s % 4)
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
            b0 = r0 + (s % 4) * B_LOCAL; b1 = min(b0 + B_LOCAL, r1)
            acc = oracle.batch_sums(P, X, y, st["w0"].value, st["w"], st["v"], b0, b1, acc)
            rows += b1 - b0
        if solver == "sgd":
            oracle.sgd_apply_sums(P, P_FEAT, st, float(rows), acc)
        else:
            oracle.ftrl_apply_sums(P, P_FEAT, st, float(rows), acc)
    return st


def _run(solver, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), solver, str(tmp_path)), nprocs=world, join=True)
    exp = _expected(solver, world)
    got = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    # replicas are bit-identical ...
    for key in ("w0", "w", "v"):
        np.testing.assert_array_equal(got[0][key], got[1][key])
    # ... and equal the single-process result up to the fp32 rounding of the exchanged sums
    assert util.rel_err(got[0]["v"], exp["v"]) < 1e-5
    assert util.rel_err(got[0]["w"], exp["w"]) < 1e-5
    assert abs(float(got[0]["w0"]) - exp["w0"].value) < 1e-6


def test_data_parallel_sgd_world2(tmp_path):
    _run("sgd", tmp_path)


def test_data_parallel_ftrl_world2(tmp_path):
    _run("ftrl", tmp_path)


def test_shard_rows_partition():
    from fmwr_amd.distributed import shard_rows
    for n in (0, 1, 7, 10_000_000):
        for world in (1, 2, 3, 8):
            cuts = [shard_rows(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
