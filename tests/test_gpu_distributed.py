"""Rehearsal of the N > 1 path on ONE GPU: two ranks share device 0 and exchange through gloo (RCCL needs one GPU per
rank; the driver runs the real N = 2/4/8 RCCL bench on an 8-GPU node).  Checks that bench.py's multi-rank path runs and
that two data-parallel replicas reproduce the single-process result on the same global batches."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from fmwr_amd import _lib as L, engine
from fmwr_amd.distributed import DataParallel, EngineStepper, shard_rows
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
n, p, z, k, B = 40000, 5000, 10, 16, 4000
r0, r1 = shard_rows(n, rank, world)
m = engine.Matrix.synthetic(r1 - r0, p, z, 7, row_offset=r0, device=0)
kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD if sys.argv[2] == "sgd" else L.SOLVER_FTRL, num_factor=k, learn_rate=0.05,
          l2_w1=1e-3, l2_v=1e-3, l1_v=1e-4 if sys.argv[2] == "ftrl" else 0.0, mode=L.MODE_MINIBATCH, batch_rows=B // world,
          state_fp64=int(sys.argv[4]), exchange_chunks=int(sys.argv[5]))
e = engine.Engine(p, **kw)
v0 = np.random.default_rng(1).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
e.set_params(0.0, None, v0)
st = EngineStepper(e, m, 0)
assert st.buffer().dtype == (torch.float64 if int(sys.argv[4]) else torch.float32) and st.buffer().numel() == e.grad_buffer()[1]
dp = DataParallel(st)
for s in range(8):
    dp.step(s % e.num_batches(m))
e.sync()
w0, w, v = e.get_params()
if rank == 0:
    np.savez(sys.argv[3], w0=w0, w=w, v=v)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("chunks", [1, 3])
@pytest.mark.parametrize("state_fp64", [0, 1])
@pytest.mark.parametrize("solver", ["sgd", "ftrl"])
def test_two_replicas_match_single_process(tmp_path, solver, state_fp64, chunks):
    from fmwr_amd import _lib as L, engine
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "dp.npz"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", str(script), ROOT, solver, str(out), str(state_fp64), str(chunks)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    # single process: the same global batches = union of the two ranks' local batches
    n, p, z, k, B = 40000, 5000, 10, 16, 4000
    halves = [engine.Matrix.synthetic(n // 2, p, z, 7, row_offset=r * (n // 2)) for r in range(2)]
    rows = [hm.export() for hm in halves]
    nb = (n // 2) // (B // 2)
    # interleave: global batch b = rank0 batch b ++ rank1 batch b
    rp = [0]; col = []; val = []; y = []
    for b in range(nb):
        for (rpi, ci, vi, yi) in rows:
            a, c = b * (B // 2), (b + 1) * (B // 2)
            col.append(ci[rpi[a]:rpi[c]]); val.append(vi[rpi[a]:rpi[c]]); y.append(yi[a:c])
            rp.extend((rpi[a + 1:c + 1] - rpi[a] + rp[-1]).tolist())
    m = engine.Matrix.from_csr(np.array(rp, np.int64), np.concatenate(col), np.concatenate(val), p, np.concatenate(y))
    kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD if solver == "sgd" else L.SOLVER_FTRL, num_factor=k, learn_rate=0.05,
              l2_w1=1e-3, l2_v=1e-3, l1_v=1e-4 if solver == "ftrl" else 0.0, mode=L.MODE_MINIBATCH, batch_rows=B, state_fp64=state_fp64)
    e = engine.Engine(p, **kw)
    e.set_params(0.0, None, np.random.default_rng(1).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64))
    for s in range(8):
        e.step(m, s % nb)
    e.sync()
    w0, w, v = e.get_params()
    scale = np.max(np.abs(v))
    tol = 1e-11 if state_fp64 else 1e-5  # fp64 state: only the association of the two ranks' sums differs
    assert np.max(np.abs(got["v"] - v)) < tol * scale
    assert np.max(np.abs(got["w"] - w)) < tol * max(np.max(np.abs(w)), 1e-3)
    assert abs(float(got["w0"]) - w0) < tol * max(1.0, abs(w0))


def test_bench_multirank_path_runs(tmp_path):
    env = dict(os.environ, FMX_BENCH_SHARED_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29518", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--rows", "400000", "--features", "50000", "--batch-rows", "32768", "--backend", "gloo", "--cpu-rows", "0"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["global_batch_rows"] == 65536 and d["config"]["tile_rows"] == 32768


def test_the_bare_n_gpu_command_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` with NO launcher (the driver's N = 1 command with N replaced): bench.py starts torch.distributed.run as a child
    process before anything touches the GPU and relays rank 0's line and the return code (VERDICT r4 missing 1)."""
    env = dict(os.environ, FMX_BENCH_SHARED_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--rows", "400000", "--features", "50000", "--batch-rows", "32768", "--backend", "gloo", "--cpu-rows", "0"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["global_batch_rows"] == 65536 and d["config"]["parallelism"] == "dp2"
    # a failing rank's code comes back through the launcher
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stream", "--backend", "gloo", "--cpu-rows", "0"],
                         capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert bad.returncode != 0


COMPACT_WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from fmwr_amd import _lib as L, engine
from fmwr_amd.distributed import DataParallel, EngineStepper, shard_rows
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
solver, out, wide, reduce = sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
# far more features than entries per step (the regime of BASELINE.json configs[3]): every tile is sparse
n, p, z, k, B = 24000, 300000, 12, 8, 3000
r0, r1 = shard_rows(n, rank, world)
rng = np.random.default_rng(100 + rank)
rows = []
for r in range(r1 - r0):   # two heavy hitters (long lists) + cold features, some shared between the ranks
    hot = [j for j, q in ((5, 0.9), (70000, 0.4)) if rng.random() < q]
    rows.append(np.unique(np.concatenate([hot, rng.integers(0, 4000, 3), rng.integers(4000, p, z - 3)])).astype(np.uint32))
rp = np.zeros(len(rows) + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
col = np.concatenate(rows); val = rng.normal(0, 1, len(col)).astype(np.float32)
y = np.where(rng.random(len(rows)) < 0.5, -1.0, 1.0).astype(np.float32)
kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD if solver.startswith("sgd") else L.SOLVER_FTRL, num_factor=k, learn_rate=0.05,
          l2_w1=1e-3, l2_v=1e-3, l1_v=1e-4 if solver != "sgd" else 0.0, l1_w1=1e-4 if solver == "sgd_l1" else 0.0,
          mode=L.MODE_MINIBATCH, batch_rows=B // world, state_fp64=wide, batch_reduce=L.REDUCE_MEAN if reduce == "mean" else L.REDUCE_SUM)
v0 = np.random.default_rng(1).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
res = {}
for exchange in ("dense", "compact"):
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    e = engine.Engine(p, **kw)
    e.set_params(0.0, None, v0)
    dp = DataParallel(EngineStepper(e, m, 0, dense=(exchange == "dense")), exchange=exchange)
    assert dp.exchange == exchange, dp.exchange
    nb = e.num_batches(m)
    moved = 0
    for s in range(7):
        dp.step(s % nb, rows_limit=(B // world - 100) if s == 3 else 0)   # one truncated step
        moved += dp.last_exchange_bytes
    e.sync()
    res[exchange] = e.get_params() + (moved,)
a, b = res["dense"], res["compact"]
assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), "compact exchange differs from the dense all-reduce"
assert np.any(a[2] != v0)
if rank == 0:
    np.savez(out, w0=b[0], w=b[1], v=b[2], dense_bytes=a[3], compact_bytes=b[3])
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("solver,wide,reduce", [("sgd", 0, "mean"), ("sgd_l1", 0, "sum"), ("ftrl", 0, "sum"), ("ftrl", 1, "mean")])
def test_compact_exchange_equals_dense_all_reduce_two_ranks(tmp_path, solver, wide, reduce):
    """configs[3]'s exchange: p >> entries per step, so each rank publishes one record per OCCURRING feature (all-gather,
    merged in rank order) instead of all-reducing the (kp + 2) * p buffer.  Two ranks on one GPU through gloo: both forms
    leave bitwise the same parameters (asserted inside the workers, heavy hitters and a truncated step included), and the
    compact form moves a small fraction of the bytes."""
    script = tmp_path / "worker.py"
    script.write_text(COMPACT_WORKER)
    out = tmp_path / "dp.npz"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", str(script), ROOT, solver, str(out), str(wide), reduce], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    assert got["compact_bytes"] * 5 < got["dense_bytes"]
    assert np.all(np.isfinite(got["v"]))


RCCL_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from fmwr_amd import _lib as L, engine
from fmwr_amd.distributed import DataParallel, EngineStepper
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))     # RCCL, one rank: every collective really runs
mode = sys.argv[2]
if mode == "dense":
    n, p, z, k, B = 40000, 5000, 10, 16, 4000
    m = engine.Matrix.synthetic(n, p, z, 7, device=0)
    kw = dict(exchange_chunks=3)
else:
    n, p, z, k, B = 6000, 400000, 12, 8, 2000                          # entries per step < features: sparse tiles, compact exchange
    m = engine.Matrix.synthetic(n, p, z, 7, device=0)
    kw = {}
mk = lambda **extra: engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_FTRL, num_factor=k, l1_v=1e-4, l2_w1=1e-3, l2_v=1e-3,
                                   mode=L.MODE_MINIBATCH, batch_rows=B, **extra)
v0 = np.random.default_rng(1).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
def run(force):
    os.environ["FMX_DP_FORCE_COLLECTIVES"] = "1" if force else "0"
    e = mk(**kw); e.set_params(0.0, None, v0)
    st = EngineStepper(e, m, 0, dense=(mode == "dense"))
    dp = DataParallel(st, exchange=mode)
    assert dp.exchange == mode and dp.collective == force
    for s in range(6):
        dp.step(s % e.num_batches(m))
    e.sync(); torch.cuda.synchronize()
    return e.get_params()
a, b = run(True), run(False)      # through RCCL (identities with one rank) / the same step with the collectives left out
same = a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
print("RCCL_WORLD1", mode, "bitwise" if same else "DIFFERENT", flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if same else 3)
'''


@pytest.mark.parametrize("mode", ["dense", "compact"])
def test_the_rccl_backend_itself_runs_the_exchange_world1(tmp_path, mode):
    """The N > 1 drivers' plumbing on the REAL backend: torch.distributed "nccl" (= RCCL) with one rank on the one GPU.  The tensors
    made from the engine's device pointers, the engine's stream as a torch ExternalStream, the asynchronous chunked all-reduce and
    the padded all-gather of records all execute (FMX_DP_FORCE_COLLECTIVES=1); with one rank they are identities, so the result must
    be bitwise that of the same steps with the collectives left out."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", str(script), ROOT, mode], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "bitwise" in r.stdout, (r.stdout[-1500:], r.stderr[-2500:])


OWNER_WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from fmwr_amd import _lib as L, engine
from fmwr_amd.distributed import DataParallel, EngineStepper, shard_rows, train_stream
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
solver, out, wide, reduce, mode = sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6]
kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD if solver.startswith("sgd") else L.SOLVER_FTRL, learn_rate=0.05,
          l2_w1=1e-3, l2_v=1e-3, l1_v=1e-4 if solver != "sgd" else 0.0, l1_w1=1e-4 if solver == "sgd_l1" else 0.0,
          mode=L.MODE_MINIBATCH, state_fp64=wide, batch_reduce=L.REDUCE_MEAN if reduce == "mean" else L.REDUCE_SUM)
res = {}
if mode == "resident":
    # far more features than entries per step (BASELINE.json configs[3]'s regime): every tile is sparse; heavy hitters (long lists)
    n, p, z, k, B = 24000, 300000, 12, 8, 3000
    r0, r1 = shard_rows(n, rank, world)
    rng = np.random.default_rng(100 + rank)
    rows = []
    for r in range(r1 - r0):
        hot = [j for j, q in ((5, 0.9), (70000, 0.4), (70001, 0.3)) if rng.random() < q]
        rows.append(np.unique(np.concatenate([hot, rng.integers(0, 4000, 3), rng.integers(4000, p, z - 3)])).astype(np.uint32))
    rp = np.zeros(len(rows) + 1, np.int64); rp[1:] = np.cumsum([len(x) for x in rows])
    col = np.concatenate(rows); val = rng.normal(0, 1, len(col)).astype(np.float32)
    y = np.where(rng.random(len(rows)) < 0.5, -1.0, 1.0).astype(np.float32)
    v0 = np.random.default_rng(1).normal(0, 0.05, (k, p)).astype(np.float32).astype(np.float64)
    for exchange in ("dense", "compact", "owner"):
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        e = engine.Engine(p, num_factor=k, batch_rows=B // world, **kw)
        e.set_params(0.0, None, v0)
        dp = DataParallel(EngineStepper(e, m, 0, dense=(exchange == "dense")), exchange=exchange)
        assert dp.exchange == exchange, dp.exchange
        nb = e.num_batches(m)
        moved = 0
        for s in range(7):
            dp.step(s % nb, rows_limit=(B // world - 100) if s == 3 else 0)   # one truncated step
            moved += dp.last_exchange_bytes
        if exchange == "owner":
            dp.pull_all(p)          # refresh the copies of the rows other ranks own
        e.sync()
        res[exchange] = e.get_params() + (moved,)
else:
    # streamed steps (fmx_source): Criteo-shaped fields, every rank its own row range of the stream
    vocab = [60000, 30000, 9000, 700, 40, 5]
    p, k, B, steps = 6 + sum(vocab), 8, 2048, 5
    total = steps * B * world - 300              # the last global step is ragged
    r0, r1 = shard_rows(total, rank, world)
    for exchange in ("compact", "owner"):
        e = engine.Engine(p, num_factor=k, batch_rows=B, **kw)
        e.init_normal(11, 0.0, 0.05)
        dp = DataParallel(EngineStepper(e, None, 0, dense=False), exchange=exchange)
        src = e.source(r1 - r0, seed=3, row_offset=r0, fields=(6, vocab, 2.0))
        trace = []
        done = train_stream(dp, src, on_step=(lambda t: trace.append(e.get_w0())) if os.environ.get("FMX_TEST_TRACE_W0") == "1" else None)
        if trace:
            print(f"W0TRACE rank {rank} {exchange}: " + " ".join(f"{x!r}" for x in trace), flush=True)
        assert done == r1 - r0
        src.close()
        if exchange == "owner":
            dp.pull_all(p)
        e.sync()
        res[exchange] = e.get_params() + (sum(dp.bytes_sent) if exchange == "owner" else 0,)
    res["dense"] = res["compact"]
a, c, o = res["dense"], res["compact"], res["owner"]
same = lambda x, y: x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2])
if not same(c, o):   # say WHERE: which features (owner = id % world), how far apart
    dw = np.flatnonzero(c[1] != o[1]); dv = np.flatnonzero(np.any(c[2] != o[2], axis=0))
    print(f"MISMATCH rank {rank}: w0 {c[0]!r} vs {o[0]!r}; w differs at {len(dw)} ids, V at {len(dv)} ids of {p}", flush=True)
    for j in list(dv[:12]) + list(dw[:6]):
        print(f"  id {j} owner {j % world} dw {o[1][j] - c[1][j]:.3e} max dV {np.max(np.abs(o[2][:, j] - c[2][:, j])):.3e} v0? {bool(np.all(o[2][:, j] == 0))}", flush=True)
assert same(c, o), "owner-sharded exchange differs from the all-gather of records"
if world == 2:
    assert same(a, o), "owner-sharded exchange differs from the dense all-reduce"
if rank == 0:
    np.savez(out, w0=o[0], w=o[1], v=o[2], dense_bytes=a[3], compact_bytes=c[3], owner_bytes=o[3])
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,solver,wide,reduce,mode", [
    (2, "sgd", 0, "mean", "resident"), (2, "sgd_l1", 0, "sum", "resident"), (2, "ftrl", 0, "sum", "resident"), (2, "ftrl", 1, "mean", "resident"),
    (3, "sgd", 0, "mean", "resident"), (3, "ftrl", 0, "sum", "resident"),
    (2, "sgd", 0, "mean", "stream"), (3, "ftrl", 0, "sum", "stream")])
def test_owner_sharded_exchange_on_the_gpu(tmp_path, world, solver, wide, reduce, mode):
    """SURVEY 8(e)(ii) / VERDICT r2 items 1-2: records routed to the rank that owns the feature (id mod N), the owner adds a
    feature's parts in rank order and applies the update to ITS slice, the rows a step reads are pulled from their owners first.
    Ranks on one GPU through gloo.  Bitwise the all-gather form for any N (the same additions in the same order) and bitwise the dense
    all-reduce for N = 2 (asserted inside the workers; heavy hitters, a truncated step, three solvers, fp64 state).  mode = stream:
    every step is a streamed tile (fmx_source) of the rank's own row range -- configs[3] as one N-GPU job."""
    script = tmp_path / "worker.py"
    script.write_text(OWNER_WORKER)
    out = tmp_path / "dp.npz"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", "29521", str(script), ROOT, solver, str(out), str(wide), reduce, mode], capture_output=True, text=True, env=env, timeout=600)
    if r.returncode != 0 and os.path.isdir(os.path.join(ROOT, "gpurun_out")):   # pytest clips long assertion messages: keep the workers' own words
        open(os.path.join(ROOT, "gpurun_out", f"fail_owner_{world}_{solver}_{mode}.txt"), "w").write(r.stdout + "\n==== stderr ====\n" + r.stderr)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = np.load(out)
    assert np.all(np.isfinite(got["v"])) and got["owner_bytes"] > 0
    if mode == "resident":
        assert got["owner_bytes"] < got["compact_bytes"] or world == 2   # 2 x (N-1)/N x records against N x records


def test_streamed_training_with_n_gpus_behind_one_handle(tmp_path):
    """fmx_train_stream on a cfg.n_gpus = 2 handle (replicas on one device): replica r streams its own half of the row range, the
    replicas exchange the occurring features' records per step; equal to the dense form bit for bit, and to ONE engine that
    trains on the interleaved global batches to fp32 rounding."""
    from fmwr_amd import _lib as L, engine
    vocab = [50000, 20000, 5000, 300, 20]
    p, k, B, steps = 4 + sum(vocab), 8, 1024, 4
    total = steps * B * 2
    kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.05, l2_w1=1e-3, l2_v=1e-3, mode=L.MODE_MINIBATCH)
    res = {}
    for form in ("compact", "dense"):
        os.environ["FMX_GROUP_EXCHANGE"] = form
        try:
            g = engine.Engine(p, batch_rows=B, n_gpus=2, gpus_share_device=1, **kw)
            g.init_normal(5, 0.0, 0.05)
            done, _ = g.train_stream(total, seed=9, fields=(4, vocab, 2.0))
            assert done == total
            res[form] = g.get_params()
        finally:
            os.environ.pop("FMX_GROUP_EXCHANGE", None)
    a, b = res["compact"], res["dense"]
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    # one engine on the interleaved global batches: global step s = rows of replica 0's step s, then replica 1's
    halves = [engine.Matrix.synthetic_fields(total // 2, 4, vocab, 2.0, 9, row_offset=r * (total // 2)) for r in range(2)]
    parts = [h.export() for h in halves]
    rp = [0]; col = []; val = []; y = []
    for s in range(steps):
        for (rpi, ci, vi, yi) in parts:
            lo, hi = s * B, (s + 1) * B
            col.append(ci[rpi[lo]:rpi[hi]]); val.append(vi[rpi[lo]:rpi[hi]]); y.append(yi[lo:hi])
            rp.extend((rpi[lo + 1:hi + 1] - rpi[lo] + rp[-1]).tolist())
    one = engine.Matrix.from_csr(np.array(rp, np.int64), np.concatenate(col), np.concatenate(val), p, np.concatenate(y))
    e = engine.Engine(p, batch_rows=2 * B, **kw)
    e.init_normal(5, 0.0, 0.05)
    for s in range(steps):
        e.step(one, s)
    e.sync()
    c = e.get_params()
    from tests import util
    assert util.rel_err(a[2], c[2]) < 1e-5 and util.rel_err(a[1], c[1]) < 1e-5 and abs(a[0] - c[0]) < 1e-5
