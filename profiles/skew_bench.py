"""Mini-batch step rate on shapes that stress phase 2's per-feature lists: few features with very long lists
(MovieLens-shaped), and a Zipf-skewed feature distribution.  Run on the GPU box: python profiles/skew_bench.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine


def run(name, rp, col, val, y, p, k, batch):
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    e = engine.Engine(p, num_factor=k, learn_rate=0.01, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=batch)
    e.set_params(0.0, None, np.random.default_rng(1).normal(0, 0.01, (k, p)))
    nb = e.num_batches(m)
    for s in range(3):
        e.step(m, s % nb)
    e.sync(); e.profile_reset(); e.profile(1)
    t0 = time.perf_counter()
    steps = 20
    for s in range(steps):
        e.step(m, s % nb)
    e.sync(); dt = time.perf_counter() - t0
    e.profile(0)
    f, fn = e.profile_get(L.KERNEL_ROWS_FORWARD); c, cn = e.profile_get(L.KERNEL_COLS_UPDATE)
    rows = min(batch, m.n)
    print(f"{name}: {rows * steps / dt / 1e6:.1f} Mex/s  rows_forward {f / max(fn,1):.3f} ms  cols_update {c / max(cn,1):.3f} ms per tile", flush=True)


rng = np.random.default_rng(0)
n = 2_000_000
# MovieLens-shaped: 943 users + 1682 items, 2 nnz/row
u = rng.integers(0, 943, n); i = rng.integers(0, 1682, n) + 943
run("movielens-shaped p=2625 z=2 k=8", np.arange(0, 2 * n + 1, 2), np.stack([u, i], 1).ravel().astype(np.uint32), np.ones(2 * n, np.float32),
    rng.integers(1, 6, n).astype(np.float32), 2625, 8, 262144)
# Zipf-skewed: 30 nnz/row, p = 1M, feature popularity ~ 1/rank^1.05 (duplicates inside a row removed)
z, p = 30, 1_000_000
n2 = 1_000_000
ranks = np.minimum(rng.zipf(1.05, (n2, z)), p) - 1
ranks.sort(axis=1)
keep = np.ones_like(ranks, bool); keep[:, 1:] = ranks[:, 1:] != ranks[:, :-1]
lens = keep.sum(1); rp = np.zeros(n2 + 1, np.int64); rp[1:] = np.cumsum(lens)
run("zipf(1.05) p=1M z<=30 k=16", rp, ranks[keep].astype(np.uint32), np.ones(int(rp[-1]), np.float32), np.where(rng.random(n2) < 0.5, -1, 1).astype(np.float32), p, 16, 262144)
