"""TDAP (the reference's default solver) in the throughput mode at configs[1]'s shape: examples/s and per-kernel time for k = 16 / 64,
next to SGD and FTRL on the same matrix and step size."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z = 10_000_000, 1_000_000, 30
m = engine.Matrix.synthetic(n, p, z, 20240001)
for k in (16, 64):
    v0 = np.random.default_rng(1).normal(0, 0.01, (k, p)).astype(np.float32).astype(np.float64)
    for name, solver, kw in (("sgd", L.SOLVER_SGD, dict(l2_w1=1e-4, l2_v=1e-4)), ("ftrl", L.SOLVER_FTRL, dict(l1_w1=1e-4, l1_v=1e-4, l2_w1=1e-4, l2_v=1e-4)),
                             ("tdap", L.SOLVER_TDAP, dict(l1_w1=1e-4, l1_v=1e-4, l2_w1=1e-4, l2_v=1e-4, gamma=1e-4))):
        B = 1_048_576
        e = engine.Engine(p, solver=solver, num_factor=k, learn_rate=0.01, mode=L.MODE_MINIBATCH, batch_rows=B, **kw)
        e.set_params(0.0, None, v0)
        nb = e.num_batches(m) - 1
        for i in range(20): e.step(m, i % nb)
        e.sync()
        e.profile_reset(); e.profile(7)
        t = time.perf_counter()
        steps = 30
        for i in range(steps): e.step(m, (20 + i) % nb)
        e.sync()
        dt = time.perf_counter() - t
        f_ms, f_n = e.profile_get(L.KERNEL_ROWS_FORWARD); u_ms, u_n = e.profile_get(L.KERNEL_COLS_UPDATE)
        w0, w, v = e.get_params()
        print(f"k={k:3d} {name:5s}: {B * steps / dt / 1e6:7.1f} M examples/s; per 524 288-row tile: rows_forward {f_ms / max(f_n, 1):.3f} ms, cols_update {u_ms / max(u_n, 1):.3f} ms; finite: {bool(np.all(np.isfinite(v)))}", flush=True)
        e.close()
