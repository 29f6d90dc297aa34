"""Determinism soak at the bench size: the same 300 steps twice (fresh engines) must give bit-identical parameters,
for SGD and FTRL, fused and tiled.  python profiles/soak.py"""
import hashlib, sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine

P, Z, K, N = 1_000_000, 30, 16, 10_000_000
m = engine.Matrix.synthetic(N, P, Z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (K, P)).astype(np.float32).astype(np.float64)
for solver, batch in ((L.SOLVER_SGD, 262144), (L.SOLVER_SGD, 1048576), (L.SOLVER_FTRL, 524288)):
    hashes = []
    for rep in range(2):
        e = engine.Engine(P, num_factor=K, solver=solver, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, l1_v=1e-4 if solver == L.SOLVER_FTRL else 0.0,
                          mode=L.MODE_MINIBATCH, batch_rows=batch)
        e.set_params(0.0, None, v0)
        nb = e.num_batches(m)
        t0 = time.perf_counter()
        for s in range(300):
            e.step(m, s % nb)
        e.sync()
        dt = time.perf_counter() - t0
        w0, w, v = e.get_params()
        assert np.all(np.isfinite(v)) and np.all(np.isfinite(w)) and np.isfinite(w0)
        hashes.append(hashlib.sha256(v.tobytes() + w.tobytes() + np.float64(w0).tobytes()).hexdigest()[:16])
        e.close()
    print(f"solver {solver} batch {batch}: {300 * batch / dt / 1e6:.0f} Mex/s, hashes {hashes}, identical: {hashes[0] == hashes[1]}", flush=True)
    assert hashes[0] == hashes[1]
print("soak ok")
