"""fmx_train_grid at configs[1]'s shape: N reference-order learners (a learning-rate grid) in one launch per 65 536 examples, one workgroup per model."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, k = 2_000_000, 1_000_000, 30, 16
m = engine.Matrix.synthetic(n, p, z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (k, p))
cnt = 200_000
for N in (1, 8, 32, 64, 128, 256, 512):
    es = [engine.Engine(p, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.002 * (1 + i % 16), l2_w1=1e-4 * (1 + i // 16), l2_v=1e-4, mode=L.MODE_SEQUENTIAL) for i in range(N)]
    for e in es: e.set_params(0.0, None, v0)
    engine.Engine.train_grid(es, m, 20_000)
    t0 = time.perf_counter()
    engine.Engine.train_grid(es, m, cnt)
    dt = time.perf_counter() - t0
    print(f"{N:4d} models: {N * cnt / dt / 1e6:8.2f} M examples/s in all ({cnt / dt / 1e6:.2f} M per model)", flush=True)
    for e in es: e.close()
