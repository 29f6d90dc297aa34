import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, k = 2_000_000, 1_000_000, 30, 16
m = engine.Matrix.synthetic_iid(n, p, z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (k, p))
order = np.arange(1, 600_001, dtype=np.int64)
out = []
for rep in range(2):
    e = engine.Engine(p, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_SEQUENTIAL)
    e.set_params(0.0, None, v0)
    e.train_order(m, order[:20000])
    t = time.perf_counter()
    e.train_order(m, order[:600_000])
    out.append(600_000 / (time.perf_counter() - t))
    w0, w, v = e.get_params()
    del e
import hashlib
print("sgd_l2 k=16 pipelined: %.0f K examples/s (runs %s), hash %s" % (max(out) / 1e3, [round(x / 1e3) for x in out], hashlib.sha1(v.tobytes()).hexdigest()[:12]))
