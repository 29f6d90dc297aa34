import sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
p, z, k = 1_000_000, 30, 16
sub = engine.Matrix.synthetic(2_000_000, p, z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (k, p)).astype(np.float32).astype(np.float64)
for B in (1024, 4096):
    e = engine.Engine(p, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
    e.set_params(0.0, None, v0)
    nb = e.num_batches(sub) - 1
    for i in range(200): e.step(sub, i % nb)
    e.sync()
    t = time.perf_counter()
    for i in range(2000): e.step(sub, (200 + i) % nb)
    t_host = time.perf_counter() - t
    e.sync()
    t_all = time.perf_counter() - t
    print(f"B={B}: host enqueue {t_host / 2000 * 1e6:.1f} us per step, end to end {t_all / 2000 * 1e6:.1f} us per step")
    e.close()
