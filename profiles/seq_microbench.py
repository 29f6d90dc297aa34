import sys, time, numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
for P, Z, K in [(10_000, 30, 16), (1_000_000, 30, 16), (1_000_000, 2, 8), (10_000, 2, 8), (10_000, 30, 2)]:
    m = engine.Matrix.synthetic(200_000, P, Z, 1)
    e = engine.Engine(P, num_factor=K, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_SEQUENTIAL)
    e.set_params(0.0, None, np.random.default_rng(1).normal(0, 0.01, (K, P)))
    e.train(m, 5000)
    t0 = time.perf_counter(); done = e.train(m, 100_000); dt = time.perf_counter() - t0
    print(P, Z, K, f"{done/dt:.0f} ex/s  {dt/done*1e6:.2f} us/ex", flush=True)
