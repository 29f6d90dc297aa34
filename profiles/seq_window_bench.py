"""Sequential-exact learners at configs[1]'s shape (p = 1M, 30 nnz, k = 16): one-wave kernel vs windowed kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, k = 2_000_000, 1_000_000, int(os.environ.get("SEQ_Z", "30")), int(os.environ.get("SEQ_K", "16"))
m = engine.Matrix.synthetic(n, p, z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (k, p))
order = np.arange(1, 400_001, dtype=np.int64)
for name, solver, kw in (("sgd_l2", L.SOLVER_SGD, dict(l2_w1=1e-4, l2_v=1e-4)), ("sgd_l1", L.SOLVER_SGD, dict(l1_w1=1e-5, l1_v=1e-5)),
                         ("ftrl", L.SOLVER_FTRL, dict(l1_w1=1e-4, l1_v=1e-4, l2_w1=1e-4, l2_v=1e-4)), ("tdap", L.SOLVER_TDAP, dict(l1_w1=1e-4, l2_v=1e-4))):
    res = {}
    for win in ("0", "1", "2"):  # one wave; windowed; windowed + pipelined (every non-TDAP shape, fitting 256 VGPRs or not)
        os.environ["FMX_SEQ_WINDOW"] = win
        e = engine.Engine(p, solver=solver, num_factor=k, learn_rate=0.01, mode=L.MODE_SEQUENTIAL, **kw)
        e.set_params(0.0, None, v0)
        e.train_order(m, order[:20000])
        t = time.perf_counter()
        if win == "2" and solver == L.SOLVER_TDAP:
            res[win] = float("nan"); continue
        cnt = 100_000 if win == "0" else 400_000
        e.train_order(m, order[:cnt])
        dt = time.perf_counter() - t
        res[win] = cnt / dt
        del e
    print(f"{name}: one wave {res['0'] / 1e3:.0f} K examples/s, windowed {res['1'] / 1e3:.0f} K ({res['1'] / res['0']:.1f}x), pipelined {res['2'] / 1e3:.0f} K ({res['2'] / res['0']:.1f}x)", flush=True)
