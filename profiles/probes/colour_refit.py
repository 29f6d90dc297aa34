"""The colouring's refit passes on i.i.d. columns at configs[4]'s size: classes, plan seconds and the feature-major sweep's time for 0 / 1 / 2 / 4 passes (FMX_COLOUR_REFIT)."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
from tests import util
N, P, Z, K, SEED = 10_000_000, 1_000_000, 30, 16, 20240001
for passes in (0, 1, 2, 4, 8):
    os.environ["FMX_COLOUR_REFIT"] = str(passes)
    m = engine.Matrix.synthetic_iid(N, P, Z, SEED, law=L.COLUMNS_UNIFORM)
    e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL, als_max_levels=-2)
    e.init_normal(SEED, 0.0, 0.01)
    e.sync(); t = time.perf_counter()
    levels, largest, _, _ = e.als_plan(m)
    tp = time.perf_counter() - t
    d_err = util.DevBuf(N)
    L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(N), d_err.ptr, C.c_int(L.LINK_NONE)))
    ts = []
    for i in range(3):
        e.sync(); t = time.perf_counter()
        e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=np.full(K, 1.0)); e.sync()
        ts.append((time.perf_counter() - t) * 1e3)
    print(f"refit passes {passes}: {levels} classes (largest {largest}), plan {tp:.2f} s (CSC + colouring), sweep {min(ts):.1f} ms", flush=True)
    e.close(); d_err.free(); m.close()
