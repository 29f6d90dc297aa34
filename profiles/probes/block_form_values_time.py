import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
from tests import util
N, P, Z, K, SEED = 10_000_000, 1_000_000, 30, 16, 20240001
for name, vals in (("one-hot", False), ("U(0,1) values", True)):
    m = engine.Matrix.synthetic(N, P, Z, SEED)
    if vals: m.synthetic_values(SEED + 1)
    e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
    e.init_normal(SEED, 0.0, 0.1)
    d_err = util.DevBuf(N)
    L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(N), d_err.ptr, C.c_int(L.LINK_NONE)))
    e.sync()
    e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=np.full(K, 1.0)); e.sync()
    t = time.perf_counter()
    for _ in range(3): e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=np.full(K, 1.0))
    e.sync()
    dt = (time.perf_counter() - t) / 3
    print(f"{name}: form {e.als_level_order_form(m)}, {dt*1e3:.1f} ms per sweep = {N/dt/1e6:.1f} M examples/s", flush=True)
    e.close(); d_err.free(); m.close()
