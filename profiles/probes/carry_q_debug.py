import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
from tests import util
N, P, Z, K, SEED = 10_000_000, 1_000_000, 30, 16, 20240001
rows = np.arange(0, P, 499, dtype=np.uint32)
for carry in (False, True):
    m = engine.Matrix.synthetic(N, P, Z, SEED).synthetic_values(SEED + 1)
    e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
    e.init_normal(SEED, 0.0, 0.1)
    if carry: e.als_carry_q(True)
    d_err = util.DevBuf(N)
    L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(N), d_err.ptr, C.c_int(L.LINK_NONE)))
    e.sync()
    for it in range(10):
        e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=np.full(K, 1.0)); e.sync()
        r = d_err.numpy(); v = e.get_rows(rows)[1]
        # the true residual from a fresh forward
        d_chk = util.DevBuf(N)
        L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(N), d_chk.ptr, C.c_int(L.LINK_NONE))); e.sync()
        print(f"carry={carry} sweep {it}: sum e^2 {np.dot(r, r):.6e}  max|V| {np.abs(v).max():.4e}  nan {np.isnan(v).sum()}  max|yhat| {np.abs(d_chk.numpy()).max():.4e}", flush=True)
        d_chk.free()
    e.close(); d_err.free(); m.close()
