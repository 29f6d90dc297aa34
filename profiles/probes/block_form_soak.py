"""Soak of the block form at 10 M x 1 M, k = 16: ten ALS sweeps (one-hot and U(0,1) values), twice (bit for bit?), against the tile form after the same ten sweeps; the residual's
sum of squares must fall sweep after sweep."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
from tests import util
N, P, Z, K, SEED = 10_000_000, 1_000_000, 30, 16, 20240001
def run(order, values, carry=False):
    if order: os.environ["FMX_ALS_ORDER"] = order
    else: os.environ.pop("FMX_ALS_ORDER", None)
    m = engine.Matrix.synthetic(N, P, Z, SEED)
    if values: m.synthetic_values(SEED + 1)
    e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
    e.init_normal(SEED, 0.0, 0.1)
    if carry: e.als_carry_q(True)
    d_err = util.DevBuf(N)
    L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(N), d_err.ptr, C.c_int(L.LINK_NONE)))
    e.sync()
    # e = y_hat - y (regression on the generator's +-1 labels)
    yy = np.zeros(N, np.float32)
    L.check(L.lib().fmx_matrix_export(m.h, C.c_int64(0), C.c_int64(N), None, None, None, yy.ctypes.data_as(C.c_void_p)))
    r0 = d_err.numpy() - yy.astype(np.float64)
    d_err.free(); d_err = util.DevBuf.from_numpy(r0)
    ss = []
    for it in range(10):
        e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=np.full(K, 1.0)); e.sync()
        r = d_err.numpy(); ss.append(float(np.dot(r, r)))
    out = (d_err.numpy(), e.get_rows(np.arange(0, P, 499, dtype=np.uint32))[1], ss, e.als_level_order_form(m))
    e.close(); d_err.free(); m.close()
    return out
for values in (False, True):
    a = run(None, values); b = run(None, values); t = run("1", values); c = run(None, values, carry=True)
    mono = all(x > y for x, y in zip(a[2], a[2][1:]))
    print(f"values={'U(0,1)' if values else 'one-hot'}: forms {a[3]}/{t[3]}; ten sweeps twice bit for bit: {np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])}; "
          f"vs the tile form: e {util.rel_err(a[0], t[0]):.2e} V {util.rel_err(a[1], t[1]):.2e}; carried q vs rebuilt: e {util.rel_err(c[0], a[0]):.2e} V {util.rel_err(c[1], a[1]):.2e}; "
          f"sum e^2 falls every sweep: {mono} ({a[2][0]:.6e} -> {a[2][-1]:.6e})", flush=True)
