"""The ALS V sweep on SURVEY 8(d)'s i.i.d. column law at configs[4]'s size: the exact schedule (the reference's feature order), the approximate groups
(cfg.als_max_levels > 0) and the coloured order (cfg.als_max_levels = -1: exact steps in the engine's own feature order)."""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
from tests import util
N, P, Z, K, SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 1_000_000, 30, 16, 20240001
ZIPF = len(sys.argv) > 2 and sys.argv[2] == "zipf"
for cap in ((0, -1) if ZIPF else (0, 64, -1, -2)):
    m = engine.Matrix.synthetic_iid(N, P, Z, SEED, law=L.COLUMNS_ZIPF if ZIPF else L.COLUMNS_UNIFORM, zipf_s=1.05)
    e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL, als_max_levels=cap)
    e.init_normal(SEED, 0.0, 0.01)
    t = time.perf_counter(); levels, largest, approx, _ = e.als_plan(m); tp = time.perf_counter() - t
    d_err = util.DevBuf(N)
    L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(N), d_err.ptr, C.c_int(L.LINK_NONE)))
    e.sync()
    e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=np.full(K, 1.0)); e.sync()
    t = time.perf_counter()
    e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=np.full(K, 1.0)); e.sync()
    dt = time.perf_counter() - t
    print(f"als_max_levels={cap}: levels {levels} (largest {largest}), kind {e.als_plan_kind(m)} (0 exact order, 1 approximate groups, 2 coloured order; als_max_levels = -2: its feature-major form), plan {tp:.2f} s, sweep {dt*1e3:.1f} ms = {N/dt/1e6:.1f} M examples/s", flush=True)
    e.close(); d_err.free(); m.close()
