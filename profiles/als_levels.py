"""configs[4] (ALS / MCMC V sweep, k = 16) against the column law of the synthetic matrix: the exact sweep needs as many
dependent launches per factor as the matrix has LEVELS.  One column per stratum (fmx_matrix_synthetic) has nnz levels;
i.i.d. uniform and Zipf columns have deep chains."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
p, z, k = 1_000_000, 30, 16
for name, make in (("stratified (one column per stratum)", lambda: engine.Matrix.synthetic(n, p, z, 3)),
                   ("i.i.d. uniform, sorted", lambda: engine.Matrix.synthetic_iid(n, p, z, 3, L.COLUMNS_UNIFORM)),
                   ("Zipf(1.05), sorted", lambda: engine.Matrix.synthetic_iid(n, p, z, 3, L.COLUMNS_ZIPF, 1.05))):
    m = make()
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
    e.set_params(0.0, None, np.random.default_rng(1).normal(0, 0.01, (k, p)))
    t = time.perf_counter()
    lv, big, ap = C.c_int64(), C.c_int64(), C.c_int32()
    L.check(L.lib().fmx_als_plan_info(e.h, m.h, C.byref(lv), C.byref(big), C.byref(ap), None))
    t_plan = time.perf_counter() - t
    err = np.zeros(n)
    t = time.perf_counter(); e.als_vsweep(m, err); t1 = time.perf_counter() - t      # includes the CSC build and the residual's two PCIe trips
    t = time.perf_counter(); e.als_vsweep(m, err); t2 = time.perf_counter() - t
    print(f"{name:38s} {lv.value:7d} levels (largest {big.value}), plan {t_plan:.2f} s, V sweep over {k} factors: first {t1:.2f} s, again {t2:.2f} s "
          f"= {40 * n * z * k / t2 / 1e12:.2f} TB/s of the 40-B/nonzero figure")
    e.close(); m.close()
