"""configs[4] (ALS / MCMC V sweep, k = 16) against the column law of the synthetic matrix: the exact sweep needs as many
dependent launches per factor as the matrix has LEVELS.  One column per stratum (fmx_matrix_synthetic) has nnz levels;
i.i.d. uniform and Zipf columns have deep chains."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
p, z, k = 1_000_000, 30, 16
cases = (("stratified (one column per stratum)", lambda: engine.Matrix.synthetic(n, p, z, 3)),
         ("i.i.d. uniform, sorted", lambda: engine.Matrix.synthetic_iid(n, p, z, 3, L.COLUMNS_UNIFORM)),
         ("Zipf(1.05), sorted", lambda: engine.Matrix.synthetic_iid(n, p, z, 3, L.COLUMNS_ZIPF, 1.05)))
for name, make in cases:
    for cap in (0, 64):      # exact schedule; approximate grouped form when more than 64 levels are needed
        m = make()
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=cap)
        e.set_params(0.0, None, np.random.default_rng(1).normal(0, 0.01, (k, p)))
        t = time.perf_counter()
        lv, big, approx, _ = e.als_plan(m)
        t_plan = time.perf_counter() - t
        err = np.random.default_rng(2).normal(0, 1, n)
        t = time.perf_counter(); e1 = e.als_vsweep(m, err); t1 = time.perf_counter() - t      # includes the residual's two PCIe trips
        t = time.perf_counter(); e2 = e.als_vsweep(m, e1); t2 = time.perf_counter() - t
        print(f"{name:36s} als_max_levels={cap:3d}: {'groups' if approx else 'levels'} {lv:6d} (largest {big:7d}), plan {t_plan:6.2f} s, V sweep over {k} factors "
              f"{t2:6.2f} s = {40 * n * z * k / t2 / 1e12:5.2f} TB/s of the 40-B/nonzero figure; sum e^2: {np.sum(err ** 2):.4g} -> {np.sum(e1 ** 2):.4g} -> {np.sum(e2 ** 2):.4g}")
        e.close(); m.close()
        if not approx and cap == 64:
            pass
