"""Side measurements quoted in DESIGN.md (not the headline bench): sequential-exact learner rate, ALS V sweep at the
configs[4] shape, host->device ingest (PCIe-inclusive) rate.  Run on the GPU box: python profiles/extra_bench.py"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from fmwr_amd import _lib as L  # noqa: E402
from fmwr_amd import engine  # noqa: E402

out = {}
P, Z, K = 1_000_000, 30, 16

# 1. sequential-exact SGD (reference order, fp64) on the configs[1] shape
m = engine.Matrix.synthetic(400_000, P, Z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (K, P))
e = engine.Engine(P, num_factor=K, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_SEQUENTIAL)
e.set_params(0.0, None, v0)
e.train(m, 20_000)
t0 = time.perf_counter(); done = e.train(m, 300_000); dt = time.perf_counter() - t0
out["sequential_sgd_k16_examples_per_s"] = done / dt
e.close()
e = engine.Engine(P, num_factor=K, solver=L.SOLVER_FTRL, l1_w1=1e-4, l1_v=1e-4, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_SEQUENTIAL)
e.set_params(0.0, None, v0)
e.train(m, 20_000)
t0 = time.perf_counter(); done = e.train(m, 200_000); dt = time.perf_counter() - t0
out["sequential_ftrl_k16_examples_per_s"] = done / dt
e.close()

# 2. host -> device ingest of a CSR (PCIe inclusive): 2M rows x 30 nnz
rp, col, val, y = m.export(0, 400_000)
t0 = time.perf_counter(); m2 = engine.Matrix.from_csr(rp, col, val, P, y); dt = time.perf_counter() - t0
out["ingest_from_csr_rows_per_s"] = 400_000 / dt
out["ingest_from_csr_GBps"] = (col.nbytes + val.nbytes + rp.nbytes + y.nbytes) / dt / 1e9
m2.close(); m.close()

# 3. ALS V sweep, configs[4] shape: 10M x 1M, k=16 (regression residual)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
m = engine.Matrix.synthetic(n, P, Z, 20240001)
e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
e.set_params(0.0, None, v0)
err = e.predict(m) - m.export()[3]
t0 = time.perf_counter(); err1 = e.als_vsweep(m, err, alpha=1.0, v_lambda=np.full(K, 1.0)); dt_first = time.perf_counter() - t0
t0 = time.perf_counter(); err2 = e.als_vsweep(m, err1, alpha=1.0, v_lambda=np.full(K, 1.0)); dt = time.perf_counter() - t0
out["als_vsweep_rows"] = n
out["als_vsweep_first_call_s"] = dt_first   # includes CSC build + level scheduling
out["als_vsweep_s"] = dt                    # includes the H2D/D2H of the residual and the level plan
out["als_vsweep_algorithmic_GBps"] = K * n * Z * 40 / dt / 1e9
out["als_sse"] = [float(np.sum(err ** 2)), float(np.sum(err1 ** 2)), float(np.sum(err2 ** 2))]
print(json.dumps(out, indent=1))
