import sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, k = 10_000_000, 1_000_000, 30, 16
t = time.perf_counter(); m = engine.Matrix.synthetic(n, p, z, 20240001); print("synthetic", time.perf_counter() - t)
e = engine.Engine(p, num_factor=k, mode=L.MODE_MINIBATCH, batch_rows=1 << 20)
t = time.perf_counter(); nb = e.num_batches(m); e.sync(); print("batch CSC build (10M x 30)", time.perf_counter() - t, nb)
e2 = engine.Engine(p, num_factor=k, mode=L.MODE_MINIBATCH, batch_rows=1 << 18)
t = time.perf_counter(); nb = e2.num_batches(m); e2.sync(); print("rebuild for another batch size", time.perf_counter() - t, nb)
rp, col, val, y = m.export(0, 2_000_000)
t = time.perf_counter(); m2 = engine.Matrix.from_csr(rp, col, val, p, y); print("from_csr 2M rows", time.perf_counter() - t)
ea = engine.Engine(p, num_factor=k, solver=L.SOLVER_ALS, task=L.TASK_REGRESSION, mode=L.MODE_SEQUENTIAL)
t = time.perf_counter(); ea.als_train(m2, 1); print("ALS first iteration incl. full CSC + level plan (2M rows)", time.perf_counter() - t)
t = time.perf_counter(); ea.als_train(m2, 1); print("ALS second iteration", time.perf_counter() - t)
