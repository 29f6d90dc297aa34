"""Phase 1 on rows of differing lengths at configs[1]'s shape: the static kernel (one lane group per row), the flat form (fm_rows_forward_flat_k) and
the pulled form, per row-length law.  usage: ragged_probe.py MIN MAX ragged|iid|strata K   (environment: FMX_ROWS_FLAT, FMX_ROWS_PULL, FMX_ROWS_SERIAL)"""
import os, sys, time
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, B = 10_000_000, 1_000_000, 30, 262_144
lo, hi, kind, k = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
m = (engine.Matrix.synthetic_ragged(n, p, float(z), 20240001, min_nnz=lo, max_nnz=hi) if kind == "ragged" else
     engine.Matrix.synthetic(n, p, z, 20240001) if kind == "strata" else engine.Matrix.synthetic_iid(n, p, z, 20240001))
e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
e.init_normal(1, 0.0, 0.01)
nb = n // B
for i in range(20): e.step(m, i % nb)
e.sync(); e.profile_reset(); e.profile(3)
t = time.perf_counter()
for i in range(40): e.step(m, (20 + i) % nb)
e.sync(); dt = time.perf_counter() - t
a, an = e.profile_get(L.KERNEL_ROWS_FORWARD); b, bn = e.profile_get(L.KERNEL_COLS_UPDATE)
print("%s [%d,%d] k=%d mean %.2f form=%d serial=%s: %.1f M examples/s, phase 1 %.4f ms, phase 2 %.4f ms, schedule %s" % (
    kind, lo, hi, k, m.nnz / m.n, m.rows_form(), os.environ.get("FMX_ROWS_SERIAL"), B * 40 / dt / 1e6, a / an, b / bn, e.rows_tune()))
