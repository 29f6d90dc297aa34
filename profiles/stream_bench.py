"""configs[3] shape on ONE GPU, streamed: the matrix does not stay resident -- chunks of rows are generated (or, for real
data, uploaded with fmx_matrix_from_csr), ingested (per-tile CSC), trained on once and dropped.  End-to-end examples/s
including generation and ingest.  p = 33 M, 39 nnz/row, k = 32, SGD, 262 144-row steps."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
p, z, k, B = 33_000_000, 39, 32, 262_144
chunk = 16 * B          # 4.2 M rows per chunk
n_chunks = 6
e = engine.Engine(p, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
e.set_params(0.0, None, None)
t_gen = t_ing = t_trn = 0.0
t0 = time.perf_counter()
for c in range(n_chunks):
    t = time.perf_counter(); m = engine.Matrix.synthetic(chunk, p, z, 20240001, row_offset=c * chunk); t_gen += time.perf_counter() - t
    t = time.perf_counter(); nb = e.num_batches(m); e.sync(); t_ing += time.perf_counter() - t
    t = time.perf_counter()
    for b in range(nb):
        e.step(m, b)
    e.sync(); t_trn += time.perf_counter() - t
    m.close()
dt = time.perf_counter() - t0
rows = n_chunks * chunk
print(f"streamed {rows / 1e6:.1f} M rows in {dt:.2f} s = {rows / dt / 1e6:.1f} M examples/s end to end "
      f"(generate {t_gen:.2f} s, ingest {t_ing:.2f} s, train {t_trn:.2f} s = {rows / t_trn / 1e6:.1f} M examples/s while training)")
