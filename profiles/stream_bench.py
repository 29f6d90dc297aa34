"""(FMX_STREAM_OVERLAP=1 in the environment: ingest on a second stream beside the running step; default: behind it on the same stream)
configs[3] shape on ONE GPU: p = 33 M, 39 nnz/row, k = 32, SGD, 262 144-row steps.
  resident : 8 M rows stay in HBM, their tile plans are built once (timed separately), steps run over them;
  streamed : fmx_train_stream -- every step's rows are generated and planned on a second stream while the previous step
             trains, trained on once and dropped (what a 4e9-row job does per GPU); the rate includes generation and ingest.
Both for the uniform generator (i.i.d. columns, ~8 M distinct features per step) and for the Criteo-shaped one
(13 dense + 26 categorical fields with power-law heads)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
p, z, k, B = 33_000_000, 39, 32, 262_144
rows = 32 * B
for name, fields in (("uniform", None), ("criteo-shaped (skew 3)", (13, engine.CRITEO_VOCAB, 3.0))):
    e = engine.Engine(p, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
    e.init_normal(1, 0.0, 0.01)
    t = time.perf_counter()
    m = engine.Matrix.synthetic(rows, p, z, 20240001) if fields is None else engine.Matrix.synthetic_fields(rows, fields[0], fields[1], fields[2], 20240001)
    e.sync(); t_gen = time.perf_counter() - t
    t = time.perf_counter(); nb = e.num_batches(m); e.sync(); t_ing = time.perf_counter() - t
    counts = [e.compact_count(m, b) for b in range(nb)]
    for b in range(4):
        e.step(m, b)
    e.sync()
    t = time.perf_counter()
    for b in range(nb):
        e.step(m, b)
    e.sync(); t_trn = time.perf_counter() - t
    m.close()
    e.train_stream(4 * B, nnz_per_row=z, seed=7, fields=fields)   # warm-up: slot and workspace allocation
    t = time.perf_counter()
    done, wait = e.train_stream(rows, nnz_per_row=z, seed=20240001, fields=fields)
    t_str = time.perf_counter() - t
    print(f"{name}: {np.mean(counts) / 1e6:.2f} M distinct features per step | resident: generate {t_gen * 1e3:.0f} ms, plan {t_ing * 1e3:.0f} ms "
          f"({rows / t_ing / 1e6:.0f} M rows/s), train {rows / t_trn / 1e6:.1f} M examples/s | "
          f"sequential ingest+train would be {rows / (t_gen + t_ing + t_trn) / 1e6:.1f} | streamed (overlapped): {done / t_str / 1e6:.1f} M examples/s end to end, "
          f"host waited {wait * 1e3:.0f} ms of {t_str * 1e3:.0f} ms for ingest")
    e.close()
