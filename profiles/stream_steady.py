"""Steady-state streamed training at configs[3]'s shape (p = 33 M, k = 32, 39 nnz/row, 262 144-row steps) through the step-by-step source
(fmx_source_*: what bench.py --workload criteo --stream times), for the uniform generator and the Criteo-shaped one: 10 warm-up steps, 60 timed.
python profiles/stream_steady.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmwr_amd import _lib as L, engine
p, z, k, B = 33_000_000, 39, 32, 262_144
for name, fields in (("uniform", None), ("criteo-shaped (skew 3)", (13, engine.CRITEO_VOCAB, 3.0))):
    e = engine.Engine(p, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
    e.init_normal(1, 0.0, 0.01)
    src = e.source(70 * B, nnz_per_row=z, seed=20240001, fields=fields)
    for i in range(70):
        if i == 10:
            e.sync(); t0 = time.perf_counter()
        m = src.next()
        e.step(m, 0)
    e.sync(); dt = time.perf_counter() - t0
    wait = src.close()
    print(f"{name}: {60 * B / dt / 1e6:.1f} M examples/s streamed in steady state ({dt / 60 * 1e3:.3f} ms per step)")
    e.close()
