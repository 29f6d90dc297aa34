"""The (column, row) sort of one-hot tiles WITHOUT a field layout: the hand-written one-field LSD sort (default) against rocprim's onesweep (FMX_PAIR_SORT=rocprim).
Plan build of a resident 10 M x 1 M matrix of i.i.d. uniform columns (38 tiles of 262 144 rows, 20-bit ids: three passes) and of ragged rows, and the streamed
steady state at configs[3]'s shape with uniform columns (25-bit ids: four 7/6-bit passes against three 9-bit ones).  python profiles/pair_sort_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmwr_amd import _lib as L, engine
which = os.environ.get("FMX_PAIR_SORT", "hand-written")
n, p, z, k, B = 10_000_000, 1_000_000, 30, 16, 262_144
for name, make in (("i.i.d. uniform, 30 per row", lambda: engine.Matrix.synthetic_iid(n, p, z, 20240001)), ("ragged Poisson(30)", lambda: engine.Matrix.synthetic_ragged(n, p, float(z), 20240001))):
    best = 1e9
    for rep in range(3):
        m = make()
        e = engine.Engine(p, num_factor=k, learn_rate=0.01, mode=L.MODE_MINIBATCH, batch_rows=B)
        e.init_normal(1, 0.0, 0.01); e.sync()
        t0 = time.perf_counter(); nb = e.num_batches(m); e.sync(); best = min(best, time.perf_counter() - t0)
        e.close(); m.close()
    print(f"{which}: plan build of 10 M x 1 M, {name}: {best * 1e3:.2f} ms for {nb} tiles ({n / best / 1e6:.0f} M rows/s)")
p3, z3, k3 = 33_000_000, 39, 32
e = engine.Engine(p3, solver=L.SOLVER_SGD, num_factor=k3, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
e.init_normal(1, 0.0, 0.01)
src = e.source(70 * B, nnz_per_row=z3, seed=20240001)
for i in range(70):
    if i == 10:
        e.sync(); t0 = time.perf_counter()
    e.step(src.next(), 0)
e.sync(); dt = time.perf_counter() - t0
src.close()
print(f"{which}: streamed uniform columns at configs[3]'s shape: {60 * B / dt / 1e6:.1f} M examples/s ({dt / 60 * 1e3:.3f} ms per step)")
