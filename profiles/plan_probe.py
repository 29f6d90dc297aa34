"""Kernel-level look at the per-tile plan build of the configs[1] matrix (10 M x 1 M, 38 tiles of 262 144 rows): run under
rocprofv3 --kernel-trace --stats.  python profiles/plan_probe.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmwr_amd import _lib as L, engine
n, p, z, k, B = 10_000_000, 1_000_000, 30, 16, 262_144
m = engine.Matrix.synthetic(n, p, z, 20240001)
e = engine.Engine(p, num_factor=k, learn_rate=0.01, mode=L.MODE_MINIBATCH, batch_rows=B)
e.init_normal(1, 0.0, 0.01); e.sync()
t0 = time.perf_counter(); nb = e.num_batches(m); e.sync(); print("plan build s", time.perf_counter() - t0, "tiles", nb)
