"""One plan build of the 10 M x 1 M headline matrix (39 tiles of 262 144 rows): wall time of fmx_num_batches; run under rocprofv3 --kernel-trace --stats the
kernels' own time is in the stats file -- the difference is allocation and launch overhead on the host."""
import sys, time
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, B = 10_000_000, 1_000_000, 30, 262_144
m = engine.Matrix.synthetic(n, p, z, 20240001)
e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=16, learn_rate=0.01, mode=L.MODE_MINIBATCH, batch_rows=B)
e.init_normal(1, 0.0, 0.01); e.sync()
t = time.perf_counter(); e.num_batches(m); e.sync(); print("plan build wall: %.2f ms" % ((time.perf_counter() - t) * 1e3))
