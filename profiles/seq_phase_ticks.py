"""Windowed sequential learner: clock ticks per group by phase (diagnostic build: profiles/variant_build.sh NAME -DFMX_SEQ_TIMING,
run with FMX_LIB_PATH=profiles/_variants/NAME/libfmx.so; the kernel prints its own averages)."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, k = 2_000_000, 1_000_000, int(os.environ.get("SEQ_Z", "30")), int(os.environ.get("SEQ_K", "16"))
m = engine.Matrix.synthetic(n, p, z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (k, p))
order = np.arange(1, 400_001, dtype=np.int64)
e = engine.Engine(p, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.01, mode=L.MODE_SEQUENTIAL, l2_w1=1e-4, l2_v=1e-4)
e.set_params(0.0, None, v0)
e.train_order(m, order[:20000]); e.sync()
t = time.perf_counter()
e.train_order(m, order); e.sync()
print(f"{len(order) / (time.perf_counter() - t) / 1e3:.0f} K examples/s", flush=True)
