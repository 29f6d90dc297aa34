"""SMatrix::scales / normalize (util/Smatrix.h:98-153) at configs[1]'s size: z-scoring 1000 of the 1 M columns of the 10 M x 30 matrix, then applying the scales."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmwr_amd import engine
m = engine.Matrix.synthetic(10_000_000, 1_000_000, 30, 20240001)
cols = np.arange(0, 1_000_000, 1000, dtype=np.int32)
for rep in range(2):
    t0 = time.perf_counter(); mean, std = m.scales(cols); t1 = time.perf_counter()
    m.normalize(mean, std); t2 = time.perf_counter()
    print(f"scales {t1 - t0:.4f} s, normalize {t2 - t1:.4f} s")
