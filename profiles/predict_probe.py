"""FMPredict at configs[1]'s size through the C ABI: forward of 10 M rows + the predictions as doubles in host memory."""
import sys, time, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmwr_amd import _lib as L, engine
n, p, z, k = 10_000_000, 1_000_000, 30, 16
m = engine.Matrix.synthetic(n, p, z, 20240001)
e = engine.Engine(p, num_factor=k, mode=L.MODE_MINIBATCH, batch_rows=262144)
e.init_normal(1, 0.0, 0.01); e.sync()
out = np.zeros(n)
for rep in range(3):
    t0 = time.perf_counter()
    L.check(L.lib().fmx_predict(e.h, m.h, out.ctypes.data_as(C.c_void_p), C.c_int(L.LINK_LOGISTIC)))
    dt = time.perf_counter() - t0
    print(f"fmx_predict of {n} rows: {dt * 1e3:.1f} ms ({n / dt / 1e6:.0f} M rows/s)")
for rep in range(2):
    t0 = time.perf_counter(); v = e.evaluate(m, L.EVAL_LL); dt = time.perf_counter() - t0
    print(f"fmx_evaluate(LL): {dt * 1e3:.1f} ms")
    t0 = time.perf_counter(); v = e.evaluate(m, L.EVAL_AUC); dt = time.perf_counter() - t0
    print(f"fmx_evaluate(AUC): {dt * 1e3:.1f} ms")
