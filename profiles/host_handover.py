"""The PCIe-inclusive rate of the boundary: bench.py's `value` starts with the rows resident in HBM; an R caller hands fm.matrix's host arrays
(value f64, col_idx i32, row_size i32, labels f64) to fmx_matrix_from_rlist.  This times, at BASELINE.json configs[1]'s size, the hand-over
(host -> device copy, conversion to the device layout, the sortedness / one-hot check), the per-tile plan build and one pass over all rows.
python profiles/host_handover.py [--rows 10000000]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmwr_amd import _lib as L, engine

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
a = ap.parse_args()
n, p, z, k, B = a.rows, 1_000_000, 30, 16, 262_144
g = engine.Matrix.synthetic(n, p, z, 20240001)
rp, col, val, y = g.export()
g.close()
value = val.astype(np.float64); col_idx = col.astype(np.int32); row_size = np.diff(rp).astype(np.int32); labels = y.astype(np.float64)
del rp, col, val, y
host_bytes = value.nbytes + col_idx.nbytes + row_size.nbytes + labels.nbytes
e = engine.Engine(p, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
import ctypes as C
v0 = np.ascontiguousarray(np.random.default_rng(1).normal(0, 0.01, (p, k)))   # R's k x p column-major matrix = p rows of k doubles
w_init = np.zeros(p)
ptr = lambda a: a.ctypes.data_as(C.c_void_p)
for rep in range(2):
    t0 = time.perf_counter(); L.check(L.lib().fmx_set_params(e.h, C.c_double(0.0), ptr(w_init), ptr(v0))); t_set = time.perf_counter() - t0
out = {}
for rep in range(3):   # (the first hand-over also pays the first touch of the pageable host arrays)
    t0 = time.perf_counter()
    m = engine.Matrix.from_rlist(value, col_idx, row_size, p, labels)
    t1 = time.perf_counter()
    nb = e.num_batches(m); e.sync()
    t2 = time.perf_counter()
    for b in range(nb):
        e.step(m, b)
    e.sync()
    t3 = time.perf_counter()
    out = {"rows": n, "host_MB": host_bytes / 1e6, "handover_s": t1 - t0, "handover_GBps": host_bytes / (t1 - t0) / 1e9, "plan_s": t2 - t1, "one_pass_s": t3 - t2,
           "one_epoch_examples_per_s_from_host_arrays": n / (t3 - t0), "two_epochs_examples_per_s_from_host_arrays": 2 * n / (t3 - t0 + (t3 - t2))}
    m.close()
w_out = np.zeros(p); v_out = np.zeros((p, k)); w0_out = C.c_double()
for rep in range(2):
    t0 = time.perf_counter(); L.check(L.lib().fmx_get_params(e.h, C.byref(w0_out), ptr(w_out), ptr(v_out))); t_get = time.perf_counter() - t0
assert np.all(np.isfinite(v_out)) and np.any(v_out != v0)
out.update({"set_params_s": t_set, "get_params_s": t_get, "params_MB": (v0.nbytes + w_init.nbytes) / 1e6,
            "note": "fmx_set_params / fmx_get_params called through the C ABI with (w0, w, V) as R holds them (V: k x p doubles)"})
print(json.dumps(out))
