"""configs[4] on SURVEY 8(d)'s i.i.d. columns in the reference's own order (cfg.als_max_levels = 0): one launch per level (FMX_ALS_PERSIST=0) against the persistent
forms (one launch per factor): "1" = the default (record-ordered, als_exact_flow_k), "counter" = als_exact_persist_k.
usage: python profiles/probes/als_exact_persist.py [rows] [sweeps] [forms, comma separated]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from fmwr_amd import _lib as L, engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
p, z, k = 1_000_000, 30, 16
m = engine.Matrix.synthetic_iid(n, p, z, 20240001, law=L.COLUMNS_UNIFORM)
for persist in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("1", "1", "counter", "0")):
    os.environ["FMX_ALS_PERSIST"] = persist
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, als_max_levels=0)
    e.init_normal(20240001, 0.0, 0.01)
    levels, largest, approx, _ = e.als_plan(m)
    d_err = torch.randn(n, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(20240001))   # (the same residual for every form: the checksums below must agree)
    ss0 = float((d_err * d_err).sum())
    e.vsweep_device(m, d_err.data_ptr(), alpha=1.0); e.sync()
    t0 = time.perf_counter()
    for _ in range(sweeps):
        e.vsweep_device(m, d_err.data_ptr(), alpha=1.0)
    e.sync()
    dt = (time.perf_counter() - t0) / sweeps
    ss1 = float((d_err * d_err).sum())
    print(f"FMX_ALS_PERSIST={persist}: levels {levels} (largest {largest}), sweep {dt * 1e3:.1f} ms = {n / dt / 1e6:.2f} M examples/s = {dt / (levels * k) * 1e6:.2f} us per level; "
          f"sum e^2 {ss0:.6e} -> {ss1:.6e}; V sample sha {__import__('hashlib').sha256(np.ascontiguousarray(e.get_rows(np.arange(0, p, 97, dtype=np.uint32))[1]).tobytes()).hexdigest()[:16]}, residual sha {__import__('hashlib').sha256(d_err.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
    e.close()
