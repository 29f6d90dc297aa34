"""Forward-only rate by slab size (rows per call of fmx_predict_device) at configs[1]."""
import sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, k = 10_000_000, 1_000_000, 30, 16
torch.cuda.set_device(0)
m = engine.Matrix.synthetic(n, p, z, 20240001)
e = engine.Engine(p, num_factor=k, mode=L.MODE_MINIBATCH, batch_rows=1 << 20)
e.set_params(0.0, None, np.random.default_rng(1).normal(0, 0.01, (k, p)))
out = torch.empty(n, dtype=torch.float64, device="cuda")
for slab in (262144, 1 << 20, 1 << 22, n):
    for link in (L.LINK_NONE, L.LINK_LOGISTIC):
        def run():
            for r0 in range(0, n, slab):
                r1 = min(n, r0 + slab)
                L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(r0), C.c_int64(r1), C.c_void_p(out.data_ptr() + 8 * r0), C.c_int(link)))
        run(); e.sync()
        t = time.perf_counter()
        for _ in range(3):
            run()
        e.sync()
        dt = (time.perf_counter() - t) / 3
        print(f"slab {slab:>9} link {link}: {n / dt / 1e6:8.1f} M rows/s  ({dt / n * 1e9:.3f} ns/row)", flush=True)
