"""Where a SMALL step's phase 1 spends its time: 100 MHz stamps from one workgroup (diagnostic build, -DFMX_TRACE).
Build:  profiles/variant_build.sh trace -DFMX_TRACE; run with FMX_LIB_PATH=profiles/_variants/trace/libfmx.so."""
import ctypes, sys
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
lib = L.lib()
p, z, k = 1_000_000, 30, 16
sub = engine.Matrix.synthetic(2_000_000, p, z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (k, p)).astype(np.float32).astype(np.float64)
names = ["start", "row_ptr known", "entries staged", "gathers done", "barrier", "S row stored", "barrier", "partials"]
for B in [int(x) for x in (sys.argv[1:] or ["1024", "4096", "262144"])]:
    e = engine.Engine(p, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
    e.set_params(0.0, None, v0)
    nb = e.num_batches(sub) - 1
    acc = np.zeros(8)
    n = 0
    for i in range(40):
        e.step(sub, i % nb); e.sync()
        buf = (ctypes.c_ulonglong * 16)()
        assert lib.fmx_debug_trace(buf) == 0
        t = np.array(buf[:8], dtype=np.float64)
        if i >= 8:
            acc += (t - t[0]) * 0.01; n += 1
    acc /= n
    print(f"B={B}: " + " | ".join(f"{nm} {acc[i]:.2f}" for i, nm in enumerate(names)) + "  (us since the workgroup started)")
    e.close()
