"""Small steps (batch_rows 1024 .. 16384) at configs[1]'s shape: step time, to be read next to rocprofv3 --kernel-trace --stats of the
same command (kernel durations against the step time: how much of a small step is inside the two kernels)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
p, z, k = 1_000_000, 30, 16
sub = engine.Matrix.synthetic(2_000_000, p, z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (k, p)).astype(np.float32).astype(np.float64)
for B in [int(x) for x in (sys.argv[1:] or ["1024", "4096", "16384"])]:
    e = engine.Engine(p, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
    e.set_params(0.0, None, v0)
    nb = e.num_batches(sub) - 1
    for i in range(50):
        e.step(sub, i % nb)
    e.sync()
    steps = 400
    t = time.perf_counter()
    for i in range(steps):
        e.step(sub, (50 + i) % nb)
    e.sync()
    dt = time.perf_counter() - t
    e.profile_reset(); e.profile(1)
    for i in range(100):
        e.step(sub, i % nb)
    e.sync()
    f_ms, f_n = e.profile_get(L.KERNEL_ROWS_FORWARD); u_ms, u_n = e.profile_get(L.KERNEL_COLS_UPDATE)
    e.profile(0)
    print(f"B={B}: {dt / steps * 1e6:.1f} us per step = {B * steps / dt / 1e6:.1f} M examples/s; HIP events: rows_forward {f_ms / max(f_n, 1) * 1e3:.1f} us, cols_update {u_ms / max(u_n, 1) * 1e3:.1f} us")
    e.close()
