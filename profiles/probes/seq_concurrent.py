"""How many reference-order learners (FMX_MODE_SEQUENTIAL: one workgroup each) run side by side on one MI355X?  T engines, T host threads, one matrix."""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, k = 2_000_000, 1_000_000, 30, 16
m = engine.Matrix.synthetic(n, p, z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (k, p))
def make(lr):
    e = engine.Engine(p, solver=L.SOLVER_SGD, num_factor=k, learn_rate=lr, mode=L.MODE_SEQUENTIAL, l2_w1=1e-4, l2_v=1e-4)
    e.set_params(0.0, None, v0)
    return e
cnt = 200_000
for T in (1, 4, 16, 32, 64):
    es = [make(0.002 * (1 + i)) for i in range(T)]
    for e in es: e.train(m, 20_000)
    def work(e): e.train(m, cnt)
    th = [threading.Thread(target=work, args=(e,)) for e in es]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"{T:3d} learners side by side: {T * cnt / dt / 1e6:7.2f} M examples/s in all ({cnt / dt / 1e6:.2f} M each)", flush=True)
    for e in es: e.close()
