"""The reference-order learner at configs[1]'s shape (10 M x 1 M stream's first 2 M rows, 30 entries per row, k = 16): the bitwise pipelined kernel against the
reassociated one (cfg.seq_reassociate: only w0 chains the examples).  usage: python profiles/probes/seq_reassoc_rate.py [solver: sgd|sgd_l1|ftrl|tdap] [k] [iid|stratified] [entries per row]"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
solver = sys.argv[1] if len(sys.argv) > 1 else "sgd"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
law = sys.argv[3] if len(sys.argv) > 3 else "iid"
n, p, z = 2_000_000, 1_000_000, int(sys.argv[4]) if len(sys.argv) > 4 else 30
m = engine.Matrix.synthetic_iid(n, p, z, 20240001) if law == "iid" else engine.Matrix.synthetic(n, p, z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (k, p))
order = np.arange(1, 400_001, dtype=np.int64)
kw = dict(num_factor=k, learn_rate=0.01, mode=L.MODE_SEQUENTIAL, l2_w1=1e-4, l2_v=1e-4)
if solver == "sgd_l1":
    kw.update(solver=L.SOLVER_SGD, l1_w1=1e-4, l1_v=1e-4)
elif solver == "ftrl":
    kw.update(solver=L.SOLVER_FTRL, l1_w1=1e-4, l1_v=1e-4)
elif solver == "tdap":
    kw.update(solver=L.SOLVER_TDAP, l1_w1=1e-4, l1_v=1e-4)
else:
    kw.update(solver=L.SOLVER_SGD)
res = {}
for re in (0, 1):
    e = engine.Engine(p, seq_reassociate=re, **kw)
    e.set_params(0.0, None, v0)
    e.train_order(m, order[:20000]); e.sync()
    t = time.perf_counter()
    e.train_order(m, order); e.sync()
    dt = time.perf_counter() - t
    res[re] = e.get_params()
    print(f"{solver} k={k} z={z} {law}: seq_reassociate={re}: {len(order) / dt / 1e3:.0f} K examples/s", flush=True)
    e.close()
sc = np.max(np.abs(res[0][2]))
print(f"max |V_re - V_bitwise| / max|V| = {np.max(np.abs(res[1][2] - res[0][2])) / sc:.3e}; w0 {res[0][0]!r} vs {res[1][0]!r}")
