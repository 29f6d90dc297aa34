"""Race hunt for the windowed sequential learner: many runs over different visiting orders of a 2 M x 1 M matrix, every
update kind, each compared bitwise with the strictly serial one-wave kernel.  Any cross-wave visibility problem between
groups (stores of group i not seen by the gathers of group i + 1) would show up as a mismatch."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import oracle
from fmwr_amd import _lib as L, engine
n, p, z = 2_000_000, 300_000, 30   # fewer features than the bench: ~20 % of the groups end early on a conflict
m = engine.Matrix.synthetic(n, p, z, 99)
bad = 0
t0 = time.time()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    k = [16, 8, 32, 64][rep % 4]
    solver, kw = [(L.SOLVER_SGD, dict(l2_w1=1e-4, l2_v=1e-4)), (L.SOLVER_SGD, dict(l1_w1=1e-5, l1_v=1e-5)),
                  (L.SOLVER_FTRL, dict(l1_w1=1e-4, l1_v=1e-4, l2_w1=1e-4, l2_v=1e-4)), (L.SOLVER_TDAP, dict(l1_w1=1e-4, l2_v=1e-4))][(rep // 4) % 4]
    order = oracle.visit_order(n, 1 + rep % 3, 120_000, seed=rep)
    v0 = np.random.default_rng(rep).normal(0, 0.01, (k, p))
    out = []
    for win in ("1", "0", "2"):  # windowed; one wave; windowed + pipelined groups (not for TDAP: "2" falls back to windowed there)
        os.environ["FMX_SEQ_WINDOW"] = win
        e = engine.Engine(p, solver=solver, num_factor=k, learn_rate=0.02, mode=L.MODE_SEQUENTIAL, **kw)
        e.set_params(0.0, None, v0)
        e.train_order(m, order)
        out.append(e.get_params())
        del e
    same = all(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True) and np.array_equal(np.asarray(c), np.asarray(b), equal_nan=True)
               for a, b, c in zip(out[0], out[1], out[2]))
    bad += not same
    print(f"rep {rep:3d} k={k:2d} solver={solver} stride={1 + rep % 3}: {'same' if same else 'MISMATCH'}", flush=True)
print(f"{bad} mismatches in {rep + 1} runs, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
