"""What the feature-major nesting (cfg.als_max_levels = -2) costs or buys in LEARNING: the ALS learner (w0, w sweep, V sweep; regression) on configs[4]'s matrix with
targets planted from a hidden FM + noise, iteration by iteration, in the reference's order (als_max_levels = 0: block form) and in (level, feature, factor) order.
Training and held-out RMSE after every iteration, seconds per iteration."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
N, P, Z, K, SEED = 10_000_000, 1_000_000, 30, 16, 20240001
NT = 1_000_000
rng = np.random.default_rng(11)
IID = len(sys.argv) > 1 and sys.argv[1] == "iid"   # SURVEY 8(d)'s i.i.d. columns: the reference's order is a chain of ~19 400 levels there (2 s per sweep); the coloured orders are what one would run
mk = (lambda n, off: engine.Matrix.synthetic_iid(n, P, Z, SEED, law=L.COLUMNS_UNIFORM, row_offset=off)) if IID else (lambda n, off: engine.Matrix.synthetic(n, P, Z, SEED, row_offset=off))
train, test = mk(N, 0), mk(NT, N)
pe = engine.Engine(P, task=L.TASK_REGRESSION, num_factor=K, mode=L.MODE_MINIBATCH, min_target=-100.0, max_target=100.0)
pe.set_params(0.1, rng.normal(0, 0.35, P), rng.normal(0, 0.12, (K, P)))
def plant(m):
    y = (pe.predict(m) + rng.normal(0, 0.5, m.n)).astype(np.float32)
    m.set_labels(y)
    return y
ytr, yte = plant(train), plant(test)
pe.close()
print(f"targets: planted FM (w ~ N(0, 0.35), V ~ N(0, 0.12)) + N(0, 0.5) noise; var(y) = {ytr.var():.3f}; the noise floor is RMSE 0.5", flush=True)
v0 = np.random.default_rng(5).normal(0, 0.01, (K, P))
runs = ((0, "reference order (factor outer): the exact schedule", 3), (-1, "coloured order, factor outer", 6), (-2, "coloured order, feature-major", 6)) if IID else \
       ((0, "reference order (factor outer), block form", 6), (-2, "feature-major: (level, feature, factor)", 6))
for cap, name, iters in runs:
    e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL, als_max_levels=cap, min_target=-100.0, max_target=100.0)
    e.set_params(0.0, None, v0)
    e.sync(); t = time.perf_counter()
    levels = e.als_plan(train)[0]
    print(f"-- {name} (plan kind {e.als_plan_kind(train)}, {levels} levels, built in {time.perf_counter() - t:.2f} s)", flush=True)
    for it in range(iters):
        e.sync(); t = time.perf_counter()
        e.als_train(train, 1, with_v=True); e.sync()
        dt = time.perf_counter() - t
        rtr = float(np.sqrt(np.mean((e.predict(train) - ytr) ** 2))); rte = float(np.sqrt(np.mean((e.predict(test) - yte) ** 2)))
        print(f"   iteration {it + 1}: {dt * 1e3:7.1f} ms   train RMSE {rtr:.4f}   held-out RMSE {rte:.4f}", flush=True)
    e.close()
