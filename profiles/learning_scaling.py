"""Which step size should an N-GPU job run?    The N-GPU job with B rows per GPU per step IS the one-GPU job with N x B rows
per step (the replicas apply the identical update from the identical sums: bitwise, tests/test_gpu_distributed.py), so its learning curve needs no
second device: train on ONE GPU with batch_rows = the GLOBAL batch.  BASELINE.json configs[1]'s own shape (10 M x 1 M, 30 per row, k = 16), labels
planted from a hidden FM (profiles/learning_curve.py), held-out log-likelihood per example after 1, 2 and 3 passes, from the same start:

  * SGD, mean gradient per coordinate per step, at the reference's learning rate lr = 0.01, at lr x sqrt(N) and at lr x N (N = global batch / 262 144);
  * FTRL (per-coordinate adaptive steps), same batches.

  python profiles/learning_scaling.py > profiles/r04_learning_scaling.txt
"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine

n_train, n_test, p, z, k = 10_000_000, 1_000_000, 1_000_000, 30, 16
rng = np.random.default_rng(11)


def plant(m, pe):
    yhat = pe.predict(m)
    prob = 1.0 / (1.0 + np.exp(-yhat))
    y = np.where(rng.random(m.n) < prob, 1.0, -1.0).astype(np.float32)
    m.set_labels(y)
    return float(np.mean(np.where(y > 0, np.log(prob + 1e-20), np.log(1 - prob + 1e-20))))


train = engine.Matrix.synthetic(n_train, p, z, 5)
test = engine.Matrix.synthetic(n_test, p, z, 5, row_offset=n_train)
pe = engine.Engine(p, num_factor=k, mode=L.MODE_MINIBATCH)
pe.set_params(0.1, rng.normal(0, 0.35, p), rng.normal(0, 0.12, (k, p)))
plant(train, pe); ll_star = plant(test, pe)
pe.close()
v0 = rng.normal(0, 0.01, (k, p)).astype(np.float32).astype(np.float64)
BASE = 262_144
print(f"planted model's held-out LL/example {ll_star:.4f}; coin flip {-np.log(2):.4f}; one pass = {n_train} examples; occurrences of a coordinate per step c = rows x {z} / {p}")
print(f"{'solver':6s} {'global rows/step':>16s} {'= N x 262144':>12s} {'c':>7s} {'lr':>8s} | held-out LL after 1 / 2 / 3 passes | M examples/s")
rows = []
for solver in ("sgd", "ftrl"):
    for G in (65_536, BASE, 2 * BASE, 4 * BASE, 8 * BASE, 16 * BASE, 32 * BASE):
        N = G / BASE
        lrs = [0.01] if solver == "ftrl" else sorted({0.01, round(0.01 * max(N, 1.0) ** 0.5, 5), round(0.01 * max(N, 1.0), 5)})
        for lr in lrs:
            kw = dict(task=L.TASK_CLASSIFICATION, num_factor=k, mode=L.MODE_MINIBATCH, batch_rows=G, batch_reduce=L.REDUCE_MEAN)
            if solver == "sgd":
                kw.update(solver=L.SOLVER_SGD, learn_rate=lr, l2_w1=1e-5, l2_v=1e-5)
            else:
                kw.update(solver=L.SOLVER_FTRL, l1_w1=1e-6, l1_v=1e-6, l2_w1=1e-5, l2_v=1e-5)
            e = engine.Engine(p, **kw)
            e.set_params(0.0, None, v0)
            nb = e.num_batches(train); e.sync()
            per_pass = (n_train // G) * G if G <= n_train else n_train
            lls, t_total, seen = [], 0.0, 0
            for _ in range(3):
                t = time.perf_counter(); seen += e.train(train, per_pass); e.sync(); t_total += time.perf_counter() - t
                lls.append(e.evaluate(test, L.EVAL_LL) / n_test)
            rows.append(dict(solver=solver, global_rows=G, n_equiv=N, c=G * z / p, lr=lr, ll=lls, rate=seen / t_total))
            print(f"{solver:6s} {G:16d} {N:12.2f} {G * z / p:7.1f} {lr if solver == 'sgd' else float('nan'):8.4f} | " + " / ".join(f"{x:8.4f}" for x in lls) + f" | {seen / t_total / 1e6:7.1f}", flush=True)
            e.close()
print(json.dumps(dict(planted=ll_star, rows=rows)))
