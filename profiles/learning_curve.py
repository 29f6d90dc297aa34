"""Does the throughput mode LEARN?    Labels are planted from a known FM: y = +1 with probability
sigmoid(y_hat*(x)) under a hidden model (w*, V*), so there is a signal to recover and a best reachable held-out log-loss (the
planted model's own).  Every configuration trains from the same start on the same rows; after each slice of training the
held-out log-loss is evaluated (evaluation time is not counted).  Reported: held-out LL against cumulative training
wall-time and examples, and the time / examples needed to get 90 % of the way from the start LL to the planted model's LL.

  python profiles/learning_curve.py [sgd|ftrl] [small|bench]
"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine

solver = sys.argv[1] if len(sys.argv) > 1 else "sgd"
shape = sys.argv[2] if len(sys.argv) > 2 else "small"
if shape == "bench":   # BASELINE.json configs[1]'s own shape: a feature is seen ~300 times per pass
    n_train, n_test, p, z, k = 10_000_000, 1_000_000, 1_000_000, 30, 16
else:                  # every feature is seen ~6 000 times per pass: the planted model is learnable within a pass
    n_train, n_test, p, z, k = 4_000_000, 400_000, 20_000, 30, 8
seed = 11
rng = np.random.default_rng(seed)

def plant(m, pe):
    yhat = pe.predict(m)
    prob = 1.0 / (1.0 + np.exp(-yhat))
    y = np.where(rng.random(m.n) < prob, 1.0, -1.0).astype(np.float32)
    m.set_labels(y)
    # log-likelihood per example of the planted model itself (the reference's LL metric is the SUM / 2 of (1+y)log p + (1-y)log(1-p))
    return float(np.mean(np.where(y > 0, np.log(prob + 1e-20), np.log(1 - prob + 1e-20))))

train = engine.Matrix.synthetic(n_train, p, z, 5)
test = engine.Matrix.synthetic(n_test, p, z, 5, row_offset=n_train)
pe = engine.Engine(p, num_factor=k, mode=L.MODE_MINIBATCH)
wstar = rng.normal(0, 0.35, p); vstar = rng.normal(0, 0.12, (k, p))
pe.set_params(0.1, wstar, vstar)
ll_star_train = plant(train, pe); ll_star = plant(test, pe)
pe.close()
v0 = rng.normal(0, 0.01, (k, p)).astype(np.float32).astype(np.float64)

def heldout(e):
    return e.evaluate(test, L.EVAL_LL) / n_test   # per-example log-likelihood (core/Evaluation.h:80-89: sum of (1+y)log p + (1-y)log(1-p), halved)

common = dict(task=L.TASK_CLASSIFICATION, num_factor=k)
if solver == "sgd":
    common.update(solver=L.SOLVER_SGD, learn_rate=0.01, l2_w1=1e-5, l2_v=1e-5)
else:
    common.update(solver=L.SOLVER_FTRL, l1_w1=1e-6, l1_v=1e-6, l2_w1=1e-5, l2_v=1e-5)

configs = [("sequential (the reference's algorithm)", dict(mode=L.MODE_SEQUENTIAL), 2_000_000)]
for B in (4096, 16384, 65536, 262144, 1_048_576):
    configs.append((f"minibatch B={B} mean", dict(mode=L.MODE_MINIBATCH, batch_rows=B, batch_reduce=L.REDUCE_MEAN), None))
if solver == "sgd":   # a larger step for the larger batches (one mean-gradient step per coordinate per batch)
    for B, lr in ((65536, 0.05), (262144, 0.1), (1_048_576, 0.2)):
        configs.append((f"minibatch B={B} mean lr={lr}", dict(mode=L.MODE_MINIBATCH, batch_rows=B, batch_reduce=L.REDUCE_MEAN, learn_rate=lr), None))
configs.append(("minibatch B=4096 sum", dict(mode=L.MODE_MINIBATCH, batch_rows=4096, batch_reduce=L.REDUCE_SUM), 4_000_000))

for name, kw, cap in configs:   # occurrences of a coordinate per step: what a MEAN step folds into one update
    if "batch_rows" in kw:
        kw["_c"] = kw["batch_rows"] * z / p
results = {"solver": solver, "planted_ll_per_example": ll_star, "shape": dict(n_train=n_train, n_test=n_test, p=p, nnz=z, k=k), "runs": []}
print(f"solver {solver}: planted model's held-out LL/example {ll_star:.4f} (ln 2 = {-np.log(2):.4f} is a coin flip)")
for name, kw, cap in configs:
    c_per_step = kw.pop("_c", 1.0)
    e = engine.Engine(p, **dict(common, **kw))
    e.set_params(0.0, None, v0)
    if kw["mode"] == L.MODE_MINIBATCH:
        e.num_batches(train); e.sync()   # the one-off inverted-index build is not training time
    ll0 = heldout(e)
    target = ll0 + 0.9 * (ll_star - ll0)
    curve = [(0.0, 0, ll0)]
    t_total, seen = 0.0, 0
    slice_rows = 250_000 if kw["mode"] == L.MODE_SEQUENTIAL else 1_000_000
    budget = cap or 3 * n_train    # examples: up to 3 passes in the throughput mode
    hit = None
    while seen < budget:
        t = time.perf_counter(); done = e.train(train, slice_rows); e.sync(); t_total += time.perf_counter() - t
        seen += done
        ll = heldout(e)
        curve.append((t_total, seen, ll))
        if not np.isfinite(ll):
            break
        if hit is None and ll >= target:
            hit = (t_total, seen)
            break
    best = max(c[2] for c in curve if np.isfinite(c[2]))
    results["runs"].append(dict(name=name, occurrences_per_coordinate_per_step=c_per_step, start_ll=ll0, best_ll=best, target_ll=target, time_to_target_s=hit[0] if hit else None,
                                examples_to_target=hit[1] if hit else None, train_examples_per_s=seen / t_total, curve=curve))
    print(f"{name:42s} c={c_per_step:7.1f} best LL {best:8.4f}  to 90% of the planted gain: " + (f"{hit[0]:7.3f} s, {hit[1] / 1e6:5.1f} M examples" if hit else "   not reached") +
          f"   ({seen / 1e6:.0f} M examples at {seen / t_total / 1e6:.1f} M/s)")
    e.close()
print(json.dumps(results))
