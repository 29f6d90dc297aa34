"""Race hunt for the two kernels whose waves meet through flags instead of barriers: the reassociated reference-order learner (tagged LDS slots) and the persistent exact
sweep (replicated global counter, sc1 pair hand-off).  Every configuration is run REPEATS times from the same start; every run must give the same bits.
usage: python profiles/probes/flag_kernels_soak.py [repeats]"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, ".")
import oracle
from fmwr_amd import _lib as L, engine
from tests import util
REPEATS = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


bad = 0
t00 = time.perf_counter()
# ---- the reassociated learner: conflict regimes from "every example meets its neighbour" to none, three register shapes, both SGD kinds, two launches per run
for p, nnz in ((40, 6), (3000, 12), (200_000, 30)):
    for name, k, l1 in (("sgd_l2 k=8", 8, 0.0), ("sgd_l2 k=16", 16, 0.0), ("sgd_l1 k=16", 16, 1e-3), ("sgd_l2 k=32", 32, 0.0)):
        n = 30_000
        rp, col, val = util.random_csr(n, p, nnz, seed=p + k, empty_rows=True, max_nnz=32)
        y = util.labels(n, 5, "classification")
        w0, w, v = util.params(p, k, 3, fp32=False)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        order = oracle.visit_order(n, 3, 80_000, seed=7)
        seen = set()
        for r in range(REPEATS):
            e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, l1_w1=l1, l1_v=l1, l2_w1=1e-3, l2_v=1e-3, learn_rate=0.03, mode=L.MODE_SEQUENTIAL, seq_reassociate=1)
            e.set_params(w0, w, v)
            e.train_order(m, order)
            g = e.get_params()
            seen.add(digest(np.float64(g[0]), g[1], g[2]))
            e.close()
        bad += len(seen) != 1
        print(f"reassociated learner, p = {p}, {nnz} per row, {name}: {REPEATS} runs of 80 000 examples, {len(seen)} distinct result(s)", flush=True)
        m.close()
# ---- the persistent exact sweep: chain-shaped plans, ALS and Gibbs, one-hot and real values
for values, gibbs in (("ones", False), ("normal", True)):
    n, p, k = 20_000, 6_000, 16
    m0 = engine.Matrix.synthetic_iid(n, p, 30, 47, law=L.COLUMNS_UNIFORM)
    rp, col, val, _ = m0.export(); m0.close()
    if values == "normal":
        val = np.random.default_rng(47).normal(0, 1, len(val)).astype(np.float32)
    y = util.labels(n, 47, "regression")
    w0, w, v = util.params(p, k, 23, stdev=0.1, fp32=False)
    err0 = np.random.default_rng(1).normal(0, 1, n)
    lam = np.linspace(2.0, 4.0, k)
    z = np.random.default_rng(11).normal(0, 1, (k, p)) if gibbs else None
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    seen = set()
    for r in range(REPEATS):
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        g = e.als_vsweep(m, err0, alpha=1.0, v_lambda=lam, std_normals=z)
        g = e.als_vsweep(m, g, alpha=1.0, v_lambda=lam, std_normals=z)
        seen.add(digest(e.get_params()[2], g))
        e.close()
    bad += len(seen) != 1
    print(f"persistent exact sweep, 20 K x 6 K i.i.d. columns ({e.k} factors, {'Gibbs' if gibbs else 'ALS'}, values {values}): {REPEATS} runs of two sweeps, {len(seen)} distinct result(s)", flush=True)
    m.close()
print(f"{'ALL IDENTICAL' if bad == 0 else f'{bad} CONFIGURATIONS DIFFER'} in {time.perf_counter() - t00:.0f} s")
sys.exit(1 if bad else 0)
