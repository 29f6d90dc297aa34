import sys
import numpy as np
sys.path.insert(0, ".")
import oracle
from tests import util
from tests.test_gpu_fuzz import _case
from fmwr_amd import _lib as L, engine
for seed in [int(a) for a in sys.argv[1:]]:
    c = _case(seed); n, p, k = c["n"], c["p"], c["k"]
    rp, col, val = util.random_csr(n, p, c["mean_nnz"], seed=seed)
    y = util.labels(n, seed, "classification" if c["task"] == oracle.CLASSIFICATION else "regression")
    w0, w, v = util.params(p, k, seed, stdev=0.2, fp32=True)
    reg = dict(l2_regw=1e-3, l2_regv=2e-3, l2_reg0=1e-3)
    if c["solver"] == "sgd_l1": reg.update(l1_regw=1e-3, l1_regv=5e-4)
    if c["solver"] == "ftrl": reg.update(l1_regw=1e-3, l1_regv=1e-3)
    P = oracle.params(task=c["task"], k=k, k0=c["k0"], k1=c["k1"], learn_rate=0.03, batch_mean=c["mean"], min_target=float(y.min()) if n else -1.0, max_target=float(y.max()) if n else 1.0, **reg)
    X = oracle.Matrix(rp, col, val, p)
    vflat = v.ravel() if k else np.zeros(1)
    kw = dict(task=c["task"], solver=L.SOLVER_FTRL if c["solver"] == "ftrl" else L.SOLVER_SGD, num_factor=k, keep_w0=int(c["k0"]), keep_w1=int(c["k1"]),
              l2_w0=P.l2_reg0, l1_w1=P.l1_regw, l2_w1=P.l2_regw, l1_v=P.l1_regv, l2_v=P.l2_regv, learn_rate=0.03, min_target=P.min_target, max_target=P.max_target)
    m = engine.Matrix.from_csr(rp, col, val, p, y)
    for wide in (0, 1):
        e = engine.Engine(p, mode=L.MODE_MINIBATCH, batch_rows=c["batch"], tile_rows=c["tile"], batch_reduce=L.REDUCE_MEAN if c["mean"] else L.REDUCE_SUM, state_fp64=wide, **kw)
        e.set_params(w0, w, v if k else None)
        total = min(3 * n + 1, 2500)
        mb = (oracle.FtrlMinibatch if c["solver"] == "ftrl" else oracle.SgdMinibatch)(P, X, y, w0, w, vflat)
        done, step, nb = 0, 0, -(-n // c["batch"])
        while done < total:
            b0 = (step % nb) * c["batch"]; rows = min(c["batch"], n - b0, total - done)
            mb.step(b0, b0 + rows); done += rows; step += 1
        assert e.train(m, total) == total
        g0, gw, gv = e.get_params()
        dv = np.max(np.abs(gv - mb.v.reshape(k, p))) / max(np.max(np.abs(mb.v)), 1e-3) if k else 0.0
        dw = np.max(np.abs(gw - mb.w)) / max(np.max(np.abs(mb.w)), 1e-3)
        d0 = abs(g0 - mb.w0.value) / max(1.0, abs(mb.w0.value))
        print("seed %d %s state_fp64=%d steps %d: rel dev V %.3g w %.3g w0 %.3g   |V|max %.3g |w|max %.3g" % (seed, c, wide, step, dv, dw, d0, np.max(np.abs(mb.v)) if k else 0, np.max(np.abs(mb.w))), flush=True)
    if n > 1:
        iters = min(2 * n + 3, 400)
        learn = oracle.ftrl_learn if c["solver"] == "ftrl" else oracle.sgd_learn
        ref = learn(P, X, y, w0, w, vflat, iters)
        es = engine.Engine(p, mode=L.MODE_SEQUENTIAL, **kw); es.set_params(w0, w, v if k else None)
        es.train(m, iters); s0, sw, sv = es.get_params()
        print("   sequential: V %.3g w %.3g w0 %.3g" % (np.max(np.abs(sv - ref["v"].reshape(k, p))) / max(np.max(np.abs(ref["v"])), 1e-3) if k else 0, np.max(np.abs(sw - ref["w"])) / max(np.max(np.abs(ref["w"])), 1e-3), abs(s0 - ref["w0"])))
