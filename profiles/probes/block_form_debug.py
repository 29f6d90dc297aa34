"""Debug aid: the V sweep of one small stratified matrix through every form (FMX_ALS_ORDER = 2, 1, 0) against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from tests import util
from tests.test_gpu_configs4 import _problem, K, Z
from fmwr_amd import _lib as L, engine

def run(values, gibbs, block_rows, n, p, orders=("2", "1", "0"), lam_lo=0.1, lam_hi=0.5):
    os.environ["FMX_ALS_TILED"] = "1"; os.environ["FMX_ALS_TILE_ROWS"] = "4096"
    if block_rows: os.environ["FMX_ALS_BLOCK_ROWS"] = str(block_rows)
    rp, col, val, y = _problem(engine, L, "stratified", n, p, 67, values)
    w0, w, v = util.params(p, K, 37, stdev=0.1, fp32=False)
    X = oracle.Matrix(rp, col, val, p)
    P = oracle.params(task=oracle.REGRESSION, k=K)
    err0 = oracle.predict_batch(P, X, w0, w, v.ravel()) - y
    lam = np.linspace(lam_lo, lam_hi, K); mu = np.linspace(-0.05, 0.05, K)
    z = np.random.default_rng(11).normal(0, 1, (K, p)) if gibbs else None
    rv, rerr, _ = oracle.als_update_v(K, X, v.ravel(), err0, alpha=1.1, v_lambda=lam, v_mu=mu, znorm=z.ravel() if gibbs else None)
    for order in orders:
        os.environ["FMX_ALS_ORDER"] = order
        e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL)
        e.set_params(w0, w, v)
        m = engine.Matrix.from_csr(rp, col, val, p, y)
        form = e.als_level_order_form(m)
        gerr = e.als_vsweep(m, err0, alpha=1.1, v_lambda=lam, v_mu=mu, std_normals=z)
        gv = e.get_params()[2]
        d = np.abs(gv - rv.reshape(K, p))
        f, j = np.unravel_index(np.argmax(d), d.shape)
        print(f"values={values} gibbs={gibbs} R={block_rows} n={n} p={p} order={order} form={form}: V rel err {util.rel_err(gv, rv.reshape(K, p)):.3e} e rel err {util.rel_err(gerr, rerr):.3e}"
              f" worst at factor {f} feature {j} (entries wrong by > 1e-9: {int(np.sum(d > 1e-9))}, first wrong factor {int(np.argmax(d.max(axis=1) > 1e-9))})", flush=True)
        e.close(); m.close()

if __name__ == "__main__":
    for lo, hi in ((0.1, 0.5), (1.5, 3.0), (10.0, 20.0), (50.0, 100.0)):
        print("lambda", lo, hi)
        run("normal", True, 1024, 16384, 30000, orders=("2", "0"), lam_lo=lo, lam_hi=hi)
        run("normal", True, 2048, 12000, 150000, orders=("2", "0"), lam_lo=lo, lam_hi=hi)
        run("ones", True, 4096, 9000, 600000, orders=("2", "0"), lam_lo=lo, lam_hi=hi)
