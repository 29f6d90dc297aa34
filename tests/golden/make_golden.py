"""Regenerates tests/golden/*.npz / kat.json.  Run from the repo root:  python tests/golden/make_golden.py

Two kinds of fixture, both DATA (inputs + expected outputs):
  kat.json        -- the reference's own known answers: SURVEY.md Appendix B (SGD/FTRL learners of the reference on a
                     6x5 matrix, produced in the survey session from the reference's headers).  Copied from tests/kat.py.
  oracle_v1.npz   -- outputs of oracle/fm_oracle.c (the CPU restatement, itself pinned to kat.json) on small seeded
                     problems covering the branches kat.json does not reach: L1 mode, regression clamp,
                     random_step > 1 (with the libc rand() visiting order recorded), FTRL l1+l2, ALS V sweep,
                     empty and single-nnz rows.  These freeze the oracle; they are NOT reference outputs.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import kat, util  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "sgd_l2_cls": dict(solver="sgd", task=oracle.CLASSIFICATION, k=8, l2_regw=1e-3, l2_regv=2e-3, l2_reg0=1e-3, learn_rate=0.05),
    "sgd_l1_cls": dict(solver="sgd", task=oracle.CLASSIFICATION, k=4, l1_regw=1e-3, l1_regv=5e-4, learn_rate=0.05),
    "sgd_l2_reg": dict(solver="sgd", task=oracle.REGRESSION, k=16, l2_regw=1e-3, l2_regv=1e-3, learn_rate=0.02),
    "sgd_rs3": dict(solver="sgd", task=oracle.CLASSIFICATION, k=4, l2_regv=1e-3, learn_rate=0.05, random_step=3),
    "ftrl_l1l2": dict(solver="ftrl", task=oracle.CLASSIFICATION, k=8, l1_regw=1e-3, l1_regv=1e-3, l2_regw=1e-2, l2_regv=1e-2),
    "tdap": dict(solver="tdap", task=oracle.CLASSIFICATION, k=4, l1_regw=1e-3, l1_regv=5e-4, l2_regw=1e-2, l2_regv=1e-2, gamma=3e-4, alpha_v=0.05),
}
LEARN = {"sgd": oracle.sgd_learn, "ftrl": oracle.ftrl_learn, "tdap": oracle.tdap_learn}
N, P, ITERS = 240, 60, 500


def problem(name, c):
    seed = sum(map(ord, name))
    rp, col, val = util.random_csr(N, P, 6, seed=seed)
    y = util.labels(N, seed, "classification" if c["task"] == oracle.CLASSIFICATION else "regression")
    w0, w, v = util.params(P, c["k"], seed, fp32=False)
    kw = {k: v_ for k, v_ in c.items() if k != "solver"}
    Pm = oracle.params(min_target=float(y.min()), max_target=float(y.max()), **kw)
    return rp, col, val, y, w0, w, v, Pm


def main():
    json.dump({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in vars(kat).items()
               if k.isupper()}, open(os.path.join(HERE, "kat.json"), "w"), indent=1)
    out = {}
    for name, c in CASES.items():
        rp, col, val, y, w0, w, v, Pm = problem(name, c)
        X = oracle.Matrix(rp, col, val, P)
        order = oracle.visit_order(N, Pm.random_step, ITERS, seed=1)
        r = LEARN[c["solver"]](Pm, X, y, w0, w, v.ravel(), ITERS, order=order)
        pred = oracle.predict_batch(Pm, X, r["w0"], r["w"], r["v"])
        for key, arr in dict(row_ptr=rp, col=col, val=val, y=y, w0_in=w0, w_in=w, v_in=v, order=order, w0=r["w0"], w=r["w"],
                             v=r["v"].reshape(c["k"], P), pred=pred).items():
            out[f"{name}/{key}"] = np.asarray(arr)
    # ALS V sweep
    rp, col, val = util.random_csr(N, P, 6, seed=77, empty_rows=False)
    y = util.labels(N, 77, "regression")
    w0, w, v = util.params(P, 3, 77, stdev=0.3, fp32=False)
    X = oracle.Matrix(rp, col, val, P)
    e0 = oracle.predict_batch(oracle.params(task=oracle.REGRESSION, k=3), X, w0, w, v.ravel()) - y
    lam, mu = np.array([0.0, 0.1, 0.5]), np.array([0.0, 0.05, -0.05])
    v1, e1, _ = oracle.als_update_v(3, X, v.ravel(), e0, alpha=1.3, v_lambda=lam, v_mu=mu)
    for key, arr in dict(row_ptr=rp, col=col, val=val, y=y, w0_in=w0, w_in=w, v_in=v, err_in=e0, v_lambda=lam, v_mu=mu, v=v1.reshape(3, P), err=e1).items():
        out[f"als/{key}"] = np.asarray(arr)
    np.savez_compressed(os.path.join(HERE, "oracle_v1.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
