"""The probit tables of the ALS / MCMC path (util/Random.h:95-124).  The reference ships
them as ~1 MB of literals; the oracle (and, with the same host code, the engine) regenerates them from their defining
formulas.  Pinned here against a committed sample of the reference's values, and against the full files where
/root/reference exists (the build container)."""
import json
import os
import re

import numpy as np
import pytest
from scipy import special

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/util"
PN_TOL, DP_TOL = 2e-15, 5e-11  # 15 printed digits; 12 printed decimals of a ratio with ~1e-10 cancellation noise near x = 5


def test_regenerated_grids_match_the_committed_sample():
    g = json.load(open(os.path.join(HERE, "golden", "probit_tables.json")))
    pn, dp = oracle.probit_tables()
    assert len(pn) == g["pnorm"]["points"] == 2861 and len(dp) == g["dpnorm"]["points"] == 40001
    assert g["pnorm"]["max"] == 5.20031455849973 and g["pnorm"]["hinv"] == 549.966731401936
    assert (g["dpnorm"]["min"], g["dpnorm"]["max"]) == (-3.0, 5.0)
    assert np.max(np.abs(pn[g["pnorm"]["index"]] - np.array(g["pnorm"]["y"]))) < PN_TOL
    d = np.abs(dp[g["dpnorm"]["index"]] - np.array(g["dpnorm"]["y"]))
    assert d.max() < DP_TOL and np.mean(d == 0) > 0.99  # all but a few points reproduce the printed value exactly


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources not present (only in the build container)")
def test_regenerated_grids_match_the_reference_files():
    def array(path, name):
        t = open(path).read()
        i = t.index(name + " "); j = t.index("{", i); k = t.index("}", j)
        return np.array([float(x) for x in t[j + 1:k].replace("\n", " ").split(",") if x.strip()])
    pn, dp = oracle.probit_tables()
    ref_pn_x, ref_pn = array(os.path.join(REF, "RandomData.h"), "_X_"), array(os.path.join(REF, "RandomData.h"), "_Y_")
    ref_dp_x, ref_dp = array(os.path.join(REF, "RandomData_.h"), "__X__"), array(os.path.join(REF, "RandomData_.h"), "__Y__")
    assert np.max(np.abs(pn - ref_pn)) < PN_TOL
    assert np.max(np.abs(dp - ref_dp)) < DP_TOL and np.mean(dp == ref_dp) > 0.99
    # the grids themselves
    assert np.max(np.abs(ref_pn_x - np.arange(2861) / 549.966731401936)) < 1e-14
    assert np.array_equal(ref_dp_x, (-30000 + 2 * np.arange(40001)) / 10000.0)


def test_fast_pnorm_semantics():
    x = np.array([0.0, 1e-9, 0.3, -0.3, 1.0, -2.5, 5.2, 5.2003, 5.21, -7.0, 40.0])
    got = oracle.fast_pnorm(x)
    true = special.ndtr(x)
    inside = np.abs(x) <= 5.20031455849973
    assert np.max(np.abs(got[inside] - true[inside])) < 2e-7      # linear interpolation error of the 1.8e-3 grid
    assert np.all(got[(x > 5.20031455849973)] == 0.999999900524235)  # "truncated", Random.h:102
    assert np.all(got[(x < -5.20031455849973)] == 1.0 - 0.999999900524235)
    assert oracle.fast_pnorm([0.0])[0] == 0.5
    xs = np.linspace(-5, 5, 2001)
    assert np.allclose(oracle.fast_pnorm(xs) + oracle.fast_pnorm(-xs), 1.0, atol=1e-15)


def test_fast_dpnorm_semantics():
    x = np.array([-3.0001, -3.0, -1.0, 0.0, 0.00013, 2.0, 4.9999, 5.0, 5.0001, 50.0])
    got = oracle.fast_dpnorm(x)
    assert got[0] == 0.0                                                          # Random.h:119
    mills = np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi) / (0.5 * special.erfc(x / np.sqrt(2)))
    inside = (x >= -3) & (x <= 5)
    assert np.max(np.abs(got[inside] - mills[inside])) < 1e-8                     # 2e-4 grid, linear interpolation
    tail = x > 5
    ax = np.abs(x)
    asym = 0.1943369 + 0.9754752 * x + 0.4136861 * np.sqrt(ax) - 0.5034295 * np.log(ax + 1e-07)
    assert np.array_equal(got[tail], asym[tail])                                  # Random.h:120


def test_als_classification_residual_and_learning():
    """calculate_error's CLASSIFICATION branch (MCMC_ALS_Learner.h:545-559) drives the ALS loop to a separating model."""
    from tests import util
    n, p, k = 600, 40, 3
    rp, col, val = util.random_csr(n, p, 6, seed=4)
    rng = np.random.default_rng(4)
    wt = rng.normal(0, 1, p)
    X = oracle.Matrix(rp, col, val, p)
    score = oracle.predict_batch(oracle.params(k=0), X, 0.1, wt, np.zeros(0))
    y = np.where(score > 0, 1.0, -1.0).astype(np.float32)
    P = oracle.params(task=oracle.CLASSIFICATION, k=k, l2_reg0=0.0)
    w0, w, v = oracle.als_learn(P, X, y, 0.0, np.zeros(p), rng.normal(0, 0.01, (k, p)).ravel(), 15)
    prob = oracle.predict_batch(P, X, w0, w, v, prob="probit")
    assert np.all((prob > 0) & (prob < 1))
    assert np.mean((prob >= 0.5) == (y > 0)) > 0.93
    # one iteration by hand: e = -dpnorm(-yhat) for y >= 0, dpnorm(yhat) otherwise, at the start (yhat = 0): -/+ 0.7978845608
    e0 = oracle.fast_dpnorm([0.0])[0]
    assert abs(e0 - np.sqrt(2 / np.pi)) < 1e-11


def test_mcmc_learner_restatement_recovers_a_linear_model():
    """fmo_mcmc_learn (MCMC_Learner with caller-drawn variates): on y = w.x + w0 + noise the chain settles on the truth and
    alpha on 1/noise^2; the same variates give the same chain; CLASSIFICATION (truncated normals from libc rand()) separates."""
    from tests import util
    n, p, k = 800, 50, 3
    rp, col, val = util.random_csr(n, p, 6, seed=3)
    X = oracle.Matrix(rp, col, val, p)
    rng = np.random.default_rng(0)
    wt = rng.normal(0, 1, p)
    y = (oracle.predict_batch(oracle.params(k=0), X, 0.3, wt, np.zeros(0)) + rng.normal(0, 0.1, n)).astype(np.float32)
    it = 30
    a1, a2 = oracle.mcmc_draw_shapes(n, p)
    assert (a1, a2) == ((1 + n) / 2, (2 + p) / 2)
    G = np.stack([rng.gamma(a1, 1.0, it), rng.gamma(a2, 1.0, it)], 1)
    Z = rng.normal(0, 1, (it, 2 + p))
    v0 = rng.normal(0, 0.01, (k, p)).ravel()
    P = oracle.params(task=oracle.REGRESSION, k=k, min_target=float(y.min()), max_target=float(y.max()))
    w0, w, v, st = oracle.mcmc_learn(P, X, y, 0.0, np.zeros(p), v0, it, G, Z)
    assert np.max(np.abs(w - wt)) < 0.05 and abs(w0 - 0.3) < 0.05 and 50 < st[0] < 200   # alpha ~ 1 / 0.1^2
    assert np.array_equal(v, v0)                                                          # V is never updated (SURVEY A-1)
    again = oracle.mcmc_learn(P, X, y, 0.0, np.zeros(p), v0, it, G, Z)
    assert again[0] == w0 and np.array_equal(again[1], w)
    ycls = np.where(y > np.median(y), 1, -1).astype(np.float32)
    Pc = oracle.params(task=oracle.CLASSIFICATION, k=k)
    c0, cw, cv, _ = oracle.mcmc_learn(Pc, X, ycls, 0.0, np.zeros(p), v0, it, G, Z, seed=5)
    prob = oracle.predict_batch(Pc, X, c0, cw, cv, prob="probit")
    assert np.mean((prob >= 0.5) == (ycls > 0)) > 0.93
    d0, dw, _, _ = oracle.mcmc_learn(Pc, X, ycls, 0.0, np.zeros(p), v0, it, G, Z, seed=5)
    assert d0 == c0 and np.array_equal(dw, cw)                                            # same rand() seed, same chain
