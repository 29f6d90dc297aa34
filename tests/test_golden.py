"""Committed golden fixtures: the oracle must keep reproducing them (CPU), and the GPU must match them (gpu)."""
import json
import os

import numpy as np
import pytest

import oracle
from tests import kat
from tests.golden import make_golden as mg

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    return np.load(os.path.join(HERE, "oracle_v1.npz"))


def test_kat_fixture_is_the_survey_appendix_b_data():
    d = json.load(open(os.path.join(HERE, "kat.json")))
    assert d["SGD_W0"] == kat.SGD_W0 and d["FTRL_W0"] == kat.FTRL_W0 and d["SGD_LL"] == kat.SGD_LL and d["COL"] == kat.COL.tolist()


@pytest.mark.parametrize("name", list(mg.CASES))
def test_oracle_reproduces_fixture(name):
    g = _load()
    c = mg.CASES[name]
    rp, col, val, y, w0, w, v, Pm = mg.problem(name, c)
    np.testing.assert_array_equal(col, g[f"{name}/col"])
    X = oracle.Matrix(rp, col, val, mg.P)
    r = mg.LEARN[c["solver"]](Pm, X, y, w0, w, v.ravel(), mg.ITERS, order=g[f"{name}/order"])
    np.testing.assert_allclose(r["v"].reshape(c["k"], mg.P), g[f"{name}/v"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(r["w"], g[f"{name}/w"], rtol=0, atol=1e-14)
    assert abs(r["w0"] - float(g[f"{name}/w0"])) < 1e-14
    if Pm.random_step > 1:  # the libc rand() stride list is part of the fixture (SURVEY A-4): glibc, state of srand(1)
        np.testing.assert_array_equal(oracle.visit_order(mg.N, Pm.random_step, mg.ITERS, seed=1), g[f"{name}/order"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mg.CASES))
def test_gpu_sequential_matches_fixture(name):
    from fmwr_amd import _lib as L, engine
    g = _load()
    c = mg.CASES[name]
    _, _, _, y, _, _, _, Pm = mg.problem(name, c)
    e = engine.Engine(mg.P, task=Pm.task, solver={"sgd": L.SOLVER_SGD, "ftrl": L.SOLVER_FTRL, "tdap": L.SOLVER_TDAP}[c["solver"]], num_factor=Pm.k, gamma=Pm.gamma,
                      l2_w0=Pm.l2_reg0, l1_w1=Pm.l1_regw, l2_w1=Pm.l2_regw, l1_v=Pm.l1_regv, l2_v=Pm.l2_regv, learn_rate=Pm.learn_rate,
                      alpha_w=Pm.alpha_w, alpha_v=Pm.alpha_v, beta_w=Pm.beta_w, beta_v=Pm.beta_v, mode=L.MODE_SEQUENTIAL,
                      min_target=Pm.min_target, max_target=Pm.max_target)
    e.set_params(float(g[f"{name}/w0_in"]), g[f"{name}/w_in"], g[f"{name}/v_in"])
    m = engine.Matrix.from_csr(g[f"{name}/row_ptr"], g[f"{name}/col"], g[f"{name}/val"], mg.P, g[f"{name}/y"])
    e.train_order(m, g[f"{name}/order"])
    w0, w, v = e.get_params()
    scale = np.max(np.abs(g[f"{name}/v"]))
    assert np.max(np.abs(v - g[f"{name}/v"])) < 1e-11 * scale
    assert np.max(np.abs(w - g[f"{name}/w"])) < 1e-11 * max(np.max(np.abs(g[f"{name}/w"])), 1e-300)
    out = e.predict(m)
    assert np.array_equal(np.sign(out), np.sign(g[f"{name}/pred"]))


@pytest.mark.gpu
def test_gpu_als_matches_fixture():
    from fmwr_amd import _lib as L, engine
    g = _load()
    e = engine.Engine(mg.P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=3, mode=L.MODE_SEQUENTIAL)
    e.set_params(float(g["als/w0_in"]), g["als/w_in"], g["als/v_in"])
    m = engine.Matrix.from_csr(g["als/row_ptr"], g["als/col"], g["als/val"], mg.P, g["als/y"])
    err = e.als_vsweep(m, g["als/err_in"], alpha=1.3, v_lambda=g["als/v_lambda"], v_mu=g["als/v_mu"])
    assert np.max(np.abs(e.get_params()[2] - g["als/v"])) < 1e-10 * np.max(np.abs(g["als/v"]))
    assert np.max(np.abs(err - g["als/err"])) < 1e-10 * np.max(np.abs(g["als/err"]))
