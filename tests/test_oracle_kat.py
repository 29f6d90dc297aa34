"""Pins the CPU oracle to the reference's known answers and to the
analytic FM identity (reference src/test/model.cpp:77-83)."""
import numpy as np

import oracle
from tests import kat


def _mat():
    return oracle.Matrix(kat.ROW_PTR, kat.COL, kat.VAL, kat.P_FEAT)


def test_sgd_kat():
    P = oracle.params(task=oracle.CLASSIFICATION, k=kat.K, l2_regw=kat.L2_REGW, l2_regv=kat.L2_REGV, learn_rate=0.05,
                      random_step=1, min_target=-1, max_target=1, eval_type=oracle.LL, trace_step=kat.TRACE_STEP,
                      conv_condition=0.0)
    r = oracle.sgd_learn(P, _mat(), kat.Y, 0.0, np.zeros(kat.P_FEAT), kat.harness_v0(), kat.MAX_ITER, trace_cap=16)
    assert r["iters"] == 50
    assert abs(r["w0"] - kat.SGD_W0) < 1e-15
    np.testing.assert_allclose(r["w"], kat.SGD_W, rtol=0, atol=1e-15)
    np.testing.assert_allclose(r["v"][: kat.P_FEAT], kat.SGD_V0, rtol=0, atol=1e-15)
    assert list(r["trace_iters"]) == kat.TRACE_ITERS
    np.testing.assert_allclose(r["trace_vals"], kat.SGD_LL, rtol=0, atol=5e-10)


def test_ftrl_kat():
    P = oracle.params(task=oracle.CLASSIFICATION, k=kat.K, l2_regw=kat.L2_REGW, l2_regv=kat.L2_REGV, l1_regw=0.001,
                      l1_regv=0.001, random_step=1, min_target=-1, max_target=1, eval_type=oracle.LL,
                      trace_step=kat.TRACE_STEP, conv_condition=0.0)
    r = oracle.ftrl_learn(P, _mat(), kat.Y, 0.0, np.zeros(kat.P_FEAT), kat.harness_v0(), kat.MAX_ITER, trace_cap=16)
    assert abs(r["w0"] - kat.FTRL_W0) < 1e-15
    assert list(r["trace_iters"]) == kat.TRACE_ITERS
    np.testing.assert_allclose(r["trace_vals"], kat.FTRL_LL, rtol=0, atol=5e-10)
    # feature 3 occurs only in row 0 (never visited, A-2) and single-nnz row 3 (pairwise grad == 0)
    assert np.all(r["v"].reshape(kat.K, kat.P_FEAT)[:, 3] == 0.0)


def test_tdap_kat():
    """TDAP is SURVEY row f-3; the reference's own numbers pin the restatement (incl. the z_w[i] bug, A-6)."""
    P = oracle.params(task=oracle.CLASSIFICATION, k=kat.K, l2_regw=kat.L2_REGW, l2_regv=kat.L2_REGV, gamma=kat.TDAP_GAMMA, random_step=1,
                      eval_type=oracle.LL, trace_step=kat.TRACE_STEP, conv_condition=0.0)
    r = oracle.tdap_learn(P, _mat(), kat.Y, 0.0, np.zeros(kat.P_FEAT), kat.harness_v0(), kat.MAX_ITER, trace_cap=16)
    assert abs(r["w0"] - kat.TDAP_W0) < 1e-15
    assert list(r["trace_iters"]) == kat.TRACE_ITERS
    np.testing.assert_allclose(r["trace_vals"], kat.TDAP_LL, rtol=0, atol=5e-10)
    assert np.all(r["v"].reshape(kat.K, kat.P_FEAT)[:, 3] == 0.0)
