"""fmx_train_grid: n reference-order learners side by side on one matrix and one visiting order (one workgroup per model, the examples' conflict plan shared):
SGD-L2 / SGD-L1 / FTRL / TDAP (the reference's default solver), the shapes the pipelined kernel takes and those that keep the windowed one.
Every model must come out bit for bit as its own fmx_train call leaves it -- which the oracle tests pin to the reference's algorithm (tests/test_gpu_train.py)."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("solver,k,z", [("sgd_l2", 16, 30), ("sgd_l2", 40, 12), ("sgd_l1", 8, 30), ("ftrl", 16, 20), ("tdap", 6, 30), ("ftrl", 40, 16), ("sgd_l1", 24, 48)])
def test_a_grid_of_models_equals_the_models_trained_one_by_one(solver, k, z):
    from fmwr_amd import _lib as L, engine
    n, p, iters = 30_000, 20_000, 90_000     # three passes over the matrix: the visiting order wraps (row 0 is never visited: SURVEY A-2)
    m = engine.Matrix.synthetic(n, p, z, 77)
    v0 = np.random.default_rng(5).normal(0, 0.05, (k, p))
    grid = []
    for i in range(5):
        kw = dict(num_factor=k, mode=L.MODE_SEQUENTIAL, task=L.TASK_CLASSIFICATION)
        if solver == "sgd_l2": kw.update(solver=L.SOLVER_SGD, learn_rate=0.01 * (1 + i), l2_w1=1e-4 * (1 + i), l2_v=1e-4)
        elif solver == "sgd_l1": kw.update(solver=L.SOLVER_SGD, learn_rate=0.02 / (1 + i), l1_w1=1e-5 * (1 + i), l1_v=1e-5)
        elif solver == "tdap": kw.update(solver=L.SOLVER_TDAP, alpha_w=0.05 * (1 + i), alpha_v=0.05, beta_w=1.0, beta_v=1.0, l1_w1=1e-3, l1_v=5e-4, l2_w1=1e-2, l2_v=1e-2, gamma=3e-4 * (1 + i))
        else: kw.update(solver=L.SOLVER_FTRL, alpha_w=0.05 * (1 + i), alpha_v=0.05, beta_w=1.0, beta_v=1.0, l1_w1=1e-4, l1_v=1e-4, l2_w1=1e-4, l2_v=1e-4)
        grid.append(kw)
    singles = []
    for kw in grid:
        e = engine.Engine(p, **kw)
        e.set_params(0.1, None, v0)
        e.train(m, iters)
        singles.append(e.get_params())
        e.close()
    es = [engine.Engine(p, **kw) for kw in grid]
    for e in es:
        e.set_params(0.1, None, v0)
    assert engine.Engine.train_grid(es, m, iters) == iters
    for e, ref in zip(es, singles):
        got = e.get_params()
        assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    # the models differ from one another (the grid is not five copies of one model)
    assert not np.array_equal(es[0].get_params()[2], es[4].get_params()[2])
    for e in es:
        e.close()
    m.close()


def test_a_grid_refuses_engines_of_different_shapes():
    from fmwr_amd import _lib as L, engine
    m = engine.Matrix.synthetic(2_000, 1_000, 10, 3)
    a = engine.Engine(1_000, num_factor=8, mode=L.MODE_SEQUENTIAL, solver=L.SOLVER_SGD)
    b = engine.Engine(1_000, num_factor=16, mode=L.MODE_SEQUENTIAL, solver=L.SOLVER_SGD)
    c = engine.Engine(1_000, num_factor=8, mode=L.MODE_MINIBATCH, solver=L.SOLVER_SGD)
    for bad in ([a, b], [a, c], [a, a]):
        with pytest.raises(L.FmxError):
            engine.Engine.train_grid(bad, m, 100)
    for e in (a, b, c):
        e.close()
    m.close()
