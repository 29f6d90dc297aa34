"""fmwr_amd -- MI355X-native engine for the hot path of evanwang1990/FMwR (FM forward, SGD / FTRL-Proximal step,
ALS V sweep) behind the C ABI of include/fmx.h.  Build the library first: python -m fmwr_amd.build"""
from .api import (ALS_solver, FTRL_solver, FmMatrix, MCMC_solver, SGD_solver, TDAP_solver, fm_matrix, fm_select, fm_track,  # noqa: F401
                  fm_train, fm_update, model_control, predict, solver_control, track_control)
from .engine import Engine, Matrix  # noqa: F401
