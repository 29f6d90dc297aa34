"""Plan time of a coloured plan (CSC of the whole matrix + the capped exact walk + the colouring) on i.i.d. columns at configs[4]'s size, with the row-wise collision
test (default) and without (FMX_COLOUR_ROWS=0), and the share of the refit passes (FMX_COLOUR_REFIT=0)."""
import os, sys, time
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
for env in ({}, {"FMX_COLOUR_ROWS": "0"}, {"FMX_COLOUR_REFIT": "0"}):
    for k in ("FMX_COLOUR_ROWS", "FMX_COLOUR_REFIT"): os.environ.pop(k, None)
    os.environ.update(env)
    m = engine.Matrix.synthetic_iid(10_000_000, 1_000_000, 30, 20240001, law=L.COLUMNS_UNIFORM)
    e = engine.Engine(1_000_000, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=16, mode=L.MODE_SEQUENTIAL, als_max_levels=-2)
    e0 = engine.Engine(1_000_000, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=16, mode=L.MODE_SEQUENTIAL, als_max_levels=64)
    e0.sync(); t = time.perf_counter(); e0.als_plan(m); t_csc = time.perf_counter() - t      # (the CSC is the matrix's: built by the first plan of any kind; a capped plan adds ~0.05 s)
    e.sync(); t = time.perf_counter(); lv = e.als_plan(m)[0]; tp = time.perf_counter() - t
    print(f"{env or 'default'}: {lv} classes; CSC + a capped exact plan {t_csc:.2f} s, then the coloured plan {tp:.2f} s", flush=True)
    e.close(); e0.close(); m.close()
