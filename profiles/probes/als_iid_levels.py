"""Sizes of the colour classes on the i.i.d. law at configs[4]'s size (FMX_COLOUR_DEBUG) and the feature-major sweep's time."""
import os, sys, time, ctypes as C
if "FMX_LIB_PATH" not in os.environ: os.environ["FMX_COLOUR_DEBUG"] = "1"
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
from tests import util
N, P, Z, K, SEED = 10_000_000, 1_000_000, 30, 16, 20240001
if os.environ.get("BALLAST_GB"):   # experiment (profiles/r05_alloc_placement.txt): hold the first GBs of device memory so that the tables land beyond them
    import torch
    _ballast = torch.empty(int(float(os.environ["BALLAST_GB"]) * (1 << 30)), dtype=torch.uint8, device="cuda")
_early = None
if os.environ.get("EARLY_GB"):   # experiment (profiles/r06_allf_two_speeds.txt): take EARLY_GB of device memory FIRST in the process and give it back right before the sweep's tables are allocated
    import torch
    _early = torch.empty(int(float(os.environ["EARLY_GB"]) * (1 << 30)), dtype=torch.uint8, device="cuda")
    _early.fill_(0)
STRAT = len(sys.argv) > 1 and sys.argv[1] == "stratified"   # configs[4]'s own generator (one column per stratum and row): ~30 colours of ~33 000 features
m = engine.Matrix.synthetic(N, P, Z, SEED) if STRAT else engine.Matrix.synthetic_iid(N, P, Z, SEED, law=L.COLUMNS_UNIFORM, zipf_s=1.05)
e = engine.Engine(P, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=K, mode=L.MODE_SEQUENTIAL, als_max_levels=-2)
e.init_normal(SEED, 0.0, 0.01)
levels, largest, approx, _ = e.als_plan(m)
print("levels", levels, "largest", largest)
d_err = util.DevBuf(N)
L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(N), d_err.ptr, C.c_int(L.LINK_NONE)))
if _early is not None:
    del _early
    torch.cuda.empty_cache()
for i in range(3):
    e.sync(); t = time.perf_counter()
    e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=np.full(K, 1.0)); e.sync()
    print("sweep ms", (time.perf_counter() - t) * 1e3)
