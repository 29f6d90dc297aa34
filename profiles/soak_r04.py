"""Round-4 soak at full size: every schedule / form variant of one job, each in a process of its own, must leave the same bits (or, where a form associates
differently by design, the same bits as itself).  (a) configs[1]: one pass over 10 M x 1 M (38 steps of 262 144 rows), phase-1 schedule serial / pipelined / as
measured, twice; (b) the same on ragged rows (Poisson(30) in [1, 64]): static kernel serial / pipelined, lane groups pulling rows -- one hash; the flat form twice --
one hash of its own; (c) configs[4]: one Gibbs sweep of 16 factors, row-tiled twice and with 4 lanes per list -- one hash.   python profiles/soak_r04.py"""
import os, subprocess, sys
CHILD = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
job = sys.argv[1]
n, p, z, k, B = 10_000_000, 1_000_000, 30, 16, 262_144
if job in ("sgd", "ragged"):
    m = engine.Matrix.synthetic(n, p, z, 20240001) if job == "sgd" else engine.Matrix.synthetic_ragged(n, p, float(z), 20240001)
    e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
    e.init_normal(1, 0.0, 0.01)
    assert e.train(m, n) == n
    w0, w, v = e.get_params()
    print("HASH", hashlib.sha256(v.tobytes() + w.tobytes() + np.float64(w0).tobytes()).hexdigest()[:16], "form", m.rows_form())
else:
    import ctypes as C
    from tests import util
    m = engine.Matrix.synthetic(n, p, z, 20240001)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC, num_factor=k)
    e.init_normal(3, 0.0, 0.05)
    d_err = util.DevBuf(n)
    L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(n), d_err.ptr, C.c_int(L.LINK_NONE)))
    rng = np.random.default_rng(5)
    d_nrm = util.DevBuf.from_numpy(rng.standard_normal(k * p))
    e.vsweep_device(m, d_err.ptr.value, alpha=1.0, v_lambda=np.full(k, 2.0), v_mu=np.zeros(k), dev_std_normals=d_nrm.ptr.value)
    e.sync()
    _, _, v = e.get_params()
    print("HASH", hashlib.sha256(v.tobytes() + d_err.numpy().tobytes()).hexdigest()[:16], "tiled", e.als_tiled(m))
'''
bad = 0
groups = (
    ("configs[1], one pass", "sgd", (("as measured", {}), ("serial", {"FMX_ROWS_SERIAL": "1"}), ("pipelined", {"FMX_ROWS_SERIAL": "0"}), ("as measured again", {}))),
    ("ragged rows, one pass", "ragged", (("static serial", {"FMX_ROWS_SERIAL": "1"}), ("static pipelined", {"FMX_ROWS_SERIAL": "0"}), ("pulled rows", {"FMX_ROWS_PULL": "1"}), ("static as measured", {}))),
    ("ragged rows, flat form", "ragged", (("flat", {"FMX_ROWS_FLAT": "1"}), ("flat pipelined", {"FMX_ROWS_FLAT": "1", "FMX_ROWS_SERIAL": "0"}), ("flat again", {"FMX_ROWS_FLAT": "1"}))),
    ("configs[4], one Gibbs sweep", "mcmc", (("tiled", {}), ("tiled again", {}), ("tiled, eight entries per round", {"FMX_ALS_SUMS_U": "8"}), ("tiled, four rows per thread", {"FMX_ALS_APPLY_ROWS": "4"}))),
)
for title, job, variants in groups:
    out = {}
    for name, env in variants:
        r = subprocess.run([sys.executable, "-c", CHILD, job], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("HASH")]
        out[name] = line[0].split(None, 1)[1] if line else "FAILED " + r.stderr[-300:]
    same = len({v.split()[0] for v in out.values()}) == 1 and not any(v.startswith("FAILED") for v in out.values())
    bad += not same
    print(f"{title}: {'identical' if same else 'DIFFERENT'} {out}", flush=True)
sys.exit(1 if bad else 0)
