"""Phase 1's two request schedules, phase 2's forms and the tile size must not change a bit: the same 120 steps at the bench size
in separate processes with the schedule pinned either way / left to the engine's timing, hashes compared.  (FMX_EMBED_MULT=0 is the one
switch that DOES change bits, by design: the embedded form keeps 22 of the S row's 24 mantissa bits -- fm_batch_kernels.hip, EmbedMode.)
python profiles/soak_schedules.py"""
import hashlib, os, subprocess, sys
CHILD = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
P, Z, K, N = 1_000_000, 30, 16, 4_000_000
solver = L.SOLVER_FTRL if sys.argv[1] == "ftrl" else L.SOLVER_SGD
m = engine.Matrix.synthetic(N, P, Z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (K, P)).astype(np.float32).astype(np.float64)
e = engine.Engine(P, num_factor=K, solver=solver, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, l1_v=1e-4 if sys.argv[1] == "ftrl" else 0.0,
                  mode=L.MODE_MINIBATCH, batch_rows=int(sys.argv[2]))
e.set_params(0.0, None, v0)
nb = e.num_batches(m)
for s in range(120):
    e.step(m, s % nb)
e.sync()
w0, w, v = e.get_params()
print("HASH", hashlib.sha256(v.tobytes() + w.tobytes() + np.float64(w0).tobytes()).hexdigest()[:16], e.rows_tune()[0])
'''
bad = 0
for solver, batch in (("sgd", 262144), ("sgd", 1048576), ("ftrl", 524288), ("sgd", 8192)):
    out = {}
    for name, env in (("engine's choice", {}), ("serial", {"FMX_ROWS_SERIAL": "1"}), ("pipelined", {"FMX_ROWS_SERIAL": "0"}),
                      ("staged phase 2", {"FMX_DIRECT_DENSE": "0", "FMX_DIRECT_LISTS": "0"})):
        r = subprocess.run([sys.executable, "-c", CHILD, solver, str(batch)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("HASH")]
        out[name] = line[0].split()[1] if line else "FAILED " + r.stderr[-300:]
    same = len(set(out.values())) == 1
    bad += not same
    print(f"{solver} batch {batch}: {'identical' if same else 'DIFFERENT'} {out}", flush=True)
sys.exit(1 if bad else 0)
