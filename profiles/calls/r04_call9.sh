#!/bin/bash
# round 4, ninth GPU call: a resident matrix of 4.5e9 entries (nnz > 2^32), rows dealt by length in phase 1 (bitwise + its gain on ragged rows),
# the default bench line with scaling_reference, timed
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_forward.py -x -q -m gpu > $O/r04_t9a.log 2>&1; rc=$?; echo "api/forward tests rc=$rc"; tail -5 $O/r04_t9a.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python3 -m pytest tests/test_gpu_nnz_2p32.py -x -q -m gpu > $O/r04_t9b.log 2>&1; rc=$?; echo "nnz > 2^32 tests rc=$rc"; tail -15 $O/r04_t9b.log
[ $rc -ne 0 ] && exit $rc
for f in 0 1; do
  FMX_SORT_ROWS=$f timeout -k 10 600 python3 - > $O/r04_ragged_sort$f.txt 2>&1 <<'PY'
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, k, B = 10_000_000, 1_000_000, 30, 16, 262_144
m = engine.Matrix.synthetic_ragged(n, p, float(z), 20240001)
e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
e.init_normal(1, 0.0, 0.01)
nb = n // B
for i in range(20): e.step(m, i % nb)
e.sync(); e.profile_reset(); e.profile(3)
t = time.perf_counter()
for i in range(40): e.step(m, (20 + i) % nb)
e.sync(); dt = time.perf_counter() - t
a, an = e.profile_get(L.KERNEL_ROWS_FORWARD); b, bn = e.profile_get(L.KERNEL_COLS_UPDATE)
print("ragged rows: %.1f M examples/s, phase 1 %.4f ms, phase 2 %.4f ms, schedule %s" % (B * 40 / dt / 1e6, a / an, b / bn, e.rows_tune()))
PY
  echo "FMX_SORT_ROWS=$f: $(cat $O/r04_ragged_sort$f.txt | tail -1)"
done
python3 - <<'PY'
import json, subprocess, time
t = time.time()
r = subprocess.run("timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err", shell=True)
print("default bench rc", r.returncode, "wall %.1f s" % (time.time() - t))
d = json.loads([l for l in open("gpurun_out/r04_bench_default.json") if l.startswith("{")][-1])
print("value %.1f M, ms/step %.4f, frac %.3f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"]))
print("scaling_reference", d.get("scaling_reference"))
r = d.get("value_ragged_rows", {})
print("ragged", r.get("value"), r.get("entry_rate_vs_fixed_length_rows"), r.get("kernel_ms"), r.get("fixed_length_kernel_ms"))
for k, v in d.get("other_configs", {}).items():
    print(k, v.get("error") or ("%.1f M, wall %.1f s" % (v["value"] / 1e6, v["wall_s"])))
PY
