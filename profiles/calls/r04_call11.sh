#!/bin/bash
# round 4, eleventh GPU call: phase 1 with lane groups pulling rows (ragged matrices): bitwise the static kernel, parity suites on ragged CSR, its gain
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_forward.py tests/test_gpu_train.py -x -q -m gpu > $O/r04_t11.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $O/r04_t11.log
[ $rc -ne 0 ] && exit $rc
cat > /tmp/ragged_probe.py <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, B = 10_000_000, 1_000_000, 30, 262_144
lo, hi, kind, k = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
m = engine.Matrix.synthetic_ragged(n, p, float(z), 20240001, min_nnz=lo, max_nnz=hi) if kind == "ragged" else engine.Matrix.synthetic_iid(n, p, z, 20240001)
e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
e.init_normal(1, 0.0, 0.01)
nb = n // B
for i in range(20): e.step(m, i % nb)
e.sync(); e.profile_reset(); e.profile(3)
t = time.perf_counter()
for i in range(40): e.step(m, (20 + i) % nb)
e.sync(); dt = time.perf_counter() - t
a, an = e.profile_get(L.KERNEL_ROWS_FORWARD); b, bn = e.profile_get(L.KERNEL_COLS_UPDATE)
print("%s [%d,%d] k=%d mean %.2f pull=%s serial=%s: %.1f M examples/s, phase 1 %.4f ms, phase 2 %.4f ms, schedule %s" % (kind, lo, hi, k, m.nnz / m.n, os.environ.get("FMX_SORT_ROWS"), os.environ.get("FMX_ROWS_SERIAL"), B * 40 / dt / 1e6, a / an, b / bn, e.rows_tune()))
PY
for cfg in "30 30 iid 16 1 x" "1 64 ragged 16 0 x" "1 64 ragged 16 1 x" "1 64 ragged 16 1 0" "1 64 ragged 16 1 1" "30 30 ragged 16 1 x" "1 64 ragged 64 0 x" "1 64 ragged 64 1 x" "1 64 ragged 8 0 x" "1 64 ragged 8 1 x"; do
  set -- $cfg
  if [ "$6" = "x" ]; then FMX_SORT_ROWS=$5 timeout -k 10 120 python3 /tmp/ragged_probe.py $1 $2 $3 $4 2>&1 | tail -1; else FMX_SORT_ROWS=$5 FMX_ROWS_SERIAL=$6 timeout -k 10 120 python3 /tmp/ragged_probe.py $1 $2 $3 $4 2>&1 | tail -1; fi
done | tee $O/r04_ragged_probe2.txt
