#!/bin/bash
# round 4, forty-third GPU call: the dgCMatrix hand-over (device transposition) test; its time on a 10 M x 1 M matrix
export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_gpu_api.py -x -q -m gpu -k "dgcmatrix or hand_over" 2>&1 | tail -5
timeout -k 10 600 python3 - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, ".")
from fmwr_amd import engine
n, p, z = 10_000_000, 1_000_000, 30
m = engine.Matrix.synthetic(n, p, z, 1)
rp, col, val, y = m.export()
m.close()
import scipy.sparse as sp
t = time.perf_counter(); csr = sp.csr_matrix((val.astype(np.float64), col.astype(np.int32), rp), shape=(n, p)); csc = csr.tocsc(); t_host = time.perf_counter() - t
print("host (scipy) CSR -> CSC of 10 M x 1 M, 3e8 entries: %.1f s  [what Matrix::t costs the reference's R side, order of magnitude]" % t_host)
t = time.perf_counter(); a = engine.Matrix.from_dgc(csc.data, csc.indices, csc.indptr, n, p, labels=y.astype(np.float64)); t_dev = time.perf_counter() - t
print("fmx_matrix_from_dgc of the same slots (3.7 GB over PCIe + device transposition + checks): %.2f s" % t_dev)
e = a.export(0, 1000)
assert np.array_equal(e[1], col[: rp[1000]]) and np.array_equal(e[0], rp[:1001])
PY
