#!/bin/bash
# round 3, fifth GPU call: field-structured plan (split), new bench lines, graph diagnostics
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs3.py tests/test_gpu_distributed.py::test_streamed_training_with_n_gpus_behind_one_handle tests/test_gpu_wide_rows.py -x -q -m gpu > $O/r3_t5.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $O/r3_t5.log
[ $rc -ge 2 ] && exit $rc
FMX_ALS_GRAPH=1 FMX_ALS_GRAPH_VERBOSE=1 timeout -k 10 200 python3 - > $O/r3_als_graph_probe.txt 2>&1 <<'PY'
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, k = 4_000_000, 1_000_000, 30, 16
m = engine.Matrix.synthetic_iid(n, p, z, 3, L.COLUMNS_UNIFORM)
e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL)
e.set_params(0.0, None, np.random.default_rng(1).normal(0, 0.01, (k, p)))
print(e.als_plan(m)[:3])
err = np.random.default_rng(2).normal(0, 1, n)
for i in range(3):
    t = time.perf_counter(); err = e.als_vsweep(m, err); print("sweep", i, time.perf_counter() - t, flush=True)
PY
echo "graph probe rc=$?"; cat $O/r3_als_graph_probe.txt
for f in 0 1; do FMX_FIELDS_SPLIT=$f timeout -k 10 200 python3 bench.py --workload criteo --stream --steps 20 > $O/r3_bench_stream_split$f.json 2> $O/r3_bench_stream_split$f.err; echo "stream split=$f rc=$?"; done
timeout -k 10 300 python3 bench.py > $O/r3_bench_sgd.json 2> $O/r3_bench_sgd.err; echo "bench sgd rc=$?"
timeout -k 10 200 python3 bench.py --features 16000000 --no-extras --cpu-rows 0 > $O/r3_bench_p16m.json 2> $O/r3_bench_p16m.err; echo "bench p16m rc=$?"
timeout -k 10 200 python3 bench.py --workload criteo > $O/r3_bench_criteo.json 2> $O/r3_bench_criteo.err; echo "bench criteo rc=$?"
python3 - <<'PY'
import json
for f in ("r3_bench_stream_split0", "r3_bench_stream_split1", "r3_bench_sgd", "r3_bench_p16m", "r3_bench_criteo"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, "%.1fM" % (d["value"] / 1e6), "%.3f ms" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], {k: (round(v["avg_launch_ms"], 4), v.get("ceiling_frac")) for k, v in d["roofline"].get("kernels", {}).items()},
              d.get("end_to_end"), d.get("value_iid_uniform"))
    except Exception as ex:
        print(f, "FAILED", ex)
PY
