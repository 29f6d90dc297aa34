#!/bin/bash
# round 4, forty-eighth GPU call: the tiles' plans built on 1 / 2 / 4 / 8 streams (FMX_PLAN_STREAMS); the tests that compare plans; the end-to-end figures
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_train.py tests/test_gpu_props.py -x -q -m gpu > $O/r04_t48.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r04_t48.log
[ $rc -ne 0 ] && exit $rc
for k in 1 2 4 8; do
FMX_PLAN_STREAMS=$k timeout -k 10 300 python3 - <<'PY'
import os, sys, time, hashlib
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine
n, p, z, B = 10_000_000, 1_000_000, 30, 262_144
out = []
for name, mk in (("strata", lambda: engine.Matrix.synthetic(n, p, z, 20240001)), ("iid", lambda: engine.Matrix.synthetic_iid(n, p, z, 20240001)),
                 ("ragged", lambda: engine.Matrix.synthetic_ragged(n, p, float(z), 20240001)), ("criteo 8M", lambda: engine.Matrix.synthetic_fields(8_000_000, 13, engine.CRITEO_VOCAB, 3.0, 20240001))):
    m = mk()
    pp = m.p
    e = engine.Engine(pp, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=16 if name != "criteo 8M" else 32, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B)
    e.init_normal(1, 0.0, 0.01); e.sync()
    t = time.perf_counter(); e.num_batches(m); e.sync(); dt = time.perf_counter() - t
    e.train(m, 4 * B); w, v = e.get_rows(np.arange(0, pp, max(1, pp // 4000), dtype=np.uint32))
    out.append("%s %.2f ms (%s)" % (name, dt * 1e3, hashlib.sha256(v.tobytes()).hexdigest()[:8]))
    e.close(); m.close()
print("plan streams %s: " % os.environ["FMX_PLAN_STREAMS"] + "; ".join(out))
PY
done | tee $O/r04_plan_streams.txt
timeout -k 10 200 python3 bench.py --cpu-rows 0 --no-other-configs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('default streams:', {k:(round(v,5) if isinstance(v,float) else v) for k,v in d['end_to_end'].items() if k!='note'})" | tee -a $O/r04_plan_streams.txt
