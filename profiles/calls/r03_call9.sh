#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_configs3.py tests/test_gpu_group.py tests/test_gpu_props.py tests/test_gpu_train.py -x -q -m gpu > $O/r3_t9.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $O/r3_t9.log; ls $O/fail_* 2>/dev/null && tail -40 $O/fail_*
[ $rc -ge 2 ] && exit $rc
timeout -k 10 200 python3 bench.py --workload criteo --stream --steps 30 > $O/r3_bench_stream_v9.json 2> $O/r3_bench_stream_v9.err; echo "stream rc=$?"
python3 - <<'PY'
import json
for f in ("r3_bench_stream_v9",):
    d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
    print(f, "%.1fM" % (d["value"] / 1e6), "%.3f ms" % d["ms_per_step"], {k: round(v["avg_launch_ms"], 4) for k, v in d["roofline"]["kernels"].items()})
PY
