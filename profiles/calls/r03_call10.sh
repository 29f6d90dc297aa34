#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_configs3.py -x -q -m gpu > $O/r3_t10.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r3_t10.log
[ $rc -ge 2 ] && exit $rc
for c in 0 1 2; do
  FMX_SORT_CFG=$c timeout -k 10 200 python3 bench.py --workload criteo --stream --steps 40 > $O/r3_bench_stream_cfg$c.json 2>/dev/null; echo "cfg $c rc=$?"
done
FMX_STREAM_OVERLAP=1 timeout -k 10 200 python3 bench.py --workload criteo --stream --steps 40 > $O/r3_bench_stream_ovl.json 2>/dev/null; echo "overlap rc=$?"
python3 - <<'PY'
import json
for f in ("r3_bench_stream_cfg0", "r3_bench_stream_cfg1", "r3_bench_stream_cfg2", "r3_bench_stream_ovl"):
    d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
    print(f, "%.1fM" % (d["value"] / 1e6), "%.3f ms" % d["ms_per_step"])
PY
