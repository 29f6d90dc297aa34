#!/bin/bash
# round 4, fifteenth GPU call: the whole GPU suite, the N = 2 rehearsals of bench.py (process per GPU over gloo, and in-library), then the round's measurement set
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/r04_gpu_suite.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -6 $O/r04_gpu_suite.log
[ $rc -ne 0 ] && exit $rc
FMX_BENCH_SHARED_DEVICE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 --backend gloo --rows 6000000 > $O/r04_bench_n2_gloo.json 2> $O/r04_bench_n2_gloo.err; echo "N=2 rehearsal (gloo, one device) rc=$?"; tail -2 $O/r04_bench_n2_gloo.err | cut -c1-300
FMX_BENCH_SHARED_DEVICE=1 timeout -k 10 300 python3 bench.py --in-library --gpus 2 --steps 6 --warmup 2 --rows 6000000 > $O/r04_bench_n2_inlib.json 2> $O/r04_bench_n2_inlib.err; echo "N=2 in-library rehearsal rc=$?"; tail -2 $O/r04_bench_n2_inlib.err | cut -c1-300
python3 - <<'PY'
import json
for f in ("r04_bench_n2_gloo", "r04_bench_n2_inlib"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, "%.1f M" % (d["value"] / 1e6), d["n_gpus"], d["config"].get("learn_rate"), d["config"].get("batch_rows_per_gpu"), d["config"].get("exchange"), d.get("group"))
    except Exception as ex:
        print(f, "FAILED", ex)
PY
