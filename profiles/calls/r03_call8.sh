#!/bin/bash
# round 3, eighth GPU call: LDS-staged head compaction, the dense-prefix pass
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_configs3.py tests/test_gpu_train.py tests/test_gpu_distributed.py tests/test_gpu_group.py tests/test_gpu_props.py -x -q -m gpu > $O/r3_t8.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $O/r3_t8.log
[ $rc -ge 2 ] && exit $rc
for pp in 1 0; do
FMX_PREFIX_PASS=$pp timeout -k 10 200 python3 bench.py --workload criteo --stream --steps 30 > $O/r3_bench_stream_pre$pp.json 2> $O/r3_bench_stream_pre$pp.err; echo "stream prefix=$pp rc=$?"
FMX_PREFIX_PASS=$pp timeout -k 10 200 python3 bench.py --workload criteo > $O/r3_bench_criteo_pre$pp.json 2> $O/r3_bench_criteo_pre$pp.err; echo "criteo prefix=$pp rc=$?"
done
python3 - <<'PY'
import json
for f in ("r3_bench_stream_pre1", "r3_bench_stream_pre0", "r3_bench_criteo_pre1", "r3_bench_criteo_pre0"):
    d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
    print(f, "%.1fM" % (d["value"] / 1e6), "%.3f ms" % d["ms_per_step"], {k: round(v["avg_launch_ms"], 4) for k, v in d["roofline"]["kernels"].items()})
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_prof_stream3 -- python3 bench.py --workload criteo --stream --steps 30 > $O/r3_bench_stream_prof3.json 2> $O/r3_prof_stream3.err; echo "rocprof stream rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3_prof_stream3/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f'{float(r["TotalDurationNs"]) / 1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Name"][:100]}')
PY
