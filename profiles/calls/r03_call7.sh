#!/bin/bash
# round 3, seventh GPU call: plan builder with its own ordered compaction, nine-bit sort passes, cheaper generator
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/r3_t7.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $O/r3_t7.log
[ $rc -ge 2 ] && exit $rc
for s9 in 1 0; do FMX_SORT9=$s9 timeout -k 10 200 python3 bench.py --workload criteo --stream --steps 30 > $O/r3_bench_stream_sort9_$s9.json 2> $O/r3_bench_stream_sort9_$s9.err; echo "stream sort9=$s9 rc=$?"; done
python3 - <<'PY'
import json
for f in ("r3_bench_stream_sort9_1", "r3_bench_stream_sort9_0"):
    d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
    print(f, "%.1fM" % (d["value"] / 1e6), "%.3f ms" % d["ms_per_step"])
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_prof_stream2 -- python3 bench.py --workload criteo --stream --steps 30 > $O/r3_bench_stream_prof2.json 2> $O/r3_prof_stream2.err; echo "rocprof stream rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3_prof_stream2/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f'{float(r["TotalDurationNs"]) / 1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Name"][:100]}')
PY
