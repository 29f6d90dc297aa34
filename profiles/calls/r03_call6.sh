#!/bin/bash
# round 3, sixth GPU call: where a streamed step spends its time (kernel trace), PMC at the Criteo shape, stride-shift A/B
export TMPDIR=/tmp
O=gpurun_out
cd /tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_prof_stream -- python3 bench.py --workload criteo --stream --steps 30 > $O/r3_bench_stream_prof.json 2> $O/r3_prof_stream.err; echo "rocprof stream rc=$?"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3_prof_stream/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f'{float(r["TotalDurationNs"]) / 1e6:9.2f} ms {int(r["Calls"]):6d} calls {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Name"][:110]}')
print("total kernel ms", tot / 1e6)
PY
for i in 1 2; do timeout -k 10 200 python3 bench.py --no-extras --cpu-rows 0 > $O/r3_bench_sgd_shift$i.json 2>/dev/null; done
export FMX_ROWS_SERIAL=0
timeout -k 10 600 bash profiles/pmc_run.sh $O/r3_pmc_criteo --workload criteo > $O/r3_pmc_criteo.log 2>&1; echo "pmc criteo rc=$?"
unset FMX_ROWS_SERIAL
python3 - <<'PY'
import json
for f in ("r3_bench_sgd_shift1", "r3_bench_sgd_shift2"):
    d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
    print(f, "%.1fM" % (d["value"] / 1e6), {k: round(v["avg_launch_ms"], 4) for k, v in d["roofline"]["kernels"].items()})
PY
