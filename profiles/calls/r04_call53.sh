#!/bin/bash
# round 4, fifty-third GPU call: kernel stats of the resident Criteo-shaped step (what phase 2's 0.36 ms is made of)
export TMPDIR=/tmp
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_criteo -- python3 bench.py --workload criteo --no-extras --cpu-rows 0 --no-other-configs --steps 40 > $O/r04_bench_criteo_under_rocprof.json 2> $O/r04_rocprof_criteo.err; echo "rc=$?"
f=$(find $O/prof_criteo -name "*kernel_stats.csv" | head -1); cp $f $O/r04_kernel_stats_criteo.csv
python3 - $f <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-84s calls %6s avg %9.1f us total %8.1f ms" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
# a timeline of two steps: start / end of every kernel relative to the first phase-1 launch after warm-up
t=$(find $O/prof_criteo -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "fm_rows_forward_k" in r["Kernel_Name"] and ", true" in r["Kernel_Name"]]
i0 = idx[len(idx) // 2]; t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 16]:
    print("%8.1f .. %8.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Kernel_Name"][:90]))
PY
rm -rf $O/prof_criteo
