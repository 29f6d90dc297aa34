#!/bin/bash
# round 4, third GPU call: kernel trace of the row-tiled sweep (which of its passes takes the time)
export TMPDIR=/tmp
O=gpurun_out
cd /tmp 2>/dev/null; cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcmc_tiled -- python3 bench.py --solver mcmc --cpu-rows 0 --no-extras --steps 3 > $O/r04_bench_mcmc_under_rocprof.json 2> $O/r04_rocprof_mcmc.err; echo "rocprof rc=$?"
f=$(find $O/prof_mcmc_tiled -name "*kernel_stats.csv" | head -1); echo $f
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s avg %10.1f us  total %8.1f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
