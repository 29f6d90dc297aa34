#!/bin/bash
# round 4, forty-ninth GPU call: how much of the plan build's wall time is kernels
export TMPDIR=/tmp
O=gpurun_out
python3 profiles/probes/plan_split.py
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_plan -- python3 profiles/probes/plan_split.py 2>/dev/null | tail -1
python3 - $(find $O/prof_plan -name "*kernel_stats.csv" | head -1) <<'PY' | tee $O/r04_plan_split.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    n = r["Name"]
    if any(s in n for s in ("synth", "init_normal", "fillBuffer")): continue
    tot += float(r["TotalDurationNs"])
    print("%-60s calls %5s total %8.3f ms" % (n[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
print("kernels of the plan build: %.2f ms" % (tot / 1e6))
PY
rm -rf $O/prof_plan
