#!/bin/bash
# per-kernel times of the level-order V sweep under env variants: bash profiles/order_ab.sh "<VAR=.. VAR=..>" ...   (one rocprofv3 kernel-stats run each)
export TMPDIR=/tmp
mkdir -p gpurun_out/order_ab
i=0
for v in "$@"; do
  i=$((i+1))
  d=gpurun_out/order_ab/run$i
  rm -rf $d
  env $v rocprofv3 --kernel-trace --stats -d $d -o s --output-format csv -- python3 bench.py --solver mcmc --cpu-rows 0 --no-extras --steps 2 --warmup 1 > $d.log 2>&1
  echo "== [$v]"
  python3 - "$d/s_kernel_stats.csv" "$d.log" <<'PY'
import csv, json, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("als_order", "als_tile", "als_rows_apply", "fm_rows_forward")):
        print("   %-60s calls %5s avg %8.1f us" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
for ln in open(sys.argv[2]):
    if ln.startswith("{"):
        d = json.loads(ln); print("   value %.1f M/s, ms_per_step %.2f (under rocprof)" % (d["value"] / 1e6, d["ms_per_step"]))
PY
done
