#!/bin/bash
# round 4, nineteenth GPU call: the sums pass walking the tiles backwards (Infinity Cache reuse between the passes)
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_configs4.py -x -q -m gpu -k "row_tiled" > $O/r04_t19.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/r04_t19.log
[ $rc -ne 0 ] && exit $rc
run() {  # backward nt
  FMX_ALS_BACKWARD=$1 FMX_ALS_NT=$2 timeout -k 10 200 python3 bench.py --solver mcmc --no-extras --cpu-rows 0 --steps 3 > $O/r04_mcmc_ab.json 2> $O/r04_mcmc_ab.err; rc=$?
  [ $rc -ne 0 ] && { echo "bench rc=$rc"; tail -5 $O/r04_mcmc_ab.err; exit $rc; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/r04_mcmc_ab.json') if l.startswith('{')][-1])
print('backward=$1 nt=$2: %.1f M ex/s, %.1f ms/step, level %.4f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms']))"
}
run 0 6
run 1 6
run 1 7
run 1 2
run 1 0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcmc_bw -- python3 bench.py --solver mcmc --cpu-rows 0 --no-extras --steps 3 > $O/r04_bench_mcmc_under_rocprof.json 2> $O/r04_rocprof_mcmc.err; echo "rocprof rc=$?"
f=$(find $O/prof_mcmc_bw -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:5]:
    print("%-70s calls %6s avg %10.1f us  total %8.1f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
