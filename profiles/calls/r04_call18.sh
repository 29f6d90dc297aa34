#!/bin/bash
# round 4, eighteenth GPU call: the lists' first rows inline in the sums pass (two dependent rounds instead of three), u16 level indices in the correction pass
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs4.py -x -q -m gpu > $O/r04_t18.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/r04_t18.log
[ $rc -ne 0 ] && exit $rc
run() {  # slots
  FMX_ALS_SLOTS=$1 timeout -k 10 200 python3 bench.py --solver $2 --no-extras --cpu-rows 0 --steps 3 > $O/r04_mcmc_ab.json 2> $O/r04_mcmc_ab.err; rc=$?
  [ $rc -ne 0 ] && { echo "bench slots=$1 rc=$rc"; tail -5 $O/r04_mcmc_ab.err; exit $rc; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/r04_mcmc_ab.json') if l.startswith('{')][-1])
print('$2 slots=$1: %.1f M ex/s, %.1f ms/step, level %.4f ms, plan %.3f s, ss %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['plan_build_s'], d['config']['residual_sum_squares'][1]))"
}
run 0 mcmc
run 1 mcmc
run 1 als
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcmc_slots -- python3 bench.py --solver mcmc --cpu-rows 0 --no-extras --steps 3 > $O/r04_bench_mcmc_under_rocprof.json 2> $O/r04_rocprof_mcmc.err; echo "rocprof rc=$?"
f=$(find $O/prof_mcmc_slots -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    print("%-70s calls %6s avg %10.1f us  total %8.1f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
