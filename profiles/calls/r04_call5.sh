#!/bin/bash
# round 4, fifth GPU call: factor-major q table + pick folded into the last correction pass, the next level's coordinates gathered by the step
# kernel, non-temporal loads per stream (bit mask): parity, then A/B at configs[4]
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs4.py tests/test_gpu_train.py -x -q -m gpu -k "not full_size" > $O/r04_t5.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/r04_t5.log
[ $rc -ne 0 ] && exit $rc
run() {  # nt rows lg apply_rows fold
  FMX_ALS_NT=$1 FMX_ALS_TILE_ROWS=$2 FMX_ALS_TILE_LG=$3 FMX_ALS_APPLY_ROWS=$4 FMX_ALS_FOLD_PREP=$5 timeout -k 10 200 python3 bench.py --solver mcmc --no-extras --cpu-rows 0 --steps 3 > $O/r04_mcmc_ab.json 2> $O/r04_mcmc_ab.err; rc=$?
  [ $rc -ne 0 ] && { echo "bench nt=$1 rows=$2 lg=$3 apply_rows=$4 rc=$rc"; tail -5 $O/r04_mcmc_ab.err; exit $rc; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/r04_mcmc_ab.json') if l.startswith('{')][-1])
print('nt=$1 rows=$2 lg=$3 apply_rows=$4 fold=$5: %.1f M ex/s, %.1f ms/step, level %.4f ms, ss %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['residual_sum_squares'][1]))"
}
run 0 131072 1 4 1
run 0 131072 1 4 0
run 2 131072 1 4 1
run 4 131072 1 4 1
run 6 131072 1 4 1
run 6 131072 1 2 1
run 6 131072 2 4 1
run 6 262144 2 4 1
run 6 262144 4 4 1
run 0 262144 2 4 1
run 2 262144 2 4 1
FMX_ALS_TILED=0 timeout -k 10 200 python3 bench.py --solver mcmc --no-extras --cpu-rows 0 --steps 3 > $O/r04_mcmc_untiled.json 2>/dev/null
python3 -c "
import json
d=json.loads([l for l in open('$O/r04_mcmc_untiled.json') if l.startswith('{')][-1])
print('untiled: %.1f M ex/s, %.1f ms/step, level %.4f ms, ss %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['residual_sum_squares'][1]))"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcmc_tiled3 -- python3 bench.py --solver mcmc --cpu-rows 0 --no-extras --steps 3 > $O/r04_bench_mcmc_under_rocprof.json 2> $O/r04_rocprof_mcmc.err; echo "rocprof rc=$?"
f=$(find $O/prof_mcmc_tiled3 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-70s calls %6s avg %10.1f us  total %8.1f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
