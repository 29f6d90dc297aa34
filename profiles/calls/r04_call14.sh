#!/bin/bash
# round 4, fourteenth GPU call: more gathers in flight per thread in the tiled sweep's passes (8 entries per round, 8 rows per thread) and tile sizes
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 120 python3 -m pytest tests/test_gpu_configs4.py -x -q -m gpu -k "row_tiled" > $O/r04_t14.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/r04_t14.log
[ $rc -ne 0 ] && exit $rc
run() {  # rows lg sums_u apply_rows
  FMX_ALS_TILE_ROWS=$1 FMX_ALS_TILE_LG=$2 FMX_ALS_SUMS_U=$3 FMX_ALS_APPLY_ROWS=$4 timeout -k 10 200 python3 bench.py --solver mcmc --no-extras --cpu-rows 0 --steps 3 > $O/r04_mcmc_ab.json 2> $O/r04_mcmc_ab.err; rc=$?
  [ $rc -ne 0 ] && { echo "bench rows=$1 lg=$2 u=$3 r=$4 rc=$rc"; tail -5 $O/r04_mcmc_ab.err; exit $rc; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/r04_mcmc_ab.json') if l.startswith('{')][-1])
print('rows=$1 lg=$2 sums_u=$3 apply_rows=$4: %.1f M ex/s, %.1f ms/step, level %.4f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms']))"
}
run 131072 1 4 4
run 131072 1 8 4
run 131072 1 4 8
run 262144 1 8 4
run 262144 1 8 8
run 262144 2 8 8
run 524288 1 8 8
run 524288 2 8 8
