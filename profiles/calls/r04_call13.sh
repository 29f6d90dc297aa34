#!/bin/bash
# round 4, thirteenth GPU call: the fused pass with a tile's sums handed out one tile after its corrections; tile sizes
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 120 python3 -m pytest tests/test_gpu_configs4.py -x -q -m gpu -k "fused_pass or row_tiled_sweep" > $O/r04_t13.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/r04_t13.log
[ $rc -ne 0 ] && exit $rc
run() {  # fuse rows wgs
  FMX_ALS_FUSE=$1 FMX_ALS_TILE_ROWS=$2 FMX_ALS_PASS_WGS=$3 timeout -k 10 200 python3 bench.py --solver mcmc --no-extras --cpu-rows 0 --steps 3 > $O/r04_mcmc_ab.json 2> $O/r04_mcmc_ab.err; rc=$?
  [ $rc -ne 0 ] && { echo "bench fuse=$1 rows=$2 wgs=$3 rc=$rc"; tail -5 $O/r04_mcmc_ab.err; exit $rc; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/r04_mcmc_ab.json') if l.startswith('{')][-1])
print('fuse=$1 rows=$2 wgs/cu=$3: %.1f M ex/s, %.1f ms/step, level %.4f ms, ss %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['residual_sum_squares'][1]))"
}
run 0 131072 8
run 1 131072 8
run 1 65536 8
run 1 32768 8
run 1 65536 6
run 1 65536 4
run 0 65536 8
