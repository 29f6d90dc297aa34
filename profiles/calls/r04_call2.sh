#!/bin/bash
# round 4, second GPU call: the probe again (its output index is fixed), parity of the row-tiled sweep, first A/B at configs[4]
export TMPDIR=/tmp
O=gpurun_out
echo "(probe: see r04_call2 first run)"; rc=0


[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs4.py -x -q -m gpu -k "not full_size" > $O/r04_t2.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $O/r04_t2.log
[ $rc -ne 0 ] && exit $rc
for cfg in "0 131072 1" "1 131072 1" "1 131072 2" "1 131072 4" "1 262144 1" "1 262144 2" "1 65536 1"; do
  set -- $cfg
  FMX_ALS_TILED=$1 FMX_ALS_TILE_ROWS=$2 FMX_ALS_TILE_LG=$3 timeout -k 10 200 python3 bench.py --solver mcmc --no-extras --cpu-rows 0 --steps 3 > $O/r04_mcmc_$1_$2_$3.json 2> $O/r04_mcmc_$1_$2_$3.err; rc=$?
  echo "bench tiled=$1 rows=$2 lg=$3 rc=$rc"
  [ $rc -ne 0 ] && { tail -5 $O/r04_mcmc_$1_$2_$3.err; exit $rc; }
  python3 -c "
import json,sys
d=json.loads([l for l in open('$O/r04_mcmc_$1_$2_$3.json') if l.startswith('{')][-1])
print('  value %.1f M ex/s, %.1f ms/step, level %.4f ms, ss %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['residual_sum_squares']))"
done
