#!/bin/bash
# round 4, twentieth GPU call: the flat form of phase 1 (rows of differing lengths): its tests, then flat against static per row-length law
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_flat_rows.py tests/test_gpu_api.py -x -q -m gpu > $O/r04_t20.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $O/r04_t20.log
[ $rc -ne 0 ] && exit $rc
for cfg in "30 30 iid 16 x x" "1 64 ragged 16 0 x" "1 64 ragged 16 1 x" "1 64 ragged 16 1 0" "1 64 ragged 16 1 1" "30 30 ragged 16 1 x" "25 35 ragged 16 0 x" "25 35 ragged 16 1 x" "1 64 ragged 64 0 x" "1 64 ragged 64 1 x" "1 64 ragged 8 0 x" "1 64 ragged 8 1 x" "1 64 ragged 32 0 x" "1 64 ragged 32 1 x"; do
  set -- $cfg
  export FMX_ROWS_FLAT=$5; [ "$5" = "x" ] && unset FMX_ROWS_FLAT
  export FMX_ROWS_SERIAL=$6; [ "$6" = "x" ] && unset FMX_ROWS_SERIAL
  timeout -k 10 120 python3 profiles/probes/ragged_probe.py $1 $2 $3 $4 2>&1 | tail -1
done | tee $O/r04_ragged_flat.txt
