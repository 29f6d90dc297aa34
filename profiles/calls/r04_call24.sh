#!/bin/bash
# round 4, twenty-fourth GPU call: one-wave workgroups on large steps with the serial request schedule
export TMPDIR=/tmp
O=gpurun_out
for cfg in "30 30 iid 16" "1 64 ragged 16" "25 35 ragged 16" "1 64 ragged 8"; do
  set -- $cfg
  for ser in 1 0; do
    FMX_ROWS_WG64=1 FMX_ROWS_SERIAL=$ser FMX_ROWS_FLAT=0 timeout -k 10 120 python3 profiles/probes/ragged_probe.py $1 $2 $3 $4 2>&1 | tail -1 | sed "s/^/wg64=1 /"
  done
done | tee $O/r04_ragged_wg64_serial.txt
FMX_ROWS_WG64=1 FMX_ROWS_SERIAL=1 timeout -k 10 200 python3 bench.py --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | cut -c1-400
