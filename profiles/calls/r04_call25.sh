#!/bin/bash
# round 4, twenty-fifth GPU call: phase 1 with TWO entries' requests outstanding per lane group (between the serial and the pipelined schedule)
export TMPDIR=/tmp
O=gpurun_out
for cfg in "30 30 iid 16" "1 64 ragged 16" "25 35 ragged 16" "1 64 ragged 8" "1 64 ragged 32"; do
  set -- $cfg
  for ser in 1 2 0; do
    FMX_ROWS_SERIAL=$ser FMX_ROWS_FLAT=0 timeout -k 10 120 python3 profiles/probes/ragged_probe.py $1 $2 $3 $4 2>&1 | tail -1
  done
done | tee $O/r04_ragged_depth2.txt
for ser in 1 2; do FMX_ROWS_SERIAL=$ser timeout -k 10 200 python3 bench.py --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | cut -c1-200; done
