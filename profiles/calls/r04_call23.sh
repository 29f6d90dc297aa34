#!/bin/bash
# round 4, twenty-third GPU call: large steps through one-wave workgroups (no workgroup-wide barrier; a wave stages its own 16 rows)
export TMPDIR=/tmp
O=gpurun_out
for cfg in "30 30 iid 16" "1 64 ragged 16" "25 35 ragged 16" "1 64 ragged 64" "1 64 ragged 8"; do
  set -- $cfg
  for w in 0 1; do
    unset FMX_ROWS_WG64; [ $w = 1 ] && export FMX_ROWS_WG64=1
    FMX_ROWS_FLAT=0 timeout -k 10 120 python3 profiles/probes/ragged_probe.py $1 $2 $3 $4 2>&1 | tail -1 | sed "s/^/wg64=$w /"
  done
done | tee $O/r04_ragged_wg64.txt
