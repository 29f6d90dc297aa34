#!/bin/bash
# round 4, twenty-seventh GPU call: phase 1 without the LDS stage and without workgroup-wide steps (experiment), 4 / 8 entries per round, both schedules
export TMPDIR=/tmp
O=gpurun_out
for cfg in "30 30 iid 16" "1 64 ragged 16" "25 35 ragged 16"; do
  set -- $cfg
  for d in 4 8; do for ser in 1 0; do
    FMX_ROWS_DIRECT=$d FMX_ROWS_SERIAL=$ser FMX_ROWS_FLAT=0 timeout -k 10 120 python3 profiles/probes/ragged_probe.py $1 $2 $3 $4 2>&1 | tail -1 | sed "s/^/direct=$d /"
  done; done
done | tee $O/r04_rows_direct.txt
for d in 4 8; do for ser in 1 0; do
  FMX_ROWS_DIRECT=$d FMX_ROWS_SERIAL=$ser timeout -k 10 200 python3 bench.py --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('headline direct=$d serial=$ser: %.1f M ex/s, phase 1 %.4f ms, phase 2 %.4f ms' % (d['value']/1e6, k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms']))"
done; done | tee -a $O/r04_rows_direct.txt
