#!/bin/bash
# round 4, twenty-sixth GPU call: phase 1 with every w load from ONE address (timing only): what the second request per nonzero costs under each schedule
export TMPDIR=/tmp
O=gpurun_out
for cfg in "30 30 iid 16" "1 64 ragged 16"; do
  set -- $cfg
  for now in 0 1; do for ser in 1 2 0; do
    unset FMX_DEBUG_NOW; [ $now = 1 ] && export FMX_DEBUG_NOW=1
    FMX_ROWS_SERIAL=$ser FMX_ROWS_FLAT=0 timeout -k 10 120 python3 profiles/probes/ragged_probe.py $1 $2 $3 $4 2>&1 | tail -1 | sed "s/^/no_w=$now /"
  done; done
done | tee $O/r04_no_w_gather.txt
for now in 0 1; do for ser in 1 0; do
  unset FMX_DEBUG_NOW; [ $now = 1 ] && export FMX_DEBUG_NOW=1
  FMX_ROWS_SERIAL=$ser timeout -k 10 200 python3 bench.py --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('headline no_w=$now serial=$ser: %.1f M ex/s, phase 1 %.4f ms, phase 2 %.4f ms' % (d['value']/1e6, k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms']))"
done; done | tee -a $O/r04_no_w_gather.txt
