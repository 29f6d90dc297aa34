#!/bin/bash
# round 4, forty-first GPU call: w in the V row's line (FMX_W_IN_ROW=1: no second request stream) under the wide kernel's schedules and the lean form
export TMPDIR=/tmp
O=gpurun_out
for wir in 0 1; do for ser in 1 2 0; do
  FMX_W_IN_ROW=$wir FMX_ROWS_SERIAL=$ser timeout -k 10 200 python3 bench.py --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('headline w_in_row=$wir serial=$ser: %.1f M ex/s, phase 1 %.4f ms, phase 2 %.4f ms' % (d['value']/1e6, k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms']))"
done; done | tee $O/r04_wir_lean.txt
for ser in 1 2; do
  FMX_ROWS_SERIAL=$ser timeout -k 10 200 python3 bench.py --no-linear --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('headline --no-linear serial=$ser: %.1f M ex/s, phase 1 %.4f ms, phase 2 %.4f ms' % (d['value']/1e6, k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms']))"
done | tee -a $O/r04_wir_lean.txt
