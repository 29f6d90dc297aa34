#!/bin/bash
# round 4, thirtieth GPU call: phase 2 with one / two S rows outstanding per lane group (the schedule question of phase 1, asked of phase 2)
export TMPDIR=/tmp
O=gpurun_out
for cs in 0 1 2; do
  FMX_COLS_SERIAL=$cs FMX_ROWS_SERIAL=1 timeout -k 10 200 python3 bench.py --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('headline cols_serial=$cs: %.1f M ex/s, phase 1 %.4f ms, phase 2 %.4f ms' % (d['value']/1e6, k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms']))"
  FMX_COLS_SERIAL=$cs timeout -k 10 200 python3 bench.py --workload criteo --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('criteo   cols_serial=$cs: %.1f M ex/s, phase 1 %.4f ms, phase 2 %.4f ms' % (d['value']/1e6, k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms']))"
  FMX_COLS_SERIAL=$cs timeout -k 10 200 python3 bench.py --solver ftrl --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('ftrl k64 cols_serial=$cs: %.1f M ex/s, phase 1 %.4f ms, phase 2 %.4f ms' % (d['value']/1e6, k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms']))"
done | tee $O/r04_cols_serial.txt
