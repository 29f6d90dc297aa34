#!/bin/bash
# round 4, thirty-eighth GPU call: the lean form of phase 1 (FMX_ROWS_SERIAL=2: one entry per step, 59 registers, eight waves per SIMD) against the wide kernel's serial schedule
export TMPDIR=/tmp
O=gpurun_out
for ser in 1 2; do
  FMX_ROWS_SERIAL=$ser timeout -k 10 200 python3 bench.py --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('headline serial=$ser: %.1f M ex/s, phase 1 %.4f ms, phase 2 %.4f ms' % (d['value']/1e6, k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms']))"
  for cfg in "30 30 iid 16" "1 64 ragged 16" "1 64 ragged 8" "1 64 ragged 32" "1 64 ragged 64"; do
    FMX_ROWS_SERIAL=$ser timeout -k 10 120 python3 profiles/probes/ragged_probe.py $cfg 2>&1 | tail -1
  done
  FMX_ROWS_SERIAL=$ser timeout -k 10 200 python3 bench.py --workload criteo --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('criteo serial=$ser: %.1f M ex/s, phase 1 %.4f ms, phase 2 %.4f ms' % (d['value']/1e6, k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms']))"
done | tee $O/r04_rows_lean.txt
