#!/bin/bash
# round 4, fiftieth GPU call: the fp64-state step's kernels (a V row is a whole 128-byte line there: how close is phase 1 to the fp32 kernel's line rate?)
export TMPDIR=/tmp
for ser in x 1 0; do
  export FMX_ROWS_SERIAL=$ser; [ $ser = x ] && unset FMX_ROWS_SERIAL
  timeout -k 10 200 python3 bench.py --state-fp64 --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('fp64 state serial=$ser: %.1f M ex/s, phase 1 %.4f ms, phase 2 %.4f ms, schedule %s' % (d['value']/1e6, k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms'], d['config']['rows_forward_schedule']))"
done | tee gpurun_out/r04_fp64_state.txt
