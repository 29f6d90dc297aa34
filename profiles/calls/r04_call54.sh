#!/bin/bash
# round 4, fifty-fourth GPU call: the Criteo-shaped step under the knobs that move work between the list-by-list kernel and the long-list kernels
export TMPDIR=/tmp
O=gpurun_out
run() {
  env "$@" timeout -k 10 200 python3 bench.py --workload criteo --no-extras --cpu-rows 0 --no-other-configs --steps 40 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('$*: %.1f M ex/s, step %.4f ms, phase 1 %.4f ms, phase 2 %.4f ms' % (d['value']/1e6, d['ms_per_step'], k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms']))"
}
{ run FMX_X=0; run FMX_LONG_SIDE=0; run FMX_LONG_MIN=16; run FMX_LONG_MIN=32; run FMX_LONG_MIN=128; run FMX_LONG_MIN=256; run FMX_LONG_MIN=1024; run FMX_DIRECT_LISTS=0; } | tee $O/r04_criteo_knobs.txt
