#!/bin/bash
# round 4, twenty-eighth GPU call: phase 1 compiled for 4 / 5 (default) / 6 / 8 waves per SIMD (amdgpu_waves_per_eu; builds under fmwr_amd/variants/, not shipped)
export TMPDIR=/tmp
O=gpurun_out
for w in 4 5 6 8; do
  unset FMX_LIB_PATH; [ $w != 5 ] && export FMX_LIB_PATH=$PWD/fmwr_amd/variants/libfmx_w$w.so
  for ser in 1 0; do
  FMX_ROWS_SERIAL=$ser timeout -k 10 200 python3 bench.py --no-extras --cpu-rows 0 --no-other-configs 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('headline waves=$w serial=$ser: %.1f M ex/s, phase 1 %.4f ms, phase 2 %.4f ms' % (d['value']/1e6, k['fm_rows_forward']['avg_launch_ms'], k['fm_cols_update']['avg_launch_ms']))"
  done
  FMX_ROWS_SERIAL=1 FMX_ROWS_FLAT=0 timeout -k 10 120 python3 profiles/probes/ragged_probe.py 1 64 ragged 16 2>&1 | tail -1 | sed "s/^/waves=$w /"
  FMX_ROWS_SERIAL=1 timeout -k 10 120 python3 profiles/probes/ragged_probe.py 30 30 iid 16 2>&1 | tail -1 | sed "s/^/waves=$w /"
done | tee $O/r04_rows_waves.txt
