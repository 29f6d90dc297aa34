#!/bin/bash
# round 4, twenty-first GPU call: where the flat form's time goes on rows that do not line up with the lane groups' ranges (timing-only debug knobs)
export TMPDIR=/tmp
O=gpurun_out
for dbg in 0 1 2 4 6; do
  for ser in 1 0; do
  FMX_FLAT_DEBUG=$dbg FMX_ROWS_FLAT=1 FMX_ROWS_SERIAL=$ser timeout -k 10 120 python3 profiles/probes/ragged_probe.py 1 64 ragged 16 2>&1 | tail -1 | sed "s/^/debug=$dbg /"
  done
done | tee $O/r04_ragged_flat_debug.txt
