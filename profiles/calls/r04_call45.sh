#!/bin/bash
# round 4, forty-fifth GPU call: phase 1's V rows loaded non-temporally (-DFMX_NT_GATHER=1 build under fmwr_amd/variants/, not shipped): do the w words keep their L2 lines then?
export TMPDIR=/tmp
O=gpurun_out
for nt in 0 1; do
  unset FMX_LIB_PATH; [ $nt = 1 ] && export FMX_LIB_PATH=$PWD/fmwr_amd/variants/libfmx_nt.so
  for ser in 1 0; do
    for cfg in "30 30 strata 16" "30 30 iid 16" "1 64 ragged 16"; do
      FMX_ROWS_SERIAL=$ser timeout -k 10 120 python3 profiles/probes/ragged_probe.py $cfg 2>&1 | tail -1 | sed "s/^/nt=$nt /"
    done
  done
done | tee $O/r04_rows_nt.txt
