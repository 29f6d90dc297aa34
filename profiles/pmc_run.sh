#!/bin/bash
# PMC passes for the two hot kernels (one counter group per pass; never mixed with trace domains other than kernel-trace).
# usage (on the GPU box, from the repo root):  bash profiles/pmc_run.sh <outdir> [bench args...]
set -e
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -- python3 bench.py --cpu-rows 0 "$@" > "$OUT/pass$i.log" 2>&1
done
python3 profiles/pmc_summarize.py "$OUT" "$@"
