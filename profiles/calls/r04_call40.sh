#!/bin/bash
# round 4, fortieth GPU call: the stratum-by-stratum gather probe with a 4-byte side word per row (as phase 1 reads w beside the V row): does the second stream cost the L2 hits?
export TMPDIR=/tmp
O=gpurun_out
for side in 0 1; do for st in 0 1; do FMX_PROBE_SIDE=$side FMX_PROBE_STRATA=$st timeout -k 10 200 python3 profiles/probes/gather_sweep.py 2>&1 | grep strata | sed "s/^/side=$side /"; done; done | tee $O/r04_gather_sweep_side.txt
