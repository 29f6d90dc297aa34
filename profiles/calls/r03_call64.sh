#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
for s in 1 0 1 0; do FMX_DETECT_FIELDS=$s timeout -k 10 600 python3 profiles/host_handover.py 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('detect=$s handover %.4f plan %.4f pass %.4f one-epoch %.1f M'%(d['handover_s'],d['plan_s'],d['one_pass_s'],d['one_epoch_examples_per_s_from_host_arrays']/1e6))"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof64 -- python3 profiles/host_handover.py > /dev/null 2>&1
