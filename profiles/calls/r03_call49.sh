#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof49 -- python3 profiles/plan_probe.py > $O/r3_plan49.txt 2>&1; cat $O/r3_plan49.txt | tail -2
