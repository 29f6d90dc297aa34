#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof59 -- python3 profiles/scales_probe.py > /dev/null 2>&1
