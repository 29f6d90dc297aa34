#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
for s in 1 0; do echo "FMX_FIELD_SORT=$s"; FMX_FIELD_SORT=$s timeout -k 10 300 python3 profiles/stream_steady.py 2>&1 | tail -2; done
FMX_STREAM_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof53 -- python3 profiles/stream_steady.py > /dev/null 2>&1
