#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs3.py tests/test_gpu_group.py -x -q -m gpu > $O/r3_t39.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/r3_t39.log
python3 bench.py --workload criteo --stream --steps 40 > $O/r3_stream39.json 2>/dev/null; cut -c100-230 $O/r3_stream39.json
FMX_STREAM_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof39 -- python3 bench.py --workload criteo --stream --steps 40 > /dev/null 2>&1
