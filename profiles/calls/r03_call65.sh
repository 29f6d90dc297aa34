#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_group.py tests/test_gpu_wide_rows.py tests/test_gpu_c_caller.py -x -q -m gpu > $O/r3_t65.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r3_t65.log
