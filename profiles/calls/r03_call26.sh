#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_group.py tests/test_gpu_tracker.py -x -q -m gpu > $O/r3_t26.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -25 $O/r3_t26.log
