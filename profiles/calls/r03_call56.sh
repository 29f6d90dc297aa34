#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py -x -q -m gpu > $O/r3_t56.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r3_t56.log
