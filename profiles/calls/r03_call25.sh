#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_train.py -x -q -m gpu -k "schedules" > $O/r3_t25.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/r3_t25.log
[ $rc -ne 0 ] && exit $rc
bash profiles/r03_call24.sh tune3
