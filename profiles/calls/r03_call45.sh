#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_train.py tests/test_gpu_group.py -x -q -m gpu > $O/r3_t45.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/r3_t45.log
nproc
timeout -k 10 600 python3 profiles/host_handover.py > $O/r3_handover.json 2> $O/r3_handover.err; echo rc=$?; cat $O/r3_handover.json; tail -3 $O/r3_handover.err
