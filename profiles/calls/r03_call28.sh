#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 800 python3 -m pytest tests/test_gpu_group.py -x -q -m gpu -l > $O/r3_t28.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/r3_t28.log; grep -n "^form \|^solver \|^n_gpus \|^key \|FmxError\|Error" $O/r3_t28.log | head
