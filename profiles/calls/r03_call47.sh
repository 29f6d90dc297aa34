#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs3.py -x -q -m gpu -k "per_field_sort_equals" > $O/r3_t47.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $O/r3_t47.log
