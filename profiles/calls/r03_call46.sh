#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py -x -q -m gpu -k "hand_over" > $O/r3_t46.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/r3_t46.log
