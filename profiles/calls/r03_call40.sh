#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/r3_t40.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r3_t40.log
