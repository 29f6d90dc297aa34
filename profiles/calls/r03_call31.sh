#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/r3_t31.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/r3_t31.log
gcc -std=c99 -Iinclude examples/kat_c.c -Lfmwr_amd -lfmx -Wl,-rpath,$PWD/fmwr_amd -lm -o /tmp/kat_c && /tmp/kat_c | tail -3
