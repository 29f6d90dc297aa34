#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/r3_t44.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/r3_t44.log
[ $rc -ne 0 ] && exit $rc
gcc -std=c99 -Iinclude examples/kat_c.c -Lfmwr_amd -lfmx -Wl,-rpath,$PWD/fmwr_amd -lm -o /tmp/kat_c && /tmp/kat_c | head -3
timeout -k 10 600 python3 profiles/host_handover.py > $O/r3_handover.json 2> $O/r3_handover.err; echo rc=$?; cat $O/r3_handover.json; tail -3 $O/r3_handover.err
