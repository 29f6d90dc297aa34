#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/r3_t63.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/r3_t63.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 profiles/host_handover.py > $O/r3_handover.json 2> $O/r3_handover.err; echo rc=$?; cat $O/r3_handover.json
