#!/bin/bash
# round 4, twenty-ninth GPU call: the whole GPU suite on the final build, then the round's measurement set (profiles/regen_r04.sh)
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/r04_t29.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/r04_t29.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 1400 bash profiles/regen_r04.sh 2>&1 | tail -20
