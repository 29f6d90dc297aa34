#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/r3_t11.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/r3_t11.log; ls $O/fail_* 2>/dev/null && tail -30 $O/fail_*
[ $rc -ne 0 ] && exit $rc
bash profiles/regen_r03.sh
