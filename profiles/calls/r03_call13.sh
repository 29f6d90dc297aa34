#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
rm -f $O/fail_*
timeout -k 10 900 python3 -m pytest tests -q -m gpu > $O/r3_t13.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $O/r3_t13.log
for f in $O/fail_*; do [ -f "$f" ] && { echo "== $f"; grep "MISMATCH\|  id " $f | head -60; }; done
[ $rc -ge 2 ] && exit $rc
bash profiles/regen_r03.sh
