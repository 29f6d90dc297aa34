#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
rm -f $O/fail_*
timeout -k 10 1000 python3 -m pytest tests -q -m gpu > $O/r3_t15.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $O/r3_t15.log
for f in $O/fail_*; do [ -f "$f" ] && { echo "== $f"; grep "MISMATCH\|  id " $f | head -60; }; done
[ $rc -ge 2 ] && exit $rc
python3 bench.py --solver ftrl --no-extras --cpu-rows 0 > $O/r3_ftrl15.json 2>$O/r3_ftrl15.err; echo "ftrl rc=$?"; cat $O/r3_ftrl15.json
python3 bench.py --no-extras --cpu-rows 0 > $O/r3_sgd15.json 2>$O/r3_sgd15.err; echo "sgd rc=$?"; cat $O/r3_sgd15.json
