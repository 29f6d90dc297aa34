#!/bin/bash
# round 4, fifty-second GPU call: the whole GPU suite, then the round's measurement set on the final build (profiles/regen_r04.sh)
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/r04_t52.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r04_t52.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 1100 bash profiles/regen_r04.sh 2>&1 | tail -16
rm -rf $O/regen4/prof_*/ $O/regen4/pmc_*/pass*
