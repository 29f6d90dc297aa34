#!/bin/bash
# round 4, forty-second GPU call: the whole GPU suite and the smoke entry on the final build
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/r04_t42.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/r04_t42.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
