#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/r3_t66.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r3_t66.log
python3 -c "
import __graft_entry__ as g
g.smoke()
" 2>&1 | tail -2
python3 bench.py 2>/dev/null | cut -c1-260
