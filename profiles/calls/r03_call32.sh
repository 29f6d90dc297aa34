#!/bin/bash
export TMPDIR=/tmp
python3 -c "
import __graft_entry__ as g
g.smoke()
print('smoke ok')
" 2>&1 | tail -3
python3 bench.py --steps 20 --warmup 5 2>/dev/null | cut -c1-200
