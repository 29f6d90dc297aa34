#!/bin/bash
# round 4, forty-fourth GPU call: the stratum-order gather probe on 128-byte rows of a 128 MB table (what w in the V row's line would let phase 1 reach)
export TMPDIR=/tmp
for st in 0 1; do
FMX_PROBE_STRATA=$st timeout -k 10 200 python3 - <<'PY'
import os, sys
sys.path.insert(0, ".")
from fmwr_amd import engine
for groups in (131_072, 262_144):
    for u in (1, 4):
        r = engine.measure_gather(128 << 20, 128, n_groups=groups, per_group=32, in_flight=u, reps=20) / 1e9
        print("128-byte rows, 128 MB: strata=%s groups %7d in flight %d: %.1f G rows/s" % (os.environ.get("FMX_PROBE_STRATA", "0"), groups, u, r))
PY
done | tee gpurun_out/r04_gather_sweep_128.txt
