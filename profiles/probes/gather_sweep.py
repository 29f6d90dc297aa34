"""Does a chip-wide sweep in step make L2 hits?  The gather probe (64 MB table of 64-byte rows, 32 fetches per lane group) with every group's fetch j taken from
stratum j of the table (FMX_PROBE_STRATA=1: the order in which phase 1 walks rows of one column per stratum) against uniformly random rows, at 1 / 2 / 4 fetches in
flight per lane group and with all groups resident at once (131 072 groups = 8 waves per SIMD) or two and four times as many."""
import os, sys
sys.path.insert(0, ".")
from fmwr_amd import engine
for groups in (131_072, 262_144, 524_288):
    for u in (1, 2, 4):
        r = engine.measure_gather(64 << 20, 64, n_groups=groups, per_group=32, in_flight=u, reps=20) / 1e9
        print("strata=%s groups %7d in flight %d: %.1f G rows/s" % (os.environ.get("FMX_PROBE_STRATA", "0"), groups, u, r))
