"""What a random 64-byte row costs the fabric: the gather probe (fmx_measure_gather, 64 MB table) with plain / non-temporal / system-scope loads and with
the table in uncached memory (FMX_PROBE_LOAD = 0..3, FMX_PROBE_UNCACHED = 1), rate only; the request sizes come from the PMC pass beside it."""
import os, sys
sys.path.insert(0, ".")
from fmwr_amd import engine
mb, rb = int(sys.argv[1]), int(sys.argv[2])
r = [engine.measure_gather(mb << 20, rb, in_flight=u) / 1e9 for u in (4, 8)]
print("table %d MB, rows of %d B, load mode %s, uncached %s: %.1f G rows/s (4 in flight), %.1f (8)" % (mb, rb, os.environ.get("FMX_PROBE_LOAD", "0"), os.environ.get("FMX_PROBE_UNCACHED", "0"), r[0], r[1]))
