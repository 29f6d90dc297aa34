"""The matrix-driven bare gather (fmx_measure_gather_matrix) on the headline generator's rows (one column per stratum) and on i.i.d. sorted columns, 1 / 4 / 8 in flight."""
import sys
sys.path.insert(0, ".")
from fmwr_amd import engine
n, p, z, B = 1_000_000, 1_000_000, 30, 262_144
for name, m in (("strata", engine.Matrix.synthetic(n, p, z, 20240001)), ("iid", engine.Matrix.synthetic_iid(n, p, z, 20240001))):
    for u in (1, 4, 8):
        r = engine.measure_gather_matrix(m, B, B, p, 64, in_flight=u, reps=10) / 1e9
        print("%s in flight %d: %.1f G rows/s" % (name, u, r))
    rp, col, _, _ = m.export(B, B + 4)
    print(name, "first rows:", [col[rp[i]:rp[i + 1]][:6].tolist() for i in range(2)])
    m.close()
