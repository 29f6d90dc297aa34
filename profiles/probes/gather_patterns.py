"""What the memory system gives the gather of phase 1 under different ORDERS of the same kind of ids (64-byte rows of a 64 MB table, configs[1]):
uniformly random ids (fmx_measure_gather) against the ids the matrices themselves name, walked as phase 1 walks them (fmx_measure_gather_matrix: lane group
g takes row g, entries in row order, U outstanding per lane): one column per stratum (the headline generator: entry j of every row lies in stratum j), i.i.d.
uniform columns sorted inside the row, and the ragged law."""
import sys
sys.path.insert(0, ".")
from fmwr_amd import engine
n, p, z, B = 2_000_000, 1_000_000, 30, 262_144
print("uniformly random ids: %.1f G rows/s (4 in flight), %.1f (8)" % tuple(engine.measure_gather(p * 64, 64, in_flight=u) / 1e9 for u in (4, 8)))
for name, m in (("strata", engine.Matrix.synthetic(n, p, z, 20240001)), ("iid sorted", engine.Matrix.synthetic_iid(n, p, z, 20240001)),
                ("ragged [1,64]", engine.Matrix.synthetic_ragged(n, p, float(z), 20240001)), ("ragged [25,35]", engine.Matrix.synthetic_ragged(n, p, float(z), 20240001, min_nnz=25, max_nnz=35))):
    r = [engine.measure_gather_matrix(m, B, B, p, 64, in_flight=u, reps=20) / 1e9 for u in (4, 8)]
    print("%-16s ids of rows %d..%d: %.1f G rows/s (4 in flight), %.1f (8)   [%.4f ms per 262144-row tile at the better]" % (name, B, 2 * B, r[0], r[1], m.nnz / m.n * B / max(r) / 1e6))
    m.close()
