"""Lengths of the feature lists of one 262 144-row tile of the Criteo-shaped generator: how the entries split between one-entry lists, short, medium and long ones."""
import sys
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import engine
B = 262_144
m = engine.Matrix.synthetic_fields(B, 13, engine.CRITEO_VOCAB, 3.0, 20240001)
rp, col, val, _ = m.export()
cnt = np.bincount(col, minlength=m.p)
occ = cnt[cnt > 0]
print("features %d, occurring %d, entries %d" % (m.p, occ.size, cnt.sum()))
edges = [1, 2, 3, 5, 9, 17, 33, 65, 129, 1025, 10**9]
for a, b in zip(edges[:-1], edges[1:]):
    sel = occ[(occ >= a) & (occ < b)]
    print("lists of %5d..%-9d entries: %8d lists (%.3f of lists), %9d entries (%.3f of entries), rounds of four: %d" % (a, b - 1, sel.size, sel.size / occ.size, sel.sum(), sel.sum() / cnt.sum(), int(np.sum((sel + 3) // 4))))
