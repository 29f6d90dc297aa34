"""What would putting a feature on the wave of its latest predecessor buy the record-ordered exact sweep (als_exact_flow_k)?  A CPU model of the sweep's timing on an i.i.d.
matrix: a record handed on through the fabric costs Lr, through the same wave's LDS Ll, a step S, a wave's own time per feature O (microseconds).
usage: python profiles/probes/als_affinity_sim.py rows columns      (results: profiles/r06_als_exact_persist.txt)"""
import numpy as np, sys, time
n, p, z = int(sys.argv[1]), int(sys.argv[2]), 30
rng = np.random.default_rng(1)
t0 = time.time()
# i.i.d. columns, sorted inside the row, no repeats (bump)
cols = np.sort(rng.integers(0, p, size=(n, z)), axis=1)
for _ in range(3):
    dup = cols[:, 1:] == cols[:, :-1]
    cols[:, 1:][dup] += 1
    cols = np.sort(np.minimum(cols, p - 1), axis=1)
pred = np.full((n, z), -1, dtype=np.int64); pred[:, 1:] = cols[:, :-1]
flat_c = cols.ravel(); flat_p = pred.ravel()
order = np.argsort(flat_c, kind="stable")
fc = flat_c[order]; fp = flat_p[order]
ptr = np.searchsorted(fc, np.arange(p + 1))
# levels
level = np.zeros(p, dtype=np.int64)
for j in range(p):
    pr = fp[ptr[j]:ptr[j + 1]]
    pr = pr[pr >= 0]
    pr = pr[pr != j]
    level[j] = (level[pr].max() + 1) if len(pr) else 0
L = level.max() + 1
pos_order = np.lexsort((np.arange(p), level))
print(f"n {n} p {p}: levels {L}, features per level {p / L:.1f}, built in {time.time() - t0:.0f} s", flush=True)

def simulate(NW, policy, Lr=1.9, Ll=0.15, S=0.35, O=1.2):
    # O: a wave's own time per feature besides the step (issue of loads / stores, statics); the wave is busy O + S after its inputs are there
    fin = np.zeros(p); wave_of = np.full(p, -1, dtype=np.int64); wfree = np.zeros(NW); last_on = np.full(NW, -1, dtype=np.int64)
    rr = 0
    nlocal = 0
    for j in pos_order:
        pr = fp[ptr[j]:ptr[j + 1]]
        pr = np.unique(pr[(pr >= 0) & (pr != j)])
        if len(pr) == 0:
            w = rr % NW; rr += 1
            ready = 0.0
        else:
            fpr = fin[pr]
            if policy == "rr":
                w = rr % NW; rr += 1
            else:
                # the wave of the latest predecessor, if this feature would be that wave's NEXT one (its record still in LDS) and the wave is not far behind
                k = pr[np.argmax(fpr)]
                w = wave_of[k]
                if last_on[w] != k or wfree[w] > fpr.max() + 0.5:
                    w = int(np.argmin(wfree))
            lat = np.where((wave_of[pr] == w) & (last_on[w] == pr), Ll, Lr)
            nlocal += int(((wave_of[pr] == w) & (last_on[w] == pr)).any())
            ready = (fpr + lat).max()
        start = max(ready, wfree[w] + O)
        fin[j] = start + S
        wfree[w] = fin[j]
        wave_of[j] = w; last_on[w] = j
    return fin.max(), nlocal
for NW in (128, 256):
    for pol in ("rr", "aff"):
        t, nl = simulate(NW, pol)
        print(f"NW {NW} {pol}: {t:.0f} us total = {t / L:.2f} us per level; features with a local predecessor {nl}", flush=True)
