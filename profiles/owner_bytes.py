"""Bytes the owner-sharded exchange moves per step and rank at BASELINE.json configs[3]'s shape, for N = 2, 4, 8 ranks -- computed on
ONE GPU from the per-rank, per-owner record counts of real streamed tiles (rank r's tile = its own row range of the stream, planned by
the same ingest the N-GPU job runs; fmx_owner_info).  Per step rank r SENDS: the ids it asks the other owners for (4 B each), the
rows it serves as an owner ((kp + 4) * 4 B each), and its records for the other owners ((kp + 4) * 4 B each); it RECEIVES the mirror
image.  The all-gather form receives N x max-count records per rank.  Run on the GPU box:  python3 profiles/owner_bytes.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fmwr_amd import _lib as L, engine  # noqa: E402

P, K, B, STEPS = 33_000_000, 32, 262_144, 3
kp = 32
rec = (kp + 4) * 4
for N in (2, 4, 8):
    C = np.zeros((STEPS, N, N), np.int64)   # [step][rank][owner]
    for r in range(N):
        e = engine.Engine(P, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=K, mode=L.MODE_MINIBATCH, batch_rows=B)
        e.owner_configure(N, r)
        src = e.source(STEPS * B, seed=20240001, row_offset=r * STEPS * B, fields=(13, engine.CRITEO_VOCAB, 3.0))
        for s in range(STEPS):
            m = src.next()
            C[s, r] = e.owner_info(m, 0)[0]
        src.close(); e.close()
    sent = np.zeros((STEPS, N)); gath = np.zeros(STEPS)
    for s in range(STEPS):
        for r in range(N):
            away = C[s, r].sum() - C[s, r, r]             # my records / requests for other owners
            serve = C[s, :, r].sum() - C[s, r, r]         # rows other ranks ask me for
            sent[s, r] = away * 4 + serve * rec + away * rec
        gath[s] = N * C[s].sum(axis=1).max() * rec
    print(f"N={N}: records per rank per step {C.sum(axis=2).mean():.0f}; owner-sharded exchange sends {sent.mean() / 1e6:.1f} MB per rank per step "
          f"(max {sent.max() / 1e6:.1f}); all-gather of records receives {gath.mean() / 1e6:.1f} MB; dense all-reduce buffer {(kp + 2) * P * 4 / 1e6:.0f} MB", flush=True)
