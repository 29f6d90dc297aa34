"""What does the timed region of bench.py cost beyond its steps?  K timed steps of the headline configuration with and without the HIP-event sampling (every 15th launch), K = 10 / 20 / 40 / 160,
and with the fence variants.  usage: python profiles/probes/timed_region_overhead.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import bench
from fmwr_amd import _lib as L, engine
args = bench.parse([])
m = bench.make_matrix(engine, L, args, args.rows, 0, 0)
B = args.batch_rows
e = engine.Engine(args.features, **bench.engine_kwargs(args, L, B, 0, 1))
v0 = np.random.default_rng(args.seed).normal(0.0, 0.01, (args.factors, args.features)).astype(np.float32)
e.set_params(0.0, None, v0.astype(np.float64))
nb = max(1, m.n // B)
e.num_batches(m); e.sync()
for i in range(21):
    e.step(m, i % nb)
e.sync()
def run(K, prof, W=5):
    for i in range(W):
        e.step(m, i % nb)
    e.sync(); torch.cuda.synchronize()
    e.profile_reset()
    if prof:
        e.profile(prof)
    t0 = time.perf_counter()
    for i in range(K):
        e.step(m, (W + i) % nb)
    e.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e.profile(0)
    return dt
for K in (10, 20, 40, 160):
    for prof in (0, 15, 7):
        ts = [run(K, prof) for _ in range(5)]
        print(f"K = {K:4d}, events every {prof:2d}th launch: {min(ts) / K * 1e3:.4f} ms per step (min of 5), {np.median(ts) / K * 1e3:.4f} median -> {B * K / np.median(ts) / 1e6:.1f} M examples/s", flush=True)
