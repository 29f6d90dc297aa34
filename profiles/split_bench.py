"""One GPU, no communication: device time of the three step forms at configs[1] --
fused fmx_step | fmx_grad + fmx_apply | fmx_grad_begin + C x (fmx_grad_chunk, fmx_apply_chunk).
The difference between the forms is what the multi-GPU split costs before any byte travels."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from fmwr_amd import _lib as L, engine

n, p, z, k, B = 10_000_000, 1_000_000, 30, 16, 1_048_576
m = engine.Matrix.synthetic(n, p, z, 20240001)
v0 = np.random.default_rng(1).normal(0, 0.01, (k, p)).astype(np.float32).astype(np.float64)
out = {}
for name, chunks in (("fused", 0), ("split", 0), ("chunked4", 4), ("chunked8", 8), ("chunked16", 16)):
    e = engine.Engine(p, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4, mode=L.MODE_MINIBATCH, batch_rows=B, exchange_chunks=chunks)
    e.set_params(0.0, None, v0)
    nb = n // B
    nc = e.grad_layout()[0]
    def step(i):
        b = i % nb
        if name == "fused":
            e.step(m, b)
        elif name == "split":
            e.grad(m, b); e.apply(0)
        else:
            e.grad_begin(m, b)
            for c in range(nc):
                e.grad_chunk(m, c)
            for c in range(nc):
                e.apply_chunk(c, 0, c == nc - 1)
    for i in range(5):
        step(i)
    e.sync()
    t0 = time.perf_counter()
    for i in range(40):
        step(5 + i)
    e.sync()
    dt = (time.perf_counter() - t0) / 40
    out[name] = {"ms_per_step": dt * 1e3, "examples_per_s": B / dt}
    print(name, out[name], flush=True)
    del e
json.dump(out, open("gpurun_out/split_bench.json", "w"), indent=1)
