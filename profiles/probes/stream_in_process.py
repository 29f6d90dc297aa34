"""Why does the streamed configs[3] line read ~6 % lower inside the default bench run (after the headline and the other side runs, in one process) than on its own?
The same main_stream call: first in a fresh process, again after a resident Criteo-shaped run, again after gc, again while a 10 M-row matrix is alive."""
import gc, sys
sys.path.insert(0, ".")
import torch
import bench
from fmwr_amd import _lib as L, engine
def stream():
    a = bench.parse(["--workload", "criteo", "--stream", "--steps", "30", "--warmup", "3", "--no-other-configs"])
    return bench.main_stream(a, 0, 0, 1)["value"] / 1e6
torch.cuda.set_device(0)
print(f"fresh process:                          {stream():.1f} M examples/s", flush=True)
print(f"again:                                  {stream():.1f}", flush=True)
a = bench.parse(["--workload", "criteo", "--steps", "30", "--warmup", "3", "--cpu-rows", "0", "--no-other-configs"])
r = bench.run_minibatch(a, 0, 0, 1)["value"] / 1e6
print(f"(resident Criteo-shaped run: {r:.1f})", flush=True)
print(f"after the resident run:                 {stream():.1f}", flush=True)
gc.collect()
print(f"after gc.collect():                     {stream():.1f}", flush=True)
args = bench.parse([])
m = bench.make_matrix(engine, L, args, args.rows, 0, 0)
print(f"with a 10 M x 1 M matrix alive:         {stream():.1f}", flush=True)
e = engine.Engine(args.features, **bench.engine_kwargs(args, L, args.batch_rows, 0, 1))
e.init_normal(1, 0.0, 0.01)
for i in range(30):
    e.step(m, i)
e.sync()
print(f"... and its engine, after 30 steps:     {stream():.1f}", flush=True)
e.close(); m.close(); gc.collect()
print(f"both closed:                            {stream():.1f}", flush=True)
# stream -> hardware-queue assignment: every engine creates its streams round-robin over the runtime's hardware queues; shift the phase with throw-away engines
for extra in (1, 1, 1, 1, 1, 1, 1, 1):
    d = engine.Engine(1000, num_factor=4, mode=L.MODE_MINIBATCH, batch_rows=128)
    d.close()
    print(f"one more engine created and closed before it: {stream():.1f}", flush=True)
