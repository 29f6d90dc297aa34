"""The streamed configs[3] line inside bench.py's default process reads ~376 M against ~400 M on its own.  What before it matters?  usage: stream_in_process2.py <what> ..."""
import gc, sys
sys.path.insert(0, ".")
import torch
import bench
from fmwr_amd import _lib as L, engine
torch.cuda.set_device(0)
def stream(tag):
    a = bench.parse(["--workload", "criteo", "--stream", "--steps", "30", "--warmup", "3", "--no-other-configs"])
    print(f"{tag}: streamed {bench.main_stream(a, 0, 0, 1)['value'] / 1e6:.1f} M examples/s", flush=True)
def mb(argv):
    a = bench.parse(argv + ["--no-other-configs"])
    a.cpu_one_core_only = True
    return bench.run_minibatch(a, 0, 0, 1)
for what in sys.argv[1:]:
    if what == "headline_noextras": mb(["--no-extras", "--cpu-rows", "0"])
    elif what == "headline_extras": mb(["--cpu-rows", "0"])
    elif what == "fp64": mb(["--state-fp64", "--no-extras", "--steps", "20", "--warmup", "3", "--cpu-rows", "0"])
    elif what == "ftrl": mb(["--solver", "ftrl", "--no-extras", "--steps", "16", "--warmup", "2", "--cpu-rows", "0"])
    elif what == "resident": mb(["--workload", "criteo", "--steps", "30", "--warmup", "3", "--cpu-rows", "0"])
    elif what == "cpu": mb(["--no-extras", "--cpu-rows", "1000000"])
    stream("after " + what)
