"""Helper for tuning runs on the GPU box: run bench.py over a grid of one option and print one line each."""
import json
import subprocess
import sys

opt, values, rest = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
for v in values:
    r = subprocess.run([sys.executable, "bench.py", "--cpu-rows", "0", "--no-extras", opt, v] + rest, capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not line:
        print(opt, v, "FAILED", r.stderr[-400:])
        continue
    d = json.loads(line[-1])
    km = {k: round(x["avg_launch_ms"], 4) for k, x in d["roofline"]["kernels"].items()}
    fr = {k: round(x["frac"], 3) for k, x in d["roofline"]["kernels"].items()}
    print(opt, v, f"{d['value'] / 1e6:.1f} Mex/s", f"{d['ms_per_step']:.3f} ms/step", "ms per tile", km, "kernel frac", fr, f"step frac {d['roofline']['frac']:.3f}", flush=True)
