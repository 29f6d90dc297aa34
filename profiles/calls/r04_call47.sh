#!/bin/bash
# round 4, forty-seventh GPU call: the seeded fuzz against the oracle on 1 500 more seeds than the suite runs (one-off)
export TMPDIR=/tmp
timeout -k 10 1000 python3 - <<'PY' 2>&1 | tail -6 | tee gpurun_out/r04_fuzz_more.txt
import sys, time, traceback
sys.path.insert(0, ".")
from tests.test_gpu_fuzz import test_fuzz_against_oracle, _case
t = time.time(); bad = []
for seed in range(150, 1650):
    try:
        test_fuzz_against_oracle(seed)
    except Exception as ex:
        bad.append(seed); print("seed", seed, _case(seed), "FAILED:", repr(ex)[:300], flush=True)
    if time.time() - t > 900: print("stopped at seed", seed); break
print("fuzz seeds 150..%d: %d failed %s in %.0f s" % (seed, len(bad), bad[:20], time.time() - t))
PY
