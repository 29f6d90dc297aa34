#!/bin/bash
# round 5, third session: the whole GPU suite, then the default bench line (the driver's command) and the configs[4] line on the block form
export TMPDIR=/tmp
O=gpurun_out/final5c
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"
python bench.py --solver mcmc --no-extras --steps 6 --warmup 1 --cpu-rows 2000000 > $O/bench_mcmc.json 2> $O/bench_mcmc.err; echo "mcmc bench rc=$?"
python bench.py --solver als --no-extras --steps 6 --warmup 1 --cpu-rows 0 > $O/bench_als.json 2> $O/bench_als.err; echo "als bench rc=$?"
python3 - <<PY
import json
for n in ("bench_default","bench_mcmc","bench_als"):
    d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1])
    print(n, round(d["value"]/1e6,1), d["ms_per_step"], d["roofline"]["frac"], (d.get("value_q_carried") or {}).get("value"), (d.get("learner_iteration") or {}).get("ms"))
    for k,v in (d.get("other_configs") or {}).items(): print("   ",k, round(v.get("value",0)/1e6,1), v.get("roofline",{}).get("frac"), (v.get("value_q_carried") or {}).get("value"))
PY
