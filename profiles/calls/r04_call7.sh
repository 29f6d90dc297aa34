#!/bin/bash
# round 4, seventh GPU call: the tiled sweep at full size (parity with the column-walking form, reproducibility), the new tests of the round,
# PMC passes of the tiled kernels, bench lines of configs[4]
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_configs4.py tests/test_gpu_group.py tests/test_gpu_api.py tests/test_gpu_wide_rows.py -x -q -m gpu > $O/r04_t7.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $O/r04_t7.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --solver mcmc > $O/r04_bench_mcmc.json 2> $O/r04_bench_mcmc.err; echo "bench mcmc rc=$?"
timeout -k 10 300 python3 bench.py --solver als > $O/r04_bench_als.json 2> $O/r04_bench_als.err; echo "bench als rc=$?"
timeout -k 10 600 bash profiles/pmc_run.sh $O/pmc_mcmc_tiled --solver mcmc --no-extras --steps 2 --warmup 1 > $O/pmc_mcmc_tiled.log 2>&1; echo "pmc rc=$?"
python3 - <<'PY'
import json
for f in ("r04_bench_mcmc", "r04_bench_als"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        r = d["roofline"]
        print(f, "%.1f M ex/s" % (d["value"] / 1e6), "%.1f ms/step" % d["ms_per_step"], "level %.4f ms" % r["avg_launch_ms"], "frac %.3f" % r["frac"], "design_frac", r.get("design_frac"), r.get("gather_ceiling", {}).get("ceiling_frac"), d.get("cpu_baseline", {}).get("value"))
    except Exception as ex:
        print(f, "FAILED", ex)
d = json.load(open("gpurun_out/pmc_mcmc_tiled/pmc_summary.json"))
for k, v in d.items():
    if isinstance(v, dict):
        print(k, {a: round(b / 1e6, 2) for a, b in v.items() if a in ("fetch_bytes_raw", "write_bytes", "traffic_bytes_per_launch", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "TCC_HIT_sum", "TCC_MISS_sum")}, "hit", round(v.get("l2_hit_rate", -1), 3))
PY
