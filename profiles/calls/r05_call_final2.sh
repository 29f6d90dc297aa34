#!/bin/bash
# round 5, closing run: the CPU-side checks the driver makes, the whole GPU suite, smoke, the default bench line (timed), kernel stats + PMC passes of configs[4]
export TMPDIR=/tmp
O=gpurun_out/final5b
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
SECONDS=0; python bench.py > gpurun_out/r05_bench_default.json 2> $O/bench.err; echo "bench rc=$? wall ${SECONDS} s"
rocprofv3 --kernel-trace --stats -d $O/ks_mcmc -o s --output-format csv -- python3 bench.py --solver mcmc --no-extras --cpu-rows 0 --steps 3 --warmup 1 > $O/ks_mcmc.log 2>&1; cp $O/ks_mcmc/s_kernel_stats.csv gpurun_out/r05_kernel_stats_mcmc.csv
bash profiles/pmc_run.sh $O/pmc_mcmc --solver mcmc --no-extras --steps 2 --warmup 1 > $O/pmc_mcmc.log 2>&1; echo "pmc mcmc rc=$?"; cp $O/pmc_mcmc/pmc_summary.json gpurun_out/r05_pmc_summary_mcmc.json
rm -rf $O/pmc_*/pass* $O/ks_*/s_kernel_trace.csv
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f M/s ms %.4f frac %.3f (with values %.3f) fabric %s" % (d["value"] / 1e6, d["ms_per_step"], r["frac"], r.get("frac_with_unread_values", 0), (r.get("fabric") or {}).get("frac")))
for k, v in d["other_configs"].items(): print(" ", k, v.get("value"), v.get("roofline", {}).get("frac"), v.get("roofline", {}).get("traffic"), v.get("error"))
PY
