#!/bin/bash
# round 5: the default bench line, rocprofv3 kernel stats of the same command's headline part, of configs[4] and of the streamed configs[3]; PMC passes of configs[4]
export TMPDIR=/tmp
O=gpurun_out/final5
mkdir -p $O
python bench.py > gpurun_out/r05_bench_default.json 2> $O/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats -d $O/ks_sgd -o s --output-format csv -- python3 bench.py --columns iid --no-extras --cpu-rows 0 > $O/ks_sgd.log 2>&1; cp $O/ks_sgd/s_kernel_stats.csv gpurun_out/r05_kernel_stats_sgd.csv
rocprofv3 --kernel-trace --stats -d $O/ks_mcmc -o s --output-format csv -- python3 bench.py --solver mcmc --no-extras --cpu-rows 0 --steps 3 --warmup 1 > $O/ks_mcmc.log 2>&1; cp $O/ks_mcmc/s_kernel_stats.csv gpurun_out/r05_kernel_stats_mcmc.csv
rocprofv3 --kernel-trace --stats -d $O/ks_stream -o s --output-format csv -- python3 bench.py --workload criteo --stream --steps 30 --warmup 3 > $O/ks_stream.log 2>&1; cp $O/ks_stream/s_kernel_stats.csv gpurun_out/r05_kernel_stats_stream.csv
rocprofv3 --kernel-trace --stats -d $O/ks_fp64 -o s --output-format csv -- python3 bench.py --columns iid --state-fp64 --no-extras --cpu-rows 0 > $O/ks_fp64.log 2>&1; cp $O/ks_fp64/s_kernel_stats.csv gpurun_out/r05_kernel_stats_sgd_fp64.csv
bash profiles/pmc_run.sh $O/pmc_mcmc --solver mcmc --no-extras --steps 2 --warmup 1 > $O/pmc_mcmc.log 2>&1; echo "pmc mcmc rc=$?"; cp $O/pmc_mcmc/pmc_summary.json gpurun_out/r05_pmc_summary_mcmc.json
rm -rf $O/pmc_*/pass* $O/ks_*/s_kernel_trace.csv
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f M/s ms %.4f frac %.3f (with values %.3f) traffic %s fabric %s" % (d["value"] / 1e6, d["ms_per_step"], r["frac"], r.get("frac_with_unread_values", 0), r.get("traffic"), (r.get("fabric") or {}).get("frac")))
for n, v in r["kernels"].items(): print(" ", n, {k: v[k] for k in ("avg_launch_ms", "frac", "ceiling_frac") if k in v})
for k in ("value_stratified_columns", "value_real_values", "value_zipf_columns", "value_ragged_rows", "scaling_reference", "sequential_exact"):
    print(" ", k, d.get(k, {}).get("value"))
print("  ttq", json.dumps(d.get("time_to_quality", {}).get("learners"))[:1200])
for k, v in d["other_configs"].items(): print(" ", k, v.get("value"), v.get("roofline", {}).get("frac"), v.get("error"))
PY
