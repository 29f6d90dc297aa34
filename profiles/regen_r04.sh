#!/bin/bash
# Round-4 measurement set (run on the GPU box from the repo root; results land in gpurun_out/regen4, the judged ones are copied into profiles/ as r04_*).
export TMPDIR=/tmp
O=gpurun_out/regen4
rm -rf $O; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_sgd.json 2> $O/bench_sgd.err; echo "bench default (configs[1] + other_configs) rc=$?"
python3 bench.py --solver mcmc > $O/bench_mcmc.json 2> $O/bench_mcmc.err; echo "bench mcmc rc=$?"
python3 bench.py --solver als > $O/bench_als.json 2> $O/bench_als.err; echo "bench als rc=$?"
python3 bench.py --solver ftrl > $O/bench_ftrl.json 2> $O/bench_ftrl.err; echo "bench ftrl rc=$?"
python3 bench.py --workload criteo > $O/bench_criteo.json 2> $O/bench_criteo.err; echo "bench criteo rc=$?"
python3 bench.py --workload criteo --stream --steps 40 > $O/bench_stream.json 2> $O/bench_stream.err; echo "bench stream rc=$?"
python3 bench.py --no-linear --no-extras --cpu-rows 0 --no-other-configs > $O/bench_no_linear.json 2>/dev/null; echo "bench --no-linear rc=$?"
# per-kernel averages of ONE schedule (see regen_r02.sh)
export FMX_ROWS_SERIAL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sgd -- python3 bench.py --cpu-rows 0 --no-extras > $O/bench_sgd_under_rocprof.json 2> $O/rocprof_sgd.err; echo "rocprof sgd rc=$?"
unset FMX_ROWS_SERIAL
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcmc -- python3 bench.py --solver mcmc --cpu-rows 0 --no-extras > $O/bench_mcmc_under_rocprof.json 2> $O/rocprof_mcmc.err; echo "rocprof mcmc rc=$?"
bash profiles/pmc_run.sh $O/pmc_mcmc --solver mcmc --no-extras --steps 2 --warmup 1 > $O/pmc_mcmc.log 2>&1; echo "pmc mcmc rc=$?"
FMX_ROWS_SERIAL=1 bash profiles/pmc_run.sh $O/pmc_sgd --no-extras > $O/pmc_sgd.log 2>&1; echo "pmc sgd rc=$?"
python3 profiles/soak_r04.py > $O/soak_r04.txt 2>&1; echo "soak r04 rc=$?"
find $O -name "*kernel_stats.csv"
