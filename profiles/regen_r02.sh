#!/bin/bash
# Round-2 measurement set (run on the GPU box from the repo root; results land in gpurun_out/regen2, the judged ones are copied
# into profiles/ as r02_*).
export TMPDIR=/tmp
O=gpurun_out/regen2
rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_sgd.json 2> $O/bench_sgd.err; echo "bench sgd rc=$?"
python3 bench.py --solver ftrl > $O/bench_ftrl.json 2> $O/bench_ftrl.err; echo "bench ftrl rc=$?"
python3 bench.py --batch-rows 1048576 --no-extras --cpu-rows 0 > $O/bench_sgd_1m.json 2>/dev/null; echo "bench sgd 1M rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sgd -- python3 bench.py --cpu-rows 0 --no-extras > $O/bench_sgd_under_rocprof.json 2> $O/rocprof_sgd.err; echo "rocprof sgd rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ftrl -- python3 bench.py --solver ftrl --cpu-rows 0 --no-extras > $O/bench_ftrl_under_rocprof.json 2> $O/rocprof_ftrl.err; echo "rocprof ftrl rc=$?"
bash profiles/pmc_run.sh $O/pmc_sgd --no-extras > $O/pmc_sgd.log 2>&1; echo "pmc sgd rc=$?"
bash profiles/pmc_run.sh $O/pmc_ftrl --solver ftrl --no-extras > $O/pmc_ftrl.log 2>&1; echo "pmc ftrl rc=$?"
find $O -name "*kernel_stats.csv" | head
