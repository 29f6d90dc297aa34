#!/bin/bash
# Round-3 measurement set (run on the GPU box from the repo root; results land in gpurun_out/regen3, the judged ones are copied
# into profiles/ as r03_*).
export TMPDIR=/tmp
O=gpurun_out/regen3
rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_sgd.json 2> $O/bench_sgd.err; echo "bench sgd rc=$?"
python3 bench.py --solver ftrl > $O/bench_ftrl.json 2> $O/bench_ftrl.err; echo "bench ftrl rc=$?"
python3 bench.py --solver mcmc > $O/bench_mcmc.json 2> $O/bench_mcmc.err; echo "bench mcmc rc=$?"
python3 bench.py --solver als > $O/bench_als.json 2> $O/bench_als.err; echo "bench als rc=$?"
python3 bench.py --workload criteo > $O/bench_criteo.json 2> $O/bench_criteo.err; echo "bench criteo rc=$?"
python3 bench.py --workload criteo --stream --steps 40 > $O/bench_stream.json 2> $O/bench_stream.err; echo "bench stream rc=$?"
FMX_STREAM_OVERLAP=0 python3 bench.py --workload criteo --stream --steps 40 > $O/bench_stream_one_ingest_stream.json 2>/dev/null; echo "bench stream (one ingest stream) rc=$?"
python3 bench.py --features 16000000 --no-extras --cpu-rows 0 > $O/bench_p16m.json 2>/dev/null; echo "bench p16m rc=$?"
python3 bench.py --in-library --no-extras --cpu-rows 0 > $O/bench_inlib.json 2>/dev/null; echo "bench in-library rc=$?"
# per-kernel averages of ONE schedule (see regen_r02.sh)
export FMX_ROWS_SERIAL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sgd -- python3 bench.py --cpu-rows 0 --no-extras > $O/bench_sgd_under_rocprof.json 2> $O/rocprof_sgd.err; echo "rocprof sgd rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcmc -- python3 bench.py --solver mcmc --cpu-rows 0 --no-extras > $O/bench_mcmc_under_rocprof.json 2> $O/rocprof_mcmc.err; echo "rocprof mcmc rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -- python3 bench.py --workload criteo --stream --steps 40 > $O/bench_stream_under_rocprof.json 2> $O/rocprof_stream.err; echo "rocprof stream rc=$?"
bash profiles/pmc_run.sh $O/pmc_sgd --no-extras > $O/pmc_sgd.log 2>&1; echo "pmc sgd rc=$?"
unset FMX_ROWS_SERIAL
python3 profiles/sweep.py --features 250000,1000000,4000000,16000000,33000000 > $O/sweep_features.txt 2>&1
python3 profiles/sweep.py --factors 4,8,16,32,64 > $O/sweep_k.txt 2>&1
find $O -name "*kernel_stats.csv"
# the boundary and the streamed steady state (round 3, later additions)
python3 profiles/host_handover.py > $O/host_handover.json 2>/dev/null; echo "host hand-over rc=$?"
python3 profiles/stream_steady.py > $O/stream_steady.txt 2>&1; echo "stream steady rc=$?"
python3 profiles/soak_r03.py > $O/soak_r03.txt 2>&1; echo "soak r03 rc=$?"
