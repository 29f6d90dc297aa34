#!/bin/bash
# Round-6 measurement set (GPU box, repo root; results land in gpurun_out/regen6, the judged ones are copied into profiles/ as r06_*).
export TMPDIR=/tmp
O=gpurun_out/regen6
rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_default.out 2> $O/bench_default.err; echo "bench default rc=$?"
tail -n 1 $O/bench_default.out > $O/bench_default_line.json; cp bench_details.json $O/bench_default_details.json
python3 bench.py --solver mcmc --sweep-iid --sweep-exact --steps 2 --warmup 1 --cpu-rows 0 > $O/bench_mcmc_iid_exact.out 2>/dev/null; echo "bench mcmc iid exact rc=$?"
# per-kernel averages: the SAME command line as the headline's timed region (no side runs), under rocprofv3 --kernel-trace --stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sgd -- python3 bench.py --cpu-rows 0 --no-extras > $O/bench_sgd_under_rocprof.out 2> $O/rocprof_sgd.err; echo "rocprof sgd rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcmc -- python3 bench.py --solver mcmc --cpu-rows 0 --no-extras > $O/bench_mcmc_under_rocprof.out 2> $O/rocprof_mcmc.err; echo "rocprof mcmc rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mcmc_iid_exact -- python3 bench.py --solver mcmc --sweep-iid --sweep-exact --steps 2 --warmup 1 --cpu-rows 0 --no-extras > $O/bench_mcmc_iid_exact_under_rocprof.out 2> $O/rocprof_mcmc_iid_exact.err; echo "rocprof mcmc iid exact rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -- python3 bench.py --workload criteo --stream --steps 30 > $O/bench_stream_under_rocprof.out 2> $O/rocprof_stream.err; echo "rocprof stream rc=$?"
python3 profiles/probes/stream_trace_gaps.py $O/prof_stream > $O/stream_queue_occupancy.txt 2>&1; echo "stream trace rc=$?"
# the reference-order learners (bitwise pipelined kernel / reassociated) under the kernel trace: one launch per 65 536 examples
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_seq -- python3 profiles/probes/seq_reassoc_rate.py sgd 16 iid > $O/seq_under_rocprof.out 2> $O/rocprof_seq.err; echo "rocprof seq rc=$?"
bash profiles/pmc_run.sh $O/pmc_sgd --no-extras > $O/pmc_sgd.log 2>&1; echo "pmc sgd rc=$?"
for d in prof_sgd prof_mcmc prof_mcmc_iid_exact prof_stream prof_seq; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_${d#prof_}.csv; done
rm -rf $O/prof_* $O/pmc_sgd/pass*/
ls -la $O
