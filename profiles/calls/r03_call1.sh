#!/bin/bash
# round 3, first GPU call: new tests (configs[4], group fixes), the ALS / MCMC bench lines, PMC passes out of the Infinity Cache
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 420 python3 -m pytest tests/test_gpu_configs4.py tests/test_gpu_group.py tests/test_gpu_api.py -x -q -m gpu > $O/r3_t1.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/r3_t1.log
[ $rc -ge 2 ] && exit $rc
timeout -k 10 200 python3 bench.py --solver mcmc > $O/r3_bench_mcmc.json 2> $O/r3_bench_mcmc.err; rc=$?; echo "bench mcmc rc=$rc"
[ $rc -ge 124 ] && exit $rc
timeout -k 10 200 python3 bench.py --solver als > $O/r3_bench_als.json 2> $O/r3_bench_als.err; rc=$?; echo "bench als rc=$rc"
[ $rc -ge 124 ] && exit $rc
export FMX_ROWS_SERIAL=1
timeout -k 10 500 bash profiles/pmc_run.sh $O/r3_pmc_p16m --features 16000000 --no-extras > $O/r3_pmc_p16m.log 2>&1; echo "pmc p16m rc=$?"
