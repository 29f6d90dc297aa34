#!/bin/bash
# round 4, first GPU call: what the memory system gives a row-tiled V sweep (probe), and the PMC passes of the untiled als_level_k
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 200 profiles/probes/bin/gather_tiled > $O/r04_gather_tiled.txt 2>&1; echo "probe rc=$?"
cat $O/r04_gather_tiled.txt
timeout -k 10 600 bash profiles/pmc_run.sh $O/pmc_mcmc_untiled --solver mcmc --no-extras --steps 2 --warmup 1 > $O/pmc_mcmc_untiled.log 2>&1; echo "pmc rc=$?"
tail -50 $O/pmc_mcmc_untiled.log
