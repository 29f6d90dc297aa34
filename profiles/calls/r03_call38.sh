#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
bash profiles/pmc_run.sh $O/pmc_p16m --features 16000000 --no-extras > $O/pmc_p16m.log 2>&1; echo "pmc p16m rc=$?"; tail -3 $O/pmc_p16m.log
bash profiles/pmc_run.sh $O/pmc_criteo --workload criteo > $O/pmc_criteo.log 2>&1; echo "pmc criteo rc=$?"; tail -3 $O/pmc_criteo.log
