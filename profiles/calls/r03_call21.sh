#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
export FMX_ROWS_SERIAL=1
PMC_SPLIT_TILES=2 bash profiles/pmc_run.sh $O/pmc_ftrl --solver ftrl --no-extras > $O/pmc_ftrl.log 2>&1; echo "pmc ftrl rc=$?"
tail -5 $O/pmc_ftrl.log
