#!/bin/bash
# round 4, thirty-ninth GPU call: counters of the lean form on the headline generator's rows (what it hits and misses against the wide kernel's two schedules)
PMC_PHASE1_ONLY_LEAN=1 timeout -k 10 800 bash profiles/pmc_phase1_schedules.sh gpurun_out/pmc_phase1_lean > gpurun_out/r04_pmc_phase1_lean.txt 2>&1; echo "pmc rc=$?"; tail -4 gpurun_out/r04_pmc_phase1_lean.txt | cut -c1-1200
