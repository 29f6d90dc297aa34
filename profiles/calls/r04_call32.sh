#!/bin/bash
# round 4, thirty-second GPU call: PMC passes behind phase 1's schedules and forms (profiles/pmc_phase1_schedules.sh)
timeout -k 10 1000 bash profiles/pmc_phase1_schedules.sh gpurun_out/pmc_phase1 2>&1 | tail -12
tail -3 gpurun_out/pmc_phase1/iid_serial.pass1.log
