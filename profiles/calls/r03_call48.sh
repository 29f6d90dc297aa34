#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 profiles/soak_r03.py > $O/r3_soak_r03.txt 2>&1; echo "soak r03 rc=$?"; cat $O/r3_soak_r03.txt
timeout -k 10 600 python3 profiles/soak.py > $O/r3_soak.txt 2>&1; echo "soak rc=$?"; tail -4 $O/r3_soak.txt
