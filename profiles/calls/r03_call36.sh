#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
python3 profiles/sweep.py --features 250000,1000000,4000000,16000000,33000000 > $O/r3_sweep_features36.txt 2>&1; tail -5 $O/r3_sweep_features36.txt | cut -c1-70
python3 profiles/sweep.py --factors 4,8,16,32,64 > $O/r3_sweep_k36.txt 2>&1; tail -5 $O/r3_sweep_k36.txt | cut -c1-70
