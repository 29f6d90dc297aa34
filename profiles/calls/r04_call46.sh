#!/bin/bash
# round 4, forty-sixth GPU call: the round's soak at full size (profiles/soak_r04.py)
export TMPDIR=/tmp
timeout -k 10 1000 python3 profiles/soak_r04.py 2>&1 | tee gpurun_out/r04_soak.txt
