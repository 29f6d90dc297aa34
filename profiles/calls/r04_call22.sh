#!/bin/bash
# round 4, twenty-second GPU call: the bare gather under the id orders of the different row laws
export TMPDIR=/tmp
timeout -k 10 300 python3 profiles/probes/gather_patterns.py 2>&1 | tee gpurun_out/r04_gather_patterns.txt
