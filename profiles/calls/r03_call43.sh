#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 600 python3 profiles/host_handover.py > gpurun_out/r3_handover.json 2> gpurun_out/r3_handover.err; echo rc=$?; cat gpurun_out/r3_handover.json; tail -3 gpurun_out/r3_handover.err
