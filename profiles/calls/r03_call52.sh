#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 profiles/stream_bench.py > $O/r3_stream_bench52.txt 2>&1; echo rc=$?; cat $O/r3_stream_bench52.txt | tail -3
FMX_FIELD_SORT=0 timeout -k 10 600 python3 profiles/stream_bench.py > $O/r3_stream_bench52_pair.txt 2>&1; echo rc=$?; cat $O/r3_stream_bench52_pair.txt | tail -3
