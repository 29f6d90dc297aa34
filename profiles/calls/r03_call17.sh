#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs3.py -x -q -m gpu > $O/r3_t17.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $O/r3_t17.log
[ $rc -ne 0 ] && exit $rc
python3 bench.py --workload criteo --stream --steps 40 > $O/r3_stream17.json 2>$O/r3_stream17.err; echo "stream rc=$?"; cut -c1-400 $O/r3_stream17.json
FMX_FIELD_SORT=0 python3 bench.py --workload criteo --stream --steps 40 > $O/r3_stream17_pairsort.json 2>/dev/null; echo "stream pair sort rc=$?"; cut -c1-400 $O/r3_stream17_pairsort.json
FMX_STREAM_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof17 -- python3 bench.py --workload criteo --stream --steps 40 > $O/r3_stream17_prof.json 2> $O/r3_prof17.err; echo "rocprof rc=$?"
