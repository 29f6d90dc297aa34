#!/bin/bash
# round 4, sixteenth GPU call: the hand-written general pair sort: plans bitwise the library sort's, then its speed
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_configs3.py tests/test_gpu_train.py tests/test_gpu_distributed.py -x -q -m gpu > $O/r04_t16.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 $O/r04_t16.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 profiles/pair_sort_probe.py > $O/r04_pair_sort.txt 2>&1; echo "probe (hand-written) rc=$?"
FMX_PAIR_SORT=rocprim timeout -k 10 300 python3 profiles/pair_sort_probe.py >> $O/r04_pair_sort.txt 2>&1; echo "probe (rocprim) rc=$?"
cat $O/r04_pair_sort.txt
