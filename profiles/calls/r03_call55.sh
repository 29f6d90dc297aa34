#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_group.py -x -q -m gpu > $O/r3_t55.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r3_t55.log
[ $rc -ne 0 ] && exit $rc
export FMX_BENCH_SHARED_DEVICE=1
for x in owner compact; do
  FMX_GROUP_EXCHANGE=$x timeout -k 10 300 python3 bench.py --in-library --workload criteo --gpus 2 --rows 4000000 --steps 20 > $O/r3_inlib55_$x.json 2>/dev/null; echo "$x rc=$?"; cut -c100-240 $O/r3_inlib55_$x.json
done
