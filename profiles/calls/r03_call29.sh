#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
export FMX_BENCH_SHARED_DEVICE=1
for x in owner compact; do
  FMX_GROUP_EXCHANGE=$x timeout -k 10 300 python3 bench.py --in-library --workload criteo --gpus 2 --rows 4000000 --steps 20 > $O/r3_inlib29_$x.json 2>$O/r3_inlib29_$x.err; echo "$x rc=$?"; cut -c1-330 $O/r3_inlib29_$x.json | cut -c90-330; tail -2 $O/r3_inlib29_$x.err
done
timeout -k 10 300 python3 bench.py --in-library --workload criteo --gpus 1 --rows 4000000 --steps 20 > $O/r3_inlib29_n1.json 2>/dev/null; echo "n1 rc=$?"; cut -c90-330 $O/r3_inlib29_n1.json
