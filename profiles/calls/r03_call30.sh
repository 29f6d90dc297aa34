#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
export FMX_BENCH_SHARED_DEVICE=1
for x in owner compact; do
  FMX_GROUP_EXCHANGE=$x rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof30_$x -- python3 bench.py --in-library --workload criteo --gpus 2 --rows 4000000 --steps 20 > $O/r3_inlib30_$x.json 2>$O/r3_inlib30_$x.err; echo "$x rc=$?"
done
