#!/bin/bash
# round 3, second GPU call: the whole GPU suite after the streaming / owner-exchange refactor, then the new bench drivers
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/r3_t2.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $O/r3_t2.log
[ $rc -ge 2 ] && exit $rc
timeout -k 10 200 python3 bench.py --workload criteo --stream --steps 20 > $O/r3_bench_stream_n1.json 2> $O/r3_bench_stream_n1.err; rc=$?; echo "bench stream n1 rc=$rc"
[ $rc -ge 124 ] && exit $rc
for ex in owner compact; do
FMX_BENCH_SHARED_DEVICE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --workload criteo --stream --exchange $ex --backend gloo --steps 10 --warmup 2 > $O/r3_bench_stream_n2_$ex.json 2> $O/r3_bench_stream_n2_$ex.err; rc=$?; echo "bench stream n2 $ex rc=$rc"
[ $rc -ge 124 ] && exit $rc
done
timeout -k 10 200 python3 bench.py --in-library --no-extras --cpu-rows 0 > $O/r3_bench_inlib_n1.json 2> $O/r3_bench_inlib_n1.err; echo "bench in-library rc=$?"
