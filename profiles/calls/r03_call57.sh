#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
export FMX_BENCH_SHARED_DEVICE=1
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --rows 4000000 > $O/r3_dp2.json 2>$O/r3_dp2.err; echo "dense rc=$?"; tail -1 $O/r3_dp2.json | cut -c1-420
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --backend gloo --workload criteo --stream --exchange owner --steps 6 --warmup 2 > $O/r3_dp2s.json 2>$O/r3_dp2s.err; echo "stream owner rc=$?"; tail -1 $O/r3_dp2s.json | cut -c1-420
