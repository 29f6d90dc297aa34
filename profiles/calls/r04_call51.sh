#!/bin/bash
# round 4, fifty-first GPU call: the driver's N-rank command at N = 2 and N = 4 with its DEFAULT arguments (10 M x 1 M, 1 048 576 rows per rank and step), ranks sharing
# this box's one device, backend gloo (the exchange goes through the host: the figure is a rehearsal of the code path, not a rate)
export TMPDIR=/tmp FMX_BENCH_SHARED_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 2 4; do
  timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) bench.py --gpus $N --steps 5 --warmup 2 --backend gloo 2> gpurun_out/r04_bench_n$N.err | tail -1 > gpurun_out/r04_bench_n${N}_rehearsal.json; echo "N=$N rc=$?"
  python3 -c "
import json
d=json.loads(open('gpurun_out/r04_bench_n${N}_rehearsal.json').read())
print('N=%d value %.1f M ex/s ms/step %.3f' % (d['n_gpus'], d['value']/1e6, d['ms_per_step']), {k:d['config'][k] for k in ('batch_rows_per_gpu','global_batch_rows','rows_per_gpu','learn_rate','parallelism','exchange') if k in d['config']})"
done
