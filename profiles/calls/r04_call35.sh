#!/bin/bash
# round 4, thirty-fifth GPU call: the schedule / form counters of phase 1 again (summary kept this time), then the default line with the size-counted fabric bytes
export TMPDIR=/tmp
timeout -k 10 600 bash profiles/pmc_phase1_schedules.sh gpurun_out/pmc_phase1 > gpurun_out/r04_pmc_phase1.txt 2>&1; echo "pmc rc=$?"; tail -4 gpurun_out/r04_pmc_phase1.txt | cut -c1-300
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_sgd.json 2> gpurun_out/r04_bench_sgd.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04_bench_sgd.json') if l.startswith('{')][-1])
r=d['roofline']; print(d['value']/1e6, d['ms_per_step'], r['frac'], r.get('fabric'))
for k,v in r['kernels'].items(): print(k, v['avg_launch_ms'], v.get('fabric'))
"
