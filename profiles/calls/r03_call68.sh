#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs3.py -x -q -m gpu > $O/r3_t68.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/r3_t68.log
for s in 0 1 0 1; do FMX_FQ_PLAIN=$s python3 bench.py --workload criteo --stream --steps 40 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain=$s stream %.1f M  %.4f ms'%(d['value']/1e6,d['ms_per_step']))"; done
for s in 0 1; do FMX_FQ_PLAIN=$s FMX_STREAM_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof68_$s -- python3 bench.py --workload criteo --stream --steps 40 > /dev/null 2>&1; done
