#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
for s in 1 0; do
  FMX_LONG_SIDE=$s python3 bench.py --workload criteo --no-extras --cpu-rows 0 > $O/r3_criteo22_$s.json 2>/dev/null; echo "side=$s criteo rc=$?"; python3 -c "
import json;d=json.loads(open('$O/r3_criteo22_$s.json').read().strip().splitlines()[-1]);print(d['value']/1e6,d['ms_per_step'],{k:v['avg_launch_ms'] for k,v in d['roofline']['kernels'].items()})"
  FMX_LONG_SIDE=$s python3 bench.py --workload criteo --stream --steps 40 > $O/r3_stream22_$s.json 2>/dev/null; echo "side=$s stream rc=$?"; cut -c100-230 $O/r3_stream22_$s.json
done
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/r3_t22.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r3_t22.log
