#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
for s in 1 0; do
  for p in 250000 500000; do
  FMX_LONG_SIDE=$s python3 bench.py --features $p --no-extras --cpu-rows 0 > $O/r3_p_$p_$s.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/r3_p_$p_$s.json').read().strip().splitlines()[-1]);print('side=$s p=$p',d['value']/1e6,d['ms_per_step'],{k:v['avg_launch_ms'] for k,v in d['roofline']['kernels'].items()})"
  done
done
FMX_LONG_SIDE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof34 -- python3 bench.py --features 250000 --no-extras --cpu-rows 0 > /dev/null 2>&1
