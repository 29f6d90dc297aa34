#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
for v in default sw3 sw5 sw6; do
  if [ $v = default ]; then unset FMX_LIB_PATH; else export FMX_LIB_PATH=$PWD/profiles/_variants/$v/libfmx.so; fi
  python3 bench.py --features 16000000 --no-extras --cpu-rows 0 > $O/r3_v51.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/r3_v51.json').read().strip().splitlines()[-1]);print('$v p16m',round(d['value']/1e6,1),{k:round(v['avg_launch_ms'],4) for k,v in d['roofline']['kernels'].items()})"
  python3 bench.py --workload criteo --no-extras --cpu-rows 0 > $O/r3_v51.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/r3_v51.json').read().strip().splitlines()[-1]);print('$v criteo',round(d['value']/1e6,1),{k:round(v['avg_launch_ms'],4) for k,v in d['roofline']['kernels'].items()})"
  python3 bench.py --features 4000000 --no-extras --cpu-rows 0 > $O/r3_v51.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/r3_v51.json').read().strip().splitlines()[-1]);print('$v p4m',round(d['value']/1e6,1),{k:round(v['avg_launch_ms'],4) for k,v in d['roofline']['kernels'].items()})"
done
