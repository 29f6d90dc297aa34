#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
FMX_W_IN_ROW_MAXKP=32 timeout -k 10 600 python3 -m pytest tests/test_gpu_wide_rows.py tests/test_gpu_configs3.py -x -q -m gpu > $O/r3_t23.log 2>&1; rc=$?; echo "tests (maxkp 32) rc=$rc"; tail -3 $O/r3_t23.log
[ $rc -ne 0 ] && exit $rc
for s in 16 32; do
  FMX_W_IN_ROW_MAXKP=$s python3 bench.py --workload criteo --no-extras --cpu-rows 0 > $O/r3_criteo23_$s.json 2>/dev/null; echo "maxkp=$s criteo rc=$?"; python3 -c "
import json;d=json.loads(open('$O/r3_criteo23_$s.json').read().strip().splitlines()[-1]);print(d['value']/1e6,d['ms_per_step'],{k:v['avg_launch_ms'] for k,v in d['roofline']['kernels'].items()})"
  FMX_W_IN_ROW_MAXKP=$s python3 bench.py --workload criteo --stream --steps 40 > $O/r3_stream23_$s.json 2>/dev/null; echo "maxkp=$s stream rc=$?"; cut -c100-230 $O/r3_stream23_$s.json
  FMX_W_IN_ROW_MAXKP=$s python3 bench.py --features 16000000 --factors 32 --no-extras --cpu-rows 0 > $O/r3_p16k32_23_$s.json 2>/dev/null; echo "maxkp=$s p16m k32 rc=$?"; python3 -c "
import json;d=json.loads(open('$O/r3_p16k32_23_$s.json').read().strip().splitlines()[-1]);print(d['value']/1e6,d['ms_per_step'],{k:(v['avg_launch_ms'],v.get('ceiling_frac')) for k,v in d['roofline']['kernels'].items()})"
done
