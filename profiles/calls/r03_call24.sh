#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
T=$1
python3 bench.py --features 16000000 --no-extras --cpu-rows 0 > $O/r3_p16m_$T.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/r3_p16m_$T.json').read().strip().splitlines()[-1]);print('p16m',d['value']/1e6,d['ms_per_step'],d['config']['rows_forward_schedule'],{k:(v['avg_launch_ms'],v.get('ceiling_frac')) for k,v in d['roofline']['kernels'].items()})"
python3 bench.py --features 33000000 --no-extras --cpu-rows 0 > $O/r3_p33m_$T.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/r3_p33m_$T.json').read().strip().splitlines()[-1]);print('p33m',d['value']/1e6,d['ms_per_step'],d['config']['rows_forward_schedule'],{k:(v['avg_launch_ms'],v.get('ceiling_frac')) for k,v in d['roofline']['kernels'].items()})"
python3 bench.py --workload criteo --no-extras --cpu-rows 0 > $O/r3_criteo_$T.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/r3_criteo_$T.json').read().strip().splitlines()[-1]);print('criteo',d['value']/1e6,d['ms_per_step'],d['config']['rows_forward_schedule'],{k:(v['avg_launch_ms'],v.get('ceiling_frac')) for k,v in d['roofline']['kernels'].items()})"
python3 bench.py --no-extras --cpu-rows 0 > $O/r3_sgd_$T.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/r3_sgd_$T.json').read().strip().splitlines()[-1]);print('sgd',d['value']/1e6,d['ms_per_step'],d['config']['rows_forward_schedule'],{k:(v['avg_launch_ms'],v.get('ceiling_frac')) for k,v in d['roofline']['kernels'].items()})"
