#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
for p in 250000 500000; do
python3 bench.py --features $p --no-extras --cpu-rows 0 > $O/r3_p35.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/r3_p35.json').read().strip().splitlines()[-1]);print('p=$p',d['value']/1e6,d['ms_per_step'])"
done
python3 bench.py --workload criteo --no-extras --cpu-rows 0 > $O/r3_c35.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/r3_c35.json').read().strip().splitlines()[-1]);print('criteo',d['value']/1e6,d['ms_per_step'])"
python3 profiles/sweep.py --features 250000,1000000,4000000,16000000,33000000 > $O/regen3/sweep_features.txt 2>&1; tail -5 $O/regen3/sweep_features.txt | cut -c1-60
