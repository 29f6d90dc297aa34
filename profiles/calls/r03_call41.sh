#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs3.py tests/test_gpu_props.py -x -q -m gpu > $O/r3_t41.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/r3_t41.log
[ $rc -ne 0 ] && exit $rc
for s in 1 0; do
FMX_FIELD_SORT=$s python3 bench.py --cpu-rows 0 > $O/r3_sgd41_$s.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$O/r3_sgd41_$s.json').read().strip().splitlines()[-1]);print('field sort $s',d['value']/1e6,d['end_to_end']['plan_build_s'],d['end_to_end']['plan_and_one_pass_s'],d['end_to_end']['one_epoch_examples_per_s']/1e6)"
done
