#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
for c in 0 1 2 3 4 5; do
  FMX_FQ_CFG=$c timeout -k 10 300 python3 -m pytest tests/test_gpu_configs3.py -x -q -m gpu -k "per_field or field_structured or streamed_training" > $O/r3_t18_$c.log 2>&1; rc=$?; echo "cfg $c tests rc=$rc"; tail -2 $O/r3_t18_$c.log
  [ $rc -ne 0 ] && exit $rc
  FMX_FQ_CFG=$c python3 bench.py --workload criteo --stream --steps 40 > $O/r3_stream18_$c.json 2>/dev/null; echo "cfg $c stream rc=$?"; cut -c100-240 $O/r3_stream18_$c.json
  FMX_FQ_CFG=$c FMX_STREAM_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof18_$c -- python3 bench.py --workload criteo --stream --steps 40 > /dev/null 2> $O/r3_prof18.err; echo "rocprof rc=$?"
done
