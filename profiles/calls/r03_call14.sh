#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
rm -f $O/fail_*
for tr in 0 1; do
FMX_TEST_TRACE_W0=$tr timeout -k 10 900 python3 -m pytest tests/test_gpu_configs3.py tests/test_gpu_configs4.py tests/test_gpu_distributed.py -q -m gpu > $O/r3_t14_$tr.log 2>&1; rc=$?; echo "trace=$tr tests rc=$rc"; tail -3 $O/r3_t14_$tr.log
for f in $O/fail_*; do [ -f "$f" ] && { echo "== $f (trace=$tr)"; grep "MISMATCH\|W0TRACE" $f | head -30; mv $f $f.tr$tr; }; done
done
