#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
rm -f $O/fail_*
for ov in 1 0 1 0; do
  FMX_STREAM_OVERLAP=$ov timeout -k 10 300 python3 -m pytest "tests/test_gpu_distributed.py::test_owner_sharded_exchange_on_the_gpu" -q -m gpu -k "stream" > $O/r3_t12_$ov.log 2>&1; echo "overlap=$ov rc=$?"; tail -2 $O/r3_t12_$ov.log
  for f in $O/fail_*; do [ -f "$f" ] && { echo "== $f (overlap=$ov)"; grep "MISMATCH\|  id " $f | head -40; mv $f $f.ov$ov.$RANDOM; }; done
done
