#!/bin/bash
# round 3, third GPU call: the w-in-row layout (bitwise tests, then A/B out of the caches), the owner exchange's bytes at N = 2/4/8
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_wide_rows.py tests/test_gpu_distributed.py tests/test_gpu_c_caller.py tests/test_gpu_configs3.py -x -q -m gpu > $O/r3_t3.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $O/r3_t3.log
[ $rc -ge 2 ] && exit $rc
for flag in 0 1; do
  FMX_W_IN_ROW=$flag timeout -k 10 300 python3 profiles/sweep.py --features 1000000,4000000,16000000,33000000 > $O/r3_wir_${flag}_k16.txt 2>&1; echo "sweep wir=$flag rc=$?"
  FMX_W_IN_ROW=$flag timeout -k 10 300 python3 profiles/sweep.py --features 16000000 --factors 8 > $O/r3_wir_${flag}_k8.txt 2>&1
done
timeout -k 10 300 python3 profiles/owner_bytes.py > $O/r3_owner_bytes.txt 2>&1; echo "owner bytes rc=$?"
cat $O/r3_wir_0_k16.txt $O/r3_wir_1_k16.txt $O/r3_wir_0_k8.txt $O/r3_wir_1_k8.txt $O/r3_owner_bytes.txt
