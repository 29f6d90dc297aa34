#!/bin/bash
# round 3, fourth GPU call: ALS sweeps (device dyn struct, graph replay of deep plans, guarded approximate form), w-in-row with full-sector w stores
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_configs4.py tests/test_gpu_train.py tests/test_gpu_wide_rows.py tests/test_gpu_tracker.py tests/test_gpu_api.py tests/test_golden.py -x -q -m gpu > $O/r3_t4.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $O/r3_t4.log
[ $rc -ge 2 ] && exit $rc
for g in 1 0; do
  FMX_ALS_GRAPH=$g timeout -k 10 300 python3 profiles/als_levels.py 4000000 > $O/r3_als_levels_graph$g.txt 2>&1; echo "als levels graph=$g rc=$?"
done
for flag in 0 1; do
  FMX_W_IN_ROW=$flag timeout -k 10 300 python3 profiles/sweep.py --features 4000000,16000000 > $O/r3_wir2_${flag}_k16.txt 2>&1
done
FMX_W_IN_ROW=1 FMX_WIR_FULL_STORE=0 timeout -k 10 300 python3 profiles/sweep.py --features 16000000 > $O/r3_wir2_1_partial_k16.txt 2>&1
cat $O/r3_als_levels_graph1.txt $O/r3_als_levels_graph0.txt $O/r3_wir2_0_k16.txt $O/r3_wir2_1_k16.txt $O/r3_wir2_1_partial_k16.txt
