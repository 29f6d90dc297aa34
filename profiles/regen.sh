#!/bin/bash
# Regenerates every measurement under profiles/ on the GPU box (run from the repo root; results land in gpurun_out/regen,
# copy what is to be judged into profiles/ afterwards).
set -e
export TMPDIR=/tmp
O=gpurun_out/regen
rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --cpu-rows 0 > $O/bench_under_rocprof.json 2> $O/rocprof.err
echo rocprof done
bash profiles/pmc_run.sh $O/pmc > $O/pmc.log 2>&1
echo pmc done
python3 profiles/sweep.py --factors 4,8,16,32,64,128 > $O/sweep_k.txt
python3 profiles/sweep.py --features 100000,1000000,4000000,16000000 > $O/sweep_features.txt
python3 profiles/sweep.py --batch-rows 65536,131072,262144,524288,1048576,2097152 > $O/sweep_batch.txt
python3 profiles/sweep.py --tile-rows 131072,262144,524288 > $O/sweep_tile.txt
echo sweeps done
python3 profiles/extra_bench.py > $O/extra.log 2>&1
python3 profiles/skew_bench.py > $O/skew.txt 2>&1
python3 profiles/split_bench.py > $O/split.txt 2>&1
echo all done
