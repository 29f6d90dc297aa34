#!/bin/bash
# round 4, eighth GPU call: ragged generator test, learning curves at the global batches of N = 1..32, the default bench line with other_configs (timed)
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py -x -q -m gpu > $O/r04_t8.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/r04_t8.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 profiles/learning_scaling.py > $O/r04_learning_scaling.txt 2> $O/r04_learning_scaling.err; echo "learning rc=$?"; grep -v "^{" $O/r04_learning_scaling.txt
s=$(date +%s.%N)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_default.json 2> $O/r04_bench_default.err; rc=$?
e=$(date +%s.%N); echo "default bench rc=$rc wall $(echo "$e - $s" | bc) s"; tail -3 $O/r04_bench_default.err
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04_bench_default.json") if l.startswith("{")][-1])
print("value %.1f M, ms/step %.4f, frac %.3f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"]))
print("end_to_end", d.get("end_to_end"))
print("ragged", d.get("value_ragged_rows"))
print("iid", d.get("value_iid_uniform"))
for k, v in d.get("other_configs", {}).items():
    if "error" in v: print(k, v); continue
    print(k, "%.1f M" % (v["value"] / 1e6), "%.3f ms" % v["ms_per_step"], "wall %.1f s" % v["wall_s"], "frac", round(v["roofline"].get("frac", 0), 3),
          {n: (round(x.get("frac") or 0, 3), round(x.get("ceiling_frac") or 0, 3)) for n, x in v["roofline"].get("kernels", {}).items()}, "cpu", v.get("cpu_baseline", {}).get("value"))
PY
