#!/bin/bash
# round 4, thirty-seventh GPU call: L2 hits of the matrix-driven bare gather (why the real rows do not get what the synthetic sweep gets)
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 200 python3 profiles/probes/gather_matrix_pmc.py 2>&1 | tail -10 | tee $O/r04_gather_matrix_pmc.txt
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $O/pmc_gm -- python3 profiles/probes/gather_matrix_pmc.py > $O/pmc_gm.log 2>&1
python3 - $O/pmc_gm <<'PY' | tee -a $O/r04_gather_matrix_pmc.txt
import csv, glob, sys
from collections import defaultdict
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "gather_matrix_k" in r["Kernel_Name"]]
    by = defaultdict(dict)
    for r in rows: by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"]); by[int(r["Dispatch_Id"])]["k"] = r["Kernel_Name"].split("gather_matrix_k")[1][:7]
    seen = defaultdict(int)
    for d in sorted(by):
        v = by[d]; seen[v["k"]] += 1
        if seen[v["k"]] in (5, 18):   # one launch of the first matrix (strata), one of the second (iid): 13 launches per (matrix, depth)
            print("dispatch", d, v["k"], "requests %.2f M, fabric reads %.2f M, L2 hit rate %.3f" % (v["TCC_REQ_sum"] / 1e6, v["TCC_EA0_RDREQ_sum"] / 1e6, v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
PY
rm -rf $O/pmc_gm
