#!/bin/bash
# round 4, thirty-sixth GPU call: the gather probe walking the table stratum by stratum in step (all lane groups resident) against random rows; L2 hits by PMC
export TMPDIR=/tmp
O=gpurun_out
for st in 0 1; do FMX_PROBE_STRATA=$st timeout -k 10 200 python3 profiles/probes/gather_sweep.py 2>&1 | grep strata; done | tee $O/r04_gather_sweep.txt
for st in 0 1; do
  export FMX_PROBE_STRATA=$st
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/pmc_sweep_$st -- python3 profiles/probes/gather_sweep.py > $O/pmc_sweep_$st.log 2>&1
  python3 - $O/pmc_sweep_$st "strata=$st" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gather_probe_k" in row["Kernel_Name"]:
            key = (row["Kernel_Name"].split("gather_probe_k")[1][:8], row["Grid_Size"])
            a = acc[key][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for key, d in sorted(acc.items()):
    v = {k: x[0] / x[1] for k, x in d.items()}
    print(sys.argv[2], key, "fabric reads %.2f M, L2 hit rate %.3f" % (v["TCC_EA0_RDREQ_sum"] / 1e6, v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
PY
done | tee -a $O/r04_gather_sweep.txt
rm -rf $O/pmc_sweep_*
