#!/bin/bash
# round 4, thirty-third GPU call: fabric request size of a random-row miss under the cache policies a load can carry, and from uncached memory
export TMPDIR=/tmp
O=gpurun_out
for unc in 0 1; do for mode in 0 1 2 3; do
  FMX_PROBE_UNCACHED=$unc FMX_PROBE_LOAD=$mode timeout -k 10 120 python3 profiles/probes/gather_granularity.py 64 64 2>&1 | tail -1
done; done | tee $O/r04_gather_granularity.txt
for cfg in "0 0" "0 1" "1 0" "0 3"; do
  set -- $cfg
  export FMX_PROBE_UNCACHED=$1 FMX_PROBE_LOAD=$2
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc_gran_$1_$2 -- python3 profiles/probes/gather_granularity.py 64 64 > $O/pmc_gran_$1_$2.log 2>&1
  python3 - $O/pmc_gran_$1_$2 "uncached=$1 mode=$2" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gather_probe_k" in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
print(sys.argv[2], "per launch of 8.4 M row fetches:", {k: int(v[0] / v[1]) for k, v in sorted(acc.items())})
PY
done | tee -a $O/r04_gather_granularity.txt
rm -rf $O/pmc_gran_*
