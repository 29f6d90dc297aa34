#!/bin/bash
# Counters behind phase 1's two request schedules and its forms on ragged rows: address translation at the CU (UTCL1), where the L2's fabric reads are
# served (DRAM or not), and the L1 -> L2 read latency.  One counter group per pass, kernel-trace only.
# usage (GPU box, repo root): bash profiles/pmc_phase1_schedules.sh <outdir>
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
run() {  # tag, then the environment and command
  tag=$1; shift
  i=0
  for grp in "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"; do
    i=$((i+1))
    ( for kv in "$@"; do case "$kv" in *=*) export "$kv";; esac; done
      rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$tag/pass$i -- python3 profiles/probes/ragged_probe.py $PROBE_ARGS > $OUT/$tag.pass$i.log 2>&1 )
  done
}
if [ -n "$PMC_PHASE1_ONLY_LEAN" ]; then
PROBE_ARGS="30 30 strata 16" run strata_serial FMX_ROWS_SERIAL=1
PROBE_ARGS="30 30 strata 16" run strata_lean FMX_ROWS_SERIAL=2
PROBE_ARGS="30 30 strata 16" run strata_pipelined FMX_ROWS_SERIAL=0
else
PROBE_ARGS="30 30 iid 16"   run iid_serial FMX_ROWS_SERIAL=1
PROBE_ARGS="30 30 iid 16"   run iid_pipelined FMX_ROWS_SERIAL=0
PROBE_ARGS="1 64 ragged 16" run ragged_static_serial FMX_ROWS_SERIAL=1 FMX_ROWS_FLAT=0
PROBE_ARGS="1 64 ragged 16" run ragged_flat_serial FMX_ROWS_SERIAL=1 FMX_ROWS_FLAT=1
fi
python3 - $OUT <<'PY'
import csv, glob, os, sys, json
from collections import defaultdict
out = sys.argv[1]
res = {}
for tag in sorted(os.listdir(out)):
    if not os.path.isdir(os.path.join(out, tag)): continue
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, tag, "pass*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "fm_rows_forward" in n and ", true" in n:
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    d = {c: v[0] / v[1] for c, v in acc.items()}
    if "TCP_UTCL1_TRANSLATION_HIT_sum" in d: d["utcl1_miss_rate"] = d["TCP_UTCL1_TRANSLATION_MISS_sum"] / max(1.0, d["TCP_UTCL1_TRANSLATION_HIT_sum"] + d["TCP_UTCL1_TRANSLATION_MISS_sum"])
    if "TCC_EA0_RDREQ_sum" in d: d["fabric_reads_from_dram_frac"] = d["TCC_EA0_RDREQ_DRAM_sum"] / max(1.0, d["TCC_EA0_RDREQ_sum"])
    if "TCP_TCC_READ_REQ_sum" in d: d["l1_to_l2_read_latency_cycles"] = d["TCP_TCC_READ_REQ_LATENCY_sum"] / max(1.0, d["TCP_TCC_READ_REQ_sum"])
    if "TCC_HIT_sum" in d: d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    res[tag] = d
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1, sort_keys=True)
for tag, d in res.items():
    print(tag, {k: (round(v, 4) if v < 100 else int(v)) for k, v in sorted(d.items())})
PY
rm -rf $OUT/*/pass*   # (the raw passes are tens of MB: the summary is what is kept)
