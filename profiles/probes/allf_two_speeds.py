"""Reduce profiles/probes/allf_two_speeds.sh: per process, the feature-major level kernel's launches -- duration (ns, kernel trace) beside GRBM_GUI_ACTIVE and SQ_BUSY_CYCLES."""
import csv, glob, os, sys
from collections import defaultdict
out, n = sys.argv[1], int(sys.argv[2])
for i in range(1, n + 1):
    dur = {}
    for f in glob.glob(os.path.join(out, f"p{i}", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "als_level_allf" in row["Kernel_Name"]:
                dur[int(row["Dispatch_Id"])] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    ctr = defaultdict(dict)
    for f in glob.glob(os.path.join(out, f"p{i}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "als_level_allf" in row["Kernel_Name"]:
                ctr[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
    ids = sorted(set(dur) & set(ctr))
    if not ids:
        print(f"process {i}: no level-kernel launches found"); continue
    ids = ids[len(ids) // 3:]   # (the first sweep of a process warms up)
    d = sum(dur[j] for j in ids) / len(ids)
    g = sum(ctr[j].get("GRBM_GUI_ACTIVE", 0.0) for j in ids) / len(ids)
    q = sum(ctr[j].get("SQ_BUSY_CYCLES", 0.0) for j in ids) / len(ids)
    print(f"process {i}: {len(ids)} level launches: {d / 1e3:8.1f} us each, GRBM_GUI_ACTIVE {g:12.0f} ({g / d:6.3f} per ns), SQ_BUSY_CYCLES {q:14.0f} ({q / d:8.3f} per ns)")
