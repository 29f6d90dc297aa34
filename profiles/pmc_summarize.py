"""Reduce rocprofv3 --pmc csv output to per-kernel, per-launch averages (hot kernels only)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
bench_args = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
# PMC_SPLIT_TILES=T: a step of T tiles launches each hot kernel T times in a fixed order (FTRL at configs[2]: the first phase-2 launch stores its
# sums, the second adds them and applies the update); the launches are then ALSO averaged per position in the step ("..._tile0", "..._tile1"),
# by their dispatch order within one pass.
split = int(os.environ.get("PMC_SPLIT_TILES", "0"))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    order = {}
    if split > 1:
        seen = defaultdict(dict)   # short kernel name -> dispatch id -> rank
        for row in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
            for key in ("fm_rows_forward", "fm_cols_update"):
                if key in row["Kernel_Name"] and not (key == "fm_rows_forward" and ", true" not in row["Kernel_Name"]):
                    seen[key].setdefault(int(row["Dispatch_Id"]), len(seen[key]))
        order = seen
    for row in rows:
        name = row["Kernel_Name"]
        if "fm_rows_forward" in name and ", true" in name: kn = "fm_rows_forward"           # the training launch (one tile): <T, LPR, TRAIN = true, WGT>
        elif "fm_rows_forward" in name: kn = "fm_rows_forward_predict"                      # bench's forward-only pass (all rows)
        elif "fm_cols_update" in name: kn = "fm_cols_update"
        elif "fm_scalar" in name: kn = "fm_scalar_update"
        elif "als_level_allf_" in name: kn = "als_level_allf"                                # the feature-major coloured sweep (cfg.als_max_levels = -2): one launch per colour, all k factors (wave / register / LDS forms together)
        elif "als_level_k" in name: kn = "als_level"                                        # configs[4]: one level of one factor (the untiled form)
        elif "als_tile_sums_k" in name: kn = "als_tile_sums"                                # the row-tiled form: per (tile, feature) sums ...
        elif "als_tile_step_k" in name: kn = "als_tile_step"                                # ... the coordinate steps of the level ...
        elif "als_rows_apply_k" in name: kn = "als_rows_apply"                              # ... and the row-major rank-1 corrections
        elif "als_order_sums_k" in name: kn = "als_order_sums"                              # the level-order form: stream + list sums + coordinate steps ...
        elif "als_order_apply_k" in name and ", true, unsigned" not in name: kn = "als_order_apply"   # ... and correct-and-permute (the once-per-factor QNEXT variant kept apart)
        elif "als_order_apply_k" in name: kn = "als_order_apply_qnext"
        elif "als_block_level_pipe_k" in name: kn = "als_block_level"                       # the block form (fm_als_blocks.hip): ONE kernel per level ...
        elif "als_block_level_k" in name: kn = "als_block_level_qin" if ", true>" in name else "als_block_level_plain"   # ... a factor's first level (takes its q in as a stream) / the unpipelined form
        elif "als_exact_persist_k" in name: kn = "als_exact_persist"                        # a deep exact plan: one launch per FACTOR (the level loop inside)
        elif "als_exact_flow_k" in name: kn = "als_exact_flow"                              # ... in its record-ordered form (the default since r6's second session)
        elif "als_q_pick_k" in name: kn = "als_q_pick"
        else: continue
        a = acc[kn][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
        if split > 1 and kn in order:
            t = order[kn][int(row["Dispatch_Id"])] % split
            a = acc[f"{kn}_tile{t}"][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
            if "Start_Timestamp" in row and "End_Timestamp" in row:
                a = acc[f"{kn}_tile{t}"]["duration_ns_under_pmc"]
                a[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); a[1] += 1
res = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in acc.items()}
for k, d in res.items():
    # MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced
    # streaming read's bytes (64-B tallies of 128-B requests).  Both the raw and the x2 figure are kept.
    if "FETCH_SIZE" in d:
        d["fetch_bytes_raw"] = d["FETCH_SIZE"] * 1024
        d["fetch_bytes_x2"] = d["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in d:
        d["write_bytes"] = d["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1)
for k, d in res.items():
    # The L2's reads from the fabric by SIZE (round 4: profiles/r04_gather_granularity.txt -- on gfx950 a miss fetches a whole 128-byte line whatever the load
    # asked for and whatever its cache policy, so a random 64-byte row costs 128 bytes of fabric bandwidth): exact read bytes instead of the x2 bound.
    if "TCC_EA0_RDREQ_128B_sum" in d and "TCC_EA0_RDREQ_sum" in d:
        r128, r64, r32 = d["TCC_EA0_RDREQ_128B_sum"], d.get("TCC_EA0_RDREQ_64B_sum", 0.0), d.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        other = max(d["TCC_EA0_RDREQ_sum"] - r128 - r64 - r32, 0.0)
        d["fabric_read_bytes"] = 128 * r128 + 64 * (r64 + other) + 32 * r32
        d["fabric_reads_128B_frac"] = r128 / max(d["TCC_EA0_RDREQ_sum"], 1.0)
        if "write_bytes" in d:
            d["fabric_bytes_per_launch"] = d["fabric_read_bytes"] + d["write_bytes"]
for k, d in res.items():
    if "fetch_bytes_x2" in d and "write_bytes" in d:
        # the figure bench.py quotes as roofline.traffic: guide's gfx950 correction (x2 on FETCH_SIZE) applied to every
        # read request -- an UPPER bound here, because only part of the reads are wide coalesced streams
        d["traffic_bytes_per_launch"] = d["fetch_bytes_x2"] + d["write_bytes"]
res["_bench_args"] = bench_args
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
