"""One wave's view of the random-row gather: few lane groups, many dependent rounds, so a round's duration is what one wave pays
(latency + whatever serialises its requests), against the throughput regime of profiles/gather_occupancy.py."""
import sys
sys.path.insert(0, ".")
from fmwr_amd import engine
for table_mb in (1, 16.8, 64, 1024):
    for groups, what in ((16, "1 wave"), (64, "1 workgroup (4 waves)"), (16 * 256, "1 wave per CU*"), (64 * 256, "1 workgroup per CU*")):
        line = f"table {table_mb:6.1f} MB, {what:24s}"
        for u in (4, 8):
            per_group = 4096
            r = engine.measure_gather(int(table_mb * 1e6) // 64 * 64, 64, groups, per_group, u, reps=5)
            us_round = groups * per_group / r * 1e6 / (per_group / u)
            line += f" | {u} in flight: {us_round:6.2f} us per round, {r / 1e6 / max(groups // 16, 1):7.1f} rows/us per wave"
        print(line)
print("* if the dispatcher spreads the workgroups evenly")
