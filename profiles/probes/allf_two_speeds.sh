#!/bin/bash
# what sets the feature-major level kernel's two speeds (29.6 or 35.9 ms per sweep, per PROCESS)?  N fresh processes of the same probe: plain first (does this
# box show both speeds?), then under rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES: the level kernel's duration in ns beside its cycles.  Equal cycles at
# different ns = clock state; different cycles = placement / scheduling.     usage (GPU box, repo root): bash profiles/probes/allf_two_speeds.sh <outdir> [N]
OUT=$1; N=${2:-5}
export TMPDIR=/tmp
mkdir -p "$OUT"
for i in $(seq 1 $N); do
  python3 profiles/probes/als_iid_levels.py stratified > "$OUT/plain$i.log" 2>&1 || exit 1
  grep "sweep ms" "$OUT/plain$i.log" | tr '\n' ' '; echo " (plain process $i)"
done
for i in $(seq 1 $N); do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d "$OUT/p$i" -- python3 profiles/probes/als_iid_levels.py stratified > "$OUT/p$i.log" 2>&1 || exit 1
  grep "sweep ms" "$OUT/p$i.log" | tr '\n' ' '; echo " (process $i under rocprofv3 --pmc)"
done
python3 profiles/probes/allf_two_speeds.py "$OUT" $N
