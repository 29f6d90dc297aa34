#!/bin/bash
# Where the feature-major level kernel's time goes (als_level_allf_reg_k, i.i.d. columns at configs[4]'s size): the sweep's time with parts of the kernel compiled out.
#   FMX_ALLF_KO bits: 1 no factor loop, 2 no q lines moved (no gathers, no stores of the 128-byte lines), 4 no e gathered or scattered.  Results are wrong by design.
set -e
cd "$(dirname "$0")/../.."
for ko in 0 1 2 4 7; do
  bash profiles/variant_build.sh allf_ko$ko -DFMX_ALLF_KO=$ko
done
for ko in 0 1 2 4 7; do
  echo "== FMX_ALLF_KO=$ko"
  FMX_LIB_PATH=profiles/_variants/allf_ko$ko/libfmx.so python profiles/probes/als_iid_levels.py 2>/dev/null | tail -1
done
