#!/bin/bash
# A/B and diagnostic builds of the library, never the product build:  profiles/variant_build.sh NAME [-DFLAG ...]
# -> profiles/_variants/NAME/libfmx.so (git-ignored; run with FMX_LIB_PATH=...).  -DFMX_TRACE adds the in-kernel stamps read by
# profiles/trace_rows_forward.py.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=profiles/_variants/$name
mkdir -p $out/obj
for f in fmx_api fm_batch_kernels fm_seq_kernels fm_ingest fm_als_kernels fm_als_tiled fm_als_blocks fm_eval_kernels fm_measure fm_group; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w "$@" -c fmwr_amd/csrc/$f.hip -o $out/obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libfmx.so $out/obj/*.o -ldl
rm -rf $out/obj
