#!/bin/bash
# Knock-outs of the block form's pipelined level kernel (fm_als_blocks.hip, -DFMX_BLK_KO=bits: compile-time, so that the register allocation of what is left is
# its own): per-level launch time of what is left.  build (CPU box): profiles/probes/block_knockouts.sh build ; run (GPU box): profiles/probes/block_knockouts.sh
KOS="${KOS:-1 2 3 4 8 12 15}"
if [ "$1" = build ]; then
  python -m fmwr_amd.build > /dev/null
  for ko in $KOS; do
    out=profiles/_variants/blk_ko$ko; mkdir -p $out
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DFMX_BLK_KO=$ko -c fmwr_amd/csrc/fm_als_blocks.hip -o $out/fm_als_blocks.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libfmx.so $(ls fmwr_amd/csrc/_obj/*.o | grep -v fm_als_blocks) $out/fm_als_blocks.o -ldl && rm $out/fm_als_blocks.o ) &
    if [ $(jobs -r | wc -l) -ge 4 ]; then wait -n; fi
  done
  wait; ls -la profiles/_variants/blk_ko*/libfmx.so; exit 0
fi
line() { tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('FMX_BLK_KO=$1: sweep %.2f ms, level launch %.1f us (avg over %d, incl. the 1 in 30 that takes q in), q build %.2f ms' % (d['ms_per_step'], r['avg_launch_ms']*1e3, r['timed_launches'], r['q_build_forward_ms']))"; }
timeout -k 10 120 python bench.py --solver mcmc --no-extras --steps 2 --warmup 1 --cpu-rows 0 2>/dev/null | line 0
for ko in $KOS; do
  FMX_LIB_PATH=$PWD/profiles/_variants/blk_ko$ko/libfmx.so timeout -k 10 120 python bench.py --solver mcmc --no-extras --steps 2 --warmup 1 --cpu-rows 0 2>/dev/null | line $ko
done
