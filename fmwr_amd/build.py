"""Builds fmwr_amd/libfmx.so (the C-ABI library of include/fmx.h) with hipcc for gfx950.

hipcc cross-compiles without a GPU; the .so stays in-tree (git-ignored) so it travels with the
repo snapshot to the GPU box.  Run:  python -m fmwr_amd.build [--force]
"""
import concurrent.futures
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libfmx.so")
SOURCES = ["fmx_api.hip", "fm_batch_kernels.hip", "fm_seq_kernels.hip", "fm_ingest.hip", "fm_als_kernels.hip", "fm_als_tiled.hip", "fm_als_blocks.hip", "fm_eval_kernels.hip", "fm_measure.hip", "fm_group.hip"]
HEADERS = [os.path.join(CSRC, "fmx_internal.h"), os.path.join(CSRC, "fm_probit.h"), os.path.join(CSRC, "fmx_test_hooks.h"), os.path.join(PKG, "..", "include", "fmx.h")]
# -ffp-contract=off: the fp64 update formulas keep the reference's operation order (no FMA fusion)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-Wno-deprecated-declarations"] + os.environ.get("FMX_EXTRA_FLAGS", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(src):
    s = os.path.join(CSRC, src)
    o = os.path.join(OBJ, src.replace(".hip", ".o"))
    if _stale(o, [s] + HEADERS):
        subprocess.check_call([_hipcc()] + FLAGS + ["-c", s, "-o", o])
    return o


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if force or _stale(LIB, objs):
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
