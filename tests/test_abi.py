"""CPU-side checks of the drop-in boundary: the library builds, loads, and exports every symbol include/fmx.h declares;
the ctypes Config mirrors the C struct; no compute call is made (no GPU here)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from fmwr_amd import build, _lib
    build.build()
    return _lib


def test_exports_every_declared_symbol():
    L = _lib()
    header = open(os.path.join(ROOT, "include", "fmx.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(fmx_[a-z0-9_]+)\s*\(", header))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    lib = L.lib()
    for s in declared:
        assert hasattr(lib, s), s
    # the test hooks live in an internal header, outside the ABI: exported, but not declared to callers
    hooks = open(os.path.join(ROOT, "fmwr_amd", "csrc", "fmx_test_hooks.h")).read()
    hooks = set(re.findall(r"\b(fmx_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hooks, flags=re.S)))
    assert hooks == set(L.TEST_HOOKS) and not (hooks & declared)
    for s in hooks:
        assert hasattr(lib, s), s


def test_config_struct_matches_header():
    L = _lib()
    cfg = L.default_config()
    assert cfg.struct_size == C.sizeof(L.Config)
    # defaults of R/fm_control.R:52-66 and R/fm_solver_control.R:91-115
    assert (cfg.num_factor, cfg.keep_w0, cfg.keep_w1) == (2, 1, 1)
    assert (cfg.learn_rate, cfg.random_step) == (0.01, 1)
    assert (cfg.alpha_w, cfg.alpha_v, cfg.beta_w, cfg.beta_v) == (0.1, 0.1, 1.0, 1.0)
    assert cfg.l2_w0 == cfg.l1_w1 == cfg.l2_w1 == cfg.l1_v == cfg.l2_v == 0.0


def test_fails_loudly_without_gpu():
    """No CPU fallback: without a device every entry point reports an error instead of computing."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    L = _lib()
    cfg = L.default_config()
    h = C.c_void_p()
    st = L.lib().fmx_engine_create(C.byref(cfg), C.c_uint64(10), C.byref(h))
    assert st != L.OK and not h.value
    assert b"no HIP device" in L.lib().fmx_last_error() or st == L.ERR_HIP


def test_dgc_hand_over_checks_its_pointers_on_the_host_and_needs_a_device_for_the_rest():
    """fmx_matrix_from_dgc (the dgCMatrix slots of R/fm_matrix.R:6-43, untransposed): the column pointers are checked before anything touches a device --
    p[0], monotonicity, the total -- and a well-formed matrix then fails for want of a GPU, it is never transposed on the host."""
    import numpy as np
    L = _lib()
    x = np.array([1.0, 2.0, 3.0]); i = np.array([0, 2, 1], np.int32)
    h = C.c_void_p()

    def call(p, nnz=3):
        p = np.asarray(p, np.int32)
        return L.lib().fmx_matrix_from_dgc(C.c_int(0), C.c_int64(3), C.c_uint32(2), C.c_int64(nnz), x.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p),
                                           p.ctypes.data_as(C.c_void_p), None, C.byref(h))
    assert call([1, 2, 3]) == L.ERR_INVALID and b"p[0]" in L.lib().fmx_last_error()
    assert call([0, 2, 1]) == L.ERR_INVALID and b"decrease at column 1" in L.lib().fmx_last_error()
    assert call([0, 2, 2]) == L.ERR_INVALID and b"stored entries" in L.lib().fmx_last_error()
    import torch
    if not torch.cuda.is_available():
        st = call([0, 2, 3])
        assert st != L.OK and not h.value


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "fmwr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                # (comments may NAME the oracle -- e.g. "defined in the oracle: fmo_tdap_apply_sums" -- code may not reach it)
                code = "\n".join(ln.split("//")[0] for ln in text.splitlines()) if f.endswith((".hip", ".h", ".cpp")) else text
                assert "oracle" not in code.lower() or f == "README.md", os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [ln for ln in bench.splitlines() if re.search(r"\bimport oracle\b|\boracle\.", ln)]
    body = bench.split("def cpu_baseline")[1].split("\ndef ")[0]
    assert all(ln in body for ln in uses), "oracle is used outside cpu_baseline() in bench.py"
