"""ctypes binding of libfmx.so (include/fmx.h).  This is the same binding a reference-side
maintainer would write (INTEGRATION.md shows the Rcpp form); Python is only the test/bench driver.

There is no fallback: if the library is missing, or no GPU is visible, calls raise.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMX_LIB_PATH") or os.path.join(_PKG, "libfmx.so")  # FMX_LIB_PATH: A/B builds of the library (tuning only)

OK, ERR_INVALID, ERR_HIP, ERR_NOGPU, ERR_STATE = 0, 1, 2, 3, 4
TASK_CLASSIFICATION, TASK_REGRESSION = 10, 20
SOLVER_MCMC, SOLVER_ALS, SOLVER_SGD, SOLVER_FTRL, SOLVER_TDAP = 100, 200, 300, 500, 600
MODE_SEQUENTIAL, MODE_MINIBATCH = 0, 1
LINK_NONE, LINK_LOGISTIC, LINK_CLAMP, LINK_PROBIT = 0, 1, 2, 3
REDUCE_MEAN, REDUCE_SUM = 0, 1
COLUMNS_UNIFORM, COLUMNS_ZIPF = 1, 2
EVAL_LL, EVAL_AUC, EVAL_ACC, EVAL_RMSE, EVAL_MSE, EVAL_MAE = 0, 111, 222, 333, 444, 555
KERNEL_ROWS_FORWARD, KERNEL_COLS_UPDATE, KERNEL_SCALAR, KERNEL_SEQ, KERNEL_ALS_SWEEP = 0, 1, 2, 3, 4

# every symbol include/fmx.h declares (tests/test_abi.py checks the library exports all of them)
SYMBOLS = [
    "fmx_last_error", "fmx_device_count", "fmx_config_default", "fmx_engine_create", "fmx_engine_destroy", "fmx_set_params",
    "fmx_get_params", "fmx_engine_save", "fmx_engine_load", "fmx_matrix_from_rlist", "fmx_matrix_from_dgc", "fmx_matrix_from_csr", "fmx_matrix_synthetic", "fmx_matrix_synthetic_fields", "fmx_matrix_synthetic_iid", "fmx_matrix_synthetic_ragged", "fmx_matrix_synthetic_values", "fmx_train_stream", "fmx_matrix_set_labels", "fmx_matrix_set_fields", "fmx_matrix_destroy",
    "fmx_matrix_info", "fmx_matrix_export", "fmx_matrix_scales", "fmx_matrix_normalize", "fmx_predict", "fmx_train", "fmx_train_grid", "fmx_train_order", "fmx_num_batches",
    "fmx_step", "fmx_grad", "fmx_grad_buffer", "fmx_grad_elem_bytes", "fmx_grad_layout", "fmx_grad_begin", "fmx_grad_chunk", "fmx_apply_chunk", "fmx_apply", "fmx_sync", "fmx_stream", "fmx_predict_device",
    "fmx_als_plan_info", "fmx_als_tiled_info", "fmx_als_order_info", "fmx_als_carry_q", "fmx_als_vsweep", "fmx_mcmc_vsweep", "fmx_als_train", "fmx_mcmc_train", "fmx_mcmc_train_from", "fmx_mcmc_v_hyper", "fmx_evaluate", "fmx_train_tracked", "fmx_trace_size", "fmx_trace_get", "fmx_trace_params",
    "fmx_profile_enable", "fmx_profile_get", "fmx_profile_reset", "fmx_rows_tune_info", "fmx_matrix_rows_form", "fmx_measure_gather", "fmx_measure_gather_occ", "fmx_measure_gather_matrix", "fmx_rccl_selftest",
    "fmx_get_rows", "fmx_set_rows", "fmx_init_normal", "fmx_compact_info", "fmx_compact_count", "fmx_compact_reserve", "fmx_grad_compact", "fmx_compact_records", "fmx_apply_compact",
    "fmx_vsweep_device", "fmx_group_info", "fmx_source_open", "fmx_source_next", "fmx_source_close",
    "fmx_apply_compact_parts", "fmx_layout_info", "fmx_owner_configure", "fmx_owner_info", "fmx_rows_pack", "fmx_rows_unpack",
]


# fmwr_amd/csrc/fmx_test_hooks.h: exported for the GPU tests, not part of the C ABI
TEST_HOOKS = ["fmx_debug_fail_next_plan_build", "fmx_debug_fail_next_comm_init", "fmx_debug_lose_next_seq_multiplier", "fmx_debug_stall_next_persistent_sweep"]


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("task", C.c_int32), ("solver", C.c_int32), ("num_factor", C.c_int32),
        ("keep_w0", C.c_int32), ("keep_w1", C.c_int32),
        ("l2_w0", C.c_double), ("l1_w1", C.c_double), ("l2_w1", C.c_double), ("l1_v", C.c_double), ("l2_v", C.c_double),
        ("learn_rate", C.c_double),
        ("alpha_w", C.c_double), ("alpha_v", C.c_double), ("beta_w", C.c_double), ("beta_v", C.c_double),
        ("random_step", C.c_int32), ("mode", C.c_int32), ("batch_rows", C.c_int64),
        ("min_target", C.c_double), ("max_target", C.c_double),
        ("device", C.c_int32), ("batch_reduce", C.c_int32), ("gamma", C.c_double), ("tile_rows", C.c_int64),
        ("state_fp64", C.c_int32), ("exchange_chunks", C.c_int32), ("n_gpus", C.c_int32), ("als_max_levels", C.c_int32), ("seq_reassociate", C.c_int32), ("gpus_share_device", C.c_int32),
    ]


class FieldsSpec(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_dense", C.c_int32), ("n_fields", C.c_int32), ("reserved", C.c_int32),
                ("field_vocab", C.c_void_p), ("skew", C.c_double), ("seed", C.c_uint64)]


class TrackConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("metric", C.c_int32), ("step_size", C.c_int64), ("convergence", C.c_double),
                ("keep_params", C.c_int32), ("reserved", C.c_int32)]


class FmxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


_lib = None


def lib():
    """Load libfmx.so; raises if it has not been built (python -m fmwr_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build the HIP library first (python -m fmwr_amd.build). "
                              "fmwr_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.fmx_last_error.restype = C.c_char_p
        for name in SYMBOLS + TEST_HOOKS:
            if name != "fmx_last_error":
                getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


def check(status):
    if status != OK:
        raise FmxError(status, lib().fmx_last_error().decode("utf-8", "replace"))


def default_config():
    cfg = Config()
    check(lib().fmx_config_default(C.byref(cfg)))
    return cfg
