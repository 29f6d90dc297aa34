/* Test hooks of libfmx.so -- NOT part of the C ABI (include/fmx.h).  The library exports them so that the GPU tests can arm a fault through ctypes;
 * nothing in the library or its drivers calls them, each is one-shot, and none changes a result.  Kept out of the public header on purpose. */
#ifndef FMX_TEST_HOOKS_H_
#define FMX_TEST_HOOKS_H_
#ifdef __cplusplus
extern "C" {
#endif
/* the next per-tile plan build of this process fails once with FMX_ERR_HIP, as an allocation failure halfway would (tests/test_gpu_api.py:
 * a failed build leaves no half-built cache behind) */
int fmx_debug_fail_next_plan_build(void);
/* the next fmx_engine_create with cfg.n_gpus > 1 fails where the communicator is initialised -- after the other replicas were created -- as
 * ncclCommInitAll failing on a later device would (tests/test_gpu_group.py: the error path tears everything down, the next create works) */
int fmx_debug_fail_next_comm_init(void);
/* the next launch of the reassociated reference-order learner (cfg.seq_reassociate) never sees the multiplier of its example 7: that worker's bounded wait gives up, every
 * other wait follows, and fmx_get_params / fmx_sync must report FMX_ERR_HIP instead of handing out the NaN (tests/test_gpu_seq_reassoc.py) */
int fmx_debug_lose_next_seq_multiplier(void);
/* in the next persistent exact sweep the plan's first feature does not hand its rows on (als_exact_flow_k: their tags stay where they were; als_exact_persist_k: it is
 * never counted): the features after it never become ready, the bounded waits give up, and the sweep must fail with FMX_ERR_HIP (tests/test_gpu_configs4.py).
 * Takes a few seconds: the waits are bounded generously */
int fmx_debug_stall_next_persistent_sweep(void);
#ifdef __cplusplus
}
#endif
#endif /* FMX_TEST_HOOKS_H_ */
