#!/usr/bin/env python3
"""bench.py -- examples/sec of the FM SGD hot path on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W

A "step" is one synchronous mini-batch SGD step (forward + gradient sums + update) over `--batch-rows` rows PER GPU of
the synthetic 10M x 1M, 30 nnz/row, k=16 workload (BASELINE.json configs[1]); the matrix is generated on the device
and is resident in HBM before the timed region.  The engine cuts a step into tiles of <= 262144 rows (two kernel
launches per tile: fm_rows_forward, fm_cols_update) with the parameters frozen across the tiles.  N > 1: one process per GPU (torch.distributed / RCCL), each rank
owns a contiguous row range, computes its gradient sums, the (k+2)*p buffer is all-reduced, every replica applies the
same update ("scaling": "weak": per-GPU rows per step are fixed).

Rank 0 prints ONE JSON line: metric/value (whole-job examples/s), the dominant kernel's roofline (HIP-event timed on
the engine's stream inside the timed region) and, at N == 1, the oracle's serial reference-order SGD timed on the
host on a bounded sample of the same rows ("cpu_baseline").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows of the whole synthetic matrix")
    ap.add_argument("--features", type=int, default=1_000_000)
    ap.add_argument("--nnz", type=int, default=30)
    ap.add_argument("--factors", type=int, default=16)
    ap.add_argument("--batch-rows", type=int, default=1_048_576,
                    help="mini-batch rows per GPU per step (processed in cache-resident tiles of <= 262144 rows)")
    ap.add_argument("--tile-rows", type=int, default=0, help="rows per tile (0: the engine's default, 262144)")
    ap.add_argument("--solver", choices=["sgd", "ftrl"], default="sgd")
    ap.add_argument("--seed", type=int, default=20240001)
    ap.add_argument("--state-fp64", action="store_true", help="experiment: fp64 parameter/optimizer state (default fp32)")
    ap.add_argument("--no-linear", action="store_true", help="experiment: keep.w1 = FALSE (no w gathers)")
    ap.add_argument("--exchange-chunks", type=int, default=0,
                    help="N > 1: blocks of features the exchange is pipelined in (1: one all-reduce of the whole buffer per step; "
                         "0: 8 blocks on 2 GPUs, where the single xGMI link is the bound and finer blocks hide more of it, 4 otherwise, "
                         "where the extra launches of finer blocks cost more than they hide: profiles/r01_split_bench.json)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl == RCCL; gloo for rehearsals)")
    ap.add_argument("--cpu-rows", type=int, default=5_000_000, help="rows of the CPU-baseline sample (0: skip)")
    return ap.parse_args()


def algorithmic_bytes(z, k, p, rows, e=4):
    """Per-launch algorithmic HBM bytes of the two hot kernels and of the SURVEY 8(d) fused step; e = bytes per state
    element (4: the fp32 state SURVEY 8(d) prices; 8 with --state-fp64)."""
    rows_fwd = rows * (z * (4 + 4 + e + e * k) + 8 + 4 + e * k + e)   # idx,val,w,V row | row_ptr, y, S row, mult
    cols_upd = rows * z * (4 + 4 + e + e * k) + p * (4 + 2 * e * k + 2 * e)  # row,val,mult,S row | bptr, V RMW, w RMW
    survey_step = rows * (z * (8 + 2 * e + 2 * e * k) + 12)
    return rows_fwd, cols_upd, survey_step


def pmc_traffic(kernel, args):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summary (profiles/r*_pmc_summary.json,
    made by profiles/pmc_run.sh with the SAME workload arguments), or None.  bench.py cannot run rocprofv3 on itself."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        a = d.get("_bench_args", [])
        def opt(name, default):
            return int(a[a.index(name) + 1]) if name in a else default
        same = (-(-opt("--batch-rows", 1_048_576) // -(-opt("--batch-rows", 1_048_576) // 262_144)) == -(-args.batch_rows // -(-args.batch_rows // 262_144)) and opt("--factors", 16) == args.factors and opt("--features", 1_000_000) == args.features
                and opt("--rows", 10_000_000) == args.rows and opt("--nnz", 30) == args.nnz and ("ftrl" in a) == (args.solver == "ftrl")
                and ("--state-fp64" in a) == bool(args.state_fp64))
        if same and kernel in d and "traffic_bytes_per_launch" in d[kernel]:
            best = (d[kernel]["traffic_bytes_per_launch"], os.path.basename(f))
    return best


def host_cpu_share():
    """Cores this process may really use: affinity mask, cgroup quota, and at most 16 (a one-GPU box's CPU share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(m, args, v0):
    """Oracle (reference-order serial SGD, one core) on the first cpu_rows rows of the same matrix."""
    import oracle
    n = min(args.cpu_rows, m.n)
    rp, col, val, y = m.export(0, n)
    X = oracle.Matrix(rp, col, val, args.features)
    P = oracle.params(task=oracle.CLASSIFICATION, k=args.factors, l2_regw=1e-4, l2_regv=1e-4, learn_rate=0.01)
    w = np.zeros(args.features)
    v = np.ascontiguousarray(v0.astype(np.float64))  # [k][p] factor-major, the reference's layout
    oracle.lib()
    t0 = time.perf_counter()
    done = oracle.sgd_pass(P, X, y, 0.0, w, v.ravel())
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "examples/s", "cores": 1, "kind": "port",
           "sample": f"rows 1..{n - 1} of the same matrix, one reference-order serial pass ({dt:.1f} s)"}
    # SURVEY 8(d) item ii, reported beside it: what all host cores give.  The reference has no parallel training, so the
    # yardstick is its example step run lock-free over row ranges (Hogwild); its forward is OpenMP over rows as shipped.
    threads = min(oracle.omp_threads(), host_cpu_share())
    w = np.zeros(args.features)
    v = np.ascontiguousarray(v0.astype(np.float64))
    t0 = time.perf_counter()
    done = oracle.omp_sgd_hogwild(P, X, y, 0.0, w, v.ravel(), threads)
    dt_h = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.omp_predict_batch(P, X, 0.0, w, v.ravel(), threads)
    dt_f = time.perf_counter() - t0
    out["all_cores"] = {"cores": threads, "hogwild_examples_per_s": done / dt_h, "forward_rows_per_s": n / dt_f,
                        "sample": f"rows 0..{n - 1}, one lock-free OpenMP pass ({dt_h:.1f} s) and one OpenMP forward ({dt_f:.1f} s)"}
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 through torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from fmwr_amd import _lib as L
    from fmwr_amd import engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # FMX_BENCH_SHARED_DEVICE=1: every rank uses device 0 (rehearsal of the N > 1 path on a one-GPU box, backend gloo)
    if os.environ.get("FMX_BENCH_SHARED_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    z, k, p = args.nnz, args.factors, args.features
    from fmwr_amd.distributed import DataParallel, EngineStepper, shard_rows
    r0, r1 = shard_rows(args.rows, rank, world)
    n_local = r1 - r0
    B = min(args.batch_rows, args.rows // world)
    m = engine.Matrix.synthetic(n_local, p, z, args.seed, row_offset=r0, device=local_rank)
    solver = L.SOLVER_SGD if args.solver == "sgd" else L.SOLVER_FTRL
    e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=solver, num_factor=k, learn_rate=0.01, l2_w1=1e-4, l2_v=1e-4,
                      l1_w1=1e-4 if args.solver == "ftrl" else 0.0, l1_v=1e-4 if args.solver == "ftrl" else 0.0,
                      mode=L.MODE_MINIBATCH, batch_rows=B, tile_rows=args.tile_rows, device=local_rank, keep_w1=0 if args.no_linear else 1,
                      state_fp64=int(args.state_fp64), exchange_chunks=(args.exchange_chunks or (8 if world == 2 else 4)) if world > 1 else 0)
    v0 = np.random.default_rng(args.seed).normal(0.0, 0.01, (k, p)).astype(np.float32)  # same V0 on every replica
    e.set_params(0.0, None, v0.astype(np.float64))
    nb_full = max(1, n_local // B)  # ragged tail batch left out so every step does the same work
    e.num_batches(m)              # builds the per-batch CSC (ingest, not timed)

    dp = DataParallel(EngineStepper(e, m, local_rank)) if world > 1 else None

    def one_step(i):
        b = i % nb_full
        if world == 1:
            e.step(m, b)      # fused: forward -> w0 step -> gradient sums + update
        else:
            dp.step(b)        # forward -> per feature block: gradient sums -> RCCL all-reduce (async) -> update

    def fence():
        e.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    fence()
    e.profile_reset()
    # HIP events around every 16th launch of each kernel, on the engine's stream, inside the timed region (an event pair is a
    # serialisation point on the stream: timing every launch costs ~5 % of the throughput it is there to explain)
    e.profile(16)
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    e.profile(0)
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # forward-only rate (Model::predict_batch + logistic link over the rank's rows, output left on the device)
    fwd_rate = None
    if world == 1:
        out_dev = torch.empty(n_local, dtype=torch.float64, device=torch.device("cuda", local_rank))
        import ctypes as C
        def forward_all():
            L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(n_local), C.c_void_p(out_dev.data_ptr()), C.c_int(L.LINK_LOGISTIC)))
        forward_all(); e.sync()
        t1 = time.perf_counter()
        for _ in range(3):
            forward_all()
        e.sync()
        fwd_rate = 3 * n_local / (time.perf_counter() - t1)

    w0, _, vv = e.get_params()
    if not (np.isfinite(w0) and np.all(np.isfinite(vv))):
        raise SystemExit("non-finite parameters after the timed region")

    if rank == 0:
        rows_step = B * world
        value = rows_step * args.steps / dt
        fwd_ms, fwd_n = e.profile_get(L.KERNEL_ROWS_FORWARD)
        upd_ms, upd_n = e.profile_get(L.KERNEL_COLS_UPDATE)
        tiles = -(-B // (args.tile_rows or (524_288 if k > 32 else 262_144)))  # fmx_api.hip effective_tile_rows()
        tile_rows = -(-B // tiles)
        eb = 8 if args.state_fp64 else 4
        b_fwd, b_upd, _ = algorithmic_bytes(z, k, p, tile_rows, eb)   # per LAUNCH: one tile
        b_step = algorithmic_bytes(z, k, p, B, eb)[2]
        kernels = {
            "fm_rows_forward": (b_fwd, fwd_ms / max(fwd_n, 1)),
            "fm_cols_update": (b_upd, upd_ms / max(upd_n, 1)),
        }
        # N > 1: phase 2 runs as one launch per (feature block, tile) plus the per-block updates, so only phase 1 keeps the
        # one-launch-per-tile byte count the roofline line is defined on
        dom = max(kernels, key=lambda name: kernels[name][1]) if world == 1 else "fm_rows_forward"
        dbytes, dms = kernels[dom]
        achieved = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
        traffic = pmc_traffic(dom, args) if world == 1 else None
        out = {
            "metric": "training examples/sec, 10Mx1M sparse FM SGD", "value": value, "unit": "examples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.state_fp64 else "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {args.rows}x{p}, {z} nnz/row, k={k}, {args.solver.upper()} mini-batch "
                                   f"(BASELINE.json configs[{1 if args.solver == 'sgd' else 2}])",
                       "batch_rows_per_gpu": B, "tile_rows": tile_rows, "global_batch_rows": rows_step, "rows_per_gpu": n_local,
                       "state": ("fp64" if args.state_fp64 else "fp32") + " V[p][k] + w[p], fp64 accumulation", "parallelism": f"dp{world}",
                       **({"exchange": f"all-reduce(sum) of {e.grad_buffer()[1] * e.grad_elem_bytes() / 1e6:.1f} MB per step in {e.grad_layout()[0]} pipelined blocks"} if world > 1 else {})},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                         "traffic_source": (f"profiles/{traffic[1]}: (FETCH_SIZE*2 + WRITE_SIZE) KiB per launch, separate --pmc passes; "
                                            "upper bound, see DESIGN.md section 6") if traffic else None,
                         "algorithmic_bytes_per_launch": dbytes, "avg_launch_ms": dms,
                         "kernels_ms": {name: kv[1] for name, kv in kernels.items()},
                         "step_algorithmic_GBps": b_step / (dt / args.steps) / 1e9 / world},
        }
        if fwd_rate is not None:
            out["forward_rows_per_s"] = fwd_rate
        if world == 1 and args.cpu_rows > 0:
            out["cpu_baseline"] = cpu_baseline(m, args, v0)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
