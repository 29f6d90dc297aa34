#!/usr/bin/env python3
"""bench.py -- examples/sec of the FM training hot path on the BASELINE.json workload.

    python bench.py --gpus N --steps K --warmup W            # configs[1]: 10M x 1M, 30 nnz/row, k = 16, SGD
    python bench.py --solver ftrl --factors 64               # configs[2]: same matrix, k = 64, FTRL (l1 + l2)

A "step" is one synchronous mini-batch step (forward + gradient sums + update) over `--batch-rows` rows PER GPU of the
synthetic matrix; the matrix is generated on the device and is resident in HBM before the timed region.  The engine cuts
a step into tiles (two kernel launches per tile: fm_rows_forward, fm_cols_update) with the parameters frozen across the
tiles.  N > 1: one process per GPU (torch.distributed / RCCL), each rank owns a contiguous row range, computes its gradient
sums, exchanges them (dense: the (k+2)*p buffer is all-reduced in pipelined blocks; compact: records of the occurring
features are all-gathered), every replica applies the same update ("scaling": "weak": per-GPU rows per step are fixed).

Rank 0's LAST stdout line is the ONE JSON line the driver parses (driver_line(): at most LINE_LIMIT = 4000 bytes -- metric, config, roofline, cpu_baseline and a
flat summary of the side runs) and the ONLY stdout line; everything else the run measured goes to stderr as lines prefixed `DETAILS ` (one key per line, long values in
numbered pieces: no line over 4 000 bytes) and is left in bench_details.json.
`value` is whole-job examples/s of the timed steps.  `roofline` prices the step in SURVEY
8(d)'s algorithmic bytes (the headline `frac`) and each of the two kernels on its own bytes and HIP-event time, next to the
measured ceiling of the access pattern (`ceiling_frac`).  At N == 1 the same line also carries what `value` leaves out:
`end_to_end` (with the one-off per-tile CSC build), `value_fp64_state`, `small_batch`, `sequential_exact` (the mode the
reference-parity claim is made in) and the oracle timed on the host (`cpu_baseline`).
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows of the whole synthetic matrix")
    ap.add_argument("--features", type=int, default=1_000_000)
    ap.add_argument("--nnz", type=int, default=30)
    ap.add_argument("--factors", type=int, default=0, help="0: 16 for sgd (configs[1]), 64 for ftrl (configs[2])")
    ap.add_argument("--batch-rows", type=int, default=0,
                    help="mini-batch rows per GPU per step (processed in cache-resident tiles).  0: SGD on one GPU 262144 -- the largest step "
                         "that learns per example like a 4096-row one at the reference's learning rate on this workload (a coordinate then "
                         "occurs ~8 times per step; profiles/r02_learning_*.txt) -- and 1048576 per GPU for FTRL and for N > 1, where the step "
                         "must be long enough to hide the exchange of the 72 MB buffer; the SGD learning rate then follows the global batch linearly "
                         "(learn_rate_for: profiles/r04_learning_scaling.txt shows the held-out loss per example kept up to 8.4 M rows per step), and the one-GPU "
                         "line carries `scaling_reference`, ONE GPU at that step size")
    ap.add_argument("--tile-rows", type=int, default=0, help="rows per tile (0: the engine's default, 524288; 262144 for k <= 8)")
    ap.add_argument("--solver", choices=["sgd", "ftrl", "als", "mcmc"], default="sgd",
                    help="als / mcmc: BASELINE.json configs[4], the V-column sweep of MCMC_ALS_Learner::update_v over the same matrix (k = 16); a step is "
                         "one sweep of all k factors over all rows (mcmc: the Gibbs draw with device-resident standard normals; als: the mean)")
    ap.add_argument("--workload", choices=["uniform", "criteo"], default="uniform",
                    help="uniform: BASELINE.json configs[1] / [2] (one column per stratum).  criteo: configs[3]'s shape -- 33 M features, 39 nnz/row "
                         "(13 dense + 26 categorical fields with power-law heads), k = 32, 262144-row steps of one sparse tile; --rows rows resident "
                         "per GPU (default 8 M); N > 1 exchanges the occurring features' records (compact exchange)")
    ap.add_argument("--columns", choices=["iid", "stratified", "zipf"], default="iid",
                    help="--workload uniform: the column law of the synthetic rows.  iid: SURVEY 8(d)'s primary law, columns i.i.d. uniform over [0, p), sorted inside the "
                         "row (fmx_matrix_synthetic_iid) -- the headline; stratified: one column per stratum of [0, p) (fmx_matrix_synthetic: what rounds 1-4 quoted `value` on, "
                         "kept in the line as `value_stratified_columns`); zipf: SURVEY 8(d)'s conflict-stress variant, exponent 1.05")
    ap.add_argument("--sweep-iid", action="store_true",
                    help="--solver als / mcmc: the V sweep on SURVEY 8(d)'s i.i.d. uniform columns in the feature-major COLOURED order (cfg.als_max_levels = -2); default: one column per stratum")
    ap.add_argument("--sweep-feature-major", action="store_true", help="--solver als / mcmc on the default one-column-per-stratum matrix: cfg.als_max_levels = -2 (the exact schedule's levels as colours, all k factors of a feature stepped together)")
    ap.add_argument("--sweep-exact", action="store_true", help="--sweep-iid: the reference's own index order on the i.i.d. law (cfg.als_max_levels = 0: the exact schedule, ~19 400 dependent levels at 10 M x 1 M) -- "
                    "configs[4] on SURVEY 8(d)'s law with the reference's numbers")
    ap.add_argument("--sweep-factor-outer", action="store_true", help="--sweep-iid: the coloured order with the reference's factor-outer nesting (cfg.als_max_levels = -1)")
    ap.add_argument("--real-values", action="store_true", help="SURVEY 8(d)'s value variant: val ~ U(0,1) instead of 1 (fmx_matrix_synthetic_values): the kernels then read the value arrays")
    ap.add_argument("--seed", type=int, default=20240001)
    ap.add_argument("--state-fp64", action="store_true", help="experiment: fp64 parameter/optimizer state (default fp32)")
    ap.add_argument("--no-linear", action="store_true", help="experiment: keep.w1 = FALSE (no w gathers)")
    ap.add_argument("--exchange", choices=["auto", "dense", "compact", "owner"], default="auto",
                    help="N > 1: dense = all-reduce of the (k+2)*p buffer; compact = all-gather of the occurring features' records "
                         "(steps of one sparse tile); owner = the records go to the rank that owns the feature (id mod N), which updates its slice and "
                         "hands current rows to whoever reads them next (SURVEY 8(e)(ii)); auto = compact where usable, owner with --stream")
    ap.add_argument("--stream", action="store_true",
                    help="--workload criteo: every step's rows are generated and planned on the fly (fmx_source; BASELINE.json configs[3]: the 4e9-row matrix never "
                         "exists), rank r streaming its own row range; N > 1 exchanges per step (--exchange owner | compact)")
    ap.add_argument("--exchange-chunks", type=int, default=0,
                    help="N > 1, dense: blocks of features the exchange is pipelined in (1: one all-reduce of the whole buffer per step; "
                         "0: 8 blocks on 2 GPUs, where the single xGMI link is the bound and finer blocks hide more of it, 4 otherwise, "
                         "where the extra launches of finer blocks cost more than they hide: profiles/r01_split_bench.json)")
    ap.add_argument("--in-library", action="store_true",
                    help="drive the N GPUs from ONE process through cfg.n_gpus (fm_group.hip: N replicas behind one handle, RCCL by dlopen) instead of one "
                         "process per GPU: `python bench.py --gpus N --in-library`, no torch.distributed.run.  N = 1 is fmx_train's own loop and must equal the default line")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl == RCCL; gloo for rehearsals)")
    ap.add_argument("--cpu-rows", type=int, default=-1, help="rows of the CPU-baseline sample (0: skip; -1: 5M for sgd, 250K for ftrl k=64: 10-20 s of one core either way)")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (fp64 state, small batches, sequential mode, ceilings)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="the default line (configs[1], one GPU) also carries `other_configs`: short runs of configs[2], configs[3] (rows resident and streamed) and configs[4] "
                         "in the same process, each with its value, kernel fractions and a small CPU sample; this flag leaves them out")
    a = ap.parse_args(argv)
    if a.workload == "criteo":
        a.features, a.nnz = 33_000_000, 39
        a.factors = a.factors or 32
        a.batch_rows = a.batch_rows or 262_144
        if a.rows == 10_000_000:
            a.rows = 8_000_000 * a.gpus
        a.no_extras = True
        if a.cpu_rows < 0:
            a.cpu_rows = 0
    if a.solver in ("als", "mcmc"):
        a.factors = a.factors or 16
        if a.steps == 40 and a.warmup == 5:   # the defaults are sized for 0.3 ms steps; a sweep of 16 factors over 3e8 entries is ~0.2 s
            a.steps, a.warmup = 5, 1
        if a.cpu_rows < 0:
            a.cpu_rows = 4_000_000   # ~10-20 s of one core
        return a
    if a.batch_rows == 0:
        # SGD, one GPU: 262144 (a coordinate occurs ~8 times per step: learns per example like 4096-row steps at the reference's
        # learning rate).  FTRL: 1048576 -- its per-coordinate adaptive step keeps learning at ~100 occurrences per step (the fastest
        # configuration of profiles/r02_learning_ftrl.txt), and the three-table sweep per tile wants the larger 524288-row tiles.
        a.batch_rows = 262_144 if (a.gpus == 1 and a.solver == "sgd") else 1_048_576
    if a.factors == 0:
        a.factors = 16 if a.solver == "sgd" else 64
    if a.cpu_rows < 0:
        a.cpu_rows = 5_000_000 if a.solver == "sgd" else max(20_000, 16_000_000 // a.factors)
    return a


def algorithmic_bytes(z, k, p, rows, e=4, ftrl=False, unit=False):
    """Per-launch algorithmic HBM bytes of the two hot kernels and of the SURVEY 8(d) fused step; e = bytes per state
    element (4: the fp32 state SURVEY 8(d) prices; 8 with --state-fp64).  FTRL keeps three tables per parameter (theta, z, n).
    unit: every stored value is 1.0f -- the kernels never read the value arrays (RowsArgs.unit / ColsArgs.unit), so the 4-byte value
    stream is NOT priced: neither in the two kernels' own bytes nor in the survey's fused step (its 8 B of (idx, val) per nonzero become 4)."""
    t = 3 if ftrl else 1
    vb = 0 if unit else 4
    rows_fwd = rows * (z * (4 + vb + e + e * k) + 8 + 4 + e * k + e)                  # idx,val,w,V row | row_ptr, y, S row, mult
    cols_upd = rows * z * (4 + vb + e + e * k) + p * (4 + t * 2 * e * k + t * 2 * e)  # row,val,mult,S row | offsets, table RMWs
    survey_step = rows * (z * (4 + vb + t * 2 * e + t * 2 * e * k) + 12)              # SGD z(16+8k)+12, FTRL z(32+24k)+12 at e = 4, values read
    return rows_fwd, cols_upd, survey_step


def tile_kp(k):
    kp = 4
    while kp < k:
        kp *= 2
    return kp


def effective_tile(B, k, tile_rows):
    want = tile_rows or (524_288 if k > 8 else 262_144)  # fmx_api.hip effective_tile_rows(): kp32 >= 16
    tiles = -(-B // want)
    return -(-B // tiles)


def pmc_traffic(kernel, args):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summaries (profiles/r*_pmc_summary*.json,
    made by profiles/pmc_run.sh with the SAME workload arguments), or None.  bench.py cannot run rocprofv3 on itself."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        a = d.get("_bench_args", [])
        def opt(name, default):
            return int(a[a.index(name) + 1]) if name in a else default
        solver = "ftrl" if "ftrl" in a else "sgd"
        crit = "--workload" in a and a[a.index("--workload") + 1] == "criteo"   # (parse() resolves the Criteo shape's own defaults)
        k = opt("--factors", 32 if crit else (16 if solver == "sgd" else 64))
        same = (effective_tile(opt("--batch-rows", 262_144 if (solver == "sgd" or crit) else 1_048_576), k, opt("--tile-rows", 0)) == effective_tile(args.batch_rows, args.factors, args.tile_rows)
                and k == args.factors and (33_000_000 if crit else opt("--features", 1_000_000)) == args.features and opt("--rows", 8_000_000 if crit else 10_000_000) == args.rows
                and (39 if crit else opt("--nnz", 30)) == args.nnz and solver == args.solver and ("--state-fp64" in a) == bool(args.state_fp64)
                and (a[a.index("--workload") + 1] if "--workload" in a else "uniform") == args.workload
                # summaries made before --columns existed (rounds 1-4) ran the stratified generator; from round 5 on the default law is i.i.d. uniform
                and (crit or (a[a.index("--columns") + 1] if "--columns" in a else ("stratified" if os.path.basename(f)[:3] < "r05" else "iid")) == args.columns)
                and ("--real-values" in a) == bool(args.real_values))
        if same and kernel in d and "traffic_bytes_per_launch" in d[kernel]:
            best = (d[kernel]["traffic_bytes_per_launch"], os.path.basename(f), d[kernel].get("fabric_bytes_per_launch"), d[kernel].get("fabric_reads_128B_frac"))
    return best


def pmc_traffic_sweep(args, tiled, ordered=False, blocks=False, fmajor=False):
    """configs[4]: HBM-side bytes of one level of one factor from the committed PMC summaries (profiles/r*_pmc_summary_mcmc*.json), or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary_mcmc*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        a = d.get("_bench_args", [])
        if ("als" in a) != (args.solver == "als") or args.rows != 10_000_000 or args.features != 1_000_000:
            continue
        if any((flag in a) != bool(getattr(args, flag[2:].replace("-", "_"))) for flag in ("--sweep-iid", "--sweep-feature-major", "--sweep-factor-outer", "--sweep-exact", "--real-values")):
            continue                                       # (another column law, nesting or value law: another kernel's counters)
        if fmajor:
            if "als_level_allf" in d and "fabric_bytes_per_launch" in d["als_level_allf"]:
                best = (d["als_level_allf"]["fabric_bytes_per_launch"], os.path.basename(f))
            continue
        if blocks:
            if "als_block_level" in d and "fabric_bytes_per_launch" in d["als_block_level"]:   # (one kernel per level; exact bytes by request size, not the x2 bound)
                best = (d["als_block_level"]["fabric_bytes_per_launch"], os.path.basename(f))
            continue
        names = ("als_order_sums", "als_order_apply") if ordered else (("als_tile_sums", "als_tile_step", "als_rows_apply") if tiled else ("als_level",))
        if all(nm in d and "traffic_bytes_per_launch" in d[nm] for nm in names) and (("als_level" in d) != tiled) and (("als_order_sums" in d) == ordered) and "als_block_level" not in d:
            best = (sum(d[nm]["traffic_bytes_per_launch"] for nm in names), os.path.basename(f))
    return best


def e_unit(m):
    """every stored value of the matrix is 1.0 (one-hot rows: the kernels never read the value arrays)"""
    _, _, val, _ = m.export(0, min(m.n, 1024))
    return bool(np.all(val == 1.0))


def shape_tag(rows, features):
    """'10Mx1M' for the configs' own shape, the actual dimensions otherwise (rehearsals and sweeps run smaller matrices through the same code)"""
    def short(v):
        for unit, name in ((1_000_000_000, "B"), (1_000_000, "M"), (1_000, "K")):
            if v >= unit and v % unit == 0:
                return f"{v // unit}{name}"
        return str(v)
    return f"{short(rows)}x{short(features)}"


def cpu_model():
    """model string of the host CPU the cpu_baseline legs ran on ("state core count and CPU model")"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine()


COLUMN_LAWS = {"iid": "columns i.i.d. uniform over [0, p), sorted inside the row (SURVEY 8(d)'s primary law; fmx_matrix_synthetic_iid)",
               "stratified": "one column per stratum of [0, p) (fmx_matrix_synthetic)",
               "zipf": "columns i.i.d. Zipf s = 1.05 over [0, p), sorted inside the row (SURVEY 8(d)'s conflict-stress variant)"}


def make_matrix(engine, L, args, n, row_offset=0, device=0, columns=None, real_values=None):
    """rows [row_offset, row_offset + n) of the uniform workload's stream under a column law / value law (defaults: the run's own)"""
    columns = columns or args.columns
    if columns == "stratified":
        m = engine.Matrix.synthetic(n, args.features, args.nnz, args.seed, row_offset=row_offset, device=device)
    else:
        m = engine.Matrix.synthetic_iid(n, args.features, args.nnz, args.seed, law=L.COLUMNS_ZIPF if columns == "zipf" else L.COLUMNS_UNIFORM, zipf_s=1.05,
                                        row_offset=row_offset, device=device)
    if args.real_values if real_values is None else real_values:
        m.synthetic_values(args.seed, row_offset)
    return m


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
    """Oracle (the reference's serial learner, one core) on the first cpu_rows rows of the same matrix."""
    import oracle
    n = min(args.cpu_rows, m.n)
    rp, col, val, y = m.export(0, n)
    p_cpu, remap_note = args.features, ""
    if args.features > 4_000_000:
        # configs[3]'s shape: the reference's dense k x p fp64 table would be 8.4 GB of host memory for a sample that touches a fraction of it.  The
        # sample's columns are renumbered onto the features that occur in it (order kept: the same work per example, a smaller table -- friendlier to the
        # CPU's caches than the real one)
        uniq, inv = np.unique(col, return_inverse=True)
        col, p_cpu = inv.astype(np.uint32), int(len(uniq))
        remap_note = f"; columns renumbered onto the {p_cpu} features the sample touches"
        v0 = np.random.default_rng(args.seed).normal(0.0, 0.01, (args.factors, p_cpu)).astype(np.float32)
    X = oracle.Matrix(rp, col, val, p_cpu)
    if args.solver in ("als", "mcmc"):   # configs[4]: one reference-order update_v sweep (MCMC_ALS_Learner.h:272-354) over a row sample
        k, p, gibbs = args.factors, args.features, args.solver == "mcmc"
        vv = np.random.default_rng(args.seed).normal(0.0, 0.01, (k, p))
        err = np.random.default_rng(args.seed + 1).normal(0.0, 1.0, n)
        zz = np.random.default_rng(args.seed + 2).normal(0.0, 1.0, k * p) if gibbs else None
        oracle.lib()
        t0 = time.perf_counter()
        oracle.als_update_v(k, X, vv.ravel(), err, alpha=1.0, v_lambda=np.full(k, 1.0) if gibbs else None, znorm=zz)
        dt = time.perf_counter() - t0
        return {"value": n / dt, "unit": "examples/s", "cores": 1, "cpu_model": cpu_model(), "kind": "port",
                "sample": f"one reference-order update_v sweep (k = {k}{', Gibbs draws' if gibbs else ''}) over rows 0..{n - 1} of the same matrix ({dt:.1f} s incl. its transpose)"}
    if args.solver == "sgd":
        P = oracle.params(task=oracle.CLASSIFICATION, k=args.factors, l2_regw=1e-4, l2_regv=1e-4, learn_rate=0.01)
    else:
        P = oracle.params(task=oracle.CLASSIFICATION, k=args.factors, l1_regw=1e-4, l1_regv=1e-4, l2_regw=1e-4, l2_regv=1e-4)
    w = np.zeros(p_cpu)
    v = np.ascontiguousarray(v0.astype(np.float64))  # [k][p] factor-major, the reference's layout
    oracle.lib()
    t0 = time.perf_counter()
    if args.solver == "sgd":
        done = oracle.sgd_pass(P, X, y, 0.0, w, v.ravel())
    else:
        done = oracle.ftrl_learn(P, X, y, 0.0, w, v.ravel(), n - 1)["iters"]
    dt = time.perf_counter() - t0
    out = {"value": done / dt, "unit": "examples/s", "cores": 1, "cpu_model": cpu_model(), "kind": "port",
           "sample": f"rows 1..{n - 1} of the same matrix, one reference-order serial {args.solver.upper()} pass ({dt:.1f} s){remap_note}"}
    if args.solver != "sgd" or remap_note or getattr(args, "cpu_one_core_only", False):
        return out
    # SURVEY 8(d) item ii, reported beside it: what all host cores give.  The reference has no parallel training, so the
    # yardstick is its example step run lock-free over row ranges (Hogwild); its forward is OpenMP over rows as shipped.
    threads = min(oracle.omp_threads(), host_cpu_share())
    w = np.zeros(p_cpu)
    v = np.ascontiguousarray(v0.astype(np.float64))
    t0 = time.perf_counter()
    done = oracle.omp_sgd_hogwild(P, X, y, 0.0, w, v.ravel(), threads)
    dt_h = time.perf_counter() - t0
    t0 = time.perf_counter()
    oracle.omp_predict_batch(P, X, 0.0, w, v.ravel(), threads)
    dt_f = time.perf_counter() - t0
    out["all_cores"] = {"cores": threads, "cpu_model": cpu_model(), "hogwild_examples_per_s": done / dt_h, "forward_rows_per_s": n / dt_f,
                        "sample": f"rows 0..{n - 1}, one lock-free OpenMP pass ({dt_h:.1f} s) and one OpenMP forward ({dt_f:.1f} s)"}
    return out


LR_BASE, LR_BASE_ROWS = 0.01, 262_144
PREWARM_S = 0.15   # seconds of untimed steps before the warm-up steps (run_minibatch)


def learn_rate_for(global_rows):
    """SGD with the MEAN gradient per coordinate per step: the reference's learning rate (0.01) holds up to 262 144 rows per step; beyond, the step is scaled
    LINEARLY with the global batch -- profiles/r04_learning_scaling.txt: held-out log-likelihood after three passes -0.633 at 262 144 rows / lr 0.01 and
    -0.633 / -0.635 / -0.640 / -0.640 / -0.641 at 2x / 4x / 8x / 16x / 32x the rows with 2x ... 32x the rate (unscaled: -0.658 ... -0.691; sqrt scaling: -0.647 ... -0.681)."""
    return LR_BASE * max(1.0, global_rows / LR_BASE_ROWS)


def engine_kwargs(args, L, B, local_rank, world, **over):
    ftrl = args.solver == "ftrl"
    kw = dict(task=L.TASK_CLASSIFICATION, solver=L.SOLVER_FTRL if ftrl else L.SOLVER_SGD, num_factor=args.factors, learn_rate=learn_rate_for(B * world), l2_w1=1e-4, l2_v=1e-4,
              l1_w1=1e-4 if ftrl else 0.0, l1_v=1e-4 if ftrl else 0.0, mode=L.MODE_MINIBATCH, batch_rows=B, tile_rows=args.tile_rows, device=local_rank,
              keep_w1=0 if args.no_linear else 1, state_fp64=int(args.state_fp64),
              exchange_chunks=(args.exchange_chunks or (8 if world == 2 else 4)) if world > 1 else 0)
    kw.update(over)
    return kw


def timed_steps(e, m, nb, steps, warmup):
    for i in range(warmup):
        e.step(m, i % nb)
    e.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        e.step(m, (warmup + i) % nb)
    e.sync()
    return time.perf_counter() - t0


def gather_ceilings(args, engine, kernels, tile_rows, v_row_elems=None, sparse_lists=0, skewed_matrix=None):
    """The access pattern's own ceiling at each kernel's table, measured in this run (fmx_measure_gather: uniformly random rows, ids
    generated in registers): rows of the V table for phase 1 (in the w-in-row layout a row is the whole 2 * kp-float line), rows of the
    tile's S table for phase 2.  `ceiling_frac` = the kernel's row gathers per second over that rate."""
    z, k, p = args.nnz, args.factors, args.features
    eb = 8 if args.state_fp64 else 4
    kp = 4
    while kp < k:
        kp *= 2
    ceil = {}
    for name, elems, table_rows in (("fm_rows_forward", v_row_elems or kp, p), ("fm_cols_update", kp, tile_rows)):
        row_bytes = min(256, max(16, elems * eb))
        lines = max((elems * eb) // row_bytes, 1)  # rows wider than 256 B are fetched as several 256-B pieces
        r = engine.measure_gather(table_rows * elems * eb, row_bytes, n_groups=tile_rows, per_group=32, in_flight=4, reps=20, device=0) / lines
        got = tile_rows * z / (kernels[name][1] * 1e-3) if kernels[name][1] > 0 else 0.0
        ceil[name] = {"table_MB": table_rows * elems * eb / 1e6, "row_bytes": elems * eb, "ceiling_rows_per_s": r, "kernel_rows_per_s": got, "ceiling_frac": got / r if r else None}
        if name == "fm_rows_forward" and skewed_matrix is not None and lines == 1:
            # skewed columns: most V-row fetches are served on-die, and uniformly random ids are no ceiling for them.  The ceiling is the bare
            # gather of the rows THIS matrix names (fmx_measure_gather_matrix: the tile's own column ids, a table of the same size, no arithmetic)
            rm = max(engine.measure_gather_matrix(skewed_matrix, 0, min(tile_rows, skewed_matrix.n), table_rows, row_bytes, in_flight=u, reps=20) for u in (4, 8))
            ceil[name].update({"uniform_ids_ceiling_rows_per_s": r, "ceiling_rows_per_s": rm, "ceiling_frac": got / rm if rm else None,
                               "ceiling_source": "bare gather of the first tile's own column ids (4 and 8 in flight, the faster)"})
    if sparse_lists > 0 and kernels["fm_cols_update"][1] > 0:
        # A sparse tile's phase 2 is not only S-row gathers: per occurring feature it reads one random row of the V table and writes it back.
        # Its floor is the sum of its parts at their own measured random-row rates (writes priced like reads): the S gathers from the tile's
        # table, two random V-table rows per list.
        rv, rs = ceil["fm_rows_forward"]["ceiling_rows_per_s"], ceil["fm_cols_update"]["ceiling_rows_per_s"]
        floor_ms = (tile_rows * z / rs + 2.0 * sparse_lists / rv) * 1e3
        c = ceil["fm_cols_update"]
        c["s_row_gather_frac"] = c["ceiling_frac"]
        c["parts_floor_ms"] = floor_ms
        c["ceiling_frac"] = floor_ms / kernels["fm_cols_update"][1]
        c["note"] = (f"sparse tile: {sparse_lists} lists; floor = entries / (S-table random-row rate) + 2 x lists / (V-table random-row rate), both measured here; "
                     "ceiling_frac = floor / measured launch time")
    return ceil


def kernel_ms(e, L):
    return {name: ms / max(cnt, 1) for name, (ms, cnt) in (("fm_rows_forward", e.profile_get(L.KERNEL_ROWS_FORWARD)), ("fm_cols_update", e.profile_get(L.KERNEL_COLS_UPDATE)))}


def timed_variant(args, L, engine, m, v0, B, steps, **over):
    """`steps` timed steps of the run's engine configuration on another matrix (schedule trials and warmup outside): value, HIP-event kernel times"""
    e = engine.Engine(m.p, **engine_kwargs(args, L, B, 0, 1, **over))
    e.set_params(0.0, None, v0.astype(np.float64))
    nb = max(1, m.n // B)
    timed_steps(e, m, nb, 16, 2)   # schedule trials
    e.profile_reset(); e.profile(5)
    dt = timed_steps(e, m, nb, steps, 2)
    e.profile(0)
    km = kernel_ms(e, L)
    e.close()
    return B * steps / dt, km


def time_to_quality(args, L, engine, v0):
    """One line that ties throughput to learning (core/Evaluation.h:80-89's log-likelihood, held out): labels planted from a hidden FM of the workload's own
    shape, every learner from the same start, one pass each; what the examples/s of each mode buy in loss."""
    n, p, z, k = args.rows, args.features, args.nnz, args.factors
    n_test = 1_000_000
    rng = np.random.default_rng(11)
    train = make_matrix(engine, L, args, n, 0)
    test = make_matrix(engine, L, args, n_test, n)
    pe = engine.Engine(p, num_factor=k, mode=L.MODE_MINIBATCH)
    pe.set_params(0.1, rng.normal(0, 0.35, p), rng.normal(0, 0.12, (k, p)))

    def plant(m):
        prob = 1.0 / (1.0 + np.exp(-pe.predict(m)))
        y = np.where(rng.random(m.n) < prob, 1.0, -1.0).astype(np.float32)
        m.set_labels(y)
        return float(np.mean(np.where(y > 0, np.log(prob + 1e-20), np.log(1 - prob + 1e-20))))

    plant(train)
    ll_star = plant(test)
    pe.close()
    out = {"planted_model_held_out_ll": ll_star, "coin_flip_ll": float(-np.log(2.0)), "train_rows": n, "held_out_rows": n_test,
           "note": "labels planted from a hidden FM (w ~ N(0, 0.35), V ~ N(0, 0.12), w0 = 0.1) on the workload's own shape; every learner starts from the bench's V0, "
                   "lr 0.01, L2 1e-5.  The reference-order learner (FMX_MODE_SEQUENTIAL with cfg.seq_reassociate, the glue's default: fp64, one update per example, its visiting order) sets the TARGET: the held-out "
                   "log-likelihood per example (core/Evaluation.h:80-89) it reaches after `examples` examples in `wall_s` seconds.  Each mini-batch learner (one MEAN-gradient "
                   "step per coordinate per batch) then trains pass by pass until it reaches that target: examples and wall seconds to get there (plan build included), "
                   "held_out_ll after every pass.  More examples for the same loss, far fewer seconds",
           "learners": {}}

    def engine_for(**kw):
        e = engine.Engine(p, task=L.TASK_CLASSIFICATION, solver=L.SOLVER_SGD, num_factor=k, learn_rate=0.01, l2_w1=1e-5, l2_v=1e-5, **kw)
        e.set_params(0.0, None, v0.astype(np.float64))
        e.sync()
        return e

    e = engine_for(mode=L.MODE_SEQUENTIAL, seq_reassociate=1)   # the reference-order learner as the glue runs it by default (round 6: 4.0 M examples/s instead of 1.65 M)
    t0 = time.perf_counter()
    done = e.train(train, 2_000_000)
    e.sync()
    dt = time.perf_counter() - t0
    target = e.evaluate(test, L.EVAL_LL) / n_test
    e.close()
    out["target_held_out_ll"] = target
    out["learners"]["sequential_exact"] = {"examples": int(done), "wall_s": dt, "held_out_ll": target}
    for name, B in (("minibatch_262144", 262_144), ("minibatch_65536", 65_536), ("minibatch_16384", 16_384)):
        e = engine_for(mode=L.MODE_MINIBATCH, batch_rows=B)
        per_pass = (n // B) * B
        seen, wall, curve, reached = 0, 0.0, [], None
        for _ in range(12):
            t0 = time.perf_counter()
            seen += e.train(train, per_pass)
            e.sync()
            wall += time.perf_counter() - t0
            ll = e.evaluate(test, L.EVAL_LL) / n_test
            curve.append(ll)
            if ll >= target:
                reached = {"examples": int(seen), "wall_s": wall, "passes": len(curve)}
                break
        out["learners"][name] = {"reached_target": reached, "held_out_ll_after_each_pass": curve, "examples": int(seen), "wall_s": wall,
                                 **({"speedup_to_target_vs_sequential": dt / reached["wall_s"]} if reached else {})}
        e.close()
    train.close(); test.close()
    return out


def side_measurements(args, L, engine, m, v0, value, csc_build_s, kernels, tile_rows):
    """What `value` does not say; every figure is measured here, on this GPU, in this run."""
    out = {}
    z, k, p = args.nnz, args.factors, args.features
    n = m.n
    ftrl = args.solver == "ftrl"
    eb = 8 if args.state_fp64 else 4
    # MEASURED, not computed: a fresh matrix and a fresh engine, the clock around { per-tile plan build of the whole matrix + one full
    # pass over its rows } (the reference's default run is two passes, R/fm_train.R:92: the second pass costs n / value more)
    B = min(args.batch_rows, n)
    m2 = make_matrix(engine, L, args, n, 0)
    e2 = engine.Engine(p, **engine_kwargs(args, L, B, 0, 1))
    e2.set_params(0.0, None, v0.astype(np.float64))
    e2.sync()
    t0 = time.perf_counter()
    nb2 = e2.num_batches(m2)
    e2.sync()
    t_plan = time.perf_counter() - t0
    for b in range(nb2):
        e2.step(m2, b)
    e2.sync()
    t_all = time.perf_counter() - t0
    for b in range(nb2):           # the reference's default run is two passes (R/fm_train.R:92): the second one, on the same clock
        e2.step(m2, b)
    e2.sync()
    t_two = time.perf_counter() - t0
    out["end_to_end"] = {"measured": True, "plan_build_s": t_plan, "plan_and_one_pass_s": t_all, "plan_and_two_passes_s": t_two, "one_epoch_examples_per_s": n / t_all,
                         "two_epochs_examples_per_s": 2 * n / t_two, "first_plan_build_s_of_the_bench_matrix": csc_build_s,
                         "note": "wall clock around the per-tile plan build of the 10 M-row matrix (device sort, once per matrix) plus one pass, and plus TWO passes, over all its "
                                 "rows -- the last ragged step and phase 1's 14 schedule-trial launches included; every figure is a measured interval of one clock"}
    e2.close(); m2.close()
    # the other column law (rounds 1-4 quoted `value` on the stratified generator; SURVEY 8(d)'s primary law is i.i.d. uniform)
    other = "stratified" if args.columns != "stratified" else "iid"
    mo = make_matrix(engine, L, args, n, 0, columns=other)
    v_o, k_o = timed_variant(args, L, engine, mo, v0, B, args.steps)
    mo.close()
    out["value_stratified_columns" if other == "stratified" else "value_iid_uniform"] = {
        "value": v_o, "unit": "examples/s", "kernel_ms": k_o, "columns": COLUMN_LAWS[other],
        "note": "the same configuration on the other column law; HIP events around every 5th launch"}
    # SURVEY 8(d)'s value variant: val ~ U(0,1).  The kernels then read the value arrays in both phases (4 bytes per nonzero each) and multiply; priced on the full
    # survey bytes (z(16+8k)+12)
    mv = make_matrix(engine, L, args, n, 0, real_values=True)
    v_v, k_v = timed_variant(args, L, engine, mv, v0, B, args.steps)
    mv.close()
    bf, bu, bs = algorithmic_bytes(z, k, p, tile_rows, eb, ftrl, unit=False)
    out["value_real_values"] = {
        "value": v_v, "unit": "examples/s", "kernel_ms": k_v,
        "roofline": {"algorithmic_bytes_per_example": bs / tile_rows, "frac": v_v * bs / tile_rows / 1e9 / HBM_PEAK_GBS,
                     "kernels": {"fm_rows_forward": {"frac": bf / (k_v["fm_rows_forward"] * 1e-3) / 1e9 / HBM_PEAK_GBS if k_v["fm_rows_forward"] > 0 else None},
                                 "fm_cols_update": {"frac": bu / (k_v["fm_cols_update"] * 1e-3) / 1e9 / HBM_PEAK_GBS if k_v["fm_cols_update"] > 0 else None}}},
        "note": "the same rows with val ~ U(0,1) (fmx_matrix_synthetic_values; util/Smatrix.h:44-61: the reference's values are real floats): both kernels stream the "
                "value arrays and the plan is built by the general sort; every other bench matrix is one-hot, where they skip that stream"}
    # SURVEY 8(d)'s conflict-stress variant: Zipf s = 1.05 columns (a few features occur in most rows: phase 2's long lists, phase 1's re-read heads)
    if args.columns != "zipf":
        mz = make_matrix(engine, L, args, n, 0, columns="zipf")
        v_z, k_z = timed_variant(args, L, engine, mz, v0, B, args.steps)
        mz.close()
        out["value_zipf_columns"] = {"value": v_z, "unit": "examples/s", "kernel_ms": k_z, "columns": COLUMN_LAWS["zipf"],
                                     "note": "the same configuration on Zipf columns: the head features' lists take the long-list kernels (segments of 1024 entries on a side stream)"}
    # SURVEY 8(d)'s ragged variant: row lengths Poisson(z) clipped to [1, 64], i.i.d. uniform columns.  The kernels give every row a fixed lane group that walks
    # the row in rounds of RU entries (padded with x = 0): what that costs against rows of exactly z entries is the ratio of the two ENTRY rates
    mr = engine.Matrix.synthetic_ragged(n, p, float(z), args.seed)
    v_rag, rag_k = timed_variant(args, L, engine, mr, v0, B, args.steps)
    mean_len = mr.nnz / mr.n
    mr.close()
    if args.columns == "iid":
        v_iid, iid_k = value, {name: ms for name, (_, ms) in kernels.items()}
    else:
        v_iid, iid_k = (v_o, k_o) if other == "iid" else (None, None)
    if v_iid:
        out["value_ragged_rows"] = {"value": v_rag, "unit": "examples/s", "nnz_per_row": {"law": f"Poisson({z}) clipped to [1, 64]", "mean": mean_len},
                                    "entries_per_s": v_rag * mean_len, "fixed_length_entries_per_s": v_iid * z, "entry_rate_vs_fixed_length_rows": (v_rag * mean_len) / (v_iid * z),
                                    "kernel_ms": rag_k, "fixed_length_kernel_ms": iid_k,
                                    "note": "fmx_matrix_synthetic_ragged against fmx_matrix_synthetic_iid (rows of exactly z entries, same column law), same engine configuration; "
                                            "HIP events around every 5th launch"}
    # the like-for-like origin of a 1 -> N curve: N > 1 runs 1 048 576 rows per GPU per step (the step has to hide the exchange of the 72 MB buffer), the headline
    # line 262 144; this is ONE GPU at the N > 1 step size, same learning-rate rule
    Bn = 1_048_576
    if args.solver == "sgd" and args.batch_rows != Bn and n >= 4 * Bn:
        ms_ = make_matrix(engine, L, args, n, 0)
        v_s, _ = timed_variant(args, L, engine, ms_, v0, Bn, max(8, args.steps // 2), learn_rate=learn_rate_for(Bn))
        ms_.close()
        out["scaling_reference"] = {"value": v_s, "unit": "examples/s", "batch_rows_per_gpu": Bn, "learn_rate": learn_rate_for(Bn),
                                    "note": "one GPU at the per-GPU step size that `bench.py --gpus N` (N > 1) runs: divide the N-GPU values by THIS figure for a like-for-like efficiency"}
    if args.state_fp64:
        return out
    # small steps (latency-bound: two dependent launches per step) on the first 2M rows
    sub_n = min(n, 2_000_000)
    sub = make_matrix(engine, L, args, sub_n, 0)
    small = {}
    for B in (4096, 16384):
        es = engine.Engine(p, **engine_kwargs(args, L, B, 0, 1))
        es.set_params(0.0, None, v0.astype(np.float64))
        nbs = es.num_batches(sub)
        steps = 300 if B == 4096 else 150
        small[str(B)] = B * steps / timed_steps(es, sub, nbs - 1, steps, 20)
        es.close()
    out["small_batch"] = dict(small, unit="examples/s", note="same engine, batch_rows = 4096 / 16384 (one fused tile per step)")
    # the mode the reference-parity claim is made in: the reference's own algorithm, one example per update, fp64
    seq = engine.Engine(p, **engine_kwargs(args, L, 1, 0, 1, mode=L.MODE_SEQUENTIAL, state_fp64=0, tile_rows=0))
    seq.set_params(0.0, None, v0.astype(np.float64))
    cnt = 200_000 if args.solver == "sgd" else 60_000
    seq.train(sub, 20_000)
    t0 = time.perf_counter()
    done = seq.train(sub, cnt)
    out["sequential_exact"] = {"value": done / (time.perf_counter() - t0), "unit": "examples/s",
                               "note": "FMX_MODE_SEQUENTIAL: the reference's per-example algorithm in its visiting order (fp64 state); "
                                       "1e-5-on-V parity with the reference CPU path is asserted in THIS mode; `value` is the mini-batch mode"}
    seq.close()
    # the same learner with the forward's sum reassociated (cfg.seq_reassociate: y_hat = w0 + (row part), only w0 chains the examples): the reference's algorithm, order and
    # precision, <= 1e-10 on V against the oracle with exact prediction signs (tests/test_gpu_seq_reassoc.py); what the glue's default `engine = "sequential"` runs for SGD
    if args.solver == "sgd":
        seq = engine.Engine(p, **engine_kwargs(args, L, 1, 0, 1, mode=L.MODE_SEQUENTIAL, state_fp64=0, tile_rows=0, seq_reassociate=1))
        seq.set_params(0.0, None, v0.astype(np.float64))
        seq.train(sub, 20_000)
        t0 = time.perf_counter()
        done = seq.train(sub, 2 * cnt)
        out["sequential_reassociated"] = {"value": done / (time.perf_counter() - t0), "unit": "examples/s",
                                          "note": "FMX_MODE_SEQUENTIAL with cfg.seq_reassociate = 1 (fm_seq_reassoc_k): one update per example in the reference's visiting order, fp64; the forward's sum is "
                                                  "w0 + (a fixed tree over the row), so the chain per example is pred = w0 + r, the multiplier, the w0 step; <= 1e-10 on V against the oracle, "
                                                  "signs exact, the same bits from run to run"}
        seq.close()
    # ... and as many of them as the chip has CUs: a grid of models (learning rate x L2, what fm.select's repeated fm.train calls walk) on ONE visiting order, one
    # workgroup per model in one launch per 65 536 examples (fmx_train_grid); every model bit for bit its own fmx_train (tests/test_gpu_train_grid.py)
    try:
        if args.solver != "sgd":
            raise KeyError("skip")   # (FTRL grids are built for k <= 16 only; the headline's SGD shape is what is timed)
        N_GRID = 256
        kw = engine_kwargs(args, L, 1, 0, 1, mode=L.MODE_SEQUENTIAL, state_fp64=0, tile_rows=0)
        grid = []
        for i in range(N_GRID):
            kwi = dict(kw)
            if args.solver == "sgd":
                kwi.update(learn_rate=kw.get("learn_rate", 0.01) * (0.25 + 0.25 * (i % 16)), l2_w1=1e-4 * (1 + i // 16))
            else:
                kwi.update(alpha_w=0.02 * (1 + i % 16), alpha_v=0.02 * (1 + i // 16))
            g = engine.Engine(p, **kwi)
            g.init_normal(args.seed, 0.0, 0.01)
            grid.append(g)
        gcnt = 100_000 if args.solver == "sgd" else 30_000
        engine.Engine.train_grid(grid, sub, 10_000)
        t0 = time.perf_counter()
        gdone = engine.Engine.train_grid(grid, sub, gcnt)
        gdt = time.perf_counter() - t0
        out["sequential_exact_grid"] = {"models": N_GRID, "value": N_GRID * gdone / gdt, "per_model": gdone / gdt, "unit": "examples/s",
                                        "note": "fmx_train_grid: 256 reference-order learners (a 16 x 16 grid of hyper-parameters) side by side, one workgroup each, one visiting order; "
                                                "each model's parameters are bit for bit what fmx_train alone gives it; `value` here counts every model's examples"}
        for g in grid:
            g.close()
    except KeyError:
        pass
    except BaseException as ex:   # a side measurement must not take the line with it
        out["sequential_exact_grid"] = {"error": f"{type(ex).__name__}: {ex}"}
    sub.close()
    if args.solver == "sgd":
        try:
            out["time_to_quality"] = time_to_quality(args, L, engine, v0)
        except BaseException as ex:   # a side measurement must not take the line with it
            out["time_to_quality"] = {"error": f"{type(ex).__name__}: {ex}"}
    return out


def main_sweep(args, rank, local_rank, world):
    """configs[4]: the ALS / Gibbs V sweep (MCMC_ALS_Learner::update_v, solver/MCMC_ALS_Learner.h:272-354).  The sweep does not
    shard (a feature's step reads the residual of every row that holds it; DESIGN.md section 7: replicas only), so N > 1 runs N
    independent replicas, each on its own 10 M-row range of the stream, and `value` adds them up."""
    import ctypes as C

    import torch
    import torch.distributed as dist
    from fmwr_amd import _lib as L
    from fmwr_amd import engine
    z, k, p, n = args.nnz, args.factors, args.features, args.rows
    gibbs = args.solver == "mcmc"
    dev = torch.device("cuda", local_rank)
    iid = bool(args.sweep_iid)
    exact_iid = iid and bool(args.sweep_exact)                                           # cfg.als_max_levels = 0 on the i.i.d. law: the reference's order, a chain of dependent levels
    fmajor = (iid and not args.sweep_factor_outer and not exact_iid) or bool(args.sweep_feature_major)   # cfg.als_max_levels = -2
    if iid:   # SURVEY 8(d)'s i.i.d. law: no field structure -- the reference's feature order is a chain of ~20 000 levels there; the coloured order (exact steps, the engine's own order) is what is timed
        m = engine.Matrix.synthetic_iid(n, p, z, args.seed, law=L.COLUMNS_UNIFORM, row_offset=rank * n, device=local_rank)
    else:
        m = engine.Matrix.synthetic(n, p, z, args.seed, row_offset=rank * n, device=local_rank)
    if args.real_values:                         # SURVEY 8(d)'s value variant: U(0, 1) instead of the one-hot 1.0 (the sweep kernels then read the value arrays)
        m.synthetic_values(args.seed + 1, row_offset=rank * n)
    e = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, device=local_rank,
                      als_max_levels=-2 if fmajor else 0 if exact_iid else -1 if iid else 0)
    e.init_normal(args.seed, 0.0, 0.01)          # V0 ~ N(0, 0.01), w0 = w = 0
    t0 = time.perf_counter()
    levels, largest, approx, _ = e.als_plan(m)   # CSC of the whole matrix + the level plan: ingest, outside the timed region
    plan_s = time.perf_counter() - t0
    d_err = torch.empty(n, dtype=torch.float64, device=dev)
    L.check(L.lib().fmx_predict_device(e.h, m.h, C.c_int64(0), C.c_int64(n), C.c_void_p(d_err.data_ptr()), C.c_int(L.LINK_NONE)))
    e.sync()
    for r0 in range(0, n, 2_000_000):            # e = y_hat - y (calculate_error, REGRESSION, :520-527); labels +-1 from the generator
        r1 = min(n, r0 + 2_000_000)
        yy = np.zeros(r1 - r0, np.float32)
        L.check(L.lib().fmx_matrix_export(m.h, C.c_int64(r0), C.c_int64(r1), None, None, None, yy.ctypes.data_as(C.c_void_p)))
        d_err[r0:r1] -= torch.from_numpy(yy).to(dev, torch.float64)
    d_z = torch.randn(k, p, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(args.seed)) if gibbs else None
    lam = np.full(k, 1.0) if gibbs else None     # MCMC: a proper prior precision (the ALS learner's init() leaves lambda = 0, SURVEY A-7)
    torch.cuda.synchronize()
    ss0 = float((d_err * d_err).sum().item())

    def sweep():
        e.vsweep_device(m, d_err.data_ptr(), alpha=1.0, v_lambda=lam, dev_std_normals=d_z.data_ptr() if gibbs else None)

    def fence():
        e.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        sweep()
    fence()
    e.profile_reset()
    e.profile(7)     # HIP events around every 7th level launch, on the engine's stream, inside the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sweep()
    fence()
    dt = time.perf_counter() - t0
    e.profile(0)
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ss1 = float((d_err * d_err).sum().item())
    if not np.isfinite(ss1) or (not gibbs and not ss1 < ss0):
        raise SystemExit(f"the sweeps did not reduce the residual: {ss0} -> {ss1}")
    if rank != 0:
        return None
    nnz = n * z
    lvl_ms, lvl_n = e.profile_get(L.KERNEL_ALS_SWEEP)
    fwd_ms, fwd_n = e.profile_get(L.KERNEL_ROWS_FORWARD)
    per_launch_ms = lvl_ms / max(lvl_n, 1)
    launches = levels * k
    tiled, tile_rows, n_tiles = e.als_tiled(m)
    ordered = bool(tiled) and e.als_level_order(m)
    blocks = ordered and e.als_level_order_form(m) == 2
    if exact_iid:
        # a deep exact plan sweeps a factor in ONE persistent launch (als_exact_flow_k): the HIP events see one launch per factor, the unit stays one level of one
        # factor -- the sweep's wall time over its levels x k dependent level steps (latency, not bytes: profiles/r06_als_exact_persist.txt)
        per_launch_ms = dt / args.steps / launches * 1e3
        lvl_n = launches * args.steps
    b_launch = 40.0 * nnz / levels               # 40 B per stored nonzero per factor; one unit = one level of one factor
    if fmajor:                                   # the feature-major form: one launch per level does all k factors of its features
        launches = levels
        b_launch *= k
    gbs = b_launch / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
    step_gbs = 40.0 * nnz * k / (dt / args.steps) / 1e9
    if exact_iid:
        form = ("the exact schedule on i.i.d. columns: the reference's index order as levels of row-disjoint features, every level a dependent step of the sweep "
                "(at most ~100 short lists: latency, not bytes); one launch per factor, als_exact_flow_k: a step takes a row's record once its tag says the row's previous "
                "feature has corrected it")
    elif iid and args.sweep_factor_outer:
        form = "als_level_k on the colours' levels (one wave per feature walks its CSC column: a random 16-byte gather and scatter of (q_f, e) per entry and factor)"
    elif iid or fmajor:
        form = ("als_level_allf_wave_k on the colours' levels: ONE WAVE per feature gathers its rows' state (e and the 128-byte line of all k values q_f, 8 lanes per line) into registers, steps the k "
                "factors there (wave sums by DPP, no barrier), writes the lines back; lists over 384 rows take a 256-thread workgroup (lines in registers to 512 rows, in LDS beyond)")
    elif blocks:
        form = ("level-order, block form: the level's (q, e) array is feature-block-major (blocks of consecutive features holding at most 8192 rows); ONE kernel per level, "
                "als_block_level_pipe_k: a resident workgroup per CU streams a block's pairs into LDS at their feature-sorted slots, sums its lists, takes the coordinate steps "
                "and corrects the pairs there, and stores them as contiguous runs into the next level's blocks while the next block's pairs are already in flight")
    elif ordered:
        form = (f"level-order: the (q, e) pairs kept in the list order of the level that consumes them next ({n_tiles} tiles of {tile_rows} rows): als_order_sums_k (streams the pairs, sums the lists, "
                "takes the coordinate steps) + als_order_apply_k (corrections, every pair written to its place in the next level's order: a permutation inside the tile's L2-resident slice)")
    elif tiled:
        form = f"row-tiled: als_tile_sums_k ({n_tiles} tiles of {tile_rows} rows, each tile's (q, e) slice gathered from its XCD's L2) + als_tile_step_k + als_rows_apply_k (row-major corrections)"
    else:
        form = "als_level_k (one wave per feature walks its CSC column: a random 16-byte gather and scatter of (q, e) per entry)"
    out = {
        "metric": f"V-sweep examples/sec, {shape_tag(n, p)} sparse FM " + ("MCMC Gibbs" if gibbs else "ALS") + " sweep over V columns",
        "value": world * n * args.steps / dt, "unit": "examples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"synthetic {n}x{p}, {z} nnz/row, i.i.d. uniform columns (SURVEY 8(d)'s law), the reference's index order (cfg.als_max_levels = 0: the exact schedule), k={k}, "
                                f"{'MCMC.solver Gibbs' if gibbs else 'ALS.solver'} sweep over V columns (BASELINE.json configs[4] on the contract's column law); a step = one sweep of all {k} factors") if exact_iid else
                               (f"synthetic {n}x{p}, {z} nnz/row, i.i.d. uniform columns (SURVEY 8(d)'s law): the COLOURED order of the sweep (cfg.als_max_levels = {-1 if args.sweep_factor_outer else -2}: every coordinate step "
                                f"exact, the features visited in the order of a colouring of the share-a-row graph instead of the reference's index order, which is a chain of ~20 000 dependent levels here"
                                + ("" if args.sweep_factor_outer else "; all k factors of a feature are stepped while its rows' state is on the chip: coordinates in (colour, feature, factor) order") + "), "
                                f"k={k}, {'MCMC.solver Gibbs' if gibbs else 'ALS.solver'} sweep over V columns; a step = one sweep of all {k} factors") if iid else
                               f"synthetic {n}x{p}, {z} nnz/row, one column per stratum of [0, p) (fmx_matrix_synthetic): cfg.als_max_levels = -2 -- the exact schedule's {z} levels serve as the colours "
                               f"(the reference's own feature order) and all k factors of a feature are stepped while its rows' state is on the chip: coordinates in (level, feature, factor) order instead of the "
                               f"reference's factor-outer nesting; every step exact, NOT the reference's numbers (configs[4] proper is the line without this flag), k={k}, "
                               f"{'MCMC.solver Gibbs' if gibbs else 'ALS.solver'} sweep over V columns; a step = one sweep of all {k} factors" if fmajor else
                               f"synthetic {n}x{p}, {z} nnz/row, one column per stratum of [0, p) (fmx_matrix_synthetic: one-column-per-field data, the shape whose exact level "
                               f"schedule is {z} levels; i.i.d. columns need thousands of dependent levels and take the approximate groups instead), k={k}, "
                               f"{'MCMC.solver Gibbs' if gibbs else 'ALS.solver'} sweep over V columns (BASELINE.json configs[4]); "
                               f"a step = one sweep of all {k} factors over all rows (every example is visited once per factor)",
                   "levels": levels, "largest_level": largest, "approximate": bool(approx) and not (iid or fmajor), "feature_order": "the reference's" if exact_iid else ("coloured, factor outer (exact steps)" if args.sweep_factor_outer else "coloured, feature-major (exact steps)") if iid else "the reference's feature order, all k factors of a feature together (exact steps; the reference nests factor outer)" if fmajor else "the reference's", "levels_per_step": launches, "levels_row_tiled": tiled, "level_order_form": ("blocks" if blocks else "tiles") if ordered else False,
                   "plan_build_s": plan_s, "residual_sum_squares": [ss0, ss1], "state": "fp64 V[p][k], fp64 (q, e) pairs per row",
                   "parallelism": f"replicas{world}" if world > 1 else "dp1"},
        "roofline": {"bound": "hbm", "kernel": "one level of one factor: " + form, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                     "traffic": None, "algorithmic_bytes_per_launch": b_launch, "avg_launch_ms": per_launch_ms, "timed_launches": lvl_n,
                     "step": {"achieved": step_gbs, "frac": step_gbs / HBM_PEAK_GBS,
                              "note": "40 B x nnz x k over the whole sweep's wall time (includes the one forward pass that builds q for all factors)"},
                     "q_build_forward_ms": fwd_ms / max(fwd_n, 1) if fwd_n else None},
    }
    tr = pmc_traffic_sweep(args, bool(tiled), ordered, blocks, fmajor)
    out["roofline"]["frac_basis"] = "algorithmic"
    if tr:
        out["roofline"]["traffic"] = tr[0]
        out["roofline"]["traffic_ratio"] = tr[0] / b_launch
        if tr[0] < 0.95 * b_launch and per_launch_ms > 0:
            # the launch moves FEWER bytes than SURVEY 8(d) prices for it (this form does not do what the formula's 40 B describe): the priced figure is a speed-up over the
            # priced design, not memory utilisation.  `frac` = counted bytes / time / peak; the priced figure stays beside it as `effective_frac`
            rf = out["roofline"]
            rf["effective_achieved"], rf["effective_frac"] = rf["achieved"], rf["frac"]
            rf["achieved"] = tr[0] / (per_launch_ms * 1e-3) / 1e9
            rf["frac"] = rf["achieved"] / HBM_PEAK_GBS
            rf["frac_basis"] = "traffic"
            rf["frac_basis_note"] = ("counter traffic of one launch (traffic_source) over this run's launch time / peak; effective_frac = SURVEY 8(d)'s 40 B per nonzero and factor over the same time, "
                                     "bytes this form does not move")
        out["roofline"]["traffic_source"] = (f"profiles/{tr[1]}: fabric bytes of one launch by request size (128 x RDREQ_128B + 64 x the other reads + WRITE_SIZE), separate --pmc passes" if blocks or fmajor else
                                             f"profiles/{tr[1]}: (FETCH_SIZE*2 + WRITE_SIZE) KiB summed over the launches of one level, separate --pmc passes; upper bound, DESIGN.md section 6")
    if fmajor:
        # what a launch has to move (design bytes), per entry of the level: the row's line of k values in and out (2 x 8 k), e in and out (2 x 8), the row id 4 (+4 value);
        # on the fabric every line is 128 bytes each way and e costs a whole line in (128) and a 64-byte write out
        vb = 0 if e_unit(m) else 4
        design = (nnz / levels) * (16 * k + 16 + 4 + vb)
        fabric = (nnz / levels) * (2 * max(128, 8 * k) + 128 + 64 + 4 + vb)
        out["roofline"]["design_bytes_per_launch"] = design
        out["roofline"]["design_frac"] = design / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if per_launch_ms > 0 else None
        out["roofline"]["fabric_estimate"] = {"bytes_per_launch": fabric, "achieved": fabric / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else None, "unit": "GB/s",
                                              "note": "128-byte lines: the q line in and out, the (q, e) pair's line in for the 8 bytes of e, a 64-byte write for them out; a level's ~1 000 features start "
                                                      "together, so its gathers, its steps and its stores follow each other chip-wide instead of overlapping (profiles/r05_allf_knockouts.txt)"}
        out["roofline"]["note"] = ("`frac` prices SURVEY 8(d)'s 40 B per nonzero and factor -- what the reference's factor-outer order moves -- against this form's launch, which does all k factors "
                                   "of its features at once and moves design_bytes_per_launch instead: frac is the contract's figure, design_frac what the kernel's own bytes make of the chip.  avg_launch_ms is a HIP-event pair around the launch: on i.i.d. columns the launch is ~34 us "
                                   "(rocprofv3 --kernel-trace --stats of this command: profiles/r05c_kernel_stats_mcmc_iid.csv, 34.1 us) and the pair adds ~4 us to it")
    elif blocks:
        # what the one pass of a level has to move (design bytes): (q, e) 16 in and 16 out, LDS slot 2, slot in destination order 2, position in the next level's array 4 (+4 value)
        vb = 0 if e_unit(m) else 4
        design = (nnz / levels) * (16 + 16 + 2 + 2 + 4 + vb)
        out["roofline"]["design_bytes_per_launch"] = design
        out["roofline"]["design_frac"] = design / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if per_launch_ms > 0 else None
    elif ordered:
        # what the two passes of a level have to move (design bytes): sums + step: (q, e) 16 (+4 value) per nonzero, the list offsets of every (tile, feature);
        # apply: (q, e) 16 in and 16 out, feature index 2, position in the next level's order 4 (+4 value) per nonzero
        vb = 0 if e_unit(m) else 4
        design = (nnz / levels) * (16 + vb) + n_tiles * largest * 4 + (nnz / levels) * (16 + 16 + 2 + 4 + vb)
        out["roofline"]["design_bytes_per_launch"] = design
        out["roofline"]["design_frac"] = design / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if per_launch_ms > 0 else None
    elif tiled:
        # what the three passes of a level have to move (design bytes): sums: (q, e) 16 + list entry 4 (+4 value) per nonzero, the per-tile sums out;
        # step: the per-tile sums in; corrections: level-major index 4 (+4 value), (q, e) 16 in and 16 out per row
        per_level_feats = largest
        vb = 0 if e_unit(m) else 4
        design = (nnz / levels) * (16 + 4 + vb) + 2 * n_tiles * per_level_feats * 16 + n * (4 + vb + 32)
        out["roofline"]["design_bytes_per_launch"] = design
        out["roofline"]["design_frac"] = design / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if per_launch_ms > 0 else None
    if not args.no_extras:
        # the untiled form's ceiling: random 16-byte (q, e) pairs from the n-row table; als_level_k reads AND writes one per entry.  The tiled form leaves
        # that ceiling behind (its gathers hit the XCD's L2): entries per second over the SAME figure shows by how much
        r = engine.measure_gather(n * 16, 16, n_groups=262_144, per_group=64, in_flight=4, reps=20, device=local_rank)
        got = (nnz / levels) / (per_launch_ms * 1e-3) if per_launch_ms > 0 else 0.0
        out["roofline"]["gather_ceiling"] = {"table_MB": n * 16 / 1e6, "row_bytes": 16, "ceiling_rows_per_s": r, "kernel_entries_per_s": got,
                                             "ceiling_frac": got / r if r else None,
                                             "note": "entries per second of a level over the measured rate of random 16-B gathers from the whole (q, e) table (the untiled form does a gather AND a scatter "
                                                     "per entry against it: at most 0.5; the row-tiled and level-order forms keep their random accesses inside L2-resident slices or LDS and are not bound by it)"}
    if (blocks or fmajor) and world == 1:
        # opt-in variant (fmx_als_carry_q): the sweep keeps q = X v_f current as it goes, so sweep t + 1 needs no forward pass to rebuild it.  NOT `value`: the reference
        # recomputes q for every factor of every sweep, and `value` above does one forward pass per sweep for it
        e.als_carry_q(True)
        sweep(); fence()          # (this one still builds the table)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sweep()
        fence()
        dtc = time.perf_counter() - t0
        e.als_carry_q(False)
        ssc = float((d_err * d_err).sum().item())
        out["value_q_carried"] = {"value": n * args.steps / dtc, "unit": "examples/s", "ms_per_step": dtc / args.steps * 1e3, "finite": bool(np.isfinite(ssc)),
                                  "note": ("fmx_als_carry_q(1): the feature-major sweep corrects every q_f of every row as it goes, so the table it leaves is X v_f of the new V; reused by the next sweep when V's 64-bit "
                                           "fingerprint is unchanged (no forward pass; rebuilt every 64th sweep); agrees with the rebuilt form to rounding (tests/test_gpu_coloured.py)") if fmajor else
                                          ("fmx_als_carry_q(1): q = X v_f is written back as each factor's pairs move on and reused by the next sweep when V's 64-bit fingerprint is "
                                           "unchanged (no forward pass; rebuilt every 64th sweep); agrees with the rebuilt form to ~1e-13 (tests/test_gpu_configs4.py)")}
    if world == 1 and not args.real_values and not args.no_extras:
        # SURVEY 8(d)'s value variant, timed once per run: the same shape with U(0, 1) values (6 144-pair blocks, the entry values in LDS beside the pairs)
        try:
            m2 = engine.Matrix.synthetic(n, p, z, args.seed, device=local_rank).synthetic_values(args.seed + 1)
            e2 = engine.Engine(p, task=L.TASK_REGRESSION, solver=L.SOLVER_MCMC if gibbs else L.SOLVER_ALS, num_factor=k, mode=L.MODE_SEQUENTIAL, device=local_rank)
            e2.init_normal(args.seed, 0.0, 0.01)
            d2 = torch.zeros(n, dtype=torch.float64, device=dev)
            L.check(L.lib().fmx_predict_device(e2.h, m2.h, C.c_int64(0), C.c_int64(n), C.c_void_p(d2.data_ptr()), C.c_int(L.LINK_NONE)))
            e2.sync()
            sw2 = lambda: e2.vsweep_device(m2, d2.data_ptr(), alpha=1.0, v_lambda=lam, dev_std_normals=d_z.data_ptr() if gibbs else None)
            sw2(); e2.sync()
            t0 = time.perf_counter()
            for _ in range(2):
                sw2()
            e2.sync()
            dtv = (time.perf_counter() - t0) / 2
            out["value_real_values"] = {"value": n / dtv, "unit": "examples/s", "ms_per_step": dtv * 1e3, "level_order_form": e2.als_level_order_form(m2)}
            e2.close(); m2.close(); del d2
        except Exception as ex:
            out["value_real_values"] = {"error": f"{type(ex).__name__}: {ex}"}
    if world == 1 and not exact_iid:
        # the whole learner iteration around the sweep -- forward, residual, w0 step, w sweep (30 levels, three-pass tiled form), V sweep (480 levels) --
        # through fmx_als_train(with_v = 1) (MCMC_ALS_Learner::learn, :91-156, with the update_v call the shipped update_all leaves out)
        try:
            e.als_train(m, 1, with_v=True); fence()
            t0 = time.perf_counter()
            e.als_train(m, 2, with_v=True); fence()
            dti = (time.perf_counter() - t0) / 2
            out["learner_iteration"] = {"ms": dti * 1e3, "examples_per_s": n / dti,
                                        "note": "one iteration of fmx_als_train(with_v = 1) at this shape: forward + residual + w0 + w sweep + V sweep, timed over two iterations"}
        except Exception as ex:
            out["learner_iteration"] = {"error": str(ex)}
    if args.cpu_rows > 0:
        out["cpu_baseline"] = cpu_baseline(m, args, None)
    return out


def main_stream(args, rank, local_rank, world):
    """configs[3] as ONE N-GPU job: p = 33 M, k = 32, Criteo-shaped rows streamed (generated, planned, trained on once, dropped), rank r
    its own row range, per step: pull current rows -> gradient sums -> records to their owners -> update (fmwr_amd/distributed.py)."""
    import torch
    import torch.distributed as dist
    from fmwr_amd import _lib as L
    from fmwr_amd import engine
    from fmwr_amd.distributed import DataParallel, EngineStepper, shard_rows, train_stream
    z, k, p, B = args.nnz, args.factors, args.features, args.batch_rows
    tune = 16 if B >= 65536 else 0
    total = (tune + args.warmup + args.steps) * B * world
    r0, r1 = shard_rows(total, rank, world)
    e = engine.Engine(p, **engine_kwargs(args, L, B, local_rank, world, exchange_chunks=0))
    e.init_normal(args.seed, 0.0, 0.01)
    exchange = "owner" if args.exchange in ("auto", "owner") else args.exchange
    dp = DataParallel(EngineStepper(e, None, local_rank, dense=False), exchange=exchange) if world > 1 else None
    src = e.source(r1 - r0, seed=args.seed, row_offset=r0, fields=(13, engine.CRITEO_VOCAB, 3.0))

    def run(steps):
        if world > 1:
            return train_stream(dp, src, steps=steps)
        done = 0
        for _ in range(steps):
            m = src.next()
            e.step(m, 0)
            done += m.n
        return done

    def fence():
        e.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run(tune + args.warmup)
    fence()
    if dp is not None and exchange == "owner":
        dp.bytes_sent.clear()
    e.profile_reset()
    e.profile(15)
    t0 = time.perf_counter()
    done = run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    e.profile(0)
    wait = src.close()
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    _, vv = e.get_rows(np.arange(rank, p, max(1, p // 100_000) * world, dtype=np.uint32))   # rows this rank owns (id mod N)
    if not np.all(np.isfinite(vv)):
        raise SystemExit("non-finite parameters after the timed region")
    moved = None
    if dp is not None and exchange == "owner":
        mine = torch.tensor([float(np.mean(dp.bytes_sent)), float(np.max(dp.bytes_sent))], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        moved = {"bytes_sent_per_step_per_rank_mean": float(torch.stack(allr)[:, 0].mean().item()), "bytes_sent_per_step_per_rank_max": float(torch.stack(allr)[:, 1].max().item())}
    elif dp is not None:
        moved = {"bytes_received_per_step_per_rank": dp.last_exchange_bytes}
    if rank != 0:
        return None
    fwd_ms, fwd_n = e.profile_get(L.KERNEL_ROWS_FORWARD)
    upd_ms, upd_n = e.profile_get(L.KERNEL_COLS_UPDATE)
    b_step = algorithmic_bytes(z, k, p, B, 4, False)[2]
    ms_step = dt / args.steps * 1e3
    step_gbs = b_step / (ms_step * 1e-3) / 1e9
    # a streamed step is ONE tile: one launch of each training kernel per step.  The roofline is priced on those two (HIP events on the engine's stream, every 15th
    # launch); generation and planning of the steps ahead run on a second stream beside them and are reported on their own
    k_ms = fwd_ms / max(fwd_n, 1) + upd_ms / max(upd_n, 1)
    k_gbs = b_step / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    return ({
        "metric": "training examples/sec, Criteo-shaped 33M-feature sparse FM SGD, streamed (configs[3])",
        "value": B * world * args.steps / dt, "unit": "examples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"Criteo-shaped synthetic STREAM, {p} features ({z} nnz/row: 13 dense + 26 categorical fields, skew 3), k={k}, SGD mini-batch: every step's "
                               f"{B} rows per GPU are generated, planned and trained on once (BASELINE.json configs[3]; its 4e9 rows = {4_000_000_000 // (B * world)} such steps)",
                   "batch_rows_per_gpu": B, "global_batch_rows": B * world, "parallelism": f"dp{world}",
                   "ingest_wait_s": wait,
                   **({"exchange": dict(form=exchange + (": records all-to-all to the owner (id mod N), owner update, rows pulled back on demand" if exchange == "owner" else ""), **moved)} if moved else {})},
        "roofline": {"bound": "hbm", "kernel": "fm_rows_forward + fm_cols_update of a streamed step (one tile: one launch each)", "achieved": k_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": k_gbs / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_example": b_step / B, "avg_launch_ms": k_ms,
                     "kernels": {"fm_rows_forward": {"avg_launch_ms": fwd_ms / max(fwd_n, 1)}, "fm_cols_update": {"avg_launch_ms": upd_ms / max(upd_n, 1)}},
                     "note": "per GPU: SURVEY 8(d) step bytes over the two training kernels' HIP-event times (they run beside the ingest stream, so their times include what its "
                             "traffic costs them)"},
        "ingest": {"ms_per_step_beyond_training_kernels": ms_step - k_ms, "step_frac_with_ingest" + ("_and_exchange" if world > 1 else ""): step_gbs / HBM_PEAK_GBS,
                   "note": "generate + split + per-field sort + directory of the step after next, on a second stream; what the step's wall time holds beyond its two "
                           "training kernels is ingest that did not hide" + (" plus the exchange" if world > 1 else "")},
    })


def main_in_library(args):
    """The other N-GPU driver: ONE process, cfg.n_gpus = N (fm_group.hip cuts the matrix into per-device shards, runs every replica's
    gradient sums on its own stream, one grouped RCCL all-reduce -- or the all-gather of records for sparse tiles -- per step).  The
    timed region is one fmx_train call of exactly `steps` global steps; the line carries the same metric / config keys as the
    process-per-GPU driver so that a SCALE run can compare the two."""
    import torch  # noqa: F401  (imported first: libfmx.so then binds to the HIP runtime torch carries; see fmwr_amd/distributed.py)
    from fmwr_amd import _lib as L
    from fmwr_amd import engine
    N = args.gpus
    z, k, p = args.nnz, args.factors, args.features
    criteo = args.workload == "criteo"
    ftrl = args.solver == "ftrl"
    B = min(args.batch_rows, args.rows // N)
    share = os.environ.get("FMX_BENCH_SHARED_DEVICE") == "1"
    if criteo:
        m = engine.Matrix.synthetic_fields(args.rows, 13, engine.CRITEO_VOCAB, 3.0, args.seed)
    else:
        m = make_matrix(engine, L, args, args.rows, 0)
    e = engine.Engine(p, **engine_kwargs(args, L, B, 0, 1, n_gpus=N, gpus_share_device=int(share and N > 1), exchange_chunks=0, learn_rate=learn_rate_for(B * N)))
    e.init_normal(args.seed, 0.0, 0.01)
    per_step = B * N
    e.train(m, per_step * (args.warmup + (16 if B >= 65536 else 0)))   # shards, plans, phase 1's schedule trials: all outside the timed region
    t0 = time.perf_counter()
    done = e.train(m, per_step * args.steps)
    dt = time.perf_counter() - t0
    assert done == per_step * args.steps
    _, vv = e.get_rows(np.arange(0, p, max(1, p // 100_000), dtype=np.uint32))
    if not np.all(np.isfinite(vv)):
        raise SystemExit("non-finite parameters after the timed region")
    b_step = algorithmic_bytes(z, k, p, B, 4, ftrl, unit=e_unit(m))[2]
    step_gbs = b_step / (dt / args.steps) / 1e9
    ginfo = e.group_info()
    return ({
        "metric": f"training examples/sec, {shape_tag(args.rows, p)} sparse FM {args.solver.upper()}" if not criteo else "training examples/sec, Criteo-shaped 33M-feature sparse FM SGD (configs[3] shape, resident rows)",
        "value": done / dt, "unit": "examples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"synthetic {args.rows}x{p}, {z} nnz/row, {'Criteo-shaped fields' if criteo else COLUMN_LAWS[args.columns]}, k={k}, {args.solver.upper()} mini-batch (BASELINE.json configs[{3 if criteo else (2 if ftrl else 1)}]{' shape, resident rows' if criteo else ''})",
                   "driver": "in-library: one process, cfg.n_gpus replicas behind one C-ABI handle (fm_group.hip)" + (" on ONE device (rehearsal)" if share and N > 1 else ""),
                   "batch_rows_per_gpu": B, "global_batch_rows": per_step, "parallelism": f"dp{N}",
                   **({} if ftrl else {"learn_rate": learn_rate_for(per_step)}),
                   **({"exchange": (os.environ.get("FMX_GROUP_EXCHANGE") or ("steps of one sparse tile: " + ("owner-sharded (peer copies of owner-major slices)" if ginfo["sparse_exchange"] == "owner"
                                                                                                           else "all-gather of the occurring features' records") + "; the dense all-reduce otherwise"))} if N > 1 else {})},
        "roofline": {"bound": "hbm", "kernel": "step = fm_rows_forward + fm_cols_update per tile", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": step_gbs / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_example": b_step / B,
                     "note": "per GPU, SURVEY 8(d) bytes over the wall time of a global step (exchange included for N > 1)"},
        "group": ginfo,
    })


LINE_LIMIT = 4000   # bytes of the LAST stdout line (the driver parses that line; r05's 26.7 KB line came back unparsed)


def _num(x):
    """floats to 6 significant digits (the line's budget is bytes)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.6g}")


def _cap(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def _frac_entry(d):
    """one `other_configs` summary entry: what a side run is worth in four figures"""
    if d is None or "error" in d:
        return d
    r = d.get("roofline", {})
    e = {"value": _num(d["value"]), "ms_per_step": _num(d["ms_per_step"]), "frac": _num(r.get("frac")), "frac_basis": r.get("frac_basis", "algorithmic")}
    for kk in ("effective_frac", "traffic_ratio"):
        if r.get(kk) is not None:
            e[kk] = _num(r[kk])
    cb = d.get("cpu_baseline")
    if isinstance(cb, dict) and "value" in cb:
        e["cpu_value"] = _num(cb["value"])
    return e


def driver_line(d):
    """The ONE line the driver parses: BASELINE.json's metric / config, `roofline`, `cpu_baseline` and a flat summary of the side runs, at most LINE_LIMIT
    bytes.  Everything else the run measured goes to stderr lines prefixed `DETAILS ` and to bench_details.json (emit())."""
    r, cfg = d.get("roofline", {}), d.get("config", {})
    line = {kk: _num(d[kk]) for kk in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if kk in d}
    line["config"] = {"workload": _cap(cfg.get("workload"), 300)}
    for kk in ("batch_rows_per_gpu", "tile_rows", "global_batch_rows", "state", "parallelism", "levels", "feature_order", "learn_rate"):
        if kk in cfg:
            line["config"][kk] = _cap(_num(cfg[kk]), 120)
    if isinstance(cfg.get("exchange"), (str, dict)):
        line["config"]["exchange"] = _cap(cfg["exchange"] if isinstance(cfg["exchange"], str) else cfg["exchange"].get("form"), 160)
    rl = {}
    for kk in ("bound", "kernel", "dominant_kernel", "achieved", "peak", "unit", "frac", "frac_basis", "effective_frac", "frac_with_unread_values", "traffic", "traffic_ratio", "traffic_source",
               "algorithmic_bytes_per_example", "algorithmic_bytes_per_launch", "avg_launch_ms"):
        if kk in r:
            rl[kk] = _cap(_num(r[kk]), 200 if kk == "kernel" else 120)
    if isinstance(rl.get("traffic_source"), str):
        rl["traffic_source"] = rl["traffic_source"].split(":")[0]
    if isinstance(r.get("fabric"), dict):
        rl["fabric_frac"] = _num(r["fabric"].get("frac"))
    if isinstance(r.get("step"), dict):
        rl["step_frac"] = _num(r["step"].get("frac"))
    if isinstance(r.get("kernels"), dict):
        rl["kernels"] = {name: {kk: _num(v[kk]) for kk in ("avg_launch_ms", "frac", "ceiling_frac") if v.get(kk) is not None} for name, v in r["kernels"].items()}
        for name, v in r["kernels"].items():
            if isinstance(v.get("fabric"), dict):
                rl["kernels"][name]["fabric_frac"] = _num(v["fabric"].get("frac"))
    line["roofline"] = rl
    cb = d.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {kk: _cap(_num(cb[kk]), 160) for kk in ("value", "unit", "cores", "cpu_model", "kind", "sample") if kk in cb}
        if isinstance(cb.get("all_cores"), dict):
            line["cpu_baseline"]["all_cores"] = {kk: _num(cb["all_cores"][kk]) for kk in ("cores", "hogwild_examples_per_s", "forward_rows_per_s") if kk in cb["all_cores"]}
    if isinstance(d.get("other_configs"), dict):
        line["other_configs"] = {name: _frac_entry(v) if not (isinstance(v, dict) and "error" in v) else {"error": _cap(v["error"], 120)} for name, v in d["other_configs"].items()}
    side = {}
    for kk in ("sequential_exact", "sequential_reassociated", "sequential_exact_grid", "value_stratified_columns", "value_iid_uniform", "value_real_values", "value_zipf_columns", "value_ragged_rows",
               "scaling_reference", "value_q_carried"):
        if isinstance(d.get(kk), dict) and "value" in d[kk]:
            side[kk] = _num(d[kk]["value"])
    if "forward_rows_per_s" in d:
        side["forward_rows_per_s"] = _num(d["forward_rows_per_s"])
    if isinstance(d.get("end_to_end"), dict):
        side["end_to_end_one_epoch"] = _num(d["end_to_end"].get("one_epoch_examples_per_s"))
    if side:
        line["side"] = side
    line["details"] = "bench_details.json; the stderr lines prefixed DETAILS"
    # the budget is a contract: shed the optional parts, in this order, rather than print a line the driver cannot read
    for shed in ("side", "other_configs"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(shed, None)
    n = len(json.dumps(line))
    if n > LINE_LIMIT:
        raise AssertionError(f"the driver's line is {n} bytes (limit {LINE_LIMIT})")
    return line


DETAIL_PIECE = 3800   # no stdout line is longer than the driver's line may be: whatever reads this output line by line, or keeps only its tail, meets nothing it cannot hold


def detail_lines(out):
    """Everything the run measured as lines `DETAILS <key> [i/n] <json text or a piece of it>`: one top-level key per line (`other_configs` one side run per line), values whose
    text is longer than DETAIL_PIECE cut into numbered pieces (concatenate the pieces of a key in order to get its JSON back: details_from_lines())."""
    items = []
    for k, v in out.items():
        if k == "other_configs" and isinstance(v, dict):
            items.extend((f"other_configs.{kk}", vv) for kk, vv in v.items())
        else:
            items.append((k, v))
    for k, v in items:
        text = json.dumps(v)
        pieces = [text[i:i + DETAIL_PIECE] for i in range(0, len(text), DETAIL_PIECE)] or [""]
        for i, piece in enumerate(pieces):
            yield f"DETAILS {k} [{i + 1}/{len(pieces)}] {piece}"


def details_from_lines(lines):
    """the inverse of detail_lines() (tests, and whoever reads a saved stdout)"""
    acc = {}
    for ln in lines:
        if not ln.startswith("DETAILS "):
            continue
        _, key, _, piece = ln.split(" ", 3)
        acc[key] = acc.get(key, "") + piece
    out = {}
    for key, text in acc.items():
        v = json.loads(text)
        if key.startswith("other_configs."):
            out.setdefault("other_configs", {})[key[len("other_configs."):]] = v
        else:
            out[key] = v
    return out


def emit(out):
    """stdout carries exactly ONE line: the compact JSON line the driver parses (the contract's "rank 0 prints ONE JSON line" -- whatever the driver does with stdout, the
    whole of it, its tail or its last line, it finds that line and nothing else).  Everything else the run measured goes to stderr as DETAILS lines (no line over
    DETAIL_PIECE + a short prefix), printed BEFORE the line, and to bench_details.json.  FMX_BENCH_DETAILS=stdout puts the DETAILS lines on stdout instead (ahead of
    the line), =off drops them."""
    try:
        with open(os.path.join(os.environ.get("FMX_BENCH_DETAILS_DIR", ROOT), "bench_details.json"), "w") as f:
            f.write(json.dumps(out) + "\n")
    except OSError:
        pass
    where = os.environ.get("FMX_BENCH_DETAILS", "stderr")
    if where != "off":
        stream = sys.stdout if where == "stdout" else sys.stderr
        for ln in detail_lines(out):
            print(ln, file=stream, flush=True)
    print(json.dumps(driver_line(out)), flush=True)


def compact_line(d):
    """what `other_configs` keeps of a full bench line"""
    if d is None:
        return None
    r = d.get("roofline", {})
    keep = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"], "dtype": d["dtype"],
            "workload": d["config"]["workload"],
            "roofline": {kk: r[kk] for kk in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_basis", "effective_frac", "traffic_ratio", "frac_note", "frac_with_unread_values", "fabric", "survey_priced_frac", "design_frac", "avg_launch_ms", "algorithmic_bytes_per_launch",
                                              "algorithmic_bytes_per_example", "traffic") if kk in r}}
    if "kernels" in r:
        keep["roofline"]["kernels"] = {name: {kk: v[kk] for kk in ("avg_launch_ms", "frac", "ceiling_frac", "hbm_priced_frac", "traffic", "fabric") if kk in v} for name, v in r["kernels"].items()}
    if "gather_ceiling" in r:
        keep["roofline"]["ceiling_frac"] = r["gather_ceiling"].get("ceiling_frac")
    for kk in ("cpu_baseline", "forward_rows_per_s", "ingest", "value_q_carried", "learner_iteration"):
        if kk in d:
            keep[kk] = d[kk]
    if isinstance(r.get("step"), dict):
        keep["roofline"]["step"] = r["step"]
    for kk in ("level_order_form", "feature_order", "plan_build_s"):
        if kk in d["config"]:
            keep[kk] = d["config"][kk]
    for kk in ("batch_rows_per_gpu", "tile_rows", "features_occurring_per_step", "levels", "levels_row_tiled", "ingest_wait_s"):
        if kk in d["config"]:
            keep[kk] = d["config"][kk]
    return keep


def in_child_process(argv):
    """`python bench.py <argv>` as a CHILD process (this process keeps its GPU context and waits; nothing is exec'd); returns the child's full object (its bench_details.json,
    written to a directory of its own) with `process: "child"` noted in its config, or None if the child failed (the caller then runs the same thing here).
    Used for the streamed configs[3] line: a streamed job is a process of its own, and inside this long process it reads ~6 % low -- the 8 GB parameter table of its engine is
    then allocated from device memory that the runs before it have cut up (profiles/r06_stream_queue_occupancy.txt: 402 M alone, 379 M after the headline's run in one
    process, 400 M after a run that had just freed a table of the same size)."""
    import subprocess
    import tempfile
    try:
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, FMX_BENCH_DETAILS_DIR=tmp, FMX_BENCH_DETAILS="off")
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                return None
            d = json.load(open(os.path.join(tmp, "bench_details.json")))
        d.setdefault("config", {})["process"] = "child: python bench.py " + " ".join(argv)
        return d
    except Exception:
        return None


def other_configs(args):
    """BASELINE.json configs[2], [3] (rows resident and streamed) and [4], each run here, in this process, right after the headline line -- the driver's
    command (`bench.py --gpus 1 --steps K --warmup W`) witnesses all five configs.  Same code paths as `bench.py --solver ftrl`,
    `--workload criteo [--stream]`, `--solver mcmc`; the GPU shapes are the configs' own, only the step counts and the CPU samples are short."""
    out = {}
    runs = [
        ("configs[1]_fp64_state", ["--state-fp64", "--no-extras", "--steps", "20", "--warmup", "3", "--cpu-rows", "0"], run_minibatch),
        ("configs[2]", ["--solver", "ftrl", "--no-extras", "--steps", "16", "--warmup", "2", "--cpu-rows", "60000"], run_minibatch),
        ("configs[3]_resident", ["--workload", "criteo", "--steps", "30", "--warmup", "3", "--cpu-rows", "120000"], run_minibatch),
        ("configs[3]_streamed", ["--workload", "criteo", "--stream", "--steps", "30", "--warmup", "3"], main_stream),
        ("configs[4]", ["--solver", "mcmc", "--no-extras", "--steps", "4", "--warmup", "1", "--cpu-rows", "2000000"], main_sweep),
        ("configs[4]_iid_exact", ["--solver", "mcmc", "--sweep-iid", "--sweep-exact", "--no-extras", "--steps", "1", "--warmup", "1", "--cpu-rows", "0"], main_sweep),
        ("configs[4]_iid_columns", ["--solver", "mcmc", "--sweep-iid", "--no-extras", "--steps", "4", "--warmup", "2", "--cpu-rows", "0"], main_sweep),
        ("configs[4]_feature_major", ["--solver", "mcmc", "--sweep-feature-major", "--no-extras", "--steps", "4", "--warmup", "2", "--cpu-rows", "0"], main_sweep),
    ]
    for name, argv, fn in runs:
        t0 = time.perf_counter()
        try:
            argv_full = argv + ["--seed", str(args.seed), "--columns", args.columns, "--no-other-configs"]
            line = None
            if fn is main_stream and os.environ.get("FMX_BENCH_STREAM_INPROCESS") != "1":
                line = in_child_process(argv_full)
            if line is None:
                a = parse(argv_full)
                a.cpu_one_core_only = True
                line = fn(a, 0, 0, 1)
            out[name] = compact_line(line)
            out[name]["wall_s"] = time.perf_counter() - t0
        except BaseException as ex:   # a failing side run must not take the headline line with it
            out[name] = {"error": f"{type(ex).__name__}: {ex}"}
    if "configs[3]_streamed" in out and "configs[3]_resident" in out and "cpu_baseline" in out["configs[3]_resident"] and "error" not in out["configs[3]_streamed"]:
        out["configs[3]_streamed"]["cpu_baseline"] = dict(out["configs[3]_resident"]["cpu_baseline"], note="the resident line's sample: the same workload")
    return out


def launch_ranks(n):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process;
    its stdout (rank 0's ONE JSON line) and stderr pass through, its return code is returned."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.in_library:
        if world != 1:
            raise SystemExit("--in-library is one process for all GPUs: start it without torch.distributed.run")
        emit(main_in_library(args))
        return
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # `python bench.py --gpus N ...` without a launcher: start one rank per GPU as CHILD processes (torch.distributed.run) and relay their output.  This
            # process has imported neither torch nor the HIP library and never touches a GPU; nothing is exec'd.
            raise SystemExit(launch_ranks(args.gpus))
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

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

    if args.solver in ("als", "mcmc") or args.stream:
        if args.stream and args.workload != "criteo":
            raise SystemExit("--stream needs --workload criteo")
        out = (main_stream if args.stream else main_sweep)(args, rank, local_rank, world)
    else:
        out = run_minibatch(args, rank, local_rank, world)
        headline = args.solver == "sgd" and args.workload == "uniform" and world == 1 and not args.state_fp64
        if out is not None and headline and not args.no_other_configs and not args.no_extras:
            out["other_configs"] = other_configs(args)
    if out is not None:
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_minibatch(args, rank, local_rank, world):
    """configs[1] / [2] / [3] with resident rows: synchronous mini-batch steps; returns rank 0's line (None on the other ranks)."""
    import torch
    import torch.distributed as dist
    from fmwr_amd import _lib as L
    from fmwr_amd import engine
    z, k, p = args.nnz, args.factors, args.features
    ftrl = args.solver == "ftrl"
    from fmwr_amd.distributed import DataParallel, EngineStepper, shard_rows
    r0, r1 = shard_rows(args.rows, rank, world)
    n_local = r1 - r0
    B = min(args.batch_rows, args.rows // world)
    criteo = args.workload == "criteo"
    if criteo:
        m = engine.Matrix.synthetic_fields(n_local, 13, engine.CRITEO_VOCAB, 3.0, args.seed, row_offset=r0, device=local_rank)
    else:
        m = make_matrix(engine, L, args, n_local, r0, local_rank)
    e = engine.Engine(p, **engine_kwargs(args, L, B, local_rank, world))
    if criteo:
        v0 = None
        e.init_normal(args.seed, 0.0, 0.01)  # drawn on the device (k x p doubles would be 8.4 GB through the host); same on every replica
    else:
        v0 = np.random.default_rng(args.seed).normal(0.0, 0.01, (k, p)).astype(np.float32)  # same V0 on every replica
        e.set_params(0.0, None, v0.astype(np.float64))
    nb_full = max(1, n_local // B)  # ragged tail batch left out so every step does the same work
    e.sync()
    t_ing = time.perf_counter()
    e.num_batches(m)              # builds the per-tile CSC (ingest: not in the timed region; reported as end_to_end)
    e.sync()
    csc_build_s = time.perf_counter() - t_ing

    dp = None
    if world > 1:
        want = args.exchange
        st = EngineStepper(e, m, local_rank, dense=(want == "dense"))
        dp = DataParallel(st, exchange={"dense": "dense", "owner": "owner"}.get(want, "compact"))
        if want == "compact" and dp.exchange != "compact":
            raise SystemExit("--exchange compact needs steps of one sparse tile on every rank")

    def one_step(i):
        b = i % nb_full
        if world == 1:
            e.step(m, b)      # fused: forward -> w0 step -> gradient sums + update
        else:
            dp.step(b)        # forward -> gradient sums -> RCCL exchange -> update

    def fence():
        e.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Before the warmup: the engine times the two schedules of phase 1 on its own first 14 large launches and keeps the faster
    # (fmx_rows_tune_info; same bits either way).  A job runs on the settled schedule for all but its first steps, so the timed
    # region does too.  The same number of steps on every rank.
    if B >= 65536:
        for i in range(16):
            one_step(i)
    # ... and the device itself settles: in a fresh process the first tens of milliseconds of heavy work run below the steady rate (clocks ramp: profiles/r05_alloc_placement.txt,
    # r06_timed_region.txt -- the same K steps read 0.332 ms each 6 ms into the process's first burst and 0.308 ms later).  Untimed steps for PREWARM_S seconds, the same
    # number on every rank, before the W warm-up steps the caller asked for; `value` is the steady-state rate whatever K and W are
    prewarm = int(os.environ.get("FMX_BENCH_PREWARM_STEPS", "-1"))
    if prewarm < 0:
        t_w = time.perf_counter()
        for i in range(8):
            one_step(i)
        e.sync()
        per = max((time.perf_counter() - t_w) / 8, 1e-5)
        prewarm = int(min(2000, PREWARM_S / per))
        if world > 1:   # every rank must take the same number of steps (each one is a collective): the largest count any rank arrived at
            t = torch.tensor([prewarm], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            prewarm = int(t.item())
    for i in range(prewarm):
        one_step(i)
    for i in range(args.warmup):
        one_step(i)
    fence()
    e.profile_reset()
    # HIP events around every 15th launch of each kernel, on the engine's stream, inside the timed region (an event pair is a
    # serialisation point on the stream: timing every launch costs ~5 % of the throughput it is there to explain; an ODD period,
    # so that the samples walk through the tiles of a multi-tile step -- the first tile of a step only stores its sums, the last
    # one also applies the update: sampling every 16th launch of a 2-tile step saw only one of the two)
    e.profile(15)
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

    if criteo:
        _, vv = e.get_rows(np.arange(0, p, max(1, p // 100_000), dtype=np.uint32))
        w0 = 0.0
    else:
        w0, _, vv = e.get_params()
    if not (np.isfinite(w0) and np.all(np.isfinite(vv))):
        raise SystemExit("non-finite parameters after the timed region")

    if rank == 0:
        rows_step = B * world
        value = rows_step * args.steps / dt
        fwd_ms, fwd_n = e.profile_get(L.KERNEL_ROWS_FORWARD)
        upd_ms, upd_n = e.profile_get(L.KERNEL_COLS_UPDATE)
        tile_rows = effective_tile(B, k, args.tile_rows)
        eb = 8 if args.state_fp64 else 4
        # per LAUNCH: one tile.  A sparse tile's phase 2 visits only the features that occur in it (their count is known from ingest)
        sparse_tiles = world == 1 and tile_rows >= B and e.compact_info(m)[2]
        p_walk = int(np.mean([e.compact_count(m, b) for b in range(min(nb_full, 8))])) if (criteo or sparse_tiles) else p
        # one-hot rows: the kernels never read the value arrays, and those bytes are not priced; Criteo-shaped rows carry 13 real values of 39
        unit = e_unit(m) and not criteo
        b_fwd, b_upd, _ = algorithmic_bytes(z, k, p_walk, tile_rows, eb, ftrl, unit)
        b_step = algorithmic_bytes(z, k, p, B, eb, ftrl, unit)[2]
        b_step_with_values = algorithmic_bytes(z, k, p, B, eb, ftrl, False)[2]
        kernels = {
            "fm_rows_forward": (b_fwd, fwd_ms / max(fwd_n, 1)),
            "fm_cols_update": (b_upd, upd_ms / max(upd_n, 1)),
        }
        # N > 1: phase 2 runs as one launch per (feature block, tile) plus the per-block updates, so only phase 1 keeps the
        # one-launch-per-tile byte count the roofline line is defined on
        dom = max(kernels, key=lambda name: kernels[name][1]) if world == 1 else "fm_rows_forward"
        per_kernel = {}
        for name, (nbytes, ms) in kernels.items():
            gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            tr = pmc_traffic(name, args) if world == 1 else None
            per_kernel[name] = {"algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                                "traffic": tr[0] if tr else None, "traffic_source": f"profiles/{tr[1]}" if tr else None}
            if tr and ms > 0:
                # the counted memory-side bytes of a launch (FETCH_SIZE x 2 + WRITE_SIZE, Infinity Cache hits included: an upper bound on HBM bytes for
                # the cache-resident shapes, close to the bytes of whole 128-byte lines for tables that live in HBM) over this run's launch time / HBM peak
                per_kernel[name]["traffic_frac_upper_bound"] = tr[0] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                if tr[2]:
                    # the L2's fabric reads counted by size (128 x TCC_EA0_RDREQ_128B + 64 x ..._64B + 32 x ..._32B) + WRITE_SIZE: on gfx950 every miss fetches a whole
                    # 128-byte line (profiles/r04_gather_granularity.txt), so a random 64-byte row costs 128 bytes of fabric bandwidth -- this is what the launch
                    # really moved, over this run's launch time / HBM peak
                    per_kernel[name]["fabric"] = {"bytes_per_launch": tr[2], "frac": tr[2] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "reads_of_128B_frac": tr[3], "source": f"profiles/{tr[1]}"}
        step_gbs = b_step / (dt / args.steps) / 1e9   # per GPU: B rows of this rank per step
        traffic = per_kernel[dom]["traffic"]
        out = {
            "metric": f"training examples/sec, {shape_tag(args.rows, p)} sparse FM {args.solver.upper()}" if not criteo else "training examples/sec, Criteo-shaped 33M-feature sparse FM SGD (configs[3] shape, resident rows)",
            "value": value, "unit": "examples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.state_fp64 else "f32", "data": "synthetic",
            "config": {"workload": (f"synthetic {args.rows}x{p}, {z} nnz/row, {COLUMN_LAWS[args.columns]}, values {'U(0,1)' if args.real_values else '1'}, k={k}, "
                                    f"{args.solver.upper()} mini-batch (BASELINE.json configs[{2 if ftrl else 1}]{', fp64 parameter state: the reference precision' if args.state_fp64 else ''})") if not criteo else
                                   (f"Criteo-shaped synthetic {args.rows}x{p} resident ({z} nnz/row: 13 dense + 26 categorical fields, skew 3), k={k}, "
                                    f"{args.solver.upper()} mini-batch (BASELINE.json configs[3]'s shape; its 4e9 rows are streamed: fmx_train_stream)"),
                       **({"features_occurring_per_step": p_walk} if (criteo or sparse_tiles) else {}),
                       "batch_rows_per_gpu": B, "tile_rows": tile_rows, "global_batch_rows": rows_step, "rows_per_gpu": n_local,
                       "batch_reduce": "mean gradient per coordinate per step (FMX_REDUCE_MEAN)",
                       **({} if ftrl else {"learn_rate": learn_rate_for(rows_step), "learn_rate_rule": "0.01 x max(1, global rows per step / 262144): linear scaling keeps the held-out loss per "
                                                                                                     "example of the one-GPU step up to 32x the rows (profiles/r04_learning_scaling.txt)"}),
                       "state": ("fp64" if args.state_fp64 else "fp32") + " V[p][k] + w[p]" + (" + z, n" if ftrl else "") + ", fp64 accumulation",
                       "rows_forward_schedule": dict(zip(("serial", "ms_serial_x6", "ms_pipelined_x6"), e.rows_tune())),
                       "parallelism": f"dp{world}",
                       **({"exchange": (f"all-reduce(sum) of {e.grad_buffer()[1] * e.grad_elem_bytes() / 1e6:.1f} MB per step in {e.grad_layout()[0]} pipelined blocks"
                                        if dp.exchange == "dense" else
                                        (f"owner-sharded: records all-to-all to the owner (id mod N), rows pulled back: {np.mean(dp.bytes_sent) / 1e6:.1f} MB sent per rank per step"
                                         if dp.exchange == "owner" else
                                         f"all-gather of the occurring features' records: {dp.last_exchange_bytes / 1e6:.1f} MB received per rank per step"))} if world > 1 else {})},
            # headline: the whole step priced in SURVEY 8(d)'s algorithmic bytes (SGD z(16+8k)+12, FTRL z(32+24k)+12 per example),
            # i.e. value x bytes/example / 8 TB/s per GPU; the two kernels on their own bytes and HIP-event times are in `kernels`
            "roofline": {"bound": "hbm", "kernel": "step = fm_rows_forward + fm_cols_update per tile", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": step_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": (f"{per_kernel[dom]['traffic_source']}: (FETCH_SIZE*2 + WRITE_SIZE) KiB per launch of {dom}, separate --pmc passes; "
                                            "an upper bound by the guide's rule, EXACT here: every fabric read of these kernels is a 128-byte line "
                                            "(TCC_EA0_RDREQ_128B / TCC_EA0_RDREQ = 1.00 in the same summary; DESIGN.md section 6.8)") if traffic else None,
                         "algorithmic_bytes_per_example": b_step / B, "dominant_kernel": dom, "kernels": per_kernel},
        }
        if unit:
            out["roofline"]["frac_with_unread_values"] = b_step_with_values / (dt / args.steps) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["bytes_note"] = (f"one-hot rows: no kernel reads the value arrays, so SURVEY 8(d)'s step bytes ({b_step_with_values / B:.0f} per example at this state width) are priced "
                                             f"without the 4-byte value stream ({b_step / B:.0f}); the two kernels' own bytes likewise (4 bytes per nonzero less in each); "
                                             "`frac_with_unread_values` is the figure rounds 1-4 printed; `value_real_values` times the value-reading kernels")
        if world == 1 and all("fabric" in v for v in per_kernel.values()):
            fb = sum(v["fabric"]["bytes_per_launch"] for v in per_kernel.values()) * (rows_step / tile_rows if tile_rows else 1)
            out["roofline"]["fabric"] = {"bytes_per_step": fb, "frac": fb / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, "line_bytes": 128,
                                         "note": "what the step's launches really move across the fabric (the L2's reads counted by size + writes, PMC) over the step time / HBM peak: "
                                                 "a miss fetches a whole 128-byte line, so the random 64-byte rows of k = 16 fp32 tables cost twice their algorithmic bytes and `frac` on "
                                                 "algorithmic bytes cannot pass 0.5 on the gather-bound part (DESIGN.md section 6.8)"}
        out["roofline"]["frac_basis"] = "algorithmic"
        if "fabric" in out["roofline"]:
            out["roofline"]["traffic_ratio"] = out["roofline"]["fabric"]["bytes_per_step"] / b_step     # counted fabric bytes of a step over its algorithmic bytes
        elif traffic:
            out["roofline"]["traffic_ratio"] = traffic / per_kernel[dom]["algorithmic_bytes_per_launch"]
        if fwd_rate is not None:
            out["forward_rows_per_s"] = fwd_rate
        if world == 1:
            # every line carries the measured ceiling of its own access pattern
            v_row = 2 * tile_kp(k) if (e.w_in_row() and not args.state_fp64) else None
            out["gather_ceiling"] = gather_ceilings(args, engine, kernels, tile_rows, v_row, p_walk if (criteo or sparse_tiles) else 0, m if criteo else None)
            for name, c in out["gather_ceiling"].items():
                out["roofline"]["kernels"][name]["ceiling_frac"] = c["ceiling_frac"]
        # A fraction above 1 says the bytes priced are not the bytes moved: the step's `frac` then switches to the
        # design's own algorithmic bytes (what the two kernels have to move per step) and the SURVEY 8(d) figure stays beside it
        rf = out["roofline"]
        tiles_per_step = -(-B // tile_rows)
        design_step = tiles_per_step * (b_fwd + b_upd)
        rf["design_bytes_per_example"] = design_step / B
        if rf["frac"] > 1.0:
            rf["survey_priced_frac"] = rf["effective_frac"] = rf["frac"]
            rf["frac_basis"] = "design"
            if "fabric" in rf:
                rf["traffic_ratio"] = rf["fabric"]["bytes_per_step"] / design_step
            rf["frac"] = design_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS
            rf["achieved"] = design_step / (dt / args.steps) / 1e9
            rf["frac_note"] = ("SURVEY 8(d) prices a read-modify-write of theta, z, n per OCCURRENCE (the reference's per-example loop); the two-phase design touches "
                               "them once per feature per tile, so the step moves design_bytes_per_example, not algorithmic_bytes_per_example: `frac` is on the former")
        for name, kk in rf["kernels"].items():
            if kk["frac"] > 1.0:
                kk["hbm_priced_frac"] = kk["frac"]
                kk["frac"] = kk.get("ceiling_frac")
                kk["frac_note"] = ("the rows this kernel gathers are served by L2 / the Infinity Cache (skewed columns: the heads are re-read on-die), so its algorithmic "
                                   "bytes over the HBM peak exceed 1; `frac` is the kernel's row rate over the measured rate of the bare gather of the same rows instead")
        if world == 1 and not args.no_extras:
            out.update(side_measurements(args, L, engine, m, v0, value, csc_build_s, kernels, tile_rows))
        if world == 1 and args.cpu_rows > 0:
            out["cpu_baseline"] = cpu_baseline(m, args, v0)
        return out
    return None


if __name__ == "__main__":
    main()
