"""Python mirror of the reference's user-facing operator interface for the hot path.

The reference's host side for this path is R + Rcpp (R/fm_train.R, R/fm_update.R, R/fm_predict.R, R/fm_control.R,
R/fm_solver_control.R, R/fm_track_control.R, R/fm_matrix.R -> src/FM.cpp).  R is not available in this image, so the
same surface is mirrored here with the same names (dots -> underscores), argument meaning, defaults and error
messages, on top of the C ABI (include/fmx.h).  The Rcpp glue a maintainer would add is in INTEGRATION.md.

ALS runs for REGRESSION and CLASSIFICATION (probit tables regenerated, csrc/fm_probit.h); the MCMC solver runs with its
Gibbs draws taken from the numpy generator of fm_train(seed=...) instead of R's (same call order, include/fmx.h).  The tracker (track.control(step_size > 0), fm.track, fm.select: row f-1) runs on the device.
"""
import warnings

import numpy as np

from . import _lib as L
from .engine import Engine, Matrix

_TASKS = {"CLASSIFICATION": L.TASK_CLASSIFICATION, "REGRESSION": L.TASK_REGRESSION}
_SOLVERS = {"SGD": L.SOLVER_SGD, "FTRL": L.SOLVER_FTRL, "ALS": L.SOLVER_ALS, "TDAP": L.SOLVER_TDAP, "MCMC": L.SOLVER_MCMC}

# R/fm_control.R:52-66
MODEL_CONTROL_DEFAULT = {
    "keep.w0": True, "L2.w0": 0.0, "keep.w1": True, "L1.w1": 0.0, "L2.w1": 0.0,
    "factor.number": 2, "v.init_mean": 0.0, "v.init_stdev": 0.01, "L1.v": 0.0, "L2.v": 0.0,
}
SGD_SOLVER_DEFAULT = {"learn_rate": 0.01, "random_step": 1}                                        # R/fm_solver_control.R:91-94
FTRL_SOLVER_DEFAULT = {"alpha_w": 0.1, "alpha_v": 0.1, "beta_w": 1.0, "beta_v": 1.0, "random_step": 1}  # :109-115
ALS_SOLVER_DEFAULT = {"alpha_0": 1.0, "gamma_0": 1.0, "beta_0": 1.0, "mu_0": 0.0, "alpha": 1.0, "w0_mean_0": 1.0,  # :64-71
                      "update_v": False}  # not in the reference: also run the V sweep its update_all leaves out


def _control_assign(default, given):
    """R/control_tools.R:1-37: overlay named arguments on the defaults, coerce integers, warn on unknown names."""
    out = dict(default)
    unknown = 0
    for name, value in given.items():
        key = name.replace("_", ".") if name.replace("_", ".") in default else name
        if key not in default:
            unknown += 1
            continue
        d = default[key]
        if isinstance(d, bool):
            if not isinstance(value, (bool, np.bool_)):
                raise TypeError(f"{key} must be logical")
        elif isinstance(d, int):
            iv = int(value)
            if iv != value:
                iv = max(d, iv)
                warnings.warn(f"{key} is not integer, it will be set as {iv}")
            value = iv
        out[key] = value
    if unknown:
        warnings.warn("some arguments are unknown...")
    return out


def model_control(task="CLASSIFICATION", **hyper):
    """model.control() -- R/fm_control.R:43-50."""
    if task not in ("CLASSIFICATION", "REGRESSION", "RANK"):
        raise ValueError("'arg' should be one of 'CLASSIFICATION', 'REGRESSION', 'RANK'")
    return {"class": "model.control", "task": task, "hyper.params": _control_assign(MODEL_CONTROL_DEFAULT, hyper)}


def SGD_solver(**kw):
    """SGD.solver() -- R/fm_solver_control.R:96-107."""
    return {"solver": "SGD", **_control_assign(SGD_SOLVER_DEFAULT, kw)}


def FTRL_solver(**kw):
    """FTRL.solver() -- R/fm_solver_control.R:117-130."""
    return {"solver": "FTRL", **_control_assign(FTRL_SOLVER_DEFAULT, kw)}


def ALS_solver(**kw):
    """ALS.solver() -- R/fm_solver_control.R:73-87 (its parameters are ignored by the reference too: SURVEY A-7)."""
    return {"solver": "ALS", **_control_assign(ALS_SOLVER_DEFAULT, kw)}


TDAP_SOLVER_DEFAULT = {"gamma": 1e-4, "alpha_w": 0.1, "alpha_v": 0.1, "random_step": 1}  # R/fm_solver_control.R:134-139


def TDAP_solver(**kw):
    """TDAP.solver() -- R/fm_solver_control.R:141-155 (row f-3; sequential mode = the shipped algorithm; mode="minibatch" = the mini-batch form defined in DESIGN.md section 4)."""
    return {"solver": "TDAP", **_control_assign(TDAP_SOLVER_DEFAULT, kw)}


def MCMC_solver(**kw):
    """MCMC.solver() -- R/fm_solver_control.R:36-62 (its parameters are ignored by the reference too: SURVEY A-7).
    The chain's Gamma / normal variates, which the reference takes from R's generator, are drawn here from the numpy
    generator seeded by fm_train(seed=...), in the reference's call order (include/fmx.h: fmx_mcmc_train)."""
    return {"solver": "MCMC", **_control_assign(ALS_SOLVER_DEFAULT, kw)}


def solver_control(max_iter=10000, solver=None):
    """solver.control() -- R/fm_solver_control.R:22-33 (default solver TDAP.solver(), as in the reference)."""
    solver = TDAP_solver() if solver is None else solver
    if solver["solver"] in ("MCMC", "ALS") and max_iter > 100:
        warnings.warn("the maximum number of iteratorions for MCMC/ALS solver is 100, so max_iter will be set to 100")
        max_iter = min(max_iter, 100)
    return {"class": "solver.control", "max_iter": int(max_iter), "solver": solver}


def track_control(step_size=-1, evaluate_metric="LL", convergence=1e-4):
    """track.control() -- R/fm_track_control.R:20-26."""
    if evaluate_metric not in ("AUC", "ACC", "LL", "RMSE", "MAE"):
        raise ValueError('evaluate.metric %in% c("AUC", "ACC", "LL", "RMSE", "MAE") is not TRUE')
    return {"class": "track.control", "max_iter": 1, "step_size": int(step_size), "evaluate.metric": evaluate_metric,
            "convergence": convergence}


class FmMatrix:
    """fm.matrix -- R/fm_matrix.R:6-43: features = list(value, col_idx, row_size, dim, size), labels."""

    def __init__(self, features, labels, feature_names):
        self.features, self.labels, self.feature_names = features, labels, feature_names

    @property
    def dim(self):
        return self.features["dim"]


def fm_matrix(data, labels=None, feature_names=None):
    """fm.matrix(): accepts a scipy.sparse matrix or a dense 2-D array (rows = cases, columns = features)."""
    import scipy.sparse as sp
    if labels is not None:
        labels = np.asarray(labels, np.float64)
        if labels.ndim != 1:
            raise ValueError("is.numeric(labels) is not TRUE")
        if np.any(np.isnan(labels)):
            raise ValueError("!any(is.na(labels)) is not TRUE")
    if sp.issparse(data) and data.format == "csc":
        # a dgCMatrix's own slots: R/fm_matrix.R:26-33 transposes on the host (Matrix::t); the slots go over as they are and the device transposes
        # (fmx_matrix_from_dgc; examples/FM_glue.cpp takes the same branch on the "col_ptr" element)
        n, p = data.shape
        if labels is not None and len(labels) != n:
            raise ValueError("length(labels) == nrow(data) is not TRUE")
        if feature_names is None:
            feature_names = [f"V{j + 1}" for j in range(p)]
        elif len(feature_names) != p:
            raise ValueError("ncol(data) == length(feature_names) is not TRUE")
        if data.nnz > np.iinfo(np.int32).max or n > np.iinfo(np.int32).max:
            raise ValueError("a dgCMatrix's slots are 32-bit: more stored entries or rows than 2^31 - 1 (pass the rows as CSR instead)")   # (astype would wrap silently)
        features = {"value": data.data.astype(np.float64), "col_idx": data.indices.astype(np.int32), "col_ptr": data.indptr.astype(np.int32),
                    "dim": (n, p), "size": int(data.nnz)}
        return FmMatrix(features, labels, list(feature_names))
    X = data.tocsr() if sp.issparse(data) else sp.csr_matrix(np.asarray(data, np.float64))
    X.sort_indices()
    n, p = X.shape
    if labels is not None and len(labels) != n:
        raise ValueError("length(labels) == nrow(data) is not TRUE")
    if feature_names is None:
        feature_names = [f"V{j + 1}" for j in range(p)]  # numpy has no colnames; R would stop("there's no feature_names")
    elif len(feature_names) != p:
        raise ValueError("ncol(data) == length(feature_names) is not TRUE")
    if p > np.iinfo(np.int32).max:
        raise ValueError("fm.matrix keeps col_idx as 32-bit integers: more columns than 2^31 - 1")   # (astype would wrap silently)
    features = {"value": X.data.astype(np.float64), "col_idx": X.indices.astype(np.int32),
                "row_size": np.diff(X.indptr).astype(np.int32), "dim": (n, p), "size": int(X.nnz)}
    return FmMatrix(features, labels, list(feature_names))


def _device_matrix(data, labels, device):
    f = data.features
    if "col_ptr" in f:
        return Matrix.from_dgc(f["value"], f["col_idx"], f["col_ptr"], f["dim"][0], f["dim"][1], labels, device=device)
    return Matrix.from_rlist(f["value"], f["col_idx"], f["row_size"], f["dim"][1], labels, device=device)


def _engine_for(controls, p, target_range, mode, batch_rows, device):
    model, solver_ctl = controls["model"], controls["solver"]
    hp, sol = model["hyper.params"], solver_ctl["solver"]
    if model["task"] not in _TASKS:
        raise NotImplementedError("task RANK is not implemented by the reference either (FM.cpp:64 -> 'unknown task...')")
    return Engine(p, task=_TASKS[model["task"]], solver=_SOLVERS[sol["solver"]], num_factor=int(hp["factor.number"]),
                  keep_w0=int(hp["keep.w0"]), keep_w1=int(hp["keep.w1"]), l2_w0=hp["L2.w0"], l1_w1=hp["L1.w1"], l2_w1=hp["L2.w1"],
                  l1_v=hp["L1.v"], l2_v=hp["L2.v"], learn_rate=sol.get("learn_rate", 0.01), alpha_w=sol.get("alpha_w", 0.1),
                  alpha_v=sol.get("alpha_v", 0.1), beta_w=sol.get("beta_w", 1.0), beta_v=sol.get("beta_v", 1.0),
                  gamma=sol.get("gamma", 1e-4), random_step=int(sol.get("random_step", 1)), mode=L.MODE_SEQUENTIAL if mode in ("sequential", "sequential_bitwise") else L.MODE_MINIBATCH,
                  seq_reassociate=int(mode == "sequential"), state_fp64=int(mode == "minibatch_fp64"), batch_rows=int(batch_rows), min_target=target_range[0], max_target=target_range[1], device=device)


def _merge_controls(data, control):
    n = data.dim[0]
    merged = {"model": model_control(), "solver": solver_control(max_iter=max(10000, 2 * n)), "track": track_control()}  # R/fm_train.R:90-93
    for c in (control or []):
        cls = c.get("class", "")
        if not cls.endswith(".control"):
            raise ValueError("control list is wrong")
        merged[cls.split(".")[0]] = c
    merged["track"]["max_iter"] = merged["solver"]["max_iter"]  # R/fm_train.R:106
    return merged


def _check_labels(data, task):
    if data.labels is None:
        raise ValueError("there are no labels in data")  # R/fm_train.R:72-74
    y = np.asarray(data.labels, np.float64)
    if task == "CLASSIFICATION":  # R/fm_train.R:112-122
        u = np.unique(y)
        if len(u) != 2:
            raise ValueError("target should have two levels")
        if np.array_equal(u, [0.0, 1.0]):
            y = np.where(y < 1, -1.0, 1.0)
        elif not np.array_equal(u, [-1.0, 1.0]):
            raise ValueError("target should be c(0, 1) or c(-1, 1)")
    return y


def _normalize_columns(normalize, p):
    """R/fm_train.R:75-87: TRUE -> every column, FALSE -> none, or an integer vector of columns (0-based here, R is 1-based)."""
    if isinstance(normalize, (bool, np.bool_)):
        return np.arange(p, dtype=np.int32) if normalize else None
    cols = np.asarray(normalize)
    if cols.dtype.kind not in "iu":
        raise TypeError("normalize should be a logical value or an integer vector")
    if cols.size and (cols.min() < 0 or cols.max() >= p):
        raise ValueError("the columns to be normalized is out of range")
    return np.sort(cols).astype(np.int32)


def _mcmc_draws(rng, n, p, iters, k0, k1):
    """Standard variates of one MCMC run in the reference's call order: per iteration the Gamma of update_alpha, the
    normal of update_w0, the Gamma of update_w_lambda, the normals of update_w_mu and update_w (fmx.h: fmx_mcmc_train)."""
    g = np.ones((iters, 2)); z = np.zeros((iters, 2 + p))
    for it in range(iters):
        g[it, 0] = rng.gamma((1.0 + n) / 2.0)
        if k0:
            z[it, 0] = rng.normal()
        if k1:
            g[it, 1] = rng.gamma((2.0 + p) / 2.0)
            z[it, 1] = rng.normal()
            z[it, 2:] = rng.normal(size=p)
    return g, z


def _train(data, controls, w0, w, v, target_range, mode, batch_rows, device, norm_cols=None, rng=None):
    p = data.dim[1]
    y = _check_labels(data, controls["model"]["task"])
    lo, hi = float(y.min()), float(y.max())  # src/FM.cpp:89-90
    if target_range is not None:               # src/FM.cpp:91-96 (fm.update widens the range)
        lo, hi = min(lo, target_range[0]), max(hi, target_range[1])
    if controls["solver"]["solver"]["solver"] in ("ALS", "MCMC"):
        mode = "sequential"  # ALS / MCMC work on the fp64 tables
    eng = _engine_for(controls, p, (lo, hi), mode, batch_rows, device)
    eng.set_params(w0, w, v)
    m = _device_matrix(data, y, device)
    mean = std = None
    if norm_cols is not None:  # src/FM.cpp:36-38: scales = m.scales(normalize)
        mean, std = m.scales(norm_cols)
    sol = controls["solver"]["solver"]["solver"]
    track = controls["track"]
    trace, convergent = None, False
    if sol == "MCMC":
        hp = controls["model"]["hyper.params"]
        iters = int(controls["solver"]["max_iter"])
        g, z = _mcmc_draws(rng if rng is not None else np.random.default_rng(), data.dim[0], p, iters, bool(hp["keep.w0"]), bool(hp["keep.w1"]))
        if track["step_size"] > 0 and iters > 0:
            # the tracker block of MCMC_ALS_Learner::learn (:96-125): the model at the START of iterations 0, step, 2 step, ... and of
            # the last one is scored (clamped predictions / fast_pnorm) and snapshotted; the chain continues call by call
            metric = getattr(L, "EVAL_" + track["evaluate.metric"])
            idx, evals, snaps, state, ii = [], [], [], None, -1
            for it in range(iters):
                ii = 0 if ii + 1 == track["step_size"] else ii + 1
                if ii == 0 or it == iters - 1:
                    idx.append(it); evals.append(eng.evaluate(m, metric))
                    a, b, c = eng.get_params(); snaps.append({"w0": a, "w": b, "v": c})
                state = eng.mcmc_train(m, 1, g[it:it + 1], z[it:it + 1]) if state is None else eng.mcmc_train_from(m, 1, g[it:it + 1], z[it:it + 1], state)
            trace = {"trace": [np.asarray(idx)] + snaps, "evaluation.train": np.asarray(evals)}
        else:
            eng.mcmc_train(m, iters, g, z)
    elif track["step_size"] > 0:  # learner->tracker.step_size > 0: Learner::learn evaluates, snapshots and may stop early
        metric = getattr(L, "EVAL_" + track["evaluate.metric"])
        r = eng.train_tracked(m, controls["solver"]["max_iter"], track["step_size"], metric, track["convergence"], keep_params=True)
        convergent = r["convergent"]
        # Tracker::save (core/Tracker.h:96-119): trace = list(record_index, {w0,w,v}...), evaluation.train
        trace = {"trace": [r["iters"]] + [{"w0": a, "w": b, "v": c} for (a, b, c) in r["params"]], "evaluation.train": r["evals"]}
    elif sol == "ALS":  # MCMC_ALS_Learner::learn; as shipped it never sweeps V (SURVEY A-1) unless als_update_v is asked for
        eng.als_train(m, controls["solver"]["max_iter"], with_v=bool(controls["solver"]["solver"].get("update_v", False)))
    else:
        eng.train(m, controls["solver"]["max_iter"])
    w0, w, v = eng.get_params()
    model = {"w0": w0, "w": w, "v": v, "model.control": controls["model"], "solver.control": controls["solver"],
             "track.control": controls["track"], "convergence": convergent}
    scales = {"mean": mean, "std": std, "model.vars": data.feature_names, "target.range": (lo, hi)}
    fit = {"class": "FM", "Model": model, "Scales": scales, "engine": {"mode": mode, "batch_rows": batch_rows, "device": device}}
    if trace is not None:
        fit["Trace"] = trace
    return fit


def fm_train(data, normalize=True, control=None, seed=None, mode="sequential", batch_rows=65536, device=0):
    """fm.train() -- R/fm_train.R:70-127.  `control` is a list of *.control objects.  V0 ~ N(v.init_mean, v.init_stdev)
    is drawn here (the reference draws it from R's RNG inside Model::init, core/Model.h:63-72); `seed` makes it repeatable.
    mode="sequential" is the reference's algorithm (its visiting order, one update per example, fp64; for SGD the forward's sum is reassociated so that only
    w0 chains the examples -- cfg.seq_reassociate: <= 1e-10 on V against the CPU restatement, 4.0 M examples/s); mode="sequential_bitwise" keeps the reference's association
    (<= 1e-11, 1.65 M examples/s); mode="minibatch" the synchronous mini-batch engine (fp32 state),
    mode="minibatch_fp64" the same engine on fp64 state."""
    if not isinstance(data, FmMatrix):
        raise TypeError("data must be a fm.matrix object")
    norm_cols = _normalize_columns(normalize, data.dim[1])
    controls = _merge_controls(data, control)
    hp = controls["model"]["hyper.params"]
    k, p = int(hp["factor.number"]), data.dim[1]
    rng = np.random.default_rng(seed)
    v0 = rng.normal(hp["v.init_mean"], hp["v.init_stdev"], (k, p)) if k > 0 else np.zeros((0, p))
    return _train(data, controls, 0.0, np.zeros(p), v0, None, mode, batch_rows, device, norm_cols, rng=rng)


def fm_update(object, data, normalize=True, max_iter=None, mode=None, batch_rows=None, device=None, rng=None):
    """fm.update() -- R/fm_update.R:18-135: continue training from a fitted FM with the controls stored on it.
    Optimizer state (FTRL z/n, SGD q/u) is NOT carried over, exactly as in the reference.
    rng: a numpy Generator standing in for R's global stream.  The reference calls fm.init() -- k*p normal draws -- BEFORE the
    warm start overwrites them (src/FM.cpp:64 precedes :66-72), so an update consumes k*p normals too: they are drawn from
    `rng` and discarded here, which leaves the stream where the reference leaves it (examples/FM_glue.cpp does the same in R)."""
    if not isinstance(object, dict) or object.get("class") != "FM":
        raise TypeError("object must be a FM object")
    if not isinstance(data, FmMatrix):
        raise TypeError("data must be a fm.matrix object")
    if list(data.feature_names) != list(object["Scales"]["model.vars"]):  # R/fm_update.R:27-37
        raise ValueError("the features in data are not the same as those in FM model")
    mdl = object["Model"]
    # normalisation settings, R/fm_update.R:39-83 (its interactive readline() branches become errors here)
    p = data.dim[1]
    sc = object["Scales"]
    model_normalized = sc["mean"] is not None
    model_cols = np.where((np.asarray(sc["mean"]) != 0) | (np.asarray(sc["std"]) != 1))[0].astype(np.int32) if model_normalized else None
    if isinstance(normalize, (bool, np.bool_)):
        if normalize:
            if not model_normalized:
                raise ValueError("all the features are not normalized in the previously saved model; pass normalize=False or the columns explicitly")
            norm_cols = model_cols  # "follow the normalization settings in the previously saved model"
        else:
            if model_normalized:
                warnings.warn("some features have been normalized in previously saved model, but those in data will not")
            norm_cols = None
    else:
        norm_cols = _normalize_columns(normalize, p)
        if model_normalized and not np.array_equal(model_cols, norm_cols):
            raise ValueError("the selected features to normalize are different from those in previously saved model")
    controls = {"model": mdl["model.control"], "solver": dict(mdl["solver.control"]), "track": mdl["track.control"]}
    controls["solver"]["max_iter"] = max(10000, 2 * data.dim[0])  # R/fm_update.R:90
    if max_iter is not None:
        controls["solver"]["max_iter"] = int(max_iter)
    eng = object.get("engine", {})
    if rng is not None:
        hp = controls["model"]["hyper.params"]
        rng.normal(hp["v.init_mean"], hp["v.init_stdev"], (int(hp["factor.number"]), p))   # Model::init's draws, discarded (src/FM.cpp:64)
    fit = _train(data, controls, mdl["w0"], mdl["w"], mdl["v"], object["Scales"]["target.range"],
                 mode or eng.get("mode", "sequential"), batch_rows or eng.get("batch_rows", 65536),
                 eng.get("device", 0) if device is None else device, norm_cols, **({"rng": rng} if rng is not None else {}))
    if object.get("Trace") is not None and fit.get("Trace") is not None:  # R/fm_update.R:125-133: traces are concatenated
        old, new = object["Trace"], fit["Trace"]
        idx = np.concatenate([np.asarray(old["trace"][0]), np.asarray(new["trace"][0]) + np.asarray(old["trace"][0])[-1]])
        fit["Trace"] = {"trace": [idx] + list(old["trace"][1:]) + list(new["trace"][1:]),
                        "evaluation.train": np.concatenate([old["evaluation.train"], new["evaluation.train"]])}
    return fit


def predict(object, newdata=None, normalize=True):
    """predict.FM() -- R/fm_predict.R:12-34 -> FMPredict (src/FM.cpp:177-214): probabilities for CLASSIFICATION
    (logistic link for SGD/FTRL/TDAP models, the probit table for ALS models), predictions clamped to the training target range for REGRESSION."""
    if newdata is None:
        raise ValueError("newdata is null")
    if not isinstance(newdata, FmMatrix):
        raise TypeError("newdata must be a fm.matrix object")
    if np.any(np.isnan(newdata.features["value"])):
        raise ValueError("there are NAs in newdata")
    if normalize and object["Scales"]["mean"] is None:
        raise ValueError("can not normalize newdata because all the variables have not been normalized in FM model")
    mdl = object["Model"]
    controls = {"model": mdl["model.control"], "solver": mdl["solver.control"], "track": mdl["track.control"]}
    device = object.get("engine", {}).get("device", 0)
    eng = _engine_for(controls, newdata.dim[1], object["Scales"]["target.range"], "sequential", 1, device)
    eng.set_params(mdl["w0"], mdl["w"], mdl["v"])
    if not normalize and object["Scales"]["mean"] is not None:
        warnings.warn("some variables in FM model are normalized, but those in newdata will not")
    m = _device_matrix(newdata, None, device)
    if normalize:  # src/FM.cpp:183-186: m.normalize(scales)
        m.normalize(object["Scales"]["mean"], object["Scales"]["std"])
    if controls["model"]["task"] != "CLASSIFICATION":
        link = L.LINK_CLAMP
    else:  # Model::predict_prob, core/Model.h:163-180: probit table for MCMC / ALS models, logistic otherwise
        link = L.LINK_PROBIT if controls["solver"]["solver"]["solver"] in ("MCMC", "ALS") else L.LINK_LOGISTIC
    return eng.predict(m, link)


def _check_track_labels(data, task, what):
    y = np.asarray(data.labels, np.float64)
    if task == "CLASSIFICATION":  # R/fm_track.R:44-53
        u = np.unique(y)
        if len(u) != 2:
            raise ValueError(f"{what}'s target should have two levels")
        if np.array_equal(u, [0.0, 1.0]):
            y = np.where(y < 1, -1.0, 1.0)
        elif not np.array_equal(u, [-1.0, 1.0]):
            raise ValueError(f"{what}'s target should be c(0, 1) or c(-1, 1)")
    return y


def _fm_track_eval(object, data, metric):
    """FMTrack (src/FM.cpp:218-258): the metric of every recorded snapshot on `data` (Tracker::report, core/Tracker.h:70-94)."""
    mdl = object["Model"]
    controls = {"model": mdl["model.control"], "solver": mdl["solver.control"], "track": mdl["track.control"]}
    task = controls["model"]["task"]
    device = object.get("engine", {}).get("device", 0)
    eng = _engine_for(controls, data.dim[1], object["Scales"]["target.range"], "sequential", 1, device)
    m = _device_matrix(data, _check_track_labels(data, task, "data"), device)
    out = []
    for snap in object["Trace"]["trace"][1:]:
        eng.set_params(snap["w0"], snap["w"], snap["v"])
        out.append(eng.evaluate(m, getattr(L, "EVAL_" + metric)))
    return np.array(out)


def fm_track(object, data=None, newdata=None, evaluate_metric="LL"):
    """fm.track() -- R/fm_track.R:28-86: the training-trace metric and the same metric on new data per snapshot."""
    if object.get("Trace") is None:
        raise ValueError("no Trace in fm object")
    task = object["Model"]["model.control"]["task"]
    if evaluate_metric not in ("LL", "AUC", "ACC", "RMSE", "MAE"):
        raise ValueError("'arg' should be one of 'LL', 'AUC', 'ACC', 'RMSE', 'MAE'")
    if (task == "CLASSIFICATION" and evaluate_metric in ("RMSE", "MAE")) or (task == "REGRESSION" and evaluate_metric in ("LL", "AUC", "ACC")):
        raise ValueError("evaluate.metric is error")
    if object["Model"]["track.control"]["evaluate.metric"] != evaluate_metric:
        if data is None:
            raise ValueError("data is missing")
        if not isinstance(data, FmMatrix):
            raise TypeError("data is not a fm.matrix object")
        val1 = _fm_track_eval(object, data, evaluate_metric)
    else:
        val1 = np.asarray(object["Trace"]["evaluation.train"])
    if newdata is None:
        raise ValueError("newdata is missing")
    if not isinstance(newdata, FmMatrix):
        raise TypeError("newdata is not a fm.matrix object")
    val2 = _fm_track_eval(object, newdata, evaluate_metric)
    return {"class": "FMTrace", "iter": np.asarray(object["Trace"]["trace"][0]), "trace.train": val1, "trace.test": val2,
            "evaluate.metric": evaluate_metric}


def fm_select(object, trace=None, best_iter=None, drop_trace=False):
    """fm.select() -- R/fm_select.R:22-66: put the best snapshot (by the test trace) into the model."""
    if trace is None:
        raise ValueError("trace is missing")
    if not isinstance(trace, dict) or trace.get("class") != "FMTrace":
        raise TypeError("trace is not a FMTrace object")
    if object.get("Trace") is None or len(object["Trace"]) <= 1:
        raise ValueError("the Trace part in object have been dropped")
    bigger_is_better = object["Model"]["track.control"]["evaluate.metric"] in ("LL", "ACC", "AUC")  # cmp(), R/fm_select.R:69-77
    better = (lambda a, b: a > b) if bigger_is_better else (lambda a, b: a < b)
    pick = np.max if bigger_is_better else np.min
    iterations = np.asarray(object["Trace"]["trace"][0])
    test, train = np.asarray(trace["trace.test"]), np.asarray(trace["trace.train"])
    if best_iter is not None:
        if best_iter < iterations[0] or best_iter > iterations[-1]:
            raise ValueError("best.iter is out of range")
        if best_iter in iterations:
            idx = int(np.where(iterations == best_iter)[0][0])
        else:  # the recorded neighbour with the better test metric (R/fm_select.R:43-48, 1-based there)
            i1 = int(np.argmin(best_iter >= iterations)) - 1
            idx = i1 if (i1 + 1 >= len(iterations) or better(test[i1], test[i1 + 1])) else i1 + 1
    else:
        cand = np.where(test == pick(test))[0]
        idx = int(cand[0]) if len(cand) == 1 else int(np.where(train == pick(train[cand]))[0][0])
    snap = object["Trace"]["trace"][idx + 1]
    out = dict(object)
    out["Model"] = dict(object["Model"], w0=snap["w0"], w=snap["w"], v=snap["v"])
    if drop_trace:
        out.pop("Trace", None)
    return out
