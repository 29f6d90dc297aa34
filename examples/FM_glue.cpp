// FM_glue.cpp -- drop-in replacement bodies for the three Rcpp entry points of evanwang1990/FMwR's src/FM.cpp
// (FM :7-173, FMPredict :177-214, FMTrack :218-258), written against include/fmx.h only.
//
// A maintainer keeps the R files, RcppExports.{cpp,R}, NAMESPACE and every list shape untouched, drops this file into src/ in
// place of FM.cpp (src/FM.h, core/, solver/, util/ are then no longer compiled) and links libfmx.so:
//
//     # src/Makevars
//     PKG_CPPFLAGS = -I/path/to/fmx/include
//     PKG_LIBS     = -L/path/to/fmx/fmwr_amd -lfmx -Wl,-rpath,/path/to/fmx/fmwr_amd
//
// R and Rcpp are not in the build image of this repository, so this translation unit is NOT compiled here; the same call
// sequences run through ctypes (fmwr_amd/api.py: tests/test_gpu_api.py, test_gpu_tracker.py) and from plain C (examples/kat_c.c:
// tests/test_gpu_c_caller.py).  Nothing below needs the reference's C++ headers.
//
// RNG contract.  The reference draws from R's generator in three places and the glue keeps every draw in its place, so that a
// script with set.seed(s) leaves R's stream exactly where the reference leaves it:
//   * Model::init (core/Model.h:63-72 via util/Dmatrix.h:143-146) draws k*p normals in [f][j] order -- ALWAYS, before a warm
//     start overwrites them (src/FM.cpp:64 precedes :66-72): fm.update() therefore consumes k*p normals too (draw and discard);
//   * the MCMC learner draws inside its loop (solver/MCMC_ALS_Learner.h:359-445): pre-drawn here per iteration, in call order;
//   * random_step > 1 strides and the MCMC truncated normals come from libc rand(), inside the library, as in the reference.
#include <Rcpp.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "fmx.h"
using namespace Rcpp;

static void fmx_check(int status) { if (status != FMX_OK) stop(fmx_last_error()); }   // END_RCPP turns it into an R error

struct Handles {   // freed on every exit path, Rcpp::stop included
  fmx_engine* e = nullptr;
  fmx_matrix* m = nullptr;
  ~Handles() { fmx_matrix_destroy(m); fmx_engine_destroy(e); }
};

static fmx_config config_from_controls(List fm_controls, List solver_controls, double min_t, double max_t) {
  fmx_config c; fmx_check(fmx_config_default(&c));
  List hp = fm_controls["hyper.params"];                         // src/FM.cpp:48-58
  c.task       = as<std::string>(fm_controls["task"]) == "REGRESSION" ? FMX_TASK_REGRESSION : FMX_TASK_CLASSIFICATION;
  c.keep_w0    = (bool)hp["keep.w0"];   c.l2_w0 = (double)hp["L2.w0"];
  c.keep_w1    = (bool)hp["keep.w1"];   c.l1_w1 = (double)hp["L1.w1"];  c.l2_w1 = (double)hp["L2.w1"];
  c.num_factor = (int)hp["factor.number"];
  c.l1_v = (double)hp["L1.v"];          c.l2_v = (double)hp["L2.v"];
  List solver = solver_controls["solver"];                       // src/FM.cpp:60-62, :123-134
  std::string s = as<std::string>(solver.attr("solver"));
  if (s == "SGD") { c.solver = FMX_SOLVER_SGD; c.learn_rate = (double)solver["learn_rate"]; c.random_step = (int)solver["random_step"]; }
  else if (s == "FTRL") { c.solver = FMX_SOLVER_FTRL;
    c.alpha_w = (double)solver["alpha_w"]; c.alpha_v = (double)solver["alpha_v"];
    c.beta_w  = (double)solver["beta_w"];  c.beta_v  = (double)solver["beta_v"]; c.random_step = (int)solver["random_step"]; }
  else if (s == "TDAP") { c.solver = FMX_SOLVER_TDAP; c.gamma = (double)solver["gamma"];
    c.alpha_w = (double)solver["alpha_w"]; c.alpha_v = (double)solver["alpha_v"]; c.random_step = (int)solver["random_step"]; }
  else if (s == "ALS") c.solver = FMX_SOLVER_ALS;                 // its R-side parameters are overridden by learner->init() in the reference too (SURVEY A-7)
  else c.solver = FMX_SOLVER_MCMC;
  c.min_target = min_t; c.max_target = max_t;                    // src/FM.cpp:89-96
  c.mode = FMX_MODE_SEQUENTIAL;                                  // default: the reference's algorithm, its visiting order, one example per update, fp64
  c.seq_reassociate = 1;                                         // SGD: y_hat summed as w0 + (row part) -- <= 1e-10 on V, signs exact, 4.0 M examples/s; "sequential_bitwise" below turns it off
  // The throughput mode is ONE optional element of solver.control -- a non-breaking extension a maintainer adds to R/fm_solver_control.R
  // (`engine = c("sequential", "sequential_bitwise", "minibatch", "minibatch_fp64")`, `batch_rows = 262144L`); lists without it behave as before:
  //   "minibatch"      synchronous mini-batches, fp32 state: 847 M examples/s at configs[1]'s shape against 4.0 M (one MI355X)
  //   "minibatch_fp64" the same with the reference's fp64 state: 616 M examples/s, 1e-5 on V guaranteed against the mini-batch restatement
  // options("FM.threads") arrives as fm_controls$nthreads (src/FM.cpp:59,97) and becomes the number of GPUs there: fmx_train shards the rows over
  // devices 0..n-1 and exchanges the gradient sums with RCCL inside the library.
  if (solver_controls.containsElementNamed("engine") && (c.solver == FMX_SOLVER_SGD || c.solver == FMX_SOLVER_FTRL || c.solver == FMX_SOLVER_TDAP)) {
    const std::string eng = as<std::string>(solver_controls["engine"]);
    if (eng == "minibatch" || eng == "minibatch_fp64") {
      c.mode = FMX_MODE_MINIBATCH;
      c.state_fp64 = eng == "minibatch_fp64";
      c.batch_rows = solver_controls.containsElementNamed("batch_rows") ? (int)solver_controls["batch_rows"] : 262144;
      int32_t n_dev = 0;
      fmx_check(fmx_device_count(&n_dev));
      const int want = (int)fm_controls["nthreads"];
      c.n_gpus = want > 1 ? (want < n_dev ? want : n_dev) : 1;
    } else if (eng == "sequential_bitwise") c.seq_reassociate = 0;   // the reference's association of the forward's sum: <= 1e-11 on V, 1.65 M examples/s
    else if (eng != "sequential") stop("solver.control(engine = ...) must be \"sequential\", \"sequential_bitwise\", \"minibatch\" or \"minibatch_fp64\"");
  }
  // ALS / MCMC: the ORDER of the coordinate sweeps is another optional element (`sweep_order = c("reference", "coloured", "feature_major")`, include/fmx.h
  // cfg.als_max_levels): "reference" keeps the reference's feature order and factor-outer nesting (its numbers); the other two take every step exactly but in an order
  // the engine chooses -- what makes matrices without field structure fast (i.i.d. columns: 5 M -> 43 M -> 213 M examples/s per sweep).
  if (solver_controls.containsElementNamed("sweep_order") && (c.solver == FMX_SOLVER_ALS || c.solver == FMX_SOLVER_MCMC)) {
    const std::string ord = as<std::string>(solver_controls["sweep_order"]);
    if (ord == "coloured") c.als_max_levels = -1;
    else if (ord == "feature_major") c.als_max_levels = -2;
    else if (ord != "reference") stop("solver.control(sweep_order = ...) must be \"reference\", \"coloured\" or \"feature_major\"");
  }
  return c;
}

static fmx_matrix* matrix_from_fm_matrix(List X, SEXP labels) {   // src/FM.cpp:31-44, util/Smatrix.h:44-61
  NumericVector value = X["value"]; IntegerVector col_idx = X["col_idx"], dim = X["dim"];
  fmx_matrix* m = nullptr;
  const double* y = Rf_isNull(labels) ? nullptr : REAL(labels);
  if (X.containsElementNamed("col_ptr")) {
    // a dgCMatrix's own slots (x, i, p: R/fm_matrix.R keeping them instead of calling Matrix::t): col_idx then holds ROW indices and the
    // rows are made on the device
    IntegerVector col_ptr = X["col_ptr"];
    // the library reads p[0 .. ncol] and i / x [0 .. nnz) on the host before it can validate anything: the lengths are checked here
    if (col_ptr.size() != dim[1] + 1) stop("col_ptr must hold ncol + 1 column pointers");
    if (col_idx.size() != value.size()) stop("the lengths of col_idx and value differ");
    if (y && Rf_xlength(labels) != dim[0]) stop("target's length is not equal the number of cases...");
    fmx_check(fmx_matrix_from_dgc(0, dim[0], (uint32_t)dim[1], value.size(), value.begin(), col_idx.begin(), col_ptr.begin(), y, &m));
    return m;
  }
  IntegerVector row_size = X["row_size"];
  fmx_check(fmx_matrix_from_rlist(0, dim[0], (uint32_t)dim[1], value.size(), value.begin(), col_idx.begin(), row_size.begin(), y, &m));
  // Optional: a matrix made from a data frame with one-hot encoded factors (numeric columns first) may name its layout -- R/fm_matrix.R would pass
  // the ranges along as X$field_base (from model.matrix's "assign" attribute) -- and the step plans are then built field by field, twice as fast:
  //   if (X.containsElementNamed("field_base")) {
  //     IntegerVector fb = X["field_base"];  std::vector<uint32_t> b(fb.begin(), fb.end());
  //     if (fmx_matrix_set_fields(m, (int32_t)b[0], (int32_t)b.size() - 1, b.data()) != FMX_OK) { /* not that layout: the general path is used */ }
  //   }
  return m;
}

static int metric_id(const std::string& name) {                   // util/Macros.h:24-29 by name, as src/FM.cpp:99-103 / :241-248 map them
  static const std::map<std::string, int> metric = {{"LL", FMX_EVAL_LL}, {"AUC", FMX_EVAL_AUC}, {"ACC", FMX_EVAL_ACC},
                                                    {"RMSE", FMX_EVAL_RMSE}, {"MSE", FMX_EVAL_MSE}, {"MAE", FMX_EVAL_MAE}};
  auto it = metric.find(name);
  if (it == metric.end()) stop("Unknown evaluation metric...");
  return it->second;
}

// Tracker::record (core/Tracker.h:54-63): the engine's current model as one list(w0, w, v)
static List snapshot(fmx_engine* e, int k, int p) {
  NumericVector w(p); NumericMatrix v(k, p); double w0 = 0.0;
  fmx_check(fmx_get_params(e, &w0, w.begin(), v.begin()));
  return List::create(_["w0"] = w0, _["w"] = w, _["v"] = v);
}

// [[Rcpp::export]]
List FM(List data_, IntegerVector normalize, List fm_controls, List solver_controls, List track_controls, List model_list) {
  List X = data_["features"]; NumericVector target = data_["labels"]; IntegerVector dim = X["dim"];
  double lo = min(target), hi = max(target);
  const bool warm = !Rf_isNull(model_list.attr("class"));
  if (warm) { NumericVector r = as<List>(model_list["Scales"]).attr("target.range"); lo = std::min(lo, r[0]); hi = std::max(hi, r[1]); }   // :91-96
  fmx_config c = config_from_controls(fm_controls, solver_controls, lo, hi);
  Handles H;
  fmx_check(fmx_engine_create(&c, (uint64_t)dim[1], &H.e));
  H.m = matrix_from_fm_matrix(X, data_["labels"]);
  List scales;
  if (normalize[0] > -1) {                                        // m.scales(normalize), src/FM.cpp:36-38
    NumericVector mean(dim[1]), sd(dim[1]);
    fmx_check(fmx_matrix_scales(H.m, normalize.begin(), normalize.size(), mean.begin(), sd.begin()));
    scales["mean"] = mean; scales["std"] = sd;
  }
  List hp = fm_controls["hyper.params"];
  const int k = c.num_factor, p = dim[1], n = dim[0];
  NumericVector w(p); NumericMatrix v(k, p); double w0 = 0.0;
  {                                                               // fm.init(), src/FM.cpp:64 -- ALWAYS drawn, also before a warm start:
    const double mu = hp["v.init_mean"], sd = hp["v.init_stdev"]; //   DMatrixDouble::init_norm fills the row-major [f][j] block front to back
    for (int f = 0; f < k; ++f)                                   //   (util/Dmatrix.h:143-146): factor OUTER, feature INNER -- the same
      for (int j = 0; j < p; ++j) v(f, j) = Rf_rnorm(mu, sd);     //   set.seed() gives the reference's own V0 and leaves R's stream where it leaves it
  }
  if (warm) {                                                     // :66-72 overwrite the draw (the k*p normals above stay consumed)
    List model = model_list["Model"]; w0 = model["w0"]; w = as<NumericVector>(model["w"]); v = as<NumericMatrix>(model["v"]);
  }
  fmx_check(fmx_set_params(H.e, w0, w.begin(), v.begin()));
  const int64_t max_iter = (int)solver_controls["max_iter"];
  const int step_size = (int)track_controls["step_size"];
  const bool track = step_size > 0;
  const int metric = track ? metric_id(as<std::string>(track_controls["evaluate.metric"])) : FMX_EVAL_LL;
  int32_t convergent = 0;
  List trace;

  if (c.solver == FMX_SOLVER_MCMC) {
    // MCMC_ALS_Learner::learn, solver/MCMC_ALS_Learner.h:91-128, for the MCMC learner.  The draws the reference takes from R inside
    // its loop are taken here, per iteration and in its call order (update_alpha, update_w0, update_w_lambda, update_w_mu, update_w:
    // :141-149); slots of switched-off updates are not drawn.  The tracker block (:96-125: at the START of iterations 0, step,
    // 2 step, ... and of the last one) evaluates the model and snapshots it; there is no convergence rule for MCMC / ALS.
    const bool k0 = c.keep_w0, k1 = c.keep_w1;
    std::vector<double> g(2), z(2 + (size_t)p);
    double state[3] = {1.0, 0.0, 0.0};                             // alpha, w_lambda, w_mu as learner->init() leaves them (:64-71)
    std::vector<int64_t> iters; std::vector<double> evals; std::vector<List> snaps;
    int ii = -1;
    for (int64_t it = 0; it < max_iter; ++it) {
      if (track) {                                                 // :96-123
        if (++ii == step_size) ii = 0;
        if (ii == 0 || it == max_iter - 1) {
          double score = 0.0;
          fmx_check(fmx_evaluate(H.e, H.m, metric, &score));      // forward + fast_pnorm / clamp + evaluates()
          iters.push_back(it); evals.push_back(score); snaps.push_back(snapshot(H.e, k, p));
        }
      }
      g[0] = Rf_rgamma((1.0 + n) / 2.0, 1.0);                      // update_alpha, :359-380
      if (k0) z[0] = norm_rand();                                  // update_w0, :160-188
      if (k1) { g[1] = Rf_rgamma((2.0 + p) / 2.0, 1.0);            // update_w_lambda, :415-445
                z[1] = norm_rand();                                // update_w_mu, :383-412
                for (int j = 0; j < p; ++j) z[2 + j] = norm_rand(); }   // update_w, :190-270
      if (it == 0) fmx_check(fmx_mcmc_train(H.e, H.m, 1, g.data(), z.data(), state));
      else fmx_check(fmx_mcmc_train_from(H.e, H.m, 1, g.data(), z.data(), state));
    }
    if (track) {
      List valid(iters.size() + 1); valid[0] = NumericVector(iters.begin(), iters.end());   // Tracker::save, core/Tracker.h:96-119
      for (size_t r = 0; r < snaps.size(); ++r) valid[r + 1] = snaps[r];
      trace = List::create(_["trace"] = valid, _["evaluation.train"] = NumericVector(evals.begin(), evals.end()));
    }
  } else if (c.solver == FMX_SOLVER_ALS && !track) {
    fmx_check(fmx_als_train(H.e, H.m, (int32_t)max_iter, /*with_v=*/0));   // as shipped: update_v is never called (SURVEY A-1)
  } else if (track) {                                              // SGD / FTRL / TDAP (solver/SGD_Learner.h:140-176) and ALS (:96-125)
    fmx_track_config t = {sizeof(fmx_track_config), metric, step_size, (double)track_controls["convergence"], /*keep_params=*/1, 0};
    fmx_check(fmx_train_tracked(H.e, H.m, max_iter, &t, nullptr, &convergent));
    int64_t nrec = 0; fmx_check(fmx_trace_size(H.e, &nrec));
    std::vector<int64_t> iters(nrec); NumericVector evals(nrec);
    fmx_check(fmx_trace_get(H.e, iters.data(), evals.begin()));
    List valid(nrec + 1); valid[0] = NumericVector(iters.begin(), iters.end());
    for (int64_t r = 0; r < nrec; ++r) {
      NumericVector sw(p); NumericMatrix sv(k, p); double s0;
      fmx_check(fmx_trace_params(H.e, r, &s0, sw.begin(), sv.begin()));
      valid[r + 1] = List::create(_["w0"] = s0, _["w"] = sw, _["v"] = sv);
    }
    trace = List::create(_["trace"] = valid, _["evaluation.train"] = evals);
  } else {
    fmx_check(fmx_train(H.e, H.m, max_iter, nullptr));            // learner->learn(data), src/FM.cpp:153
  }

  fmx_check(fmx_get_params(H.e, &w0, w.begin(), v.begin()));
  List md = List::create(_["w0"] = w0, _["w"] = w, _["v"] = v);   // Model::save_model, :156-161
  md.attr("model.control") = fm_controls; md.attr("solver.control") = solver_controls;
  md.attr("track.control") = track_controls; md.attr("convergence") = (bool)convergent;
  scales["model.vars"] = X.attr("feature_names"); scales.attr("target.range") = NumericVector::create(lo, hi);
  List res = List::create(_["Model"] = md, _["Scales"] = scales); res.attr("class") = "FM";
  if (track) res["Trace"] = trace;
  return res;
}

// [[Rcpp::export]]
NumericVector FMPredict(List newdata, bool normalize, List model_list, int max_threads) {
  (void)max_threads;                                              // no OpenMP team: the forward runs on the GPU
  List model = model_list["Model"]; List scales = model_list["Scales"];
  NumericVector r = scales.attr("target.range");
  fmx_config c = config_from_controls(model.attr("model.control"), model.attr("solver.control"), r[0], r[1]);
  List X = newdata["features"]; IntegerVector dim = X["dim"];
  Handles H;
  fmx_check(fmx_engine_create(&c, (uint64_t)dim[1], &H.e));
  NumericVector w = model["w"]; NumericMatrix v = model["v"];
  fmx_check(fmx_set_params(H.e, (double)model["w0"], w.begin(), v.begin()));   // fm.load_model, :191-194
  H.m = matrix_from_fm_matrix(X, R_NilValue);
  if (normalize) {                                                // m.normalize(scales), src/FM.cpp:183-186
    NumericVector mean = scales["mean"], sd = scales["std"];
    fmx_check(fmx_matrix_normalize(H.m, mean.begin(), sd.begin()));
  }
  NumericVector out(dim[0]);
  const int link = c.task != FMX_TASK_CLASSIFICATION ? FMX_LINK_CLAMP                                  // :204-210
                 : (c.solver == FMX_SOLVER_ALS || c.solver == FMX_SOLVER_MCMC) ? FMX_LINK_PROBIT       // Model::predict_prob, core/Model.h:163-180
                                                                               : FMX_LINK_LOGISTIC;
  fmx_check(fmx_predict(H.e, H.m, out.begin(), link));
  return out;
}

// [[Rcpp::export]]
NumericVector FMTrack(List newdata, List model_list, bool normalize, String type, int max_threads) {   // src/FM.cpp:218-258
  (void)max_threads;
  List model = model_list["Model"]; List scales = model_list["Scales"];
  NumericVector r = scales.attr("target.range");
  fmx_config c = config_from_controls(model.attr("model.control"), model.attr("solver.control"), r[0], r[1]);
  List X = newdata["features"]; IntegerVector dim = X["dim"];
  Handles H;
  fmx_check(fmx_engine_create(&c, (uint64_t)dim[1], &H.e));
  H.m = matrix_from_fm_matrix(X, newdata["labels"]);              // :222-233
  if (normalize) {                                                // m.normalize(scales), :226-228
    NumericVector mean = scales["mean"], sd = scales["std"];
    fmx_check(fmx_matrix_normalize(H.m, mean.begin(), sd.begin()));
  }
  const int eval_type = metric_id(std::string(type.get_cstring()));   // :241-248
  List trace = model_list["Trace"]; List valid = trace["trace"];      // Tracker::load, core/Tracker.h:121-134
  const int nrec = valid.size() - 1;                                  // valid[0] holds the record indices
  // Tracker::load / report overwrite only the parts the model keeps (k0 / k1 / num_factor > 0) and leave the loaded model's values
  // otherwise (core/Tracker.h:76-79, :128-132): every snapshot starts from the final model
  const double w0_final = model["w0"]; NumericVector w_final = model["w"]; NumericMatrix v_final = model["v"];
  NumericVector out(nrec);
  for (int i = 0; i < nrec; ++i) {                                    // Tracker::report, core/Tracker.h:70-94
    List snap = valid[i + 1];
    const double w0 = c.keep_w0 ? (double)snap["w0"] : w0_final;
    NumericVector w = c.keep_w1 ? as<NumericVector>(snap["w"]) : w_final;
    NumericMatrix v = c.num_factor > 0 ? as<NumericMatrix>(snap["v"]) : v_final;
    fmx_check(fmx_set_params(H.e, w0, w.begin(), v.begin()));
    fmx_check(fmx_evaluate(H.e, H.m, eval_type, &out[i]));            // forward + the task's link + evaluates(): one call
  }
  return out;                                                         // tracker.evaluations_of_test
}
