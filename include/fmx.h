/*
 * fmx.h -- C ABI of the MI355X-native Factorization Machine engine (libfmx.so).
 *
 * This is the drop-in boundary for ONE path of evanwang1990/FMwR: the degree-2 FM forward and the
 * SGD / FTRL-Proximal / TDAP training step, plus the ALS / MCMC learners (V-column sweep, w0 / w loops,
 * probit tables) and the tracker that call the same forward.  It replaces what the Rcpp entry
 * points FM() / FMPredict() do between unmarshalling the R lists and marshalling the result
 * (reference src/FM.cpp:7-173, :177-214), i.e. the seam
 *
 *     learner->init(); learner->learn(data);        src/FM.cpp:145,153   (core/Learner.h:49-51)
 *     fm.predict_batch(..) / fm.predict_prob(..)    src/FM.cpp:199-203   (core/Model.h:106-180)
 *
 * Plain pointers and sizes only; no C++/torch types; nothing throws across the boundary: every call
 * returns an int status (FMX_OK == 0) and fmx_last_error() gives the message (the glue turns a
 * non-zero status into Rcpp::stop(msg), as END_RCPP does for the reference, src/RcppExports.cpp:11,22).
 * The caller keeps ownership of every host pointer it passes; no pointer is retained after return
 * (the reference deep-copies too: util/Smatrix.h:53-60, util/Dvector.h:89-99).
 *
 * A handle is used from one host thread at a time (R is single-threaded; Model::predict is not
 * re-entrant either, core/Model.h:26-27).
 */
#ifndef FMX_H_
#define FMX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMX_OK 0
#define FMX_ERR_INVALID 1  /* bad argument / unsupported configuration */
#define FMX_ERR_HIP 2      /* HIP runtime error (message holds hipGetErrorString) */
#define FMX_ERR_NOGPU 3    /* no usable device: the engine has NO CPU fallback */
#define FMX_ERR_STATE 4    /* call order / handle state */

/* task and solver ids are the reference's (util/Macros.h:11-21) */
#define FMX_TASK_CLASSIFICATION 10
#define FMX_TASK_REGRESSION 20
#define FMX_SOLVER_MCMC 100  /* util/Macros.h:17; trains through fmx_mcmc_train */
#define FMX_SOLVER_ALS 200
#define FMX_SOLVER_SGD 300
#define FMX_SOLVER_FTRL 500
#define FMX_SOLVER_TDAP 600

/* FMX_MODE_SEQUENTIAL: the reference's algorithm as is -- one example per update, visited in the
 *   reference's order (solver/SGD_Learner.h:86-88), fp64 state.  The parity mode.
 * FMX_MODE_MINIBATCH : synchronous mini-batches of batch_rows examples, fp32 state, gradient sums per
 *   coordinate.  The throughput mode; equals the reference step at batch_rows == 1. */
#define FMX_MODE_SEQUENTIAL 0
#define FMX_MODE_MINIBATCH 1

/* How a mini-batch combines the per-example gradients of one coordinate (c = its occurrences in the batch):
 * FMX_REDUCE_MEAN: one reference step with the MEAN gradient G/c (and one lazy-L2 / L1 / FTRL update) per touched
 *   coordinate per batch -- stable for any batch size with the reference's learning rates; the default.
 * FMX_REDUCE_SUM : the SUM G with c-fold decay -- what processing the c examples one after the other with frozen
 *   gradients does; first-order equal to the reference's pass, but dense coordinates (w0!) overshoot once
 *   learn_rate * c is not small.  Both are the reference's example step at batch_rows == 1. */
#define FMX_REDUCE_MEAN 0
#define FMX_REDUCE_SUM 1

/* output transform of fmx_predict */
#define FMX_LINK_NONE 0      /* raw y_hat: Model::predict_batch, core/Model.h:106-161 */
#define FMX_LINK_LOGISTIC 1  /* 1/(1+exp(-y_hat)): Model::predict_prob, core/Model.h:173-178 */
#define FMX_LINK_CLAMP 2     /* clamp to [min_target, max_target]: src/FM.cpp:202-210 */
#define FMX_LINK_PROBIT 3    /* fast_pnorm(y_hat), the table-driven Phi of MCMC / ALS models: core/Model.h:166-171 */

/* Mirrors the three R control lists (R/fm_control.R:52-66 model.control, R/fm_solver_control.R:91-115
 * SGD.solver / FTRL.solver) as FM() reads them by key (src/FM.cpp:48-63, :97-144). */
typedef struct fmx_config {
  uint32_t struct_size;    /* = sizeof(fmx_config); ABI guard */
  int32_t task;            /* model.control(task)                       */
  int32_t solver;          /* attr(solver, "solver")                    */
  int32_t num_factor;      /* hyper.params$factor.number   (default 2)  */
  int32_t keep_w0;         /* hyper.params$keep.w0                      */
  int32_t keep_w1;         /* hyper.params$keep.w1                      */
  double l2_w0;            /* hyper.params$L2.w0                        */
  double l1_w1, l2_w1;     /* hyper.params$L1.w1, L2.w1                 */
  double l1_v, l2_v;       /* hyper.params$L1.v, L2.v                   */
  double learn_rate;       /* SGD.solver(learn_rate = 0.01)             */
  double alpha_w, alpha_v; /* FTRL.solver(alpha_w = .1, alpha_v = .1)   */
  double beta_w, beta_v;   /* FTRL.solver(beta_w = 1, beta_v = 1)       */
  int32_t random_step;     /* SGD/FTRL.solver(random_step = 1L)         */
  int32_t mode;            /* FMX_MODE_*                                */
  int64_t batch_rows;      /* mini-batch rows per step per GPU          */
  double min_target;       /* learner->min_target, src/FM.cpp:89-96     */
  double max_target;       /* learner->max_target                       */
  int32_t device;          /* HIP device ordinal                        */
  int32_t batch_reduce;    /* FMX_REDUCE_* (mini-batch mode)            */
  double gamma;            /* TDAP.solver(gamma = 1e-4): decay rate (alpha_w, alpha_v shared with FTRL) */
  int64_t tile_rows;       /* 0: default.  A step of batch_rows rows is processed in tiles of at most this many
                              rows (parameters frozen across the tiles, sums accumulated): keeps the per-tile
                              tables cache resident for large batches.  Same result up to the rounding of the
                              partial sums to the state type between tiles (fp32 unless state_fp64).           */
  int32_t state_fp64;      /* mini-batch mode: 0 = fp32 parameter/optimizer tables (default, half the HBM traffic),
                              1 = the fp64 tables of the sequential mode (the reference's precision, core/Model.h:26-42;
                              per-row sums and the exchange buffer become fp64 too).
                              WHICH MODES GUARANTEE north_star's "1e-5 relative on V" against the CPU restatement of the same
                              algorithm: FMX_MODE_SEQUENTIAL (measured <= 1e-11, prediction signs exact) and FMX_MODE_MINIBATCH with
                              state_fp64 = 1 (<= 6e-7 on every case tried, 1 650 fuzz seeds incl. diverging runs).  With fp32 state
                              the bar holds on runs that do not amplify rounding -- all targeted tests and 1 641 of the 1 650 seeds --
                              and NOT in general: on nine seeds the dynamics amplify the fp32 storage rounding beyond 1e-4, two of
                              them short and tame (41 steps, |V| <= 286: w off by 1.8e-4; 10 steps, |V| <= 8.9: V off by 2.9e-4;
                              profiles/r04_fuzz_more.txt; kept as fp64-state regression cases in tests/test_gpu_fuzz.py).  bench.py's
                              headline is the fp32 mode (BASELINE.json configs[1] asks for fp32); the fp64-state figure and its
                              roofline are printed beside it (other_configs."configs[1]_fp64_state").                          */
  int32_t exchange_chunks; /* 0/1: the exchange buffer is one block.  n > 1: it is laid out in n blocks of consecutive
                              features so that a multi-GPU driver can pipeline the exchange (fmx_grad_begin/_chunk/
                              _apply_chunk): the all-reduce of one block overlaps the gradient sums of the next.   */
  int32_t n_gpus;          /* 0/1: one GPU.  N > 1 (FMX_MODE_MINIBATCH): fmx_train shards the matrix's rows over the devices
                              device .. device+N-1, one replica each, and all-reduces the gradient sums between steps with
                              RCCL -- what the reference's `nthreads` (options("FM.threads"), src/FM.cpp:59,97) becomes
                              here.  batch_rows stays "rows per step per GPU".  fmx_get_params / fmx_predict use replica 0. */
  int32_t als_max_levels;  /* ALS / MCMC sweeps: the ORDER of the coordinate steps (history and measurements: DESIGN.md section 6; no further values will be added).
                               0  the exact schedule: levels of row-disjoint features, the reference's index-order Gauss-Seidel reproduced (its numbers).
                                  Deep plans (columns without field structure: thousands of dependent levels) sweep a factor in ONE launch (als_exact_persist_k).
                              L>0 a matrix that needs more than L levels is swept in the reference's own approximate parallel form instead
                                  (solver/MCMC_ALS_Learner.h:200-268: the features of a group step against one snapshot of the residual), under a guard that
                                  falls back to 0 if the residual rises.  For one-column-per-field data both forms coincide.
                              -1  the COLOURED order: every step exact, the features visited in (colour, index) order of a deterministic colouring of the
                                  "share a row" graph -- the reference's numbers on the matrix relabelled in that order (fmx_als_plan_info's level_of gives it),
                                  not on the matrix as given.  A matrix whose exact schedule is shallow (<= ~128 levels) keeps those levels as its colours:
                                  -1 is then bit for bit 0.
                              -2  the coloured order nested FEATURE-MAJOR: all k factors of a feature stepped together, coordinates in (colour, feature, factor)
                                  order (the reference nests factor outer); every step exact.  Applies to plans of light lists of at most 1 024 rows whose rows'
                                  lines fit the LDS at this k (577 rows at k = 17..32, 296 at k <= 64); other plans run as -1 (fmx_als_plan_info: kind 2).        */
  int32_t seq_reassociate; /* FMX_MODE_SEQUENTIAL, SGD (L2 or cumulative L1) on rows of at most 32 (64 at k <= 32) ascending columns.  0 (default): the forward's sum in
                              the reference's association, y_hat = ((w0 + w_j1 x_j1) + ...) + 0.5 (s_1^2 - q_1) + ... (core/Model.h:77-100) -- the oracle's
                              bits up to the device exp().  1: the same formula summed as w0 + (row part), the row part a fixed tree; only w0 then
                              chains one example to the next (solver/SGD_Learner.h:100-112).  The reference's algorithm, visiting order and
                              precision; the last bits of y_hat differ (<= 1e-10 on V against the oracle, prediction signs exact on every parity
                              case; the same bits from run to run).  4.0 M examples/s against 1.65 M at configs[1]'s shape.  FTRL, TDAP and other row
                              shapes ignore the flag (they run the bitwise kernels).                                                              */
  int32_t gpus_share_device; /* 1: all N replicas live on `device` and exchange through a device kernel instead of RCCL
                              (rehearsals and tests on a one-GPU box; same sums, same order of ranks)                  */
} fmx_config;

typedef struct fmx_engine fmx_engine; /* parameters + optimizer state on one GPU */
typedef struct fmx_matrix fmx_matrix; /* a device-resident fm.matrix (CSR + labels [+ per-batch CSC]) */

/* Message of the last failing call on this thread ("" if none). */
const char* fmx_last_error(void);

/* Number of HIP devices this process sees (what cfg.n_gpus may count up to); FMX_ERR_NOGPU without one. */
int fmx_device_count(int32_t* count);

/* Fill *cfg with the reference's defaults (R/fm_control.R:52-66, R/fm_solver_control.R:91-115). */
int fmx_config_default(fmx_config* cfg);

/* ---- engine: replaces `Model fm; fm.init(); learner = new XXX_Learner(); learner->init()`
 *      (src/FM.cpp:47-64, :78-145).  Parameters start at w0 = 0, w = 0, V = 0: V0 is an INPUT
 *      (fmx_set_params), because the reference draws it from R's RNG (util/Dmatrix.h:143-146). */
int fmx_engine_create(const fmx_config* cfg, uint64_t num_features, fmx_engine** out);
int fmx_engine_destroy(fmx_engine* e);

/* w: [p] or NULL (zeros); v: the R NumericMatrix k x p, column-major, i.e. v[f + j*k] (what
 * Model::save_model / load_model exchange, core/Model.h:182-227) or NULL (zeros).
 * Warm start of fm.update(): src/FM.cpp:66-72.  Optimizer state is reset, as learner->init() does. */
int fmx_set_params(fmx_engine* e, double w0, const double* w, const double* v);
int fmx_get_params(fmx_engine* e, double* w0, double* w, double* v);

/* Sparse access to the parameters (p = 33 M, k = 32 is 8.4 GB of doubles: a full fmx_get_params is the wrong tool there):
 * rows of the features ids[0..n): w -> w[i], V -> v[f + i*k] (the same k x n column-major shape as fmx_set_params).
 * fmx_set_rows leaves every other row and the optimizer state alone; a NULL w or v keeps that part of the rows. */
int fmx_get_rows(fmx_engine* e, const uint32_t* ids, int64_t n, double* w, double* v);
int fmx_set_rows(fmx_engine* e, const uint32_t* ids, int64_t n, const double* w, const double* v);
/* w0 = 0, w = 0, V ~ N(mean, stdev) drawn ON THE DEVICE (Philox4x32-10 keyed by seed, feature and factor pair; Box-Muller):
 * the shape of Model::init (core/Model.h:63-72) for synthetic workloads too large to stage through the host.  It is not R's
 * generator: runs that must reproduce the reference draw V0 in the glue and pass it to fmx_set_params. */
int fmx_init_normal(fmx_engine* e, uint64_t seed, double mean, double stdev);

/* ---- checkpoint (the reference keeps a model only as an R list and drops the optimizer state on fm.update):
 * parameters AND optimizer state (SGD-L1 q/u, FTRL z/n, TDAP u/nu/delta/h/z) to a file and back.  The loading engine must
 * have the same feature count, factor count, solver kind and mode.  Format: 64-byte header ("FMX1", version, shape),
 * the device scalars, then the tables as stored on the device (little endian). */
int fmx_engine_save(fmx_engine* e, const char* path);
int fmx_engine_load(fmx_engine* e, const char* path);

/* ---- data: replaces `SMatrix<float> m; m.assign(X); DVector<float> tg; tg.assign(target)`
 *      (src/FM.cpp:31-44).  All inputs are host pointers; the matrix is copied to HBM. */

/* R's fm.matrix layout (R/fm_matrix.R:25-34): value f64[nnz], col_idx i32[nnz] 0-based, row_size i32[n];
 * narrowed to f32 / u32 like util/Smatrix.h:44-61.  labels may be NULL (prediction). */
int fmx_matrix_from_rlist(int device, int64_t n, uint32_t p, int64_t nnz, const double* value,
                          const int32_t* col_idx, const int32_t* row_size, const double* labels,
                          fmx_matrix** out);
/* A dgCMatrix as it lies in R (Matrix package slots: x f64[nnz], i i32[nnz] 0-based ROW indices, p i32[ncol + 1] column pointers, Dim = (nrow, ncol)).
 * R/fm_matrix.R:26-33 transposes it on the host first (`Matrix::t(data)`, then the list above); here the slots go over as they are and the transposition
 * is a device sort by row (stable: the columns of a row come out ascending).  Row indices out of range, decreasing pointers and a pointer total other than nnz
 * are refused; nrow and nnz must be below 2^32 (one sort over all entries).  labels may be NULL. */
int fmx_matrix_from_dgc(int device, int64_t nrow, uint32_t ncol, int64_t nnz, const double* x, const int32_t* i, const int32_t* p, const double* labels,
                        fmx_matrix** out);
/* Plain CSR: row_ptr i64[n+1], col u32[nnz], val f32[nnz], y f32[n] or NULL. */
int fmx_matrix_from_csr(int device, int64_t n, uint32_t p, const int64_t* row_ptr, const uint32_t* col,
                        const float* val, const float* y, fmx_matrix** out);
/* Synthetic workload generated on the device: rows [row_offset, row_offset+n) of a
 * stream keyed by (seed, global row id): nnz_per_row stratified-uniform sorted distinct columns, value 1,
 * label +-1.  Shard-independent: a rank asks for its own row range. */
int fmx_matrix_synthetic(int device, int64_t n, uint32_t p, int32_t nnz_per_row, uint64_t seed,
                         int64_t row_offset, fmx_matrix** out);
/* Criteo-shaped synthetic rows (BASELINE.json configs[3]): n_dense always-present features (ids
 * 0..n_dense-1, value in [0,1)) followed by one one-hot feature from each of n_fields categorical fields (<= 64); field f owns
 * the next field_vocab[f] ids, the id inside a field is floor(vocab * u^skew) with u uniform (skew = 1: uniform; larger: a
 * power-law head, as in click logs).  Number of features = n_dense + sum(field_vocab); every row holds n_dense + n_fields
 * entries, ascending.  Keyed by (seed, global row id) like fmx_matrix_synthetic. */
typedef struct fmx_fields_spec {
  uint32_t struct_size;  /* = sizeof(fmx_fields_spec) */
  int32_t n_dense;
  int32_t n_fields;
  int32_t reserved;
  const uint32_t* field_vocab; /* [n_fields] */
  double skew;
  uint64_t seed;
} fmx_fields_spec;
int fmx_matrix_synthetic_fields(int device, int64_t n, const fmx_fields_spec* spec, int64_t row_offset, fmx_matrix** out);
/* Field-structured rows -- what fm.matrix makes of a data frame whose factor columns are one-hot encoded (R/fm_matrix.R: model.matrix keeps a
 * factor's dummy columns together): every row holds the n_dense always-present columns 0 .. n_dense-1 (any values), then exactly ONE id of each of
 * n_fields (<= 64) categorical fields, field c's ids lying in [field_base[c], field_base[c + 1]) with value 1 (field_base[0] = n_dense,
 * field_base[n_fields] = the feature count).  The layout is checked on the device (FMX_ERR_INVALID if a row differs); the inverted index of a step is
 * then built field by field -- a column's ids inside its field are sorted on ceil(log2 vocabulary) bits, the dense columns are not sorted at all --
 * instead of by one sort of all column ids: same plan, same results, about twice the planning rate.  The generators set it themselves,
 * and every uploaded matrix is looked at for it (rows of one length whose entry positions have disjoint ascending column ranges, values 1 outside a
 * leading run of always-present columns): this call is for a caller who knows the vocabularies to be wider than the ids that occur, or wants the check. */
int fmx_matrix_set_fields(fmx_matrix* m, int32_t n_dense, int32_t n_fields, const uint32_t* field_base);
/* Replace the labels of a device-resident matrix (y: f32[n] on the host): e.g. labels planted from a known model. */
int fmx_matrix_set_labels(fmx_matrix* m, const float* y);
/* SURVEY 8(d)'s other column laws (fmx_matrix_synthetic draws one column per stratum of [0, p)): nnz_per_row (<= 64) columns
 * i.i.d. over [0, p), sorted inside the row, repeats bumped to the next id.  law = FMX_COLUMNS_UNIFORM, or FMX_COLUMNS_ZIPF with
 * exponent `zipf_s` > 1 (1.05: a few features occur in most rows). */
#define FMX_COLUMNS_UNIFORM 1
#define FMX_COLUMNS_ZIPF 2
int fmx_matrix_synthetic_iid(int device, int64_t n, uint32_t p, int32_t nnz_per_row, uint64_t seed, int64_t row_offset, int32_t law,
                             double zipf_s, fmx_matrix** out);
/* SURVEY 8(d)'s ragged variant: row lengths Poisson(mean_nnz) clipped to [min_nnz, max_nnz] (the survey's "Poisson(30) clipped to [1, 64]"), columns
 * i.i.d. uniform over [0, p), sorted inside the row, repeats bumped; values 1, labels +-1; shard independent (keyed by the global row). */
int fmx_matrix_synthetic_ragged(int device, int64_t n, uint32_t p, double mean_nnz, int32_t min_nnz, int32_t max_nnz, uint64_t seed, int64_t row_offset,
                                fmx_matrix** out);
/* SURVEY 8(d)'s value variant: every stored value of a resident matrix redrawn uniform in (0, 1), keyed by (seed, global row = row_offset + r, entry) like the
 * generators above (a shard draws what the whole matrix would).  The matrix stops being one-hot: its plans are dropped and the kernels read the value arrays
 * from here on (util/Smatrix.h:44-61: the reference's values are real floats). */
int fmx_matrix_synthetic_values(fmx_matrix* m, uint64_t seed, int64_t row_offset);
int fmx_matrix_destroy(fmx_matrix* m);
int fmx_matrix_info(const fmx_matrix* m, int64_t* n, uint32_t* p, int64_t* nnz);
/* Copy rows [r0, r1) back to the host (row_ptr is rebased to 0); any pointer may be NULL. */
int fmx_matrix_export(const fmx_matrix* m, int64_t r0, int64_t r1, int64_t* row_ptr, uint32_t* col,
                      float* val, float* y);

/* ---- preprocessing on the device
 * SMatrix::scales (util/Smatrix.h:98-135, called at src/FM.cpp:36-38): z-score the stored entries of the listed columns
 * (ascending 0-based ids, as R passes `normalize - 1`) in place; mean/std: f64[p] outputs = Scales$mean / Scales$std. */
int fmx_matrix_scales(fmx_matrix* m, const int32_t* norm_columns, int64_t n_norm, double* mean, double* std);
/* SMatrix::normalize (util/Smatrix.h:137-153, src/FM.cpp:183-186): apply a fitted model's Scales to new data. */
int fmx_matrix_normalize(fmx_matrix* m, const double* mean, const double* std);

/* ---- the hot path */

/* Model::predict_batch / predict_prob (+ clamp) for every row; out: f64[n] on the host. */
int fmx_predict(fmx_engine* e, const fmx_matrix* m, double* out, int link);

/* Learner::learn(data): run max_iter EXAMPLES (the reference counts examples, solver/SGD_Learner.h:168-173).
 * SEQUENTIAL: rows visited as the reference visits them (libc rand() strides when random_step > 1).
 * MINIBATCH : consecutive batches of batch_rows rows, wrapping at the end of the matrix.
 * examples_done (may be NULL) receives the number actually processed. */
int fmx_train(fmx_engine* e, fmx_matrix* m, int64_t max_iter, int64_t* examples_done);
/* A GRID of models trained side by side in the reference's own algorithm (SGD_Learner::learn / FTRL_Learner::learn, one update per example in the reference's
 * visiting order).  NO COUNTERPART IN THE REFERENCE: its R/fm_select.R:22-66 picks the best SNAPSHOT of one fit's trace, it does not train a grid; this entry is
 * what a user's own loop of fm.train() calls over hyper-parameters becomes -- ONE launch per 65 536 examples with one workgroup per model (the reference-order
 * learner is a single workgroup bound by its scalar chain, DESIGN.md section 4: one model cannot use more of the chip, 64 or 256 models can).  An extension, not
 * a row of SURVEY section 8.  Every engine keeps its own parameters, optimizer state and hyper-parameters (learn_rate, regularisers, alpha / beta ...);
 * the engines share the feature count, factor.number, solver, task, device and random_step = 1 (ONE visiting order), and the matrix.  Each model's result is bit for
 * bit what fmx_train(engine, m, max_iter) alone gives it.  SGD (L1 / L2), FTRL and TDAP (the reference's default solver); rows of at most 32 entries (64 at k <= 32) with
 * ascending columns -- the shapes the windowed learners take. */
int fmx_train_grid(fmx_engine* const* engines, int32_t n_engines, fmx_matrix* m, int64_t max_iter, int64_t* examples_done);
/* Same, but with an explicit visiting order (row ids, SEQUENTIAL mode only). */
int fmx_train_order(fmx_engine* e, fmx_matrix* m, const int64_t* order, int64_t count);

/* Streamed training (BASELINE.json configs[3]: 4e9 rows x 33 M features do not fit the reference's uint32 offsets,
 * util/Smatrix.h:10-17, nor any memory): the rows [row_offset, row_offset + total_rows) of a synthetic stream are produced
 * step by step -- batch_rows rows are generated and their inverted index is built two steps ahead of the step that trains on
 * them (on a second stream beside the running step; FMX_STREAM_OVERLAP=0: behind it on the engine's stream -- DESIGN.md 6.5, 6.7); each step is trained
 * on once and dropped.  spec == NULL: the uniform generator of fmx_matrix_synthetic with nnz_per_row entries; otherwise the
 * Criteo-shaped one (nnz_per_row ignored).  Needs batch_rows <= the tile size (one tile per step).  ingest_wait_s (may be
 * NULL): host seconds spent waiting for a tile's counts.  On a cfg.n_gpus > 1 handle replica r streams rows
 * [r T / N, (r + 1) T / N) of the range on its own device and the replicas exchange per step (records of the occurring features
 * for sparse tiles, the dense buffer otherwise). */
int fmx_train_stream(fmx_engine* e, const fmx_fields_spec* spec, int32_t nnz_per_row, uint64_t seed, int64_t row_offset,
                     int64_t total_rows, int64_t* examples_done, double* ingest_wait_s);

/* The same stream step by step, for a driver that exchanges between the gradient sums and the update of every step (one process
 * per GPU: fmwr_amd/distributed.py; rank r opens rows [r T / N, (r + 1) T / N) -- the generators are keyed by the global row id).
 *   fmx_source_open   generates and plans the first two steps (enqueued on the engine's stream)
 *   fmx_source_next   hands out the next step as a matrix of ONE batch (step index 0 for fmx_step / fmx_grad / fmx_grad_compact /
 *                     fmx_owner_info ...), valid until the call after the next one; waits (host) only for that tile's counts, and
 *                     enqueues the generation and planning of the step after the next.  *step_matrix == NULL at the end.
 *   fmx_source_close  waits for the engine's stream and frees the stream; ingest_wait_s as above. */
typedef struct fmx_source fmx_source;
int fmx_source_open(fmx_engine* e, const fmx_fields_spec* spec, int32_t nnz_per_row, uint64_t seed, int64_t row_offset, int64_t total_rows,
                    fmx_source** out);
int fmx_source_next(fmx_source* s, fmx_matrix** step_matrix, int64_t* rows);
int fmx_source_close(fmx_source* s, double* ingest_wait_s);

/* ---- tracker (core/Tracker.h, the evaluation blocks of solver/SGD_Learner.h:140-176 and FTRL_Learner.h:118-154) */

/* metric ids are the reference's (util/Macros.h:24-29); evaluates() picks by task (core/Evaluation.h:20-41) */
#define FMX_EVAL_LL 0
#define FMX_EVAL_AUC 111
#define FMX_EVAL_ACC 222
#define FMX_EVAL_RMSE 333
#define FMX_EVAL_MSE 444
#define FMX_EVAL_MAE 555

/* track.control() (R/fm_track_control.R:20-26) as FM() plumbs it (src/FM.cpp:99-103) */
typedef struct fmx_track_config {
  uint32_t struct_size;  /* = sizeof(fmx_track_config) */
  int32_t metric;        /* FMX_EVAL_*: evaluate.metric */
  int64_t step_size;     /* evaluate + snapshot every step_size examples (> 0) */
  double convergence;    /* stop after 3 consecutive relative changes <= this (conv_condition) */
  int32_t keep_params;   /* 1: snapshot (w0, w, V) at every record like Tracker::record (core/Tracker.h:54-63); 0: metrics only */
  int32_t reserved;
} fmx_track_config;

/* Metric of the engine's current parameters on a data set: forward + link (probability for CLASSIFICATION, clamp to
 * the target range for REGRESSION) + evaluates().  What Tracker::report computes per snapshot (core/Tracker.h:70-94). */
int fmx_evaluate(fmx_engine* e, const fmx_matrix* m, int metric, double* out);

/* Learner::learn with the tracker on: as fmx_train, plus an evaluation on the training matrix after example 0,
 * step_size, 2*step_size, ... and after the last one; stops early when converged.  The trace stays in the engine
 * until the next fmx_train_tracked / fmx_set_params.  cfg.n_gpus > 1 (mini-batch learners): the record rule is applied to the
 * example indices a GLOBAL step covers (n_gpus * batch_rows of them) and the model is looked at on replica 0, over all of m. */
int fmx_train_tracked(fmx_engine* e, fmx_matrix* m, int64_t max_iter, const fmx_track_config* track,
                      int64_t* examples_done, int32_t* convergent);
int fmx_trace_size(fmx_engine* e, int64_t* n_records);
/* iters: the example index of each record (Trace$trace[[1]]), evals: Trace$evaluation.train */
int fmx_trace_get(fmx_engine* e, int64_t* iters, double* evals);
/* snapshot `record` (needs keep_params): same layouts as fmx_get_params */
int fmx_trace_params(fmx_engine* e, int64_t record, double* w0, double* w, double* v);

/* ---- step-level interface (what fmx_train loops over; used by bench.py and the multi-GPU driver).
 * All of these enqueue on the engine's stream and return without waiting; fmx_sync waits. */

/* number of steps (batches of batch_rows rows) the matrix splits into (builds the per-tile CSC on first use) */
int fmx_num_batches(fmx_engine* e, fmx_matrix* m, int64_t* n_batches);
/* one full mini-batch step on this GPU: forward -> gradient sums -> update, rows of batch `batch`
 * (rows_limit > 0 truncates the batch to its first rows_limit rows). */
int fmx_step(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit);
/* multi-GPU split of the same step: local gradient sums into the exchange buffer ... */
int fmx_grad(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit);
/* ... device pointer / element count of that buffer, for an in-place all-reduce(sum); elements are fp32, or fp64
 * with cfg.state_fp64 (fmx_grad_elem_bytes says which: 4 or 8) ... */
int fmx_grad_buffer(fmx_engine* e, void** dev_ptr, int64_t* n_floats);
int fmx_grad_elem_bytes(const fmx_engine* e, int32_t* bytes);
/* Pipelined form of the same split (cfg.exchange_chunks > 1).  The buffer is n_chunks blocks of chunk_elems elements,
 * block c holding the sums of features [c*chunk_features, (c+1)*chunk_features), followed at tail_offset by 4 elements
 * {sum of multipliers, sum of their squares, rows / 4096, rows % 4096} (two parts so that an fp32 sum over the ranks stays exact):
 *   fmx_grad_begin          forward of the whole step (all its tiles), writes the tail        -> all-reduce the tail
 *   fmx_grad_chunk(c)       gradient sums of block c over all tiles                            -> all-reduce block c (async)
 *   fmx_apply_chunk(c,..)   update of block c's features from the reduced block; `last` != 0 on the final call of the
 *                           step also applies the w0 / penalty-level update (every block reads the step's start scalars).
 * Results are identical to fmx_grad + fmx_apply (same sums in the same order). */
int fmx_grad_layout(fmx_engine* e, int64_t* n_chunks, int64_t* chunk_features, int64_t* chunk_elems, int64_t* tail_offset);
int fmx_grad_begin(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit);
int fmx_grad_chunk(fmx_engine* e, fmx_matrix* m, int64_t chunk);
int fmx_apply_chunk(fmx_engine* e, int64_t chunk, int64_t global_rows, int32_t last);
/* ... and the update from the (reduced) buffer; global_rows = rows of the whole global batch, or <= 0 to take the
 * count that travelled in the buffer's tail (each rank's fmx_grad wrote its own row count there; the all-reduce summed them). */
int fmx_apply(fmx_engine* e, int64_t global_rows);
/* ---- compact exchange (SURVEY 8(e) "collective sizing", BASELINE.json configs[3]: p = 33 M, k = 32 -> a dense buffer of 4.5 GB
 * per step).  When a step is one tile holding fewer entries than there are features (fmx_compact_info says `usable`), the
 * gradient sums of the features that OCCUR in the step are published as records instead:
 *     record = G[kp] | (Q[kp]: FTRL with FMX_REDUCE_SUM) | Gw | Qw | cnt | feature id      (record_elems elements, fp32 or fp64)
 * in ascending feature order, plus a 4-element tail {sum of multipliers, sum of squares, rows / 4096, rows % 4096}.
 *   fmx_grad_compact      forward + gradient sums of the step -> this rank's records and tail (enqueued)
 *   fmx_compact_records   device pointers of the records / the tail, and the record count (known on the host from ingest)
 *   -- the driver all-reduces the tail (sum) and all-gathers the records (padded to the largest count) --
 *   fmx_apply_compact     the gathered parts (part r: counts[r] records starting at record r * stride_records) are merged by
 *                         feature id, a feature's parts added in rank order, and the update is applied once per feature.
 * Two ranks give bitwise the result of the dense fmx_grad / all-reduce / fmx_apply step (tests/test_gpu_distributed.py). */
int fmx_compact_info(fmx_engine* e, fmx_matrix* m, int64_t* record_elems, int64_t* capacity, int32_t* usable);
int fmx_compact_count(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t* n_records); /* records step `batch` publishes */
int fmx_compact_reserve(fmx_engine* e, int64_t capacity); /* room for `capacity` records (>= every rank's own capacity) */
int fmx_grad_compact(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit);
int fmx_compact_records(fmx_engine* e, void** dev_records, int64_t* n_records, void** dev_tail);
int fmx_apply_compact(fmx_engine* e, const void* dev_records, const int64_t* counts, int32_t n_parts, int64_t stride_records,
                      int64_t global_rows);
/* The parts of fmx_apply_compact at explicit positions: part r holds counts[r] records starting at record starts[r] of the buffer
 * (what an all-to-all with uneven splits leaves behind). */
int fmx_apply_compact_parts(fmx_engine* e, const void* dev_records, const int64_t* counts, const int64_t* starts, int32_t n_parts,
                            int64_t global_rows);

/* ---- owner-sharded exchange (SURVEY 8(e) option (ii), BASELINE.json configs[3] on N GPUs).  Feature j BELONGS to rank j mod N: the
 * owner holds its current (V row, w) and its optimizer state; the other ranks hold copies that are refreshed when a step needs
 * them.  Per step, on every rank (the collectives are the driver's: fmwr_amd/distributed.py):
 *     ids    = the step's occurring features in owner-major order, counts[o] of them owned by rank o     (fmx_owner_info)
 *     pull   : all-to-all of the ids to their owners; an owner packs the rows asked for (fmx_rows_pack), all-to-all back,
 *              the asking rank stores them (fmx_rows_unpack): every row the step reads is the owner's current one
 *     sums   : fmx_grad_compact -- with fmx_owner_configure(N > 1) its records come out in the same owner-major order, so the
 *              part for owner o is one contiguous slice
 *     push   : all-to-all of the record slices to their owners (+ all-reduce of the 4-element tail)
 *     update : fmx_apply_compact_parts on the received parts: a feature's parts are added in rank order and the update is applied
 *              once, by its owner.
 * Per rank and step this moves about 2 x (N-1)/N x records instead of the all-gather's N x records, and equals it bit for bit
 * (the same additions in the same order).  w0 and the other scalars stay replicated (the tail is all-reduced).  After training,
 * a rank's copy of a feature it does not own is as old as the last step of its own that used it. */
int fmx_owner_configure(fmx_engine* e, int32_t n_owners, int32_t rank);
/* counts: i64[n_owners] records of step `batch` per owner; *dev_ids: u32[sum(counts)] the ids in owner-major order (device; valid
 * until the matrix's plans change, for a streamed step until the call after the next fmx_source_next) */
int fmx_owner_info(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t* counts, void** dev_ids);
/* (V row | w 0 0 0) of n features in the state's element type: row_elems = kp + 4 elements per feature (device buffers) */
int fmx_rows_pack(fmx_engine* e, const void* dev_ids_u32, int64_t n, void* dev_rows, int64_t* row_elems);
int fmx_rows_unpack(fmx_engine* e, const void* dev_ids_u32, int64_t n, const void* dev_rows);

int fmx_sync(fmx_engine* e);
/* the hipStream_t the engine launches on (as void*), so a caller can order its own work after it */
int fmx_stream(fmx_engine* e, void** stream);
/* device-side forward: y_hat (f64) for rows [r0, r1) into a device buffer the caller owns */
int fmx_predict_device(fmx_engine* e, const fmx_matrix* m, int64_t r0, int64_t r1, void* dev_out_f64, int link);

/* ---- ALS V-column sweep (solver/MCMC_ALS_Learner.h:272-354, ALS branch, one attribute group):
 * error: f64[n] residual on entry (y_hat - y, :520-527), updated in place; v_lambda, v_mu: f64[k] or NULL (zeros). */
int fmx_als_vsweep(fmx_engine* e, fmx_matrix* m, double* error, double alpha, const double* v_lambda,
                   const double* v_mu);
/* The MCMC (Gibbs) form of the same sweep (do_sample, :329-331): every coordinate is drawn from N(mean, var) instead of set
 * to the mean.  The reference draws with R's Rf_rnorm inside the loop; here the caller pre-draws the standard normals in the
 * same order -- std_normals: f64[k][p], element (f, j) at f*p + j, e.g. rnorm(k*p) under the same seed -- so the engine
 * needs no RNG and reproduces the reference's chain.  The hyper-prior draws (update_v_lambda / update_v_mu) stay with the
 * caller, who passes their current values in v_lambda / v_mu. */
int fmx_mcmc_vsweep(fmx_engine* e, fmx_matrix* m, double* error, double alpha, const double* v_lambda,
                    const double* v_mu, const double* std_normals);

/* Both sweeps with the residual RESIDENT on the engine's device (what a learner's loop and bench.py --solver als / mcmc use: no
 * host transfer per sweep): dev_error_f64 f64[n], updated in place; dev_std_normals_f64 f64[k][p] (element (f, j) at f*p + j) or
 * NULL for the ALS form.  v_lambda / v_mu stay host pointers (k scalars).  Returns when the sweep has finished. */
int fmx_vsweep_device(fmx_engine* e, fmx_matrix* m, void* dev_error_f64, double alpha, const double* v_lambda, const double* v_mu,
                      const void* dev_std_normals_f64);

/* The exact sweeps process the features in LEVELS (features of a level share no row, levels in ascending order reproduce the
 * reference's index-order Gauss-Seidel): how many levels (or, with cfg.als_max_levels exceeded, groups of the approximate
 * form: `approximate` = 1; with cfg.als_max_levels = -1 or -2, colours of the coloured order: `approximate` = 2, or 3 when the V sweep of a -2 plan nests feature-major -- light lists of at most 1 024 rows; 2 there means it nests factor outer, as -1) this matrix needs, the size of the largest, and every
 * feature's level / group / colour. */
int fmx_als_plan_info(fmx_engine* e, fmx_matrix* m, int64_t* levels, int64_t* largest_level, int32_t* approximate,
                      int32_t* level_of_feature /* [p] or NULL */);
/* Wide levels of an exact plan (every feature of the level holds at most 4096 entries, at least 2048 features: the fields of one-column-per-
 * field data) are swept in a ROW-TILED form on matrices of 2 M rows or more (fm_als_tiled.hip: per-tile sums against an L2-resident slice of
 * the (q, e) pairs, the coordinate steps, a row-major correction pass) instead of walking CSC columns against the whole table.  Same
 * arithmetic per entry; the two sums of a coordinate step associate differently (1e-10 against the column-walking form).  Reports how many
 * levels take that form (0: none), the rows per tile and the number of tiles; builds the plan if need be.  FMX_ALS_TILED=0 / 1 in the
 * environment forbids / forces the form wherever a level qualifies (1: any size, any width -- tests), FMX_ALS_TILE_ROWS sets the tile. */
int fmx_als_tiled_info(fmx_engine* e, fmx_matrix* m, int32_t* levels_tiled, int64_t* tile_rows, int32_t* n_tiles);
/* A COMPLETE tiled plan -- every level of the plan is a tiled one and every row holds exactly one feature of every level: one-column-per-field data, BASELINE.json
 * configs[4]'s shape -- lets the V sweep keep the (q, e) pairs physically in the list order of the level that consumes them next (the LEVEL-ORDER form,
 * fm_als_tiled.hip): per level one kernel streams the pairs, sums the lists and takes the coordinate steps (no per-tile partial sums), one kernel applies the
 * corrections and writes every pair to its place in the next level's order (a permutation inside the tile's L2-resident slice: the one random 16-byte access
 * per stored nonzero that is left).  Same arithmetic per entry as the other forms, sums associated in (tile, entry) order: 1e-10 against them and the oracle,
 * bitwise run to run.  Where, in addition, every feature's list fits a block of 8 192 rows (6 144 with real values), the sweep takes the BLOCK form
 * (fm_als_blocks.hip): the level's array is feature-block-major and ONE kernel per level streams a block's pairs into LDS, sums its lists, takes the coordinate
 * steps, corrects the pairs there and stores them as contiguous runs into the next level's blocks -- the pairs are read once per level and nothing waits for
 * another workgroup (202 against 120 M examples/s at configs[4]).  *level_order = 2: the block form (V sweep and w sweep), 1: the tile form (V sweep; the w sweep keeps the three-pass form), 0: neither.  FMX_ALS_ORDER=1 keeps the tile form, 0 forbids both. */
int fmx_als_order_info(fmx_engine* e, fmx_matrix* m, int32_t* level_order);
/* Opt-in (default off): carry q = X v_f from one V sweep to the next.  The reference recomputes q_f from scratch for every factor of every sweep
 * (solver/MCMC_ALS_Learner.h:283-300); here one forward pass builds it for all factors (38 GB of V-row gathers at configs[4]: 7 of a sweep's 49 ms).  But a sweep
 * itself keeps q current -- every correction of v_fj is applied to the rows' q (:341-350) -- so when a factor's last level is done its pairs hold X v_f for the NEW
 * v_f.  With on = 1 the block form writes that back into the table as the pairs move on, and the next V sweep on the same plan skips the forward pass if the V
 * table is bit for bit what the sweep left (a 64-bit fingerprint; set_params, training steps, another matrix or a rebuilt plan all force the rebuild, as does every
 * 64th sweep, against rounding drift: each carried sweep adds ~1e-16 relative per level).  Results agree with the rebuilt form to ~1e-13: within the 1e-10 of the
 * oracle tests, not bit for bit (the carried q keeps an ABSOLUTE rounding floor of ~1e-16 x the largest |q| since the last rebuild: a sweep that drives q towards zero by
 * many orders of magnitude sees it; ten sweeps at configs[4]: 1e-15 from the rebuilt form, profiles/r05_block_soak.txt).  The feature-major sweep
 * (cfg.als_max_levels = -2) keeps its row-major table current by construction and honours the switch the same way (the key also holds the matrix's value
 * generation: redrawn or rescaled values force the rebuild): 10 M x 1 M, k = 16: 223 -> 259 M examples/s on i.i.d. columns, 334 -> 404 M on field data.  Other
 * forms of the sweep ignore the switch. */
int fmx_als_carry_q(fmx_engine* e, int32_t on);

/* The ALS learner's training loop (MCMC_ALS_Learner::learn, :91-156; REGRESSION): max_iter times { forward; residual;
 * w0 update (:162-188); w sweep (:190-270, the exact one-thread form) }.  As shipped the reference never sweeps V (its
 * update_v call is commented out, :151-155): with_v = 0 reproduces that, with_v = 1 adds the V sweep after the w sweep.
 * The R-side ALS.solver parameters are overridden by learner->init() in the reference and do not enter (alpha = 1,
 * lambdas = 0).  Needs an FMX_MODE_SEQUENTIAL engine (fp64 tables). */
int fmx_als_train(fmx_engine* e, fmx_matrix* m, int32_t max_iter, int32_t with_v);
/* MCMC_ALS_Learner::learn for the MCMC learner (solver/MCMC_ALS_Learner.h:91-156, hyper-parameter draws :359-445), one
 * attribute group, REGRESSION and CLASSIFICATION.  The reference draws from R's generator inside the loop; a library has
 * no access to it, so the CALLER pre-draws, in the reference's call order, per iteration:
 *   std_gammas [max_iter][2]     standard (scale 1) Gamma variates of shape (1 + n)/2 and (1 + p + 1)/2
 *                                (update_alpha, update_w_lambda: Rf_rgamma(a, s) == s * such a variate),
 *   std_normals[max_iter][2 + p] standard normals for w0, w_mu and w[0..p)  (Rf_rnorm(m, s) == m + s * z).
 * Under R:  set.seed(s); for (it in 1:max_iter) { g1 <- rgamma(1, (1+n)/2); z0 <- rnorm(1); g2 <- rgamma(1, (2+p)/2);
 * zmu <- rnorm(1); zw <- rnorm(p) }  reproduces the reference's chain (slots of switched-off updates are not drawn).
 * The CLASSIFICATION residual subtracts truncated normals drawn from libc rand() row by row, as the reference does
 * (util/Random.h:20-93; one host thread).  V is never updated, as shipped (SURVEY A-1).  state_out[3]: alpha, w_lambda, w_mu. */
int fmx_mcmc_train(fmx_engine* e, fmx_matrix* m, int32_t max_iter, const double* std_gammas, const double* std_normals, double* state_out);

/* The same chain continued: state_io[3] = {alpha, w_lambda, w_mu} on entry (what an earlier call returned) and on return.  Lets the
 * caller interleave evaluations with iterations -- the tracker block of MCMC_ALS_Learner::learn (:96-125) is, per record point,
 * fmx_evaluate(...) followed by fmx_mcmc_train_from(e, m, 1, gammas + 2*it, normals + (2+p)*it, state). */
int fmx_mcmc_train_from(fmx_engine* e, fmx_matrix* m, int32_t max_iter, const double* std_gammas, const double* std_normals, double* state_io);
/* V hyper-priors of the MCMC / ALS learners: update_v_lambda then update_v_mu (solver/MCMC_ALS_Learner.h:448-517; in the shipped
 * code their caller is commented out together with the V sweep, :151-155), one attribute group.  v_lambda, v_mu: f64[k] in/out
 * (what fmx_mcmc_vsweep / fmx_als_vsweep take).  sample != 0 (MCMC): std_gammas[k] standard Gamma variates of shape (2 + p)/2,
 * std_normals[k] standard normals, in factor order -- under R: g <- rgamma(k, (2+p)/2); z <- rnorm(k).  sample == 0 (ALS): the
 * means, no variates.  The shipped update_v_mu sums v(f, attr_group[i]) instead of v(f, i) (:462): kept. */
int fmx_mcmc_v_hyper(fmx_engine* e, const double* std_gammas, const double* std_normals, double* v_lambda, double* v_mu, int32_t sample);

/* ---- measurement: HIP-event timing of each kernel on the engine's stream (bench.py roofline leg). */
#define FMX_KERNEL_ROWS_FORWARD 0 /* phase 1: V-row gather + forward + grad multiplier */
#define FMX_KERNEL_COLS_UPDATE 1  /* phase 2: per-feature gradient sums + update */
#define FMX_KERNEL_SCALAR 2       /* w0 reduction/update */
#define FMX_KERNEL_SEQ 3          /* sequential-exact learner */
#define FMX_KERNEL_ALS_SWEEP 4    /* one level (or group) of one factor of an ALS / MCMC sweep: als_level_k and its heavy-column forms */
#define FMX_KERNEL_COUNT 8
/* on == 0: off; on == n > 0: time every n-th launch of each kernel (n = 1: all; sampling keeps the events' own cost,
 * a few microseconds of stream time per timed launch, out of the measured throughput). */
int fmx_profile_enable(fmx_engine* e, int on);
int fmx_profile_get(fmx_engine* e, int kernel, double* total_ms, int64_t* launches);
int fmx_profile_reset(fmx_engine* e);
/* Phase 1's request schedule for large steps (>= 65536 rows per launch), which the engine picks by timing its own first 14
 * such launches: *serial = 1 one entry's requests outstanding per lane group, 0 four entries', -1 not decided yet;
 * ms_serial / ms_pipelined = the six timed launches of each.  The choice never changes a result.  FMX_ROWS_SERIAL=0/1 pins it. */
int fmx_rows_tune_info(fmx_engine* e, int32_t* serial, double* ms_serial, double* ms_pipelined);
/* Which form of phase 1 large steps (and large forward passes) take on this matrix: 0 one lane group per row (the product form), 1 the flat form
 * (FMX_ROWS_FLAT=1 on rows of differing lengths: the entries of a block of rows as one stream cut evenly over the lane groups, a row's pieces combined in
 * entry order; replaces the per-row loop of core/Model.h:83-97), 2 lane groups pulling rows (FMX_ROWS_PULL=1).  Forms 1 and 2 are measurement records
 * (both slower, profiles/r04_ragged_forms.txt); form 1 associates a row's sums differently from forms 0 and 2 (same parity bars, other last bits).  Under it a
 * row's bits depend on the matrix it is launched on: the shards of a cfg.n_gpus handle are re-based copies whose blocks of rows are cut elsewhere, so an N-GPU run
 * and a one-GPU run of the same data differ in last bits under FMX_ROWS_FLAT=1 (they are bitwise equal under the product form), and the switch is read at every
 * launch -- set it before the first call and leave it. */
int fmx_matrix_rows_form(const fmx_matrix* m, int32_t* form);
/* How the engine laid out its parameter tables: elements between consecutive features' V rows, and whether a feature's linear weight
 * sits inside its V row (fp32 mini-batch tables of at most 16 padded factors, from 3 M features up: out of the caches a nonzero then
 * costs one memory request instead of two; FMX_W_IN_ROW=0/1 in the environment overrides).  Never changes a result. */
int fmx_layout_info(fmx_engine* e, int32_t* v_row_stride, int32_t* w_in_row);
/* A cfg.n_gpus > 1 handle: replicas, whether they share one device (rehearsal), the ordered device pairs (a, b), a != b, and for how many of them
 * direct peer access could be enabled at creation (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess: the shards, fmx_set_params and the
 * owner-sharded exchange move data with peer copies, which go over xGMI only then), and the exchange a step of one sparse tile takes unless
 * FMX_GROUP_EXCHANGE says otherwise (1: all-gather of the occurring features' records; 2: owner-sharded -- the default only where the replicas share
 * a device until a run on two or more devices has confirmed it bitwise).  A one-GPU handle reports n = 1 and zeros. */
int fmx_group_info(fmx_engine* e, int32_t* n_replicas, int32_t* share_device, int32_t* peer_pairs, int32_t* peer_pairs_direct, int32_t* sparse_exchange);
/* RCCL smoke test for cfg.n_gpus > 1: loads librccl, ncclCommInitAll over devices 0..n-1, one grouped fp32 and fp64
 * all-reduce(sum) of 1000 elements per rank on per-device streams, checked against the closed form; max_err = largest deviation. */
int fmx_rccl_selftest(int32_t n, double* max_err);
/* What the memory system gives the hot kernels' access pattern and nothing else: uniformly random rows of row_bytes bytes
 * (16..256, a power of two) from a table of table_bytes bytes; ids are generated in registers, row_bytes / 16 lanes fetch a row,
 * in_flight (1, 2, 4 or 8) rows outstanding per lane, n_groups lane groups each summing per_group rows, `reps` launches timed with
 * HIP events.  bench.py reports kernel rows/s divided by this figure as "ceiling_frac".  Environment switches of the probe (measurement only; profiles/
 * r04_gather_granularity.txt, r04_phase1_l2.txt): FMX_PROBE_STRATA=1 fetch j of every lane group comes from stratum j of the table (the order in which phase 1 walks
 * rows of one column per stratum); FMX_PROBE_SIDE=1 a 4-byte word of a second table under the same id beside every row (phase 1's w); FMX_PROBE_LOAD=1/2/3 non-temporal /
 * system-scope / both buffer loads; FMX_PROBE_UNCACHED=1 the table in hipDeviceMallocUncached memory. */
int fmx_measure_gather(int device, int64_t table_bytes, int32_t row_bytes, int64_t n_groups, int32_t per_group, int32_t in_flight,
                       int32_t reps, double* rows_per_s);
/* the same probe held to at most 160 KiB / lds_bytes workgroups per CU: how many requests in flight the ceiling needs */
int fmx_measure_gather_occ(int device, int64_t table_bytes, int32_t row_bytes, int64_t n_groups, int32_t per_group, int32_t in_flight,
                           int32_t reps, int32_t lds_bytes, double* rows_per_s);

/* the same gather driven by a MATRIX: lane group g fetches the table row of every column id of row r0 + g (rows [r0, r0 + nrows) of m) from
 * a scratch table of table_rows rows -- phase 1's own access stream with the arithmetic stripped away.  For skewed columns (most fetches
 * served on-die) this, not the uniformly random probe, is the ceiling bench.py divides by. */
int fmx_measure_gather_matrix(fmx_matrix* m, int64_t r0, int64_t nrows, int64_t table_rows, int32_t row_bytes, int32_t in_flight,
                              int32_t reps, double* rows_per_s);

#ifdef __cplusplus
}
#endif
#endif /* FMX_H_ */
