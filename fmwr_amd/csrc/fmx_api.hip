// Host side of libfmx.so: the extern "C" entry points of include/fmx.h.
// What FM() / FMPredict() do around learner->learn() and fm.predict_batch() in the reference
// (src/FM.cpp:7-173, :177-214) minus the R list (un)marshalling, which stays in the Rcpp glue.
// There is no CPU fallback anywhere in this library: without a usable HIP device every entry point fails.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <atomic>
#include <memory>

#include "fmx_internal.h"
#include "fmx_test_hooks.h"
#include "fm_probit.h"

namespace fmx {

static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
}

void prof_begin(fmx_engine* e, int kernel) {
  e->prof_open = 0;
  if (!e->profile) return;
  if ((e->prof_seen[kernel]++ % e->profile) != 0) return;  // sampled: events between kernels cost ~µs of stream time
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  (void)hipEventRecord(a, e->stream);
  e->prof_pending.push_back({kernel, {a, b}});
  e->prof_open = 1;
}

void prof_end(fmx_engine* e) {
  if (!e->prof_open || e->prof_pending.empty()) return;
  e->prof_open = 0;
  (void)hipEventRecord(e->prof_pending.back().second.second, e->stream);
}

static int prof_collect(fmx_engine* e) {
  if (e->prof_pending.empty()) return FMX_OK;
  FMX_HIP(hipStreamSynchronize(e->stream));
  for (auto& pr : e->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.second.first, pr.second.second) == hipSuccess) {
      e->prof_ms[pr.first] += ms;
      e->prof_n[pr.first] += 1;
    }
    (void)hipEventDestroy(pr.second.first);
    (void)hipEventDestroy(pr.second.second);
  }
  e->prof_pending.clear();
  return FMX_OK;
}

static int use_device(int device) {
  int count = 0;
  hipError_t err = hipGetDeviceCount(&count);
  if (err != hipSuccess || count <= 0) {
    set_error("no HIP device available (%s); libfmx has no CPU fallback", err == hipSuccess ? "device count 0" : hipGetErrorString(err));
    return FMX_ERR_NOGPU;
  }
  FMX_CHECK(device >= 0 && device < count, FMX_ERR_INVALID, "device %d out of range (0..%d)", device, count - 1);
  FMX_HIP(hipSetDevice(device));
  return FMX_OK;
}

template <typename T>
static int dev_alloc_zero(T** ptr, size_t count) {
  const size_t bytes = (count ? count : 1) * sizeof(T);
  FMX_HIP(hipMalloc(ptr, bytes));
  FMX_HIP(hipMemset(*ptr, 0, bytes));
  return FMX_OK;
}

// solver/SGD_Learner.h:44-59 (regularisation mode) and FM.cpp:87-144 (plumbing) as one struct for the kernels
static int make_hyper(const fmx_config& c, Hyper* h) {
  h->task = c.task;
  h->k0 = c.keep_w0 != 0;
  h->k1 = c.keep_w1 != 0;
  h->lr = c.learn_rate;
  h->reg0 = c.l2_w0;
  h->l1w = c.l1_w1; h->l1v = c.l1_v; h->l2w = c.l2_w1; h->l2v = c.l2_v;
  h->alpha_w = c.alpha_w; h->alpha_v = c.alpha_v; h->beta_w = c.beta_w; h->beta_v = c.beta_v;
  h->min_t = c.min_target; h->max_t = c.max_target;
  h->mean = (c.batch_reduce == FMX_REDUCE_MEAN);
  h->egamma = std::exp(-c.gamma);  // solver/TDAP_Learner.h:83
  h->gamma = c.gamma;
  if (c.solver == FMX_SOLVER_FTRL || c.solver == FMX_SOLVER_TDAP) {
    h->kind = c.solver == FMX_SOLVER_FTRL ? UPD_FTRL : UPD_TDAP;
    h->regw = 0; h->regv = 0;
  } else {
    bool l1 = false;
    if (c.l1_w1 > 0 || c.l1_v > 0) {  // SGD_Learner.h:46-51: any L1 > 0 => L1 rates are used, L2 dropped
      l1 = true;
      h->regw = c.l1_w1; h->regv = c.l1_v;
    } else {
      h->regw = c.l2_w1; h->regv = c.l2_v;
    }
    if (c.task != FMX_TASK_CLASSIFICATION) l1 = false;  // SGD_Learner.h:57-59 (the L1 rates then act as L2, SURVEY A-9)
    h->kind = l1 ? UPD_SGD_L1 : UPD_SGD_L2;
  }
  h->decay_w = 1.0 - h->lr * h->regw;
  h->decay_v = 1.0 - h->lr * h->regv;
  h->log_decay_w = h->decay_w > 0 ? std::log(h->decay_w) : 0.0;
  h->log_decay_v = h->decay_v > 0 ? std::log(h->decay_v) : 0.0;
  return FMX_OK;
}

static bool seq_mode(const fmx_engine* e) { return e->cfg.mode == FMX_MODE_SEQUENTIAL; }

static int reset_optimizer_state(fmx_engine* e) {
  const size_t p = e->p;
  double zeros[SC_COUNT] = {0};
  double w0 = 0;
  FMX_HIP(hipMemcpy(&w0, e->scal + SC_W0, sizeof(double), hipMemcpyDeviceToHost));
  zeros[SC_W0] = w0;
  FMX_HIP(hipMemcpy(e->scal, zeros, sizeof(zeros), hipMemcpyHostToDevice));
  if (e->sV) FMX_HIP(hipMemset(e->sV, 0, p * e->kp32 * sizeof(float)));
  if (e->sw) FMX_HIP(hipMemset(e->sw, 0, p * sizeof(float)));
  if (e->nV) FMX_HIP(hipMemset(e->nV, 0, p * e->kp32 * sizeof(float)));
  if (e->nw) FMX_HIP(hipMemset(e->nw, 0, p * sizeof(float)));
  for (float* t : {e->t1V, e->t2V, e->t3V}) if (t) FMX_HIP(hipMemset(t, 0, p * e->kp32 * sizeof(float)));
  for (float* t : {e->t1w, e->t2w, e->t3w}) if (t) FMX_HIP(hipMemset(t, 0, p * sizeof(float)));
  if (e->dsV) FMX_HIP(hipMemset(e->dsV, 0, p * e->kp64 * sizeof(double)));
  if (e->dsw) FMX_HIP(hipMemset(e->dsw, 0, p * sizeof(double)));
  if (e->dnV) FMX_HIP(hipMemset(e->dnV, 0, p * e->kp64 * sizeof(double)));
  if (e->dnw) FMX_HIP(hipMemset(e->dnw, 0, p * sizeof(double)));
  for (double* t : {e->dt1V, e->dt2V, e->dt3V}) if (t) FMX_HIP(hipMemset(t, 0, p * e->kp64 * sizeof(double)));
  for (double* t : {e->dt1w, e->dt2w, e->dt3w}) if (t) FMX_HIP(hipMemset(t, 0, p * sizeof(double)));
  return FMX_OK;
}

// S / multiplier rows for `s_rows` rows (one tile, or a whole step for the chunked exchange) and phase 1's per-workgroup
// partial sums for one step; grow-only
static int ensure_workspace(fmx_engine* e, int64_t s_rows, int64_t step_rows, int64_t tiles_per_step) {
  const int rpw = WG_THREADS / mb_lpr(e);
  // sum over tiles of ceil(rows_t / rpw); a small step's one-wave workgroups (rows_wg_threads) write at most 2048 of them (4096 with four lane groups per row)
  const int64_t partials = (step_rows + rpw - 1) / rpw + (tiles_per_step > 0 ? tiles_per_step : 1) + 8 + 4096;
  if (s_rows <= e->ws_rows && partials <= e->ws_partials) return FMX_OK;
  FMX_HIP(hipStreamSynchronize(e->stream));
  (void)hipFree(e->S); (void)hipFree(e->amul); (void)hipFree(e->partials);
  e->S = nullptr; e->amul = nullptr; e->partials = nullptr; e->ws_rows = 0; e->ws_partials = 0;
  FMX_HIP(hipMalloc(&e->S, (size_t)s_rows * mb_kp(e) * mb_elem(e)));
  FMX_HIP(hipMalloc(&e->amul, (size_t)s_rows * mb_elem(e)));
  FMX_HIP(hipMalloc(&e->partials, (size_t)partials * 2 * sizeof(double)));
  e->ws_rows = s_rows;
  e->ws_partials = partials;
  return FMX_OK;
}

static int ensure_gbuf(fmx_engine* e) {
  if (e->gbuf) return FMX_OK;
  // blocks of F features: GV [F][kp] | GW [F] | CNT [F] | (QV [F][kp] | QW [F]: only FTRL with FMX_REDUCE_SUM needs
  // sum(g^2)); then tail[4].  One block holding all p features unless cfg.exchange_chunks > 1.
  const bool has_q = exchange_has_q(e);
  const int64_t chunks = e->cfg.exchange_chunks > 1 ? e->cfg.exchange_chunks : 1;
  const int64_t per = ((int64_t)e->p + chunks - 1) / chunks;
  e->gb_feats = chunks > 1 ? (per + 63) / 64 * 64 : ((int64_t)e->p + 3) / 4 * 4;  // keeps every plane 16-byte aligned
  e->gb_blocks = ((int64_t)e->p + e->gb_feats - 1) / e->gb_feats;
  e->gb_block_elems = e->gb_feats * mb_kp(e) * (has_q ? 2 : 1) + e->gb_feats * (has_q ? 3 : 2);
  e->gbuf_floats = e->gb_blocks * e->gb_block_elems + 4;
  FMX_HIP(hipMalloc(&e->gbuf, (size_t)e->gbuf_floats * mb_elem(e)));
  FMX_HIP(hipMemset(e->gbuf, 0, (size_t)e->gbuf_floats * mb_elem(e)));
  FMX_HIP(hipDeviceSynchronize());
  return FMX_OK;
}

// The reference's visiting order, solver/SGD_Learner.h:86-88 with util/Random.h:20-24,126-132: strides from libc rand()
// (unseeded in the reference, SURVEY A-4), row 0 skipped when random_step == 1 (A-2).
static uint32_t random_select(int n) {
  if (n == 1) return 1;
  return (uint32_t)((rand() / ((double)RAND_MAX + 1)) * n + 1);
}

// Resumable form of the loop `for(;;) for (i = random_select(step); i < n; i += random_select(step))`: next(count) appends the
// next `count` visited rows, drawing from libc rand() exactly as many times and in the same order as the reference does, so
// a long run can be produced (uploaded, trained on) in bounded pieces instead of one max_iter-sized array.
struct VisitOrder {
  int64_t n;
  int step;
  uint64_t i = 0;
  bool in_pass = false;
  int idle = 0;
  VisitOrder(int64_t n_, int step_) : n(n_), step(step_) {}
  // false when nothing can ever be visited (n <= 1)
  bool pending = false;  // the stride after the last visited row is drawn only when the next row is asked for (the reference
                         // leaves its loop by `break` BEFORE the increment: the libc stream must not run one draw ahead)
  bool next(int64_t count, std::vector<int64_t>* out) {
    out->clear();
    out->reserve((size_t)count);
    size_t pass_start = 0;
    while ((int64_t)out->size() < count) {
      if (!in_pass) { i = random_select(step); in_pass = true; pending = false; pass_start = out->size(); }
      else if (pending) { i += random_select(step); pending = false; }
      if (i < (uint64_t)n) {
        out->push_back((int64_t)i);
        pending = true;
        idle = 0;
      } else {
        in_pass = false;  // the pass ended: the outer for(;;) starts the next one with a fresh first stride
        if (out->size() == pass_start && ++idle > 1000) return false;
      }
    }
    return true;
  }
};

void free_matrix(fmx_matrix* m) {
  if (!m) return;
  (void)hipFree(m->row_ptr); (void)hipFree(m->col); (void)hipFree(m->val); (void)hipFree(m->y);
  drop_plans(m);
  (void)hipFree(m->col_ptr); (void)hipFree(m->crow); (void)hipFree(m->cval); (void)hipFree(m->als_feats); (void)hipFree(m->als_heavy); (void)hipFree(m->als_level_ptr_dev); (void)hipFree(m->als_rank);
  (void)hipFree(m->als_vh); (void)hipFree(m->als_vh_seg0); (void)hipFree(m->als_vseg_feat); (void)hipFree(m->als_vseg_b); (void)hipFree(m->als_vseg_e); (void)hipFree(m->als_vh_work);
  als_tiled_free(m);
  delete m;
}

static int alloc_matrix(int device, int64_t n, uint32_t p, int64_t nnz, bool labels, fmx_matrix** out) {
  FMX_CHECK(out != nullptr, FMX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMX_CHECK(n >= 0 && nnz >= 0, FMX_ERR_INVALID, "negative size");
  FMX_TRY(use_device(device));
  std::unique_ptr<fmx_matrix, void (*)(fmx_matrix*)> m(new fmx_matrix(), free_matrix);
  static std::atomic<uint64_t> next_uid{1};
  m->uid = next_uid.fetch_add(1);
  m->device = device; m->n = n; m->p = p; m->nnz = nnz; m->has_labels = labels;
  FMX_HIP(hipMalloc(&m->row_ptr, ((size_t)n + 1) * sizeof(int64_t)));
  // one spare entry, like the reference's over-read guard (SURVEY A-13); keeps zero-nnz matrices allocatable too
  FMX_HIP(hipMalloc(&m->col, ((size_t)nnz + 1) * sizeof(uint32_t)));
  FMX_HIP(hipMalloc(&m->val, ((size_t)nnz + 1) * sizeof(float)));
  FMX_HIP(hipMalloc(&m->y, ((size_t)n + 1) * sizeof(float)));
  FMX_HIP(hipMemset(m->y, 0, ((size_t)n + 1) * sizeof(float)));
  *out = m.release();
  return FMX_OK;
}

static int check_pair(const fmx_engine* e, const fmx_matrix* m) {
  FMX_CHECK(e && m, FMX_ERR_INVALID, "NULL handle");
  // core/Model.h:115: "number of input's features is not correct..."
  FMX_CHECK((uint64_t)m->p == e->p, FMX_ERR_INVALID, "number of input's features is not correct...");
  FMX_CHECK(m->device == e->cfg.device, FMX_ERR_INVALID, "matrix lives on device %d, engine on %d", m->device, e->cfg.device);
  return FMX_OK;
}

static int forward_rows(fmx_engine* e, const fmx_matrix* m, int64_t r0, int64_t r1, double* d_out, int link) {
  FMX_CHECK(link >= FMX_LINK_NONE && link <= FMX_LINK_PROBIT, FMX_ERR_INVALID, "unknown link %d", link);
  if (link == FMX_LINK_PROBIT) FMX_TRY(ensure_probit(e));
  RowsArgs a{};  // launch_rows_forward cuts the range into launches of 262 144 rows
  a.row_ptr = m->row_ptr; a.col = m->col; a.val = m->val; a.y = nullptr;
  a.r0 = r0; a.nrows = r1 - r0;
  if (wide_state(e)) { a.V = e->dV; a.w = e->dw; a.vs = e->kp64; a.ws = 1; }
  else { a.V = e->V; a.w = mb_wbase(e); a.vs = e->vstride32; a.ws = mb_wstride(e); }
  a.scal = e->scal;
  a.yhat = d_out;
  a.link = link;
  a.pn_y = e->probit;
  a.unit = m->unit_values;
  a.sort_rows = rows_ragged(m);
  a.flat = rows_flat(m); a.nmat = m->n;
  FMX_TRY(launch_rows_forward(e, a, false, wide_state(e)));
  return FMX_OK;
}

// util/Random.h:95-124's tables (fm_probit.h), generated from their formulas and uploaded on first use
int ensure_probit(fmx_engine* e) {
  if (e->probit) return FMX_OK;
  std::vector<double> t((size_t)PN_POINTS + 1 + DP_POINTS + 1);
  for (int i = 0; i < PN_POINTS; ++i) t[(size_t)i] = 0.5 * std::erfc(-pn_x(i) / std::sqrt(2.0));
  t[PN_POINTS] = t[PN_POINTS - 1];  // x == MAX reads one past the shipped table
  double* dp = t.data() + PN_POINTS + 1;
  for (int i = 0; i < DP_POINTS; ++i) {
    const double x = dp_x(i);
    // the shipped values carry this formula's cancellation and 12 decimals
    const double r = std::exp(-0.5 * x * x) / std::sqrt(2.0 * 3.14159265358979323846) / (1.0 - 0.5 * std::erfc(-x / std::sqrt(2.0)));
    dp[i] = std::round(r * 1e12) / 1e12;
  }
  dp[DP_POINTS] = dp[DP_POINTS - 1];
  FMX_HIP(hipMalloc(&e->probit, t.size() * sizeof(double)));
  FMX_HIP(hipMemcpy(e->probit, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
  return FMX_OK;
}

// rows-per-tile actually used: a step of batch_rows rows is cut into ceil(batch_rows / tile) equal-ish tiles
static int64_t effective_tile_rows(const fmx_engine* e) {
  // measured optimum (profiles/r02_sweep_tile.txt, 1M-row steps at configs[1]'s shape): 524288 rows from k = 16 up (k = 16: 865 ->
  // 910 M ex/s against 262144-row tiles, k = 32: 583 -> 651; the per-tile sweep over all p features is paid half as often), 262144
  // below (k = 8: 990 against 891 -- the rows are 32 bytes and the tile's S table is what has to stay cached); 1048576 loses everywhere
  const int64_t want = e->cfg.tile_rows > 0 ? e->cfg.tile_rows : (e->kp32 >= 16 ? 524288 : 262144);
  if (e->cfg.batch_rows <= want) return e->cfg.batch_rows;
  const int64_t tiles = (e->cfg.batch_rows + want - 1) / want;
  return (e->cfg.batch_rows + tiles - 1) / tiles;
}

struct TileRun {
  int64_t tile;    // global tile index
  int64_t r0;      // first row
  int64_t nrows;   // rows taking part
};

// the tiles of step `batch`, cut at rows_limit (> 0: only the first rows_limit rows of the step take part)
static int step_tiles(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit, std::vector<TileRun>* out, int64_t* step_rows) {
  FMX_CHECK(!seq_mode(e), FMX_ERR_STATE, "step interface needs FMX_MODE_MINIBATCH");
  FMX_CHECK(m->has_labels, FMX_ERR_STATE, "there are no labels in data");  // R/fm_train.R:72-74
  const int64_t tile = effective_tile_rows(e);
  FMX_TRY(build_batch_csc(m, e->cfg.batch_rows, tile, e->stream));
  FMX_CHECK(batch >= 0 && batch < m->n_batches, FMX_ERR_INVALID, "batch %lld out of range (0..%lld)", (long long)batch, (long long)m->n_batches - 1);
  const int64_t step_cap = e->cfg.batch_rows < m->n ? e->cfg.batch_rows : (m->n > 0 ? m->n : 1);  // rows of the largest step
  FMX_TRY(ensure_workspace(e, tile < step_cap ? tile : step_cap, step_cap, (step_cap + tile - 1) / tile));
  out->clear();
  int64_t left = rows_limit > 0 ? rows_limit : (int64_t)1 << 62;
  int64_t total = 0;
  for (int64_t t = m->step_first_tile[(size_t)batch]; t < m->step_first_tile[(size_t)batch + 1] && left > 0; ++t) {
    const auto& pl = m->plans[(size_t)t];
    int64_t nrows = pl.nrows;
    if (nrows > left) nrows = left;
    out->push_back({t, pl.r0, nrows});
    left -= nrows;
    total += nrows;
  }
  *step_rows = total;
  return FMX_OK;
}

static int rows_phase(fmx_engine* e, fmx_matrix* m, const TileRun& t, int64_t step_rows, int64_t partial_offset, int64_t* n_partials, int64_t s_row0 = 0) {
  RowsArgs a{};
  a.row_ptr = m->row_ptr; a.col = m->col; a.val = m->val; a.y = m->y;
  a.r0 = t.r0; a.nrows = t.nrows;
  a.V = mb_vbase(e); a.w = mb_wbase(e); a.vs = mb_vstride(e); a.ws = mb_wstride(e);
  a.scal = e->scal;
  a.S = (char*)e->S + (size_t)s_row0 * mb_kp(e) * mb_elem(e);
  a.amul = (char*)e->amul + (size_t)s_row0 * mb_elem(e);
  a.partials = e->partials + 2 * partial_offset;
  a.unit = m->unit_values;
  a.sort_rows = rows_ragged(m);
  a.wg_threads = rows_wg_threads(step_rows, mb_lpr(e));
  a.split = rows_split(step_rows, mb_lpr(e));
  a.flat = rows_flat(m); a.nmat = m->n;
  const int rpw = a.wg_threads / (mb_lpr(e) * (a.wg_threads == 64 && a.split == 4 ? 4 : 1));
  *n_partials = (a.flat == 1 && !a.sort_rows && a.wg_threads != 64) ? rows_flat_blocks(t.r0, t.nrows, rpw) : (t.nrows + rpw - 1) / rpw;
  return launch_rows_forward(e, a, true, mb_wide(e));
}

// phase-2 arguments of a tile.  sparse_ok: the launch touches no dense exchange buffer, so it may walk only the lists of
// the features that occur in the tile; otherwise it walks one list per feature (a dense directory is made on first use).
static int cols_args(fmx_engine* e, fmx_matrix* m, const TileRun& t, bool sparse_ok, ColsArgs* out) {
  const auto& pl = m->plans[(size_t)t.tile];
  ColsArgs c{};
  c.brow = m->brow + pl.base;
  c.bval = m->bval + pl.base;
  c.rows_active = (uint32_t)t.nrows;
  c.walk = 1;
  c.unit = m->unit_values;
  if (pl.feat && sparse_ok) {
    c.tfeat = pl.feat; c.toff = pl.soff; c.n_tfeat = pl.n_lists;
    c.trow0 = pl.row0; c.tval0 = pl.val0;
    c.list_entries = pl.cnt;
  } else {
    if (!pl.off) FMX_TRY(plan_ensure_dense(m, t.tile, e->stream));
    c.bptr = m->plans[(size_t)t.tile].off;
  }
  *out = c;
  return FMX_OK;
}

// heavy hitters of a tile: the long-list plan and its partial-sum buffer
static int long_args(fmx_engine* e, fmx_matrix* m, int64_t tile, LongArgs* la, ColsArgs* c) {
  const auto& pl = m->plans[(size_t)tile];
  if (pl.n_long == 0) return FMX_OK;
  const int64_t need = m->max_long_seg * (2 * (int64_t)mb_kp(e) + 4);
  if (need > e->long_partial_cap) {
    FMX_HIP(hipStreamSynchronize(e->stream));
    (void)hipFree(e->long_partial); e->long_partial = nullptr; e->long_partial_cap = 0;
    FMX_HIP(hipMalloc(&e->long_partial, (size_t)need * sizeof(double)));
    e->long_partial_cap = need;
  }
  *la = LongArgs{pl.lfeat, pl.lpos, pl.lseg_ptr, pl.seg_list, pl.seg_begin, pl.seg_end, e->long_partial, (int64_t)pl.n_long, (int64_t)pl.n_seg};
  c->long_min = list_long_min();
  return FMX_OK;
}

// run the tiles of one step: every tile accumulates; `finish_local` applies the update after the last tile (single GPU),
// otherwise the sums (and the partial-sum tail) are left in the exchange buffer for the all-reduce
static int run_step(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit, bool finish_local, bool empty_share = false) {
  std::vector<TileRun> tiles;
  int64_t step_rows = 0;
  FMX_TRY(step_tiles(e, m, batch, rows_limit, &tiles, &step_rows));
  if (empty_share) { tiles.clear(); step_rows = 0; }
  if (tiles.empty()) {
    if (finish_local) return FMX_OK;
    tiles.push_back({m->step_first_tile[(size_t)batch], 0, 0});  // an empty share still has to publish zeros
  }
  const bool single = tiles.size() == 1;
  if (!(single && finish_local)) FMX_TRY(ensure_gbuf(e));
  int64_t partials = 0;
  for (size_t i = 0; i < tiles.size(); ++i) {
    int64_t np = 0;
    FMX_TRY(rows_phase(e, m, tiles[i], step_rows, partials, &np));
    partials += np;
    const bool last = i + 1 == tiles.size();
    // a tile that is a whole fused step walks only the features occurring in it (the exchange buffer is dense: tiles
    // that read or write it visit every feature)
    ColsArgs c{};
    FMX_TRY(cols_args(e, m, tiles[i], single && finish_local, &c));
    c.load_gbuf = i > 0;
    c.store_gbuf = !(last && finish_local);
    c.apply = last && finish_local;
    c.scalar = !last ? SCALAR_NONE : (finish_local ? SCALAR_FUSED : SCALAR_PUBLISH);
    c.n_partials = last ? partials : 0;
    c.global_rows = (double)step_rows;
    LongArgs la{};
    FMX_TRY(long_args(e, m, tiles[i].tile, &la, &c));
    FMX_TRY(launch_cols_update(e, c, la));
  }
  return FMX_OK;
}

// ---- chunked exchange (cfg.exchange_chunks > 1): phase 1 of the whole step first, then phase 2 one feature block at a
// time over all the step's tiles, so that the all-reduce of a block overlaps phase 2 of the next one ---------------------
static int grad_begin(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit) {
  std::vector<TileRun> tiles;
  int64_t step_rows = 0;
  FMX_TRY(step_tiles(e, m, batch, rows_limit, &tiles, &step_rows));
  FMX_TRY(ensure_gbuf(e));
  // the whole step's S / multipliers stay resident until its last block has been walked
  const int64_t cap = e->cfg.batch_rows < m->n ? e->cfg.batch_rows : (m->n > 0 ? m->n : 1);
  FMX_TRY(ensure_workspace(e, cap, cap, (int64_t)tiles.size()));
  e->open_tiles.clear();
  int64_t partials = 0, s_row0 = 0;
  for (const TileRun& t : tiles) {
    int64_t np = 0;
    FMX_TRY(rows_phase(e, m, t, step_rows, partials, &np, s_row0));
    partials += np;
    e->open_tiles.push_back({t.tile, t.r0, t.nrows, s_row0});
    s_row0 += t.nrows;
  }
  e->open_rows = step_rows;
  e->open_partials = partials;
  e->open_matrix = m;
  e->open_generation = m->plan_generation;
  // the tail {sum mult, sum mult^2, rows, 0} is known after phase 1: publish it now so that it can travel first
  ColsArgs c{};
  c.f0 = c.f1 = (uint32_t)e->p;  // no features: workgroup 0's scalar work only
  c.scalar = SCALAR_PUBLISH;
  c.n_partials = partials;
  c.global_rows = (double)step_rows;
  return launch_cols_update(e, c, LongArgs{});
}

static int grad_block(fmx_engine* e, fmx_matrix* m, int64_t block) {
  FMX_CHECK(e->open_matrix == m && m != nullptr, FMX_ERR_STATE, "fmx_grad_chunk needs a preceding fmx_grad_begin on the same matrix");
  // the open step holds tile indices into the matrix's plans: another engine's tiling, fmx_matrix_scales / _normalize or a
  // failed rebuild since fmx_grad_begin replaced them
  FMX_CHECK(e->open_generation == m->plan_generation, FMX_ERR_STATE, "the matrix's tile plans changed since fmx_grad_begin (another tiling, or scales / normalize): begin the step again");
  FMX_CHECK(block >= 0 && block < e->gb_blocks, FMX_ERR_INVALID, "chunk %lld out of range (0..%lld)", (long long)block, (long long)e->gb_blocks - 1);
  const uint32_t f0 = (uint32_t)(block * e->gb_feats);
  const uint32_t f1 = (uint32_t)(((block + 1) * e->gb_feats < (int64_t)e->p) ? (block + 1) * e->gb_feats : (int64_t)e->p);
  if (e->open_tiles.empty()) {  // an empty share still publishes zeros
    ColsArgs c{};
    c.f0 = f0; c.f1 = f1; c.store_gbuf = 1;
    return launch_cols_update(e, c, LongArgs{});
  }
  for (size_t i = 0; i < e->open_tiles.size(); ++i) {
    const auto& ot = e->open_tiles[i];
    const TileRun t{ot.tile, ot.r0, ot.nrows};
    ColsArgs c{};
    FMX_TRY(cols_args(e, m, t, false, &c));
    c.f0 = f0; c.f1 = f1;
    c.s_row0 = ot.s_row0;
    c.load_gbuf = i > 0;
    c.store_gbuf = 1;
    c.global_rows = (double)e->open_rows;
    LongArgs la{};
    FMX_TRY(long_args(e, m, t.tile, &la, &c));
    FMX_TRY(launch_cols_update(e, c, la));
  }
  return FMX_OK;
}

static int apply_block(fmx_engine* e, int64_t block, int64_t global_rows, bool last) {
  FMX_CHECK(e->gbuf != nullptr, FMX_ERR_STATE, "fmx_apply_chunk needs a preceding fmx_grad_begin");
  FMX_CHECK(block >= 0 && block < e->gb_blocks, FMX_ERR_INVALID, "chunk %lld out of range (0..%lld)", (long long)block, (long long)e->gb_blocks - 1);
  ColsArgs c{};
  c.f0 = (uint32_t)(block * e->gb_feats);
  c.f1 = (uint32_t)(((block + 1) * e->gb_feats < (int64_t)e->p) ? (block + 1) * e->gb_feats : (int64_t)e->p);
  c.load_gbuf = 1;
  c.apply = 1;
  // every block reads the step's START scalars (penalty level, w0 ...): only the last launch writes the next ones
  c.scalar = last ? SCALAR_FROM_TAIL : SCALAR_NONE;
  c.global_rows = (double)global_rows;
  if (last) e->open_matrix = nullptr;
  return launch_cols_update(e, c, LongArgs{});
}

// ---- compact exchange: steps of ONE sparse tile publish a record per occurring feature instead of the dense buffer -----
static int ensure_compact(fmx_engine* e, int64_t cap) {
  const bool has_q = exchange_has_q(e);
  e->rec_elems = mb_kp(e) * (has_q ? 2 : 1) + 4;
  if (!e->ctail) {
    FMX_HIP(hipMalloc(&e->ctail, 4 * mb_elem(e)));
    // ON THE ENGINE'S STREAM: a null-stream memset is not ordered against a non-blocking stream, and when the first fmx_grad_compact of an
    // engine is also the call that creates the tail (no fmx_compact_reserve before it: the owner-sharded driver), the zero fill could land
    // AFTER the kernel had published the step's tail -- one rank then applied its first step with a zero tail (w0 = -0, found in round 3
    // through a per-step trace of w0 across three ranks: replicas that disagree after a collective that hands everyone the same bytes)
    FMX_HIP(hipMemsetAsync(e->ctail, 0, 4 * mb_elem(e), e->stream));
  }
  if (cap > e->crec_cap) {
    FMX_HIP(hipStreamSynchronize(e->stream));
    (void)hipFree(e->crec); e->crec = nullptr; e->crec_cap = 0;
    FMX_HIP(hipMalloc(&e->crec, (size_t)cap * e->rec_elems * mb_elem(e)));
    e->crec_cap = cap;
  }
  return FMX_OK;
}

// most records one step of this matrix can publish
static int64_t compact_capacity(const fmx_matrix* m) {
  int64_t cap = 1;
  for (const auto& pl : m->plans) { const int64_t c = pl.feat ? (int64_t)pl.cap_lists : (int64_t)m->p; if (c > cap) cap = c; }
  return cap;
}

// the owner-major order of a resident tile, built on first use (a streamed tile's comes with its ingest: fmx_source_open)
static int ensure_owner_plan(fmx_engine* e, fmx_matrix* m, int64_t tile) {
  auto& pl = m->plans[(size_t)tile];
  if (pl.own_n == e->owner_parts) return FMX_OK;
  FMX_CHECK(pl.feat != nullptr, FMX_ERR_STATE, "the owner-sharded exchange needs sparse tiles (fewer entries than features per tile)");
  OwnerWorkspace ws;
  FMX_TRY(ws.reserve(pl.n_lists, e->stream));
  FMX_TRY(plan_owner_alloc(pl, pl.n_lists));
  FMX_TRY(plan_owner_build(pl, ws, e->owner_parts, pl.n_lists, e->stream));
  uint32_t h[OWNERS_MAX + 2];
  FMX_HIP(hipMemcpyAsync(h, pl.own_counts, sizeof(h), hipMemcpyDeviceToHost, e->stream));
  FMX_HIP(hipStreamSynchronize(e->stream));  // (also keeps the workspace alive until the sort has run)
  for (int o = 0; o <= OWNERS_MAX; ++o) pl.own_counts_h[o] = h[o];
  return FMX_OK;
}

static int grad_compact(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit, bool empty_share = false) {
  std::vector<TileRun> tiles;
  int64_t step_rows = 0;
  FMX_TRY(step_tiles(e, m, batch, rows_limit, &tiles, &step_rows));
  if (empty_share) { tiles.clear(); step_rows = 0; }
  FMX_CHECK(tiles.size() <= 1, FMX_ERR_STATE, "the compact exchange needs steps of one tile (batch_rows <= tile_rows)");
  if (tiles.empty()) tiles.push_back({m->step_first_tile[(size_t)batch], 0, 0});  // an empty share publishes no record, but its tail
  const auto& pl = m->plans[(size_t)tiles[0].tile];
  FMX_CHECK(pl.feat != nullptr, FMX_ERR_STATE, "the compact exchange needs sparse tiles (fewer entries than features per tile); use fmx_grad for dense ones");
  // room for this step's records; grows only when a step needs more than was reserved (fmx_compact_reserve: the pointer a
  // driver holds stays valid as long as it reserved enough)
  FMX_TRY(ensure_compact(e, (int64_t)pl.n_lists > 0 ? (int64_t)pl.n_lists : 1));
  int64_t np = 0;
  FMX_TRY(rows_phase(e, m, tiles[0], step_rows, 0, &np));
  ColsArgs c{};
  FMX_TRY(cols_args(e, m, tiles[0], true, &c));
  c.store_compact = 1;
  c.compact_tail = 1;
  c.scalar = SCALAR_PUBLISH;
  c.n_partials = np;
  c.global_rows = (double)step_rows;
  if (e->owner_parts > 1) {  // owner-sharded exchange: records in owner-major order
    FMX_TRY(ensure_owner_plan(e, m, tiles[0].tile));
    c.rec_pos = m->plans[(size_t)tiles[0].tile].own_pos;
  }
  LongArgs la{};
  FMX_TRY(long_args(e, m, tiles[0].tile, &la, &c));
  e->crec_count = pl.dcounts;  // n_lists as the plan builder left it on the device
  e->crec_n = (int64_t)pl.n_lists;
  return launch_cols_update(e, c, la);
}

int use_device_public(int device) { return use_device(device); }
int alloc_matrix_public(int device, int64_t n, uint32_t p, int64_t nnz, bool labels, fmx_matrix** out) { return alloc_matrix(device, n, p, nnz, labels, out); }
// a replica whose share of a (truncated) step is empty still publishes zeros and its tail
int group_grad_empty(fmx_engine* e, fmx_matrix* m, int64_t batch) { return run_step(e, m, batch, 0, false, true); }
// rows == 0: an empty share (records with zero counts, a zero tail)
int group_grad_compact(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows) { return grad_compact(e, m, batch, rows, rows <= 0); }

}  // namespace fmx

using namespace fmx;

extern "C" {

const char* fmx_last_error(void) { return g_error.c_str(); }

int fmx_device_count(int32_t* count) {
  FMX_CHECK(count != nullptr, FMX_ERR_INVALID, "count is NULL");
  int c = 0;
  const hipError_t err = hipGetDeviceCount(&c);
  *count = (err == hipSuccess) ? c : 0;
  if (err != hipSuccess || c <= 0) { set_error("no HIP device available (%s); libfmx has no CPU fallback", err == hipSuccess ? "device count 0" : hipGetErrorString(err)); return FMX_ERR_NOGPU; }
  return FMX_OK;
}

int fmx_config_default(fmx_config* cfg) {
  FMX_CHECK(cfg != nullptr, FMX_ERR_INVALID, "cfg is NULL");
  memset(cfg, 0, sizeof(*cfg));
  cfg->struct_size = sizeof(fmx_config);
  cfg->task = FMX_TASK_CLASSIFICATION;  // R/fm_control.R:43 (first of match.arg)
  cfg->solver = FMX_SOLVER_SGD;
  cfg->num_factor = 2;                  // R/fm_control.R:60
  cfg->keep_w0 = 1; cfg->keep_w1 = 1;   // R/fm_control.R:54,57
  cfg->learn_rate = 0.01;               // R/fm_solver_control.R:91-94
  cfg->alpha_w = 0.1; cfg->alpha_v = 0.1; cfg->beta_w = 1.0; cfg->beta_v = 1.0;  // :109-115
  cfg->random_step = 1;
  cfg->mode = FMX_MODE_SEQUENTIAL;
  cfg->batch_rows = 65536;
  cfg->min_target = -1.0; cfg->max_target = 1.0;
  cfg->device = 0;
  cfg->batch_reduce = FMX_REDUCE_MEAN;
  cfg->gamma = 1e-4;                    // R/fm_solver_control.R:134-139
  cfg->n_gpus = 1;                      // options("FM.threads") defaults to 1 too (R/fm_set_threads.R:19-22)
  return FMX_OK;
}

int fmx_engine_create(const fmx_config* cfg, uint64_t num_features, fmx_engine** out) {
  FMX_CHECK(out != nullptr, FMX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMX_CHECK(cfg != nullptr, FMX_ERR_INVALID, "cfg is NULL");
  FMX_CHECK(cfg->struct_size == sizeof(fmx_config), FMX_ERR_INVALID, "fmx_config size mismatch (%u vs %zu): header/library skew",
            cfg->struct_size, sizeof(fmx_config));
  FMX_CHECK(cfg->task == FMX_TASK_CLASSIFICATION || cfg->task == FMX_TASK_REGRESSION, FMX_ERR_INVALID, "unknown task...");
  FMX_CHECK(cfg->solver == FMX_SOLVER_SGD || cfg->solver == FMX_SOLVER_FTRL || cfg->solver == FMX_SOLVER_ALS || cfg->solver == FMX_SOLVER_TDAP ||
                cfg->solver == FMX_SOLVER_MCMC,
            FMX_ERR_INVALID, "Unknown solver...");  // src/FM.cpp:85
  FMX_CHECK(cfg->num_factor >= 0 && cfg->num_factor <= 128, FMX_ERR_INVALID, "factor.number must be in 0..128 (got %d)", cfg->num_factor);
  FMX_CHECK(cfg->mode == FMX_MODE_SEQUENTIAL || cfg->mode == FMX_MODE_MINIBATCH, FMX_ERR_INVALID, "unknown mode %d", cfg->mode);
  FMX_CHECK(cfg->random_step >= 1, FMX_ERR_INVALID, "random_step must be >= 1");
  FMX_CHECK(cfg->n_gpus >= 0, FMX_ERR_INVALID, "n_gpus must be >= 0");
  FMX_CHECK(cfg->als_max_levels >= -2, FMX_ERR_INVALID, "als_max_levels must be >= 0, or -1 / -2 (the coloured orders)");
  FMX_CHECK(cfg->batch_reduce == FMX_REDUCE_MEAN || cfg->batch_reduce == FMX_REDUCE_SUM, FMX_ERR_INVALID, "unknown batch_reduce %d", cfg->batch_reduce);
  FMX_CHECK(num_features > 0 && num_features < (1ull << 32), FMX_ERR_INVALID, "number of features must be in 1..2^32-1");
  if (cfg->mode == FMX_MODE_MINIBATCH) FMX_CHECK(cfg->batch_rows >= 1 && cfg->tile_rows >= 0, FMX_ERR_INVALID, "batch_rows must be >= 1 and tile_rows >= 0");
  FMX_TRY(use_device(cfg->device));

  std::unique_ptr<fmx_engine, int (*)(fmx_engine*)> e(new fmx_engine(), fmx_engine_destroy);
  e->cfg = *cfg;
  e->p = num_features;
  e->k = cfg->num_factor;
  e->kp32 = pad_factor(e->k, 4);
  e->kp64 = pad_factor(e->k, 2);
  FMX_TRY(make_hyper(*cfg, &e->hyper));
  FMX_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  FMX_TRY(dev_alloc_zero(&e->scal_base, (size_t)2 * SC_COUNT));
  e->scal = e->scal_base;
  e->scal_next = e->scal_base + SC_COUNT;
  const size_t p = (size_t)e->p;
  if (cfg->mode == FMX_MODE_MINIBATCH && !cfg->state_fp64) {
    {  // layout of the V / w tables (fmx_internal.h: w_in_row)
      const char* v = getenv("FMX_W_IN_ROW");
      const bool want = v ? v[0] == '1' : num_features >= 3000000ull;  // measured (profiles/r03_wir_ab.txt): 1 M features -0.4 %, 4 M +8 %, 16 M +6 %, 33 M +6 %
      e->w_in_row = (want && e->kp32 <= WIR_MAX_KP && e->k > 0) ? 1 : 0;
      e->vstride32 = e->w_in_row ? 2 * e->kp32 : e->kp32;
    }
    FMX_TRY(dev_alloc_zero(&e->V, p * e->vstride32));
    if (!e->w_in_row) FMX_TRY(dev_alloc_zero(&e->w, p));
    if (e->hyper.kind != UPD_SGD_L2) { FMX_TRY(dev_alloc_zero(&e->sV, p * e->kp32)); FMX_TRY(dev_alloc_zero(&e->sw, p)); }
    if (e->hyper.kind == UPD_FTRL || e->hyper.kind == UPD_TDAP) { FMX_TRY(dev_alloc_zero(&e->nV, p * e->kp32)); FMX_TRY(dev_alloc_zero(&e->nw, p)); }
    if (e->hyper.kind == UPD_TDAP) {
      FMX_TRY(dev_alloc_zero(&e->t1V, p * e->kp32)); FMX_TRY(dev_alloc_zero(&e->t1w, p));
      FMX_TRY(dev_alloc_zero(&e->t2V, p * e->kp32)); FMX_TRY(dev_alloc_zero(&e->t2w, p));
      FMX_TRY(dev_alloc_zero(&e->t3V, p * e->kp32)); FMX_TRY(dev_alloc_zero(&e->t3w, p));
    }
  } else {
    FMX_TRY(dev_alloc_zero(&e->dV, p * e->kp64));
    FMX_TRY(dev_alloc_zero(&e->dw, p));
    if (e->hyper.kind != UPD_SGD_L2) { FMX_TRY(dev_alloc_zero(&e->dsV, p * e->kp64)); FMX_TRY(dev_alloc_zero(&e->dsw, p)); }
    if (e->hyper.kind == UPD_FTRL || e->hyper.kind == UPD_TDAP) { FMX_TRY(dev_alloc_zero(&e->dnV, p * e->kp64)); FMX_TRY(dev_alloc_zero(&e->dnw, p)); }
    if (e->hyper.kind == UPD_TDAP) {
      FMX_TRY(dev_alloc_zero(&e->dt1V, p * e->kp64)); FMX_TRY(dev_alloc_zero(&e->dt1w, p));
      FMX_TRY(dev_alloc_zero(&e->dt2V, p * e->kp64)); FMX_TRY(dev_alloc_zero(&e->dt2w, p));
      FMX_TRY(dev_alloc_zero(&e->dt3V, p * e->kp64)); FMX_TRY(dev_alloc_zero(&e->dt3w, p));
    }
  }
  FMX_HIP(hipDeviceSynchronize());  // the zero fills ran on the null stream; the engine stream does not wait for it
  if (cfg->n_gpus > 1) FMX_TRY(group_create(e.get()));  // (a failure destroys what was built: the unique_ptr's deleter)
  *out = e.release();
  return FMX_OK;
}

int fmx_engine_destroy(fmx_engine* e) {
  if (!e) return FMX_OK;
  group_destroy(e);
  (void)hipSetDevice(e->cfg.device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (auto& pr : e->prof_pending) { (void)hipEventDestroy(pr.second.first); (void)hipEventDestroy(pr.second.second); }
  if (e->rows_tune.events) for (auto& ev : e->rows_tune.ev) (void)hipEventDestroy(ev);
  (void)hipFree(e->scal_base);
  (void)hipFree(e->V); (void)hipFree(e->w); (void)hipFree(e->sV); (void)hipFree(e->sw); (void)hipFree(e->nV); (void)hipFree(e->nw);
  (void)hipFree(e->t1V); (void)hipFree(e->t1w); (void)hipFree(e->t2V); (void)hipFree(e->t2w); (void)hipFree(e->t3V); (void)hipFree(e->t3w);
  (void)hipFree(e->dV); (void)hipFree(e->dw); (void)hipFree(e->dsV); (void)hipFree(e->dsw); (void)hipFree(e->dnV); (void)hipFree(e->dnw);
  (void)hipFree(e->dt1V); (void)hipFree(e->dt1w); (void)hipFree(e->dt2V); (void)hipFree(e->dt2w); (void)hipFree(e->dt3V); (void)hipFree(e->dt3w);
  (void)hipFree(e->seq_b); (void)hipFree(e->seq_len); (void)hipFree(e->seq_y);
  (void)hipFree(e->seq_packed); (void)hipFree(e->seq_conf); (void)hipFree(e->seq_keys); (void)hipFree(e->seq_sort_tmp);
  (void)hipFree(e->long_partial); (void)hipFree(e->probit);
  (void)hipFree(e->S); (void)hipFree(e->amul); (void)hipFree(e->partials); (void)hipFree(e->gbuf);
  (void)hipFree(e->crec); (void)hipFree(e->ctail); merge_ws_free(e->merge); (void)hipFree(e->als_qe_new); (void)hipFree(e->als_Q); (void)hipFree(e->als_qe); (void)hipFree(e->als_dyn); (void)hipFree(e->als_backup); (void)hipFree(e->als_tile_ws); (void)hipFree(e->als_lo[0]); (void)hipFree(e->als_lo[1]); (void)hipFree(e->als_hash_word); (void)hipFree(e->als_lam_mu); (void)hipFree(e->als_persist_ctl); (void)hipFree(e->als_rec);
  als_graph_free(e->als_graph_w); als_graph_free(e->als_graph_v);
  if (e->side_fork) (void)hipEventDestroy(e->side_fork);
  if (e->side_join) (void)hipEventDestroy(e->side_join);
  if (e->side) (void)hipStreamDestroy(e->side);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return FMX_OK;
}

int fmx_set_params(fmx_engine* e, double w0, const double* w, const double* v) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  FMX_TRY(use_device(e->cfg.device));
  FMX_HIP(hipStreamSynchronize(e->stream));
  FMX_HIP(hipMemcpy(e->scal + SC_W0, &w0, sizeof(double), hipMemcpyHostToDevice));
  // R's vectors go over as they are (V: k x p column-major = p rows of k doubles) and are narrowed and laid out by kernels (fm_ingest.hip:
  // params_to_device); null = zeros, with no p-sized host buffer either way (p = 33 M, k = 32: 8.4 GB of doubles)
  e->als_q_invalidate();
  FMX_TRY(params_to_device(e, w, v));
  e->trace_iters.clear(); e->trace_evals.clear(); e->trace_params.clear();
  FMX_TRY(reset_optimizer_state(e));  // learner->init() zeroes q/u (SGD_Learner.h:61-69) and z/n (FTRL_Learner.h:50-55)
  FMX_HIP(hipDeviceSynchronize());
  if (e->group) FMX_TRY(group_set_params(e, w0, w, v));
  return FMX_OK;
}

// the reassociated reference-order learner's bounded waits gave up in an earlier launch: its parameters are part-way through a chunk (never seen; a silent NaN otherwise)
static int seq_abort_check(fmx_engine* e) {
  if (!e->cfg.seq_reassociate && !getenv("FMX_SEQ_REASSOC")) return FMX_OK;
  double flag = 0.0;
  FMX_HIP(hipMemcpy(&flag, e->scal + SC_SEQ_ABORT, sizeof(double), hipMemcpyDeviceToHost));
  FMX_CHECK(flag == 0.0, FMX_ERR_HIP, "the reassociated sequential learner gave up waiting inside a launch (its workgroup's waves did not all run?): the parameters are not a valid state");
  return FMX_OK;
}

int fmx_get_params(fmx_engine* e, double* w0, double* w, double* v) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  FMX_TRY(use_device(e->cfg.device));
  FMX_HIP(hipStreamSynchronize(e->stream));
  FMX_TRY(seq_abort_check(e));
  if (w0) FMX_HIP(hipMemcpy(w0, e->scal + SC_W0, sizeof(double), hipMemcpyDeviceToHost));
  FMX_TRY(params_from_device(e, w, v));   // widened and packed on the device, copied down in pieces (fm_ingest.hip)
  return FMX_OK;
}

int fmx_init_normal(fmx_engine* e, uint64_t seed, double mean, double stdev) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  FMX_TRY(use_device(e->cfg.device));
  FMX_TRY(fmx_set_params(e, 0.0, nullptr, nullptr));  // w0 = 0, w = 0 (core/Model.h:63-72), optimizer state reset
  FMX_TRY(init_normal(e, seed, mean, stdev));
  FMX_HIP(hipStreamSynchronize(e->stream));
  if (e->group) FMX_TRY(group_init_normal(e, seed, mean, stdev));  // the same V0 on every replica
  return FMX_OK;
}

static int rows_io(fmx_engine* e, const uint32_t* ids, int64_t n, double* w, double* v, bool set) {
  FMX_CHECK(e != nullptr && n >= 0 && (n == 0 || ids), FMX_ERR_INVALID, "bad argument");
  for (int64_t i = 0; i < n; ++i) FMX_CHECK((uint64_t)ids[i] < e->p, FMX_ERR_INVALID, "feature id %u out of range", ids[i]);
  FMX_TRY(use_device(e->cfg.device));
  if (n == 0) return FMX_OK;
  uint32_t* d_ids = nullptr; double *d_w = nullptr, *d_v = nullptr;
  const size_t kk = (size_t)(e->k > 0 ? e->k : 1);
  int st = FMX_OK;
  auto body = [&]() -> int {
    FMX_HIP(hipMalloc(&d_ids, (size_t)n * sizeof(uint32_t)));
    FMX_HIP(hipMalloc(&d_w, (size_t)n * sizeof(double)));
    FMX_HIP(hipMalloc(&d_v, (size_t)n * kk * sizeof(double)));
    FMX_HIP(hipMemcpy(d_ids, ids, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (set) {
      // a NULL part keeps the rows' current values: read them first
      FMX_TRY(rows_copy(e, d_ids, n, d_w, d_v, false));
      FMX_HIP(hipStreamSynchronize(e->stream));
      if (w) FMX_HIP(hipMemcpy(d_w, w, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      if (v && e->k > 0) FMX_HIP(hipMemcpy(d_v, v, (size_t)n * kk * sizeof(double), hipMemcpyHostToDevice));
    }
    FMX_TRY(rows_copy(e, d_ids, n, d_w, d_v, set));
    FMX_HIP(hipStreamSynchronize(e->stream));
    if (!set) {
      if (w) FMX_HIP(hipMemcpy(w, d_w, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
      if (v && e->k > 0) FMX_HIP(hipMemcpy(v, d_v, (size_t)n * kk * sizeof(double), hipMemcpyDeviceToHost));
    }
    return FMX_OK;
  };
  st = body();
  (void)hipFree(d_ids); (void)hipFree(d_w); (void)hipFree(d_v);
  return st;
}

int fmx_get_rows(fmx_engine* e, const uint32_t* ids, int64_t n, double* w, double* v) { return rows_io(e, ids, n, w, v, false); }
int fmx_set_rows(fmx_engine* e, const uint32_t* ids, int64_t n, const double* w, const double* v) {
  if (e) e->als_q_invalidate();
  FMX_TRY(rows_io(e, ids, n, const_cast<double*>(w), const_cast<double*>(v), true));
  if (e->group) FMX_TRY(group_set_rows(e, ids, n, w, v));  // every replica holds the full model
  return FMX_OK;
}

namespace fmx {

struct CkptHeader {
  char magic[4];
  uint32_t version;
  uint64_t p;
  int32_t k, kp, mode, kind;
  uint32_t scalars;
  uint32_t reserved[7];
};
static_assert(sizeof(CkptHeader) == 64, "checkpoint header is 64 bytes");

// every device table of the engine with its element size and count, in file order
static void ckpt_tables(fmx_engine* e, std::vector<std::pair<void*, size_t>>* out) {
  const size_t p = (size_t)e->p;
  auto add = [&](void* ptr, size_t bytes) { if (ptr) out->push_back({ptr, bytes}); };
  add(e->V, p * (e->vstride32 ? e->vstride32 : e->kp32) * sizeof(float)); add(e->w, p * sizeof(float));  // (w-in-row layout: no separate w table)
  add(e->sV, p * e->kp32 * sizeof(float)); add(e->sw, p * sizeof(float));
  add(e->nV, p * e->kp32 * sizeof(float)); add(e->nw, p * sizeof(float));
  add(e->t1V, p * e->kp32 * sizeof(float)); add(e->t1w, p * sizeof(float));
  add(e->t2V, p * e->kp32 * sizeof(float)); add(e->t2w, p * sizeof(float));
  add(e->t3V, p * e->kp32 * sizeof(float)); add(e->t3w, p * sizeof(float));
  add(e->dV, p * e->kp64 * sizeof(double)); add(e->dw, p * sizeof(double));
  add(e->dsV, p * e->kp64 * sizeof(double)); add(e->dsw, p * sizeof(double));
  add(e->dnV, p * e->kp64 * sizeof(double)); add(e->dnw, p * sizeof(double));
  add(e->dt1V, p * e->kp64 * sizeof(double)); add(e->dt1w, p * sizeof(double));
  add(e->dt2V, p * e->kp64 * sizeof(double)); add(e->dt2w, p * sizeof(double));
  add(e->dt3V, p * e->kp64 * sizeof(double)); add(e->dt3w, p * sizeof(double));
}

// every per-feature table of the engine as (base, bytes per feature); params_only: the V rows and w alone (fm_group.hip: the owner-sharded
// exchange refreshes the copies of features a replica does not own)
extern "C++" void engine_tables(fmx_engine* e, std::vector<std::pair<void*, size_t>>* out, bool params_only) {
  std::vector<std::pair<void*, size_t>> all;
  ckpt_tables(e, &all);
  for (auto& t : all) {
    const bool param = t.first == (void*)e->V || t.first == (void*)e->w || t.first == (void*)e->dV || t.first == (void*)e->dw;
    if (!params_only || param) out->push_back({t.first, t.second / (size_t)(e->p ? e->p : 1)});
  }
}

static CkptHeader ckpt_header(const fmx_engine* e) {
  CkptHeader h{};
  memcpy(h.magic, "FMX1", 4);
  h.version = 1; h.p = e->p; h.k = e->k; h.kp = wide_state(e) ? e->kp64 : e->kp32; h.mode = e->cfg.mode; h.kind = e->hyper.kind;
  h.scalars = SC_COUNT;
  h.reserved[0] = mb_wide(e) ? 1u : 0u;  // mini-batch state kept in the fp64 tables
  // reserved[1]: how the FILE holds the fp32 parameters.  0 (every file written since round 4): canonical -- V[p][kp] then w[p] -- whatever layout the
  // engine keeps on the device (the w-in-row layout is a tuning choice made from p, k and FMX_W_IN_ROW at engine creation: it must not leak into the
  // format).  1 (round-3 files of w-in-row engines): one table [p][2 kp] with w in slot kp of its row and no w table; still read, by either layout.
  h.reserved[1] = 0u;
  return h;
}

// The fp32 parameters between the device layout and the file's.  Pieces of PIECE features: device rows are `ds` floats apart (kp, or 2 kp with w in slot kp),
// file rows `fs` floats apart (kp canonical; 2 kp for a round-3 w-in-row file, whose w travels in the row).  w of a canonical file is a separate array.
constexpr size_t CKPT_PIECE = 1u << 18;
static bool ckpt_save_params32(fmx_engine* e, FILE* f) {
  const size_t p = (size_t)e->p, kp = (size_t)e->kp32, ds = (size_t)e->vstride32;
  std::vector<float> dev(CKPT_PIECE * ds), rows(CKPT_PIECE * kp), w(p);
  for (size_t j0 = 0; j0 < p; j0 += CKPT_PIECE) {
    const size_t c = p - j0 < CKPT_PIECE ? p - j0 : CKPT_PIECE;
    if (hipMemcpy(dev.data(), e->V + j0 * ds, c * ds * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return false;
    for (size_t j = 0; j < c; ++j) {
      memcpy(&rows[j * kp], &dev[j * ds], kp * sizeof(float));
      w[j0 + j] = dev[j * ds + kp];
    }
    if (fwrite(rows.data(), sizeof(float), c * kp, f) != c * kp) return false;
  }
  return fwrite(w.data(), sizeof(float), p, f) == p;
}
// file_wir: the file holds rows of 2 kp floats with w inside (and no w array)
static bool ckpt_load_params32(fmx_engine* e, FILE* f, bool file_wir) {
  const size_t p = (size_t)e->p, kp = (size_t)e->kp32, ds = (size_t)e->vstride32, fs = file_wir ? 2 * kp : kp;
  std::vector<float> dev(CKPT_PIECE * ds), rows(CKPT_PIECE * fs), w(p, 0.f);
  if (!file_wir && e->w_in_row) {   // canonical file into w-in-row tables: w is needed while the rows are laid out -- it sits BEHIND them in the file
    const long at = ftell(f);
    if (at < 0 || fseek(f, (long)(p * kp * sizeof(float)), SEEK_CUR) != 0 || fread(w.data(), sizeof(float), p, f) != p || fseek(f, at, SEEK_SET) != 0) return false;
  }
  for (size_t j0 = 0; j0 < p; j0 += CKPT_PIECE) {
    const size_t c = p - j0 < CKPT_PIECE ? p - j0 : CKPT_PIECE;
    if (fread(rows.data(), sizeof(float), c * fs, f) != c * fs) return false;
    std::fill(dev.begin(), dev.begin() + c * ds, 0.f);
    for (size_t j = 0; j < c; ++j) {
      memcpy(&dev[j * ds], &rows[j * fs], kp * sizeof(float));
      if (file_wir) w[j0 + j] = rows[j * fs + kp];
      if (e->w_in_row) dev[j * ds + kp] = w[j0 + j];
    }
    if (hipMemcpy(e->V + j0 * ds, dev.data(), c * ds * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return false;
  }
  if (!file_wir && e->w_in_row) { if (fseek(f, (long)(p * sizeof(float)), SEEK_CUR) != 0) return false; }   // (w was read ahead)
  else if (!file_wir) { if (fread(w.data(), sizeof(float), p, f) != p) return false; }
  if (!e->w_in_row && hipMemcpy(e->w, w.data(), p * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return false;
  return true;
}

}  // namespace fmx

int fmx_engine_save(fmx_engine* e, const char* path) {
  FMX_CHECK(e != nullptr && path != nullptr, FMX_ERR_INVALID, "NULL argument");
  if (e->group) FMX_TRY(group_make_replicated(e));   // (after owner-sharded steps the optimizer state of a feature is current at its owner only)
  FMX_TRY(use_device(e->cfg.device));
  FMX_HIP(hipStreamSynchronize(e->stream));
  FILE* f = fopen(path, "wb");
  FMX_CHECK(f != nullptr, FMX_ERR_INVALID, "cannot open %s for writing", path);
  const CkptHeader h = ckpt_header(e);
  double scal[SC_COUNT];
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && hipMemcpy(scal, e->scal, sizeof(scal), hipMemcpyDeviceToHost) == hipSuccess &&
            fwrite(scal, sizeof(scal), 1, f) == 1;
  std::vector<std::pair<void*, size_t>> tabs;
  ckpt_tables(e, &tabs);
  std::vector<char> buf;
  for (auto& t : tabs) {
    if (!ok) break;
    if (e->w_in_row && t.first == (void*)e->V) { ok = ckpt_save_params32(e, f); continue; }   // canonical: V[p][kp] then w[p]
    buf.resize(t.second);
    ok = hipMemcpy(buf.data(), t.first, t.second, hipMemcpyDeviceToHost) == hipSuccess && fwrite(buf.data(), 1, t.second, f) == t.second;
  }
  ok = (fclose(f) == 0) && ok;
  FMX_CHECK(ok, FMX_ERR_INVALID, "writing %s failed", path);
  return FMX_OK;
}

int fmx_engine_load(fmx_engine* e, const char* path) {
  FMX_CHECK(e != nullptr && path != nullptr, FMX_ERR_INVALID, "NULL argument");
  FMX_TRY(use_device(e->cfg.device));
  FMX_HIP(hipStreamSynchronize(e->stream));
  e->als_q_invalidate();
  FILE* f = fopen(path, "rb");
  FMX_CHECK(f != nullptr, FMX_ERR_INVALID, "cannot open %s", path);
  CkptHeader h{}, want = ckpt_header(e);
  bool ok = fread(&h, sizeof(h), 1, f) == 1;
  if (!ok || memcmp(h.magic, "FMX1", 4) != 0 || h.version != 1) { fclose(f); set_error("%s is not an fmx checkpoint", path); return FMX_ERR_INVALID; }
  if (h.p != want.p || h.k != want.k || h.kp != want.kp || h.mode != want.mode || h.kind != want.kind || h.scalars != want.scalars ||
      h.reserved[0] != want.reserved[0]) {
    fclose(f);
    set_error("checkpoint shape (p=%llu k=%d padded to %d, mode=%d kind=%d, %s state) does not match the engine (p=%llu k=%d padded to %d, mode=%d kind=%d, %s state)",
              (unsigned long long)h.p, h.k, h.kp, h.mode, h.kind, h.reserved[0] ? "fp64" : "fp32", (unsigned long long)want.p, want.k, want.kp, want.mode, want.kind,
              want.reserved[0] ? "fp64" : "fp32");
    return FMX_ERR_INVALID;
  }
  const bool file_wir = h.reserved[1] == 1u;   // a round-3 file of a w-in-row engine: rows of 2 kp floats, w inside, no w array
  if (h.reserved[1] > 1u || (file_wir && e->V == nullptr)) { fclose(f); set_error("%s: unknown parameter layout %u in the checkpoint header", path, h.reserved[1]); return FMX_ERR_INVALID; }
  double scal[SC_COUNT];
  ok = fread(scal, sizeof(scal), 1, f) == 1 && hipMemcpy(e->scal, scal, sizeof(scal), hipMemcpyHostToDevice) == hipSuccess;
  std::vector<std::pair<void*, size_t>> tabs;
  ckpt_tables(e, &tabs);
  std::vector<char> buf;
  for (auto& t : tabs) {
    if (!ok) break;
    if (t.first == (void*)e->V && (e->w_in_row || file_wir)) { ok = ckpt_load_params32(e, f, file_wir); continue; }   // either layout on either side
    if (t.first == (void*)e->w && file_wir) continue;                                                                 // (its w came with the rows)
    buf.resize(t.second);
    ok = fread(buf.data(), 1, t.second, f) == t.second && hipMemcpy(t.first, buf.data(), t.second, hipMemcpyHostToDevice) == hipSuccess;
  }
  fclose(f);
  FMX_CHECK(ok, FMX_ERR_INVALID, "reading %s failed (truncated?)", path);
  e->trace_iters.clear(); e->trace_evals.clear(); e->trace_params.clear();
  FMX_HIP(hipDeviceSynchronize());
  if (e->group) FMX_TRY(group_load(e, path));  // every replica resumes from the same file
  return FMX_OK;
}

int fmx_matrix_from_csr(int device, int64_t n, uint32_t p, const int64_t* row_ptr, const uint32_t* col, const float* val,
                        const float* y, fmx_matrix** out) {
  FMX_CHECK(out != nullptr, FMX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMX_CHECK(row_ptr != nullptr && n >= 0, FMX_ERR_INVALID, "row_ptr is NULL or n < 0");
  FMX_CHECK(row_ptr[0] == 0, FMX_ERR_INVALID, "row_ptr[0] must be 0");
  const int64_t nnz = row_ptr[n];
  FMX_CHECK(nnz >= 0, FMX_ERR_INVALID, "row_ptr ends in a negative count");
  FMX_CHECK(nnz == 0 || (col && val), FMX_ERR_INVALID, "col/val is NULL");
  fmx_matrix* m = nullptr;
  FMX_TRY(alloc_matrix(device, n, p, nnz, y != nullptr, &m));
  // the arrays go over as they are and are checked on the device (fm_ingest.hip: ingest_host_arrays).  The reference leaves the column check
  // commented out (core/Model.h:88); on a GPU an out-of-range column is a fault, so it is enforced
  uint64_t bad[2] = {~0ull, ~0ull};
  int64_t total = 0;
  int st = ingest_host_arrays(m, val, false, col, false, nullptr, row_ptr, y, false, bad, &total);
  if (st == FMX_OK && bad[1] != ~0ull) { set_error("row_ptr decreases at row %llu", (unsigned long long)bad[1]); st = FMX_ERR_INVALID; }
  if (st == FMX_OK && bad[0] != ~0ull) {
    set_error("Length of x is greater then then number of attributes... (col %u at %llu, p=%u: out of range)", col[bad[0]], (unsigned long long)bad[0], p);
    st = FMX_ERR_INVALID;
  }
  if (st == FMX_OK) st = check_rows_sorted(m);
  if (st != FMX_OK) { free_matrix(m); return st; }
  *out = m;
  return FMX_OK;
}

int fmx_matrix_from_rlist(int device, int64_t n, uint32_t p, int64_t nnz, const double* value, const int32_t* col_idx,
                          const int32_t* row_size, const double* labels, fmx_matrix** out) {
  FMX_CHECK(out != nullptr, FMX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMX_CHECK(n >= 0 && nnz >= 0 && (n == 0 || row_size), FMX_ERR_INVALID, "the length of input's row_size is not correct...");
  FMX_CHECK(nnz == 0 || (value && col_idx), FMX_ERR_INVALID, "value/col_idx is NULL");
  // util/Smatrix.h:53-60 narrows value f64 -> f32 and col_idx i32 -> u32 and prefix-sums row_size into row_idx in a host loop; here the R
  // vectors are copied as they are and all of that happens on the device (fm_ingest.hip: ingest_host_arrays)
  fmx_matrix* m = nullptr;
  FMX_TRY(alloc_matrix(device, n, p, nnz, labels != nullptr, &m));
  uint64_t bad[2] = {~0ull, ~0ull};
  int64_t total = 0;
  int st = ingest_host_arrays(m, value, true, col_idx, true, row_size, nullptr, labels, true, bad, &total);
  if (st == FMX_OK && bad[1] != ~0ull) { set_error("negative row_size at row %llu", (unsigned long long)bad[1]); st = FMX_ERR_INVALID; }
  if (st == FMX_OK && total != nnz) {
    set_error("the length of input's row_size is not correct... (sum %lld, size %lld)", (long long)total, (long long)nnz);
    st = FMX_ERR_INVALID;
  }
  if (st == FMX_OK && bad[0] != ~0ull) { set_error("col_idx %d out of range at %llu", col_idx[bad[0]], (unsigned long long)bad[0]); st = FMX_ERR_INVALID; }
  if (st == FMX_OK) st = check_rows_sorted(m);
  if (st != FMX_OK) { free_matrix(m); return st; }
  *out = m;
  return FMX_OK;
}

int fmx_matrix_from_dgc(int device, int64_t nrow, uint32_t ncol, int64_t nnz, const double* x, const int32_t* i, const int32_t* p, const double* labels,
                        fmx_matrix** out) {
  FMX_CHECK(out != nullptr, FMX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMX_CHECK(nrow >= 0 && nnz >= 0 && p != nullptr, FMX_ERR_INVALID, "negative size or NULL column pointers");
  FMX_CHECK(nnz == 0 || (x && i), FMX_ERR_INVALID, "x/i is NULL");
  FMX_CHECK(nrow < (1LL << 32) && nnz < (1LL << 32), FMX_ERR_INVALID, "a dgCMatrix is transposed by one device sort over all its entries: fewer than 2^32 rows and stored entries (this one: %lld, %lld)",
            (long long)nrow, (long long)nnz);
  FMX_CHECK(p[0] == 0, FMX_ERR_INVALID, "p[0] must be 0");
  std::vector<int32_t> col_size((size_t)ncol);
  for (uint32_t j = 0; j < ncol; ++j) {
    FMX_CHECK(p[j + 1] >= p[j], FMX_ERR_INVALID, "the column pointers decrease at column %u", j);
    col_size[j] = p[j + 1] - p[j];
  }
  FMX_CHECK((int64_t)p[ncol] == nnz, FMX_ERR_INVALID, "the column pointers end at %d, not at the number of stored entries (%lld)", p[ncol], (long long)nnz);
  // The slots ARE the row-major arrays of the TRANSPOSE (its rows = the columns): ingest that (the same device path as fmx_matrix_from_rlist), build its
  // column-major form with the whole-matrix sort of the ALS path -- which is the row-major form of the matrix itself, columns ascending inside a row
  // because the sort is stable -- and move it into a matrix of its own.
  fmx_matrix* t = nullptr;
  FMX_TRY(alloc_matrix(device, (int64_t)ncol, (uint32_t)nrow, nnz, false, &t));
  std::unique_ptr<fmx_matrix, void (*)(fmx_matrix*)> tg(t, free_matrix);
  uint64_t bad[2] = {~0ull, ~0ull};
  int64_t total = 0;
  FMX_TRY(ingest_host_arrays(t, x, true, i, true, col_size.data(), nullptr, nullptr, true, bad, &total));
  FMX_CHECK(bad[0] == ~0ull, FMX_ERR_INVALID, "row index %d out of range at entry %llu (%lld rows)", i[bad[0]], (unsigned long long)bad[0], (long long)nrow);
  FMX_TRY(build_full_csc(t, nullptr));
  // the transpose's row-major arrays are done with: give them back before the result is allocated (otherwise three copies of the matrix are resident at once)
  (void)hipFree(t->col); (void)hipFree(t->val); t->col = nullptr; t->val = nullptr;
  fmx_matrix* m = nullptr;
  FMX_TRY(alloc_matrix(device, nrow, ncol, nnz, labels != nullptr, &m));
  std::unique_ptr<fmx_matrix, void (*)(fmx_matrix*)> mg(m, free_matrix);
  FMX_HIP(hipMemcpy(m->row_ptr, t->col_ptr, ((size_t)nrow + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice));
  if (nnz > 0) {
    FMX_HIP(hipMemcpy(m->col, t->crow, (size_t)nnz * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    FMX_HIP(hipMemcpy(m->val, t->cval, (size_t)nnz * sizeof(float), hipMemcpyDeviceToDevice));
  }
  tg.reset();
  if (labels && nrow > 0) {
    std::vector<float> y((size_t)nrow);
    for (int64_t r = 0; r < nrow; ++r) y[(size_t)r] = (float)labels[r];  // util/Smatrix.h narrows the same way (DVector<float>::assign)
    FMX_HIP(hipMemcpy(m->y, y.data(), (size_t)nrow * sizeof(float), hipMemcpyHostToDevice));
  }
  FMX_TRY(check_rows_sorted(m));
  *out = mg.release();
  return FMX_OK;
}

int fmx_matrix_synthetic(int device, int64_t n, uint32_t p, int32_t nnz_per_row, uint64_t seed, int64_t row_offset, fmx_matrix** out) {
  FMX_CHECK(out != nullptr, FMX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMX_CHECK(n >= 0 && nnz_per_row >= 1 && (uint32_t)nnz_per_row <= p, FMX_ERR_INVALID, "need 1 <= nnz_per_row <= p");
  fmx_matrix* m = nullptr;
  FMX_TRY(alloc_matrix(device, n, p, n * nnz_per_row, true, &m));
  int st = generate_synthetic(m, nnz_per_row, seed, row_offset);
  if (st != FMX_OK) { free_matrix(m); return st; }
  *out = m;
  return FMX_OK;
}

static int fields_from_spec(const fmx_fields_spec* spec, FieldSpec* fs, uint64_t* p_out) {
  FMX_CHECK(spec != nullptr && spec->struct_size == sizeof(fmx_fields_spec), FMX_ERR_INVALID, "bad fmx_fields_spec");
  FMX_CHECK(spec->n_dense >= 0 && spec->n_fields >= 0 && spec->n_fields <= FMX_MAX_FIELDS && spec->n_dense + spec->n_fields >= 1, FMX_ERR_INVALID,
            "need 0 <= n_fields <= %d and at least one entry per row", FMX_MAX_FIELDS);
  FMX_CHECK(spec->n_fields == 0 || spec->field_vocab != nullptr, FMX_ERR_INVALID, "field_vocab is NULL");
  FMX_CHECK(spec->skew >= 1.0, FMX_ERR_INVALID, "skew must be >= 1");
  fs->n_dense = spec->n_dense; fs->n_fields = spec->n_fields; fs->skew = spec->skew;
  uint64_t at = (uint64_t)spec->n_dense;
  for (int f = 0; f < spec->n_fields; ++f) {
    FMX_CHECK(spec->field_vocab[f] >= 1, FMX_ERR_INVALID, "field %d has an empty vocabulary", f);
    fs->base[f] = (uint32_t)at; fs->vocab[f] = spec->field_vocab[f];
    at += spec->field_vocab[f];
    FMX_CHECK(at < (1ull << 32), FMX_ERR_INVALID, "more than 2^32-1 features");
  }
  *p_out = at;
  return FMX_OK;
}

int fmx_matrix_synthetic_fields(int device, int64_t n, const fmx_fields_spec* spec, int64_t row_offset, fmx_matrix** out) {
  FMX_CHECK(out != nullptr, FMX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FieldSpec fs{};
  uint64_t p = 0;
  FMX_TRY(fields_from_spec(spec, &fs, &p));
  FMX_CHECK(n >= 0, FMX_ERR_INVALID, "negative size");
  const int z = fs.n_dense + fs.n_fields;
  fmx_matrix* m = nullptr;
  FMX_TRY(alloc_matrix(device, n, (uint32_t)p, n * z, true, &m));
  int st = generate_fields_async(m, n, fs, spec->seed, row_offset, nullptr);
  if (st == FMX_OK && hipDeviceSynchronize() != hipSuccess) { set_error("generator failed"); st = FMX_ERR_HIP; }
  if (st != FMX_OK) { free_matrix(m); return st; }
  m->rows_sorted = 1;
  m->max_row_len = z;
  m->fixed_row_len = z;
  m->unit_values = (fs.n_dense == 0) ? 1 : 0;  // the dense features carry values in [0, 1)
  m->dense_prefix = fs.n_dense;
  if (fs.n_fields > 0) { m->field_base.assign(fs.base, fs.base + fs.n_fields); m->field_base.push_back((uint32_t)p); }
  *out = m;
  return FMX_OK;
}

int fmx_matrix_set_fields(fmx_matrix* m, int32_t n_dense, int32_t n_fields, const uint32_t* field_base) {
  FMX_CHECK(m != nullptr && field_base != nullptr, FMX_ERR_INVALID, "NULL argument");
  FMX_CHECK(n_dense >= 0 && n_fields >= 1 && n_fields <= FMX_MAX_FIELDS && n_dense + n_fields <= 64, FMX_ERR_INVALID, "need n_dense >= 0, 1 <= n_fields <= %d and at most 64 entries per row", FMX_MAX_FIELDS);
  return matrix_set_fields(m, n_dense, n_fields, field_base);
}

int fmx_matrix_synthetic_iid(int device, int64_t n, uint32_t p, int32_t nnz_per_row, uint64_t seed, int64_t row_offset, int32_t law, double zipf_s,
                             fmx_matrix** out) {
  FMX_CHECK(out != nullptr, FMX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMX_CHECK(n >= 0 && nnz_per_row >= 1 && nnz_per_row <= 64 && (uint32_t)nnz_per_row <= p, FMX_ERR_INVALID, "need 1 <= nnz_per_row <= min(64, p)");
  FMX_CHECK(law == FMX_COLUMNS_UNIFORM || (law == FMX_COLUMNS_ZIPF && zipf_s > 1.0), FMX_ERR_INVALID, "law must be FMX_COLUMNS_UNIFORM, or FMX_COLUMNS_ZIPF with zipf_s > 1");
  fmx_matrix* m = nullptr;
  FMX_TRY(alloc_matrix(device, n, p, n * nnz_per_row, true, &m));
  int st = generate_iid_async(m, n, nnz_per_row, seed, row_offset, law, zipf_s, nullptr);
  if (st == FMX_OK && hipDeviceSynchronize() != hipSuccess) { set_error("generator failed"); st = FMX_ERR_HIP; }
  if (st != FMX_OK) { free_matrix(m); return st; }
  m->rows_sorted = 1;
  m->max_row_len = nnz_per_row;
  m->fixed_row_len = nnz_per_row;
  { const char* v = getenv("FMX_UNIT_VALUES"); m->unit_values = !(v && v[0] == '0'); }
  *out = m;
  return FMX_OK;
}

int fmx_matrix_synthetic_ragged(int device, int64_t n, uint32_t p, double mean_nnz, int32_t min_nnz, int32_t max_nnz, uint64_t seed, int64_t row_offset,
                                fmx_matrix** out) {
  FMX_CHECK(out != nullptr, FMX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMX_CHECK(n >= 0 && mean_nnz > 0.0 && mean_nnz <= 64.0 && min_nnz >= 0 && min_nnz <= max_nnz && max_nnz <= 64 && (uint32_t)max_nnz <= p, FMX_ERR_INVALID,
            "need 0 < mean_nnz <= 64 and 0 <= min_nnz <= max_nnz <= min(64, p)");
  return generate_ragged(device, n, p, mean_nnz, min_nnz, max_nnz, seed, row_offset, out);
}

int fmx_matrix_set_labels(fmx_matrix* m, const float* y) {
  FMX_CHECK(m != nullptr && (y != nullptr || m->n == 0), FMX_ERR_INVALID, "NULL argument");
  FMX_TRY(use_device(m->device));
  if (m->n > 0) FMX_HIP(hipMemcpy(m->y, y, (size_t)m->n * sizeof(float), hipMemcpyHostToDevice));
  m->has_labels = 1;
  m->value_generation++;
  return FMX_OK;
}

int fmx_matrix_synthetic_values(fmx_matrix* m, uint64_t seed, int64_t row_offset) {
  FMX_CHECK(m != nullptr, FMX_ERR_INVALID, "NULL matrix");
  FMX_CHECK(row_offset >= 0, FMX_ERR_INVALID, "row_offset must not be negative");
  FMX_TRY(use_device(m->device));
  return matrix_values_uniform(m, seed, row_offset);
}

int fmx_matrix_destroy(fmx_matrix* m) {
  if (m) (void)hipSetDevice(m->device);
  free_matrix(m);
  return FMX_OK;
}

int fmx_matrix_info(const fmx_matrix* m, int64_t* n, uint32_t* p, int64_t* nnz) {
  FMX_CHECK(m != nullptr, FMX_ERR_INVALID, "NULL matrix");
  if (n) *n = m->n;
  if (p) *p = m->p;
  if (nnz) *nnz = m->nnz;
  return FMX_OK;
}

int fmx_matrix_export(const fmx_matrix* m, int64_t r0, int64_t r1, int64_t* row_ptr, uint32_t* col, float* val, float* y) {
  FMX_CHECK(m != nullptr, FMX_ERR_INVALID, "NULL matrix");
  FMX_CHECK(r0 >= 0 && r0 <= r1 && r1 <= m->n, FMX_ERR_INVALID, "row range [%lld,%lld) out of bounds", (long long)r0, (long long)r1);
  FMX_TRY(use_device(m->device));
  std::vector<int64_t> rp((size_t)(r1 - r0) + 1);
  FMX_HIP(hipMemcpy(rp.data(), m->row_ptr + r0, rp.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
  const int64_t base = rp[0], cnt = rp.back() - base;
  if (row_ptr) for (size_t i = 0; i < rp.size(); ++i) row_ptr[i] = rp[i] - base;
  if (col && cnt) FMX_HIP(hipMemcpy(col, m->col + base, (size_t)cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (val && cnt) FMX_HIP(hipMemcpy(val, m->val + base, (size_t)cnt * sizeof(float), hipMemcpyDeviceToHost));
  if (y && r1 > r0) FMX_HIP(hipMemcpy(y, m->y + r0, (size_t)(r1 - r0) * sizeof(float), hipMemcpyDeviceToHost));
  return FMX_OK;
}

int fmx_matrix_scales(fmx_matrix* m, const int32_t* norm_columns, int64_t n_norm, double* mean, double* std) {
  FMX_CHECK(m != nullptr && mean != nullptr && std != nullptr, FMX_ERR_INVALID, "NULL argument");
  FMX_CHECK(n_norm >= 0 && (n_norm == 0 || norm_columns), FMX_ERR_INVALID, "norm_columns is NULL");
  FMX_TRY(use_device(m->device));
  std::vector<uint8_t> listed((size_t)m->p, 0);
  // util/Smatrix.h:113-124 walks the columns in ascending order against an ascending id list; ids that are out of order
  // or out of range are never matched there (R/fm_train.R:84-86 rejects them before)
  int64_t i = 0;
  for (uint32_t c = 0; c < m->p && i < n_norm; ++c)
    if ((int64_t)c == (int64_t)norm_columns[i]) { listed[c] = 1; ++i; }
  return matrix_scales(m, listed.data(), mean, std);
}

int fmx_matrix_normalize(fmx_matrix* m, const double* mean, const double* std) {
  FMX_CHECK(m != nullptr && mean != nullptr && std != nullptr, FMX_ERR_INVALID, "NULL argument");
  FMX_TRY(use_device(m->device));
  return matrix_normalize(m, mean, std);
}

int fmx_predict(fmx_engine* e, const fmx_matrix* m, double* out, int link) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(out != nullptr || m->n == 0, FMX_ERR_INVALID, "out is NULL");
  FMX_CHECK(link >= FMX_LINK_NONE && link <= FMX_LINK_PROBIT, FMX_ERR_INVALID, "unknown link %d", link);
  FMX_TRY(use_device(e->cfg.device));
  if (m->n == 0) return FMX_OK;
  double* d = nullptr;
  FMX_HIP(hipMalloc(&d, (size_t)m->n * sizeof(double)));
  int st = forward_rows(e, m, 0, m->n, d, link);
  if (st == FMX_OK && hipStreamSynchronize(e->stream) != hipSuccess) { set_error("forward kernel failed"); st = FMX_ERR_HIP; }
  if (st == FMX_OK && hipMemcpy(out, d, (size_t)m->n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { set_error("copy of predictions failed"); st = FMX_ERR_HIP; }
  (void)hipFree(d);
  return st;
}

int fmx_predict_device(fmx_engine* e, const fmx_matrix* m, int64_t r0, int64_t r1, void* dev_out_f64, int link) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(r0 >= 0 && r0 <= r1 && r1 <= m->n && dev_out_f64, FMX_ERR_INVALID, "bad row range or NULL output");
  FMX_TRY(use_device(e->cfg.device));
  return forward_rows(e, m, r0, r1, (double*)dev_out_f64, link);
}

int fmx_train_order(fmx_engine* e, fmx_matrix* m, const int64_t* order, int64_t count) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(seq_mode(e), FMX_ERR_STATE, "an explicit visiting order needs FMX_MODE_SEQUENTIAL");
  FMX_CHECK(e->cfg.solver != FMX_SOLVER_ALS && e->cfg.solver != FMX_SOLVER_MCMC, FMX_ERR_STATE, "ALS / MCMC engines train through fmx_als_train / fmx_mcmc_train");
  FMX_CHECK(m->has_labels, FMX_ERR_STATE, "there are no labels in data");
  FMX_CHECK(count >= 0 && (count == 0 || order), FMX_ERR_INVALID, "order is NULL");
  for (int64_t i = 0; i < count; ++i) FMX_CHECK(order[i] >= 0 && order[i] < m->n, FMX_ERR_INVALID, "order[%lld]=%lld out of range", (long long)i, (long long)order[i]);
  if (count == 0) return FMX_OK;
  FMX_TRY(use_device(e->cfg.device));
  int64_t* d = nullptr;
  FMX_HIP(hipMalloc(&d, (size_t)count * sizeof(int64_t)));
  int st = FMX_OK;
  if (hipMemcpy(d, order, (size_t)count * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) { set_error("upload of the visiting order failed"); st = FMX_ERR_HIP; }
  if (st == FMX_OK) st = launch_seq_learn(e, m, d, count);
  if (st == FMX_OK) {
    hipError_t err = hipStreamSynchronize(e->stream);
    if (err != hipSuccess) { set_error("sequential learner failed: %s", hipGetErrorString(err)); st = FMX_ERR_HIP; }
  }
  (void)hipFree(d);
  return st;
}

int fmx_train(fmx_engine* e, fmx_matrix* m, int64_t max_iter, int64_t* examples_done) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(max_iter >= 0, FMX_ERR_INVALID, "max_iter must be >= 0");
  FMX_CHECK(e->cfg.solver != FMX_SOLVER_ALS && e->cfg.solver != FMX_SOLVER_MCMC, FMX_ERR_STATE, "ALS / MCMC engines train through fmx_als_train / fmx_mcmc_train");
  FMX_CHECK(m->has_labels, FMX_ERR_STATE, "there are no labels in data");
  if (examples_done) *examples_done = 0;
  if (max_iter == 0 || m->n == 0) return FMX_OK;
  if (e->group) return group_train(e, m, max_iter, examples_done);
  if (seq_mode(e)) {
    // the visiting order is produced, uploaded and trained on in pieces: host and device memory stay bounded however many
    // examples max_iter asks for (the reference counts examples, not passes)
    const int64_t PIECE = 1 << 22;
    VisitOrder vo(m->n, e->cfg.random_step);
    std::vector<int64_t> order;
    int64_t done = 0;
    while (done < max_iter) {
      const int64_t want = max_iter - done < PIECE ? max_iter - done : PIECE;
      const bool alive = vo.next(want, &order);
      if (!order.empty()) FMX_TRY(fmx_train_order(e, m, order.data(), (int64_t)order.size()));
      done += (int64_t)order.size();
      if (!alive) break;
    }
    if (examples_done) *examples_done = done;
    return FMX_OK;
  }
  int64_t nb = 0;
  FMX_TRY(fmx_num_batches(e, m, &nb));
  int64_t done = 0;
  for (int64_t step = 0; done < max_iter; ++step) {
    const int64_t batch = step % nb;
    const int64_t b0 = batch * m->batch_rows;
    int64_t rows = (b0 + m->batch_rows <= m->n) ? m->batch_rows : m->n - b0;
    if (rows > max_iter - done) rows = max_iter - done;
    FMX_TRY(fmx_step(e, m, batch, rows));
    done += rows;
  }
  FMX_TRY(fmx_sync(e));
  if (examples_done) *examples_done = done;
  return FMX_OK;
}

// n models trained side by side on one matrix in the reference's visiting order (fm_seq_kernels.hip: one workgroup per model, the examples' plan shared)
int fmx_train_grid(fmx_engine* const* engines, int32_t n_engines, fmx_matrix* m, int64_t max_iter, int64_t* examples_done) {
  FMX_CHECK(engines != nullptr && n_engines >= 1, FMX_ERR_INVALID, "no engines");
  FMX_CHECK(max_iter >= 0, FMX_ERR_INVALID, "max_iter must be >= 0");
  if (examples_done) *examples_done = 0;
  fmx_engine* e0 = engines[0];
  for (int32_t b = 0; b < n_engines; ++b) {
    fmx_engine* e = engines[b];
    FMX_TRY(check_pair(e, m));
    FMX_CHECK(seq_mode(e), FMX_ERR_STATE, "grid training runs the reference-order learner: create every engine with FMX_MODE_SEQUENTIAL");
    FMX_CHECK(e->cfg.solver != FMX_SOLVER_ALS && e->cfg.solver != FMX_SOLVER_MCMC, FMX_ERR_STATE, "ALS / MCMC engines train through fmx_als_train / fmx_mcmc_train");
    FMX_CHECK(!e->group, FMX_ERR_STATE, "grid training takes single-device engines");
    FMX_CHECK(e->p == e0->p && e->k == e0->k && e->hyper.kind == e0->hyper.kind && e->cfg.device == e0->cfg.device && e->cfg.task == e0->cfg.task, FMX_ERR_INVALID,
              "the engines of a grid share the feature count, factor.number, solver, task and device (engine %d differs)", (int)b);
    FMX_CHECK(e->cfg.random_step <= 1, FMX_ERR_INVALID, "grid training shares ONE visiting order: random_step > 1 draws a model's own strides from libc rand()");
    for (int32_t c = 0; c < b; ++c) FMX_CHECK(engines[c] != e, FMX_ERR_INVALID, "engine %d appears twice in the grid", (int)b);
  }
  FMX_CHECK(m->has_labels, FMX_ERR_STATE, "there are no labels in data");
  if (max_iter == 0 || m->n == 0) return FMX_OK;
  FMX_TRY(use_device(e0->cfg.device));
  for (int32_t b = 1; b < n_engines; ++b) FMX_HIP(hipStreamSynchronize(engines[b]->stream));   // their tables are about to be used from engines[0]'s stream
  const int64_t PIECE = 1 << 22;
  VisitOrder vo(m->n, e0->cfg.random_step);
  std::vector<int64_t> order;
  int64_t done = 0;
  while (done < max_iter) {
    const int64_t want = max_iter - done < PIECE ? max_iter - done : PIECE;
    const bool alive = vo.next(want, &order);
    if (!order.empty()) {
      int64_t* d = nullptr;
      FMX_HIP(hipMalloc(&d, order.size() * sizeof(int64_t)));
      int st = FMX_OK;
      if (hipMemcpy(d, order.data(), order.size() * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) { set_error("upload of the visiting order failed"); st = FMX_ERR_HIP; }
      if (st == FMX_OK) st = launch_seq_learn_grid(engines, (int)n_engines, m, d, (int64_t)order.size());   // (waits for the stream)
      (void)hipFree(d);
      FMX_TRY(st);
    }
    done += (int64_t)order.size();
    if (!alive) break;
  }
  if (examples_done) *examples_done = done;
  return FMX_OK;
}

// ------------------------------------------------------------------------------------------------ streamed training
// A stream of training steps over generated rows (BASELINE.json configs[3]: 4e9 rows never exist at once).  Three slots, ingest two
// steps ahead: the engine's stream then holds  plan(t+1) | train(t) | plan(t+2) | train(t+1) ...  and the counts the host waits
// for (a tile's launch sizes) belong to a plan that finished BEFORE the step now running -- with two slots the host waited for
// plan(t+1) behind train(t) and the GPU idled from the end of that plan until the host had woken up and enqueued train(t+1).
// Measured (profiles/r02_stream.txt): uniform columns 107 -> 113 M examples/s.
//
// Where the next step's tile is generated and planned: on a second stream beside the running step (default since round 3), or on
// the engine's own stream right behind it (FMX_STREAM_OVERLAP=0).  Round 2 measured the overlapped form 15-20 % SLOWER at
// configs[3]'s shape: the sort's streaming passes (10.2 M x 12 B x 4 passes) pushed the step's gather tables out of the Infinity
// Cache.  With the field-structured plan of round 3 (6.8 M x 8 B x 3 passes, no select passes) the balance turned: 272 against
// 257 M examples/s (profiles/r03_stream_overlap.txt).  In both forms the host only waits for the tile's counts, one step ahead.
struct fmx_source {
  static constexpr int NSLOT = 3;
  struct Slot { fmx_matrix* m = nullptr; hipEvent_t ingested = nullptr, trained = nullptr; uint32_t* h_counts = nullptr; int used = 0; };
  fmx_engine* e = nullptr;
  Slot slot[NSLOT];
  hipStream_t ingest = nullptr;
  bool own_stream = false;
  fmx::PlanWorkspace ws;
  fmx::OwnerWorkspace ows;
  bool has_spec = false;
  fmx::FieldSpec fs{};
  int z = 0;
  uint64_t seed = 0, p = 0;
  int64_t row_offset = 0, total_rows = 0, B = 0, steps = 0;
  int64_t next_t = 0;       // step the next fmx_source_next hands out
  int64_t ingested_to = 0;  // steps whose ingest has been enqueued
  int owners = 0;
  double waited = 0.0;
  ~fmx_source() {
    for (auto& s : slot) {
      if (s.ingested) (void)hipEventDestroy(s.ingested);
      if (s.trained) (void)hipEventDestroy(s.trained);
      if (s.h_counts) (void)hipHostFree(s.h_counts);
      fmx::free_matrix(s.m);
    }
    if (ingest && own_stream) (void)hipStreamDestroy(ingest);
  }
};

namespace fmx {

constexpr int STREAM_COUNTS = 4 + OWNERS_MAX + 2;  // a tile's {n_lists, n_long, n_seg, -} and its lists per owner (+ total)

static int stream_ingest(fmx_source* S, int64_t t) {
  fmx_source::Slot& s = S->slot[t % fmx_source::NSLOT];
  fmx_matrix* m = s.m;
  const int64_t rows = (t + 1) * S->B <= S->total_rows ? S->B : S->total_rows - t * S->B;
  if (s.used) FMX_HIP(hipStreamWaitEvent(S->ingest, s.trained, 0));  // the slot's previous step must have finished with its arrays
  m->n = rows; m->nnz = rows * S->z;
  // Criteo-shaped steps: the generator writes the rows AND what the plan builder's split pass would make of them (fm_ingest.hip: synth_fields_split_k);
  // FMX_STREAM_FUSED=0 keeps the two kernels (A/B runs; the tests compare the forms)
  const std::vector<uint32_t>* fbase = m->field_base.empty() ? nullptr : &m->field_base;
  const char* fe = getenv("FMX_STREAM_FUSED");
  const bool fused = S->has_spec && rows > 0 && !(fe && fe[0] == '0') && plan_fields_split_applies(S->ws, m->unit_values, S->z, m->dense_prefix, fbase) &&
                     S->fs.n_dense == m->dense_prefix;
  if (fused) FMX_TRY(generate_fields_split_async(m, rows, S->fs, S->seed, S->row_offset + t * S->B, S->ingest, S->ws.keys_out, m->brow, m->bval, reinterpret_cast<uint32_t*>(S->ws.vals_in)));
  else if (S->has_spec) FMX_TRY(generate_fields_async(m, rows, S->fs, S->seed, S->row_offset + t * S->B, S->ingest));
  else FMX_TRY(generate_synthetic_async(m, rows, S->z, S->seed, S->row_offset + t * S->B, S->ingest));
  auto& pl = m->plans[0];
  pl.r0 = 0; pl.nrows = rows; pl.base = 0; pl.cnt = rows * S->z;
  FMX_TRY(plan_build(pl, S->ws, (uint32_t)S->p, m->row_ptr, m->col, m->val, m->brow, m->bval, S->ingest, m->unit_values, S->z, m->dense_prefix, fbase, fused));
  FMX_HIP(hipMemcpyAsync(s.h_counts, pl.dcounts, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, S->ingest));
  if (S->owners > 1 && pl.feat) {  // the owner-major order of the tile's lists: the count is still on the device, so the whole directory is sorted
    FMX_TRY(plan_owner_build(pl, S->ows, S->owners, pl.own_cap, S->ingest));
    FMX_HIP(hipMemcpyAsync(s.h_counts + 4, pl.own_counts, (OWNERS_MAX + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, S->ingest));
  }
  FMX_HIP(hipEventRecord(s.ingested, S->ingest));
  s.used = 1;
  return FMX_OK;
}

}  // namespace fmx

int fmx_source_open(fmx_engine* e, const fmx_fields_spec* spec, int32_t nnz_per_row, uint64_t seed, int64_t row_offset, int64_t total_rows, fmx_source** out) {
  FMX_CHECK(out != nullptr, FMX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  FMX_CHECK(!seq_mode(e), FMX_ERR_STATE, "streamed training runs in FMX_MODE_MINIBATCH");
  FMX_CHECK(e->cfg.solver != FMX_SOLVER_ALS && e->cfg.solver != FMX_SOLVER_MCMC, FMX_ERR_STATE, "ALS / MCMC engines train through fmx_als_train / fmx_mcmc_train");
  FMX_CHECK(total_rows >= 0 && row_offset >= 0, FMX_ERR_INVALID, "total_rows and row_offset must be >= 0");
  std::unique_ptr<fmx_source> S(new fmx_source());
  S->e = e;
  S->p = e->p;
  S->z = nnz_per_row;
  S->seed = seed;
  if (spec) {
    FMX_TRY(fields_from_spec(spec, &S->fs, &S->p));
    S->has_spec = true;
    S->seed = spec->seed;
    S->z = S->fs.n_dense + S->fs.n_fields;
  } else {
    FMX_CHECK(nnz_per_row >= 1 && (uint64_t)nnz_per_row <= e->p, FMX_ERR_INVALID, "need 1 <= nnz_per_row <= p");
  }
  FMX_CHECK(S->p == e->p, FMX_ERR_INVALID, "number of input's features is not correct...");
  const int64_t B = e->cfg.batch_rows;
  FMX_CHECK(B <= effective_tile_rows(e), FMX_ERR_STATE, "streamed training needs steps of one tile (batch_rows <= %lld)", (long long)effective_tile_rows(e));
  FMX_TRY(use_device(e->cfg.device));
  S->B = B; S->row_offset = row_offset; S->total_rows = total_rows; S->steps = (total_rows + B - 1) / B;
  S->owners = e->owner_parts;
  const int64_t cap_cnt = B * S->z;
  static const bool overlap = [] { const char* v = getenv("FMX_STREAM_OVERLAP"); return !(v && v[0] == '0'); }();
  if (overlap) {
    // FMX_STREAM_PRIO=1 gives the ingest stream the LOWEST priority the device offers (the ingest of step t + 2 has two steps of slack, the training kernels
    // none).  MEASURED SLOWER, hence opt-in: the training kernels do get faster (0.62 against 0.67 ms per step) but the ingest then falls behind and the step
    // waits for it (353 against 385 M examples/s, profiles/r05_stream_prio.txt) -- the ingest stream is itself on the critical path.
    int least = 0, greatest = 0;
    const char* pe = getenv("FMX_STREAM_PRIO");
    if ((pe && pe[0] == '1') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
      FMX_HIP(hipStreamCreateWithPriority(&S->ingest, hipStreamNonBlocking, least));
    else { (void)hipGetLastError(); FMX_HIP(hipStreamCreateWithFlags(&S->ingest, hipStreamNonBlocking)); }
    S->own_stream = true;
  }
  else S->ingest = e->stream;
  FMX_TRY(S->ws.reserve(cap_cnt, (uint32_t)S->p, S->ingest));
  const bool dense = cap_cnt >= (int64_t)S->p;
  for (auto& s : S->slot) {
    FMX_TRY(alloc_matrix(e->cfg.device, B, (uint32_t)S->p, cap_cnt, true, &s.m));
    s.m->rows_sorted = 1; s.m->max_row_len = S->z; s.m->fixed_row_len = S->z;
    s.m->unit_values = (S->has_spec && S->fs.n_dense > 0) ? 0 : 1;  // the uniform generator writes 1.0f everywhere, the Criteo-shaped one has dense values
    s.m->dense_prefix = S->has_spec ? S->fs.n_dense : 0;
    if (S->has_spec && S->fs.n_fields > 0) { s.m->field_base.assign(S->fs.base, S->fs.base + S->fs.n_fields); s.m->field_base.push_back((uint32_t)S->p); }
    else if (!S->has_spec && s.m->unit_values) strata_bounds((uint32_t)S->p, (int32_t)S->z, &s.m->field_base);
    FMX_HIP(hipMalloc(&s.m->brow, (size_t)cap_cnt * sizeof(uint32_t)));
    FMX_HIP(hipMalloc(&s.m->bval, (size_t)cap_cnt * sizeof(float)));
    s.m->plans.resize(1);
    FMX_TRY(plan_alloc(s.m->plans[0], (uint32_t)S->p, cap_cnt, dense));
    if (S->owners > 1 && !dense) FMX_TRY(plan_owner_alloc(s.m->plans[0], s.m->plans[0].cap_lists));
    s.m->step_first_tile = {0, 1};
    s.m->n_batches = 1; s.m->batch_rows = B; s.m->tile_rows = effective_tile_rows(e);   // build_batch_csc sees a finished cache
    FMX_HIP(hipEventCreateWithFlags(&s.ingested, hipEventDisableTiming));
    FMX_HIP(hipEventCreateWithFlags(&s.trained, hipEventDisableTiming));
    FMX_HIP(hipHostMalloc(&s.h_counts, STREAM_COUNTS * sizeof(uint32_t)));
    memset(s.h_counts, 0, STREAM_COUNTS * sizeof(uint32_t));
  }
  if (S->owners > 1 && !dense) FMX_TRY(S->ows.reserve(S->slot[0].m->plans[0].cap_lists, S->ingest));
  FMX_HIP(hipDeviceSynchronize());
  for (int64_t t = 0; t < 2 && t < S->steps; ++t) { FMX_TRY(stream_ingest(S.get(), t)); S->ingested_to = t + 1; }
  *out = S.release();
  return FMX_OK;
}

int fmx_source_next(fmx_source* S, fmx_matrix** step_matrix, int64_t* rows) {
  FMX_CHECK(S != nullptr && step_matrix != nullptr, FMX_ERR_INVALID, "NULL argument");
  fmx_engine* e = S->e;
  FMX_TRY(use_device(e->cfg.device));
  *step_matrix = nullptr;
  if (rows) *rows = 0;
  const int64_t t = S->next_t;
  if (t > 0) {
    // everything the caller enqueued for step t - 1 (its gradient kernels read the slot's arrays) is on the engine's stream by now
    FMX_HIP(hipEventRecord(S->slot[(t - 1) % fmx_source::NSLOT].trained, e->stream));
    if (S->ingested_to < S->steps && S->ingested_to <= t + 1) { FMX_TRY(stream_ingest(S, S->ingested_to)); S->ingested_to++; }
  }
  if (t >= S->steps) return FMX_OK;
  fmx_source::Slot& s = S->slot[t % fmx_source::NSLOT];
  timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  FMX_HIP(hipEventSynchronize(s.ingested));  // the host needs the tile's counts (launch sizes); the engine's stream keeps running meanwhile
  clock_gettime(CLOCK_MONOTONIC, &t1);
  S->waited += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  auto& pl = s.m->plans[0];
  // a dense directory added on demand for the slot's PREVIOUS tile (a launch that walked the dense exchange buffer) describes that
  // tile, not this one: drop it, plan_ensure_dense makes the new one if a launch asks again
  if (pl.off && !pl.off_in_pool) { (void)hipFree(pl.off); pl.off = nullptr; }
  plan_set_counts(pl, (uint32_t)S->p, s.h_counts);
  if (pl.own_n > 0) for (int o = 0; o <= OWNERS_MAX; ++o) pl.own_counts_h[o] = s.h_counts[4 + o];
  s.m->max_long_seg = pl.n_seg;
  s.m->plan_generation++;
  FMX_HIP(hipStreamWaitEvent(e->stream, s.ingested, 0));
  *step_matrix = s.m;
  if (rows) *rows = s.m->n;
  S->next_t = t + 1;
  return FMX_OK;
}

int fmx_source_close(fmx_source* S, double* ingest_wait_s) {
  if (!S) return FMX_OK;
  (void)hipSetDevice(S->e->cfg.device);
  hipError_t e1 = hipStreamSynchronize(S->e->stream), e2 = S->ingest ? hipStreamSynchronize(S->ingest) : hipSuccess;
  if (ingest_wait_s) *ingest_wait_s = S->waited;
  delete S;
  FMX_CHECK(e1 == hipSuccess && e2 == hipSuccess, FMX_ERR_HIP, "a streamed step failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  return FMX_OK;
}

int fmx_train_stream(fmx_engine* e, const fmx_fields_spec* spec, int32_t nnz_per_row, uint64_t seed, int64_t row_offset, int64_t total_rows,
                     int64_t* examples_done, double* ingest_wait_s) {
  if (examples_done) *examples_done = 0;
  if (ingest_wait_s) *ingest_wait_s = 0.0;
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  if (e->group) return group_train_stream(e, spec, nnz_per_row, seed, row_offset, total_rows, examples_done, ingest_wait_s);
  fmx_source* S = nullptr;
  FMX_TRY(fmx_source_open(e, spec, nnz_per_row, seed, row_offset, total_rows, &S));
  int64_t done = 0;
  int st = FMX_OK;
  for (;;) {
    fmx_matrix* m = nullptr;
    int64_t rows = 0;
    st = fmx_source_next(S, &m, &rows);
    if (st != FMX_OK || m == nullptr) break;
    st = run_step(e, m, 0, 0, true);
    if (st != FMX_OK) break;
    done += rows;
  }
  const int st2 = fmx_source_close(S, ingest_wait_s);
  if (st == FMX_OK) st = st2;
  if (examples_done) *examples_done = done;
  return st;
}

// ------------------------------------------------------------------------------------------------ tracker
namespace fmx {

// the evaluation block of solver/SGD_Learner.h:143-155: prediction with the task's link, then tracker.evaluate
static int track_eval(fmx_engine* e, const fmx_matrix* m, int metric, double* d_yhat, double* score) {
  // Model::predict_prob, core/Model.h:163-180: MCMC / ALS models answer through the probit table, the others logistic
  const int link = e->cfg.task == FMX_TASK_REGRESSION ? FMX_LINK_CLAMP : ((e->cfg.solver == FMX_SOLVER_ALS || e->cfg.solver == FMX_SOLVER_MCMC) ? FMX_LINK_PROBIT : FMX_LINK_LOGISTIC);
  FMX_TRY(forward_rows(e, m, 0, m->n, d_yhat, link));
  return evaluate_device(e, d_yhat, m->y, m->n, metric, score);
}

static int track_record(fmx_engine* e, int64_t iter, double score, bool keep) {  // Tracker::record, core/Tracker.h:54-63
  e->trace_iters.push_back(iter);
  e->trace_evals.push_back(score);
  if (keep) {
    fmx_engine::Snapshot s;
    s.w.resize((size_t)e->p);
    s.v.resize((size_t)e->p * (e->k > 0 ? e->k : 1));
    FMX_TRY(fmx_get_params(e, &s.w0, s.w.data(), s.v.data()));
    e->trace_params.push_back(std::move(s));
  }
  return FMX_OK;
}

}  // namespace fmx

int fmx_evaluate(fmx_engine* e, const fmx_matrix* m, int metric, double* out) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(out != nullptr, FMX_ERR_INVALID, "out is NULL");
  FMX_CHECK(m->has_labels, FMX_ERR_STATE, "there are no labels in data");
  FMX_TRY(use_device(e->cfg.device));
  double* d = nullptr;
  FMX_HIP(hipMalloc(&d, (size_t)(m->n > 0 ? m->n : 1) * sizeof(double)));
  int st = track_eval(e, m, metric, d, out);
  (void)hipFree(d);
  return st;
}

int fmx_train_tracked(fmx_engine* e, fmx_matrix* m, int64_t max_iter, const fmx_track_config* track, int64_t* examples_done,
                      int32_t* convergent) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(track != nullptr && track->struct_size == sizeof(fmx_track_config), FMX_ERR_INVALID, "bad fmx_track_config");
  FMX_CHECK(track->step_size > 0, FMX_ERR_INVALID, "step_size must be > 0 (use fmx_train when the tracker is off)");
  FMX_CHECK(max_iter >= 0, FMX_ERR_INVALID, "max_iter must be >= 0");
  FMX_CHECK(m->has_labels, FMX_ERR_STATE, "there are no labels in data");
  FMX_CHECK(e->group == nullptr || (e->cfg.solver != FMX_SOLVER_ALS && !seq_mode(e)), FMX_ERR_STATE,
            "with n_gpus > 1 the tracker follows the mini-batch learners only (ALS and the sequential mode run on one GPU)");
  FMX_TRY(use_device(e->cfg.device));
  e->trace_iters.clear(); e->trace_evals.clear(); e->trace_params.clear();
  if (examples_done) *examples_done = 0;
  if (convergent) *convergent = 0;
  if (max_iter == 0 || m->n == 0) return FMX_OK;

  // Tracker::init (core/Tracker.h:41-52): at most MAX_REC = 10000 records, else the step is widened
  int64_t step = track->step_size;
  {
    const int64_t MAX_REC = 10000;
    int64_t record_times = (int64_t)std::ceil(((double)max_iter - 0.5) / (double)step) + 1;
    if (record_times > MAX_REC) step = (int64_t)((double)(max_iter + 1) / (double)MAX_REC) + 1;
  }
  const bool keep = track->keep_params != 0;
  double* d_yhat = nullptr;
  FMX_HIP(hipMalloc(&d_yhat, (size_t)m->n * sizeof(double)));
  int64_t* d_order = nullptr;
  int st = FMX_OK;
  int conv_times = 0;
  double old_score = 0.0;
  int64_t done = 0;
  auto after_eval = [&](int64_t iter, double score) {  // solver/SGD_Learner.h:157-164
    if (iter > step && std::fabs((score - old_score) / (old_score + 1e-30)) <= track->convergence) conv_times++;
    else conv_times = 0;
    old_score = score;
    return track_record(e, iter, score, keep);
  };

  if (e->cfg.solver == FMX_SOLVER_ALS) {
    // MCMC_ALS_Learner::learn, :96-125: the tracker looks at the model at the START of iterations 0, step, 2 step, ... and of the
    // last one (clamped predictions, or fast_pnorm for CLASSIFICATION); no convergence rule
    if (!seq_mode(e)) { set_error("ALS runs on the fp64 tables: create the engine with FMX_MODE_SEQUENTIAL"); st = FMX_ERR_STATE; }
    int64_t ii = -1;
    for (int64_t it = 0; st == FMX_OK && it < max_iter; ++it) {
      if (++ii == step) ii = 0;
      if (ii == 0 || it == max_iter - 1) {
        double score = 0.0;
        st = track_eval(e, m, track->metric, d_yhat, &score);
        if (st == FMX_OK) st = track_record(e, it, score, keep);
      }
      if (st == FMX_OK) st = launch_als_train(e, m, 1, 0);
      done = it + 1;
    }
  } else if (seq_mode(e)) {
    std::vector<int64_t> order;
    { VisitOrder vo(m->n, e->cfg.random_step); (void)vo.next(max_iter, &order); }  // (the tracker's record points index into the whole order)
    const int64_t count = (int64_t)order.size();
    if (count > 0) {
      if (hipMalloc(&d_order, (size_t)count * sizeof(int64_t)) != hipSuccess ||
          hipMemcpy(d_order, order.data(), (size_t)count * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("upload of the visiting order failed"); st = FMX_ERR_HIP;
      }
    }
    int64_t pos = 0;
    while (st == FMX_OK && pos < count) {
      // next evaluation: after example `next`, the first index >= pos with next % step == 0, or the last example
      int64_t next = (pos % step == 0) ? pos : (pos / step + 1) * step;
      if (next > count - 1) next = count - 1;
      st = launch_seq_learn(e, m, d_order + pos, next - pos + 1);
      double score = 0.0;
      if (st == FMX_OK) st = track_eval(e, m, track->metric, d_yhat, &score);
      if (st == FMX_OK) st = after_eval(next, score);
      pos = next + 1;
      done = pos;
      if (conv_times >= 3) { if (convergent) *convergent = 1; break; }  // SGD_Learner.h:169-172
    }
  } else if (e->group) {
    // N replicas behind the handle: the same record rule, applied to the example indices a GLOBAL step covers (n_gpus x batch_rows of them); the
    // model is looked at on replica 0 -- the handle itself -- over the whole matrix
    const GroupStepHook hook = [&](int64_t first, int64_t last, bool* stop) -> int {
      const bool hit = (last / step) * step >= first || last == max_iter - 1;
      if (!hit) return FMX_OK;
      double score = 0.0;
      FMX_TRY(track_eval(e, m, track->metric, d_yhat, &score));
      FMX_TRY(after_eval(last, score));
      if (conv_times >= 3) { if (convergent) *convergent = 1; *stop = true; }
      return FMX_OK;
    };
    st = group_train(e, m, max_iter, &done, &hook);
  } else {
    int64_t nb = 0;
    st = fmx_num_batches(e, m, &nb);
    for (int64_t s = 0; st == FMX_OK && done < max_iter; ++s) {
      const int64_t batch = s % nb;
      const int64_t b0 = batch * m->batch_rows;
      int64_t rows = (b0 + m->batch_rows <= m->n) ? m->batch_rows : m->n - b0;
      if (rows > max_iter - done) rows = max_iter - done;
      st = fmx_step(e, m, batch, rows);
      const int64_t first = done, last = done + rows - 1;
      done += rows;
      // the step covered example indices [first, last]: evaluate if one of them is a record point
      const bool hit = (last / step) * step >= first || last == max_iter - 1;  // a multiple of step in [first, last], or the end
      if (st == FMX_OK && hit) {
        double score = 0.0;
        st = track_eval(e, m, track->metric, d_yhat, &score);
        if (st == FMX_OK) st = after_eval(last, score);
        if (conv_times >= 3) { if (convergent) *convergent = 1; break; }
      }
    }
  }
  if (st == FMX_OK) st = fmx_sync(e);
  (void)hipFree(d_yhat); (void)hipFree(d_order);
  if (examples_done) *examples_done = done;
  return st;
}

int fmx_trace_size(fmx_engine* e, int64_t* n_records) {
  FMX_CHECK(e != nullptr && n_records != nullptr, FMX_ERR_INVALID, "NULL argument");
  *n_records = (int64_t)e->trace_iters.size();
  return FMX_OK;
}

int fmx_trace_get(fmx_engine* e, int64_t* iters, double* evals) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  for (size_t i = 0; i < e->trace_iters.size(); ++i) {
    if (iters) iters[i] = e->trace_iters[i];
    if (evals) evals[i] = e->trace_evals[i];
  }
  return FMX_OK;
}

int fmx_trace_params(fmx_engine* e, int64_t record, double* w0, double* w, double* v) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  FMX_CHECK(record >= 0 && record < (int64_t)e->trace_params.size(), FMX_ERR_INVALID, "no snapshot %lld (the trace holds %zu)", (long long)record, e->trace_params.size());
  const auto& s = e->trace_params[(size_t)record];
  if (w0) *w0 = s.w0;
  if (w) memcpy(w, s.w.data(), s.w.size() * sizeof(double));
  if (v && e->k > 0) memcpy(v, s.v.data(), (size_t)e->p * e->k * sizeof(double));
  return FMX_OK;
}

int fmx_num_batches(fmx_engine* e, fmx_matrix* m, int64_t* n_batches) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(!seq_mode(e), FMX_ERR_STATE, "batches exist only in FMX_MODE_MINIBATCH");
  FMX_TRY(use_device(e->cfg.device));
  FMX_TRY(build_batch_csc(m, e->cfg.batch_rows, effective_tile_rows(e), e->stream));
  if (n_batches) *n_batches = m->n_batches;
  return FMX_OK;
}

int fmx_step(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(!group_outside(e), FMX_ERR_STATE, "this handle drives %d GPUs (cfg.n_gpus): the step-level calls would change replica 0 alone; train it with fmx_train", e->cfg.n_gpus);
  FMX_TRY(use_device(e->cfg.device));
  return run_step(e, m, batch, rows_limit, true);
}

int fmx_grad(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(!group_outside(e), FMX_ERR_STATE, "this handle drives %d GPUs (cfg.n_gpus): the step-level calls would change replica 0 alone; train it with fmx_train", e->cfg.n_gpus);
  FMX_TRY(use_device(e->cfg.device));
  return run_step(e, m, batch, rows_limit, false);
}

int fmx_grad_buffer(fmx_engine* e, void** dev_ptr, int64_t* n_floats) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  FMX_CHECK(!seq_mode(e), FMX_ERR_STATE, "the exchange buffer exists only in FMX_MODE_MINIBATCH");
  FMX_TRY(use_device(e->cfg.device));
  FMX_TRY(ensure_gbuf(e));
  if (dev_ptr) *dev_ptr = e->gbuf;
  if (n_floats) *n_floats = e->gbuf_floats;
  return FMX_OK;
}

int fmx_grad_layout(fmx_engine* e, int64_t* n_chunks, int64_t* chunk_features, int64_t* chunk_elems, int64_t* tail_offset) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  FMX_CHECK(!seq_mode(e), FMX_ERR_STATE, "the exchange buffer exists only in FMX_MODE_MINIBATCH");
  FMX_TRY(use_device(e->cfg.device));
  FMX_TRY(ensure_gbuf(e));
  if (n_chunks) *n_chunks = e->gb_blocks;
  if (chunk_features) *chunk_features = e->gb_feats;
  if (chunk_elems) *chunk_elems = e->gb_block_elems;
  if (tail_offset) *tail_offset = e->gb_blocks * e->gb_block_elems;
  return FMX_OK;
}

int fmx_grad_begin(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(!group_outside(e), FMX_ERR_STATE, "this handle drives %d GPUs (cfg.n_gpus): the step-level calls would change replica 0 alone; train it with fmx_train", e->cfg.n_gpus);
  FMX_TRY(use_device(e->cfg.device));
  return grad_begin(e, m, batch, rows_limit);
}

int fmx_grad_chunk(fmx_engine* e, fmx_matrix* m, int64_t chunk) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(!group_outside(e), FMX_ERR_STATE, "this handle drives %d GPUs (cfg.n_gpus): the step-level calls would change replica 0 alone; train it with fmx_train", e->cfg.n_gpus);
  FMX_TRY(use_device(e->cfg.device));
  return grad_block(e, m, chunk);
}

int fmx_apply_chunk(fmx_engine* e, int64_t chunk, int64_t global_rows, int32_t last) {
  FMX_CHECK(e != nullptr && !seq_mode(e), FMX_ERR_STATE, "fmx_apply_chunk needs a mini-batch engine");
  FMX_CHECK(!group_outside(e), FMX_ERR_STATE, "this handle drives %d GPUs (cfg.n_gpus): the step-level calls would change replica 0 alone; train it with fmx_train", e->cfg.n_gpus);
  FMX_TRY(use_device(e->cfg.device));
  return apply_block(e, chunk, global_rows, last != 0);
}

int fmx_grad_elem_bytes(const fmx_engine* e, int32_t* bytes) {
  FMX_CHECK(e != nullptr && bytes != nullptr, FMX_ERR_INVALID, "NULL argument");
  *bytes = (int32_t)mb_elem(e);
  return FMX_OK;
}

int fmx_apply(fmx_engine* e, int64_t global_rows) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  FMX_CHECK(!seq_mode(e) && e->gbuf, FMX_ERR_STATE, "fmx_apply needs a preceding fmx_grad");
  FMX_CHECK(!group_outside(e), FMX_ERR_STATE, "this handle drives %d GPUs (cfg.n_gpus): the step-level calls would change replica 0 alone; train it with fmx_train", e->cfg.n_gpus);
  FMX_TRY(use_device(e->cfg.device));
  ColsArgs c{};
  c.load_gbuf = 1;
  c.apply = 1;
  c.scalar = SCALAR_FROM_TAIL;
  c.global_rows = (double)global_rows;
  return launch_cols_update(e, c, LongArgs{});
}

int fmx_compact_info(fmx_engine* e, fmx_matrix* m, int64_t* record_elems, int64_t* capacity, int32_t* usable) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(!seq_mode(e), FMX_ERR_STATE, "the compact exchange exists only in FMX_MODE_MINIBATCH");
  FMX_TRY(use_device(e->cfg.device));
  FMX_TRY(build_batch_csc(m, e->cfg.batch_rows, effective_tile_rows(e), e->stream));
  const bool has_q = exchange_has_q(e);
  bool ok = !m->plans.empty();
  for (int64_t s = 0; s < m->n_batches && ok; ++s) ok = m->step_first_tile[(size_t)s + 1] - m->step_first_tile[(size_t)s] <= 1;
  for (const auto& pl : m->plans) ok = ok && pl.feat != nullptr;
  if (record_elems) *record_elems = mb_kp(e) * (has_q ? 2 : 1) + 4;
  if (capacity) *capacity = compact_capacity(m);
  if (usable) *usable = ok ? 1 : 0;
  return FMX_OK;
}

int fmx_compact_count(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t* n_records) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(!seq_mode(e) && n_records != nullptr, FMX_ERR_STATE, "the compact exchange exists only in FMX_MODE_MINIBATCH");
  FMX_TRY(use_device(e->cfg.device));
  FMX_TRY(build_batch_csc(m, e->cfg.batch_rows, effective_tile_rows(e), e->stream));
  FMX_CHECK(batch >= 0 && batch < m->n_batches, FMX_ERR_INVALID, "batch %lld out of range", (long long)batch);
  const int64_t t0 = m->step_first_tile[(size_t)batch], t1 = m->step_first_tile[(size_t)batch + 1];
  FMX_CHECK(t1 - t0 <= 1 && (t1 == t0 || m->plans[(size_t)t0].feat), FMX_ERR_STATE, "step %lld is not one sparse tile", (long long)batch);
  *n_records = t1 > t0 ? (int64_t)m->plans[(size_t)t0].n_lists : 0;
  return FMX_OK;
}

int fmx_compact_reserve(fmx_engine* e, int64_t capacity) {
  FMX_CHECK(e != nullptr && !seq_mode(e) && capacity >= 0, FMX_ERR_INVALID, "bad argument");
  FMX_TRY(use_device(e->cfg.device));
  return ensure_compact(e, capacity > 0 ? capacity : 1);
}

int fmx_grad_compact(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows_limit) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(!group_outside(e), FMX_ERR_STATE, "this handle drives %d GPUs (cfg.n_gpus): the step-level calls would change replica 0 alone; train it with fmx_train", e->cfg.n_gpus);
  FMX_TRY(use_device(e->cfg.device));
  return grad_compact(e, m, batch, rows_limit);
}

int fmx_compact_records(fmx_engine* e, void** dev_records, int64_t* n_records, void** dev_tail) {
  FMX_CHECK(e != nullptr && e->ctail != nullptr, FMX_ERR_STATE, "fmx_compact_records needs a preceding fmx_grad_compact (or fmx_compact_reserve)");
  if (dev_records) *dev_records = e->crec;
  if (n_records) *n_records = e->crec_n;
  if (dev_tail) *dev_tail = e->ctail;
  return FMX_OK;
}

int fmx_apply_compact(fmx_engine* e, const void* dev_records, const int64_t* counts, int32_t n_parts, int64_t stride_records, int64_t global_rows) {
  FMX_CHECK(e != nullptr && !seq_mode(e), FMX_ERR_STATE, "fmx_apply_compact needs a mini-batch engine");
  FMX_CHECK(!group_outside(e), FMX_ERR_STATE, "this handle drives %d GPUs (cfg.n_gpus): the step-level calls would change replica 0 alone; train it with fmx_train", e->cfg.n_gpus);
  FMX_CHECK(e->ctail != nullptr, FMX_ERR_STATE, "fmx_apply_compact needs a preceding fmx_grad_compact");
  FMX_CHECK(counts != nullptr && n_parts >= 1 && stride_records >= 0 && (dev_records != nullptr || stride_records == 0), FMX_ERR_INVALID, "bad record parts");
  FMX_TRY(use_device(e->cfg.device));
  int64_t total = 0;
  FMX_TRY(merge_records(e, dev_records, counts, nullptr, n_parts, stride_records, &total));
  const uint32_t *pos, *roff, *rfeat, *d_n;
  merge_result(e, &pos, &roff, &rfeat, &d_n);
  return launch_apply_records(e, dev_records, pos, roff, rfeat, d_n, total, global_rows);
}

int fmx_apply_compact_parts(fmx_engine* e, const void* dev_records, const int64_t* counts, const int64_t* starts, int32_t n_parts, int64_t global_rows) {
  FMX_CHECK(e != nullptr && !seq_mode(e), FMX_ERR_STATE, "fmx_apply_compact_parts needs a mini-batch engine");
  FMX_CHECK(!group_outside(e), FMX_ERR_STATE, "this handle drives %d GPUs (cfg.n_gpus): train it with fmx_train", e->cfg.n_gpus);
  FMX_CHECK(e->ctail != nullptr, FMX_ERR_STATE, "fmx_apply_compact_parts needs a preceding fmx_grad_compact");
  FMX_CHECK(counts != nullptr && starts != nullptr && n_parts >= 1, FMX_ERR_INVALID, "bad record parts");
  FMX_TRY(use_device(e->cfg.device));
  int64_t total = 0;
  for (int r = 0; r < n_parts; ++r) total += counts[r];
  FMX_CHECK(dev_records != nullptr || total == 0, FMX_ERR_INVALID, "dev_records is NULL");
  FMX_TRY(merge_records(e, dev_records, counts, starts, n_parts, 0, &total));
  const uint32_t *pos, *roff, *rfeat, *d_n;
  merge_result(e, &pos, &roff, &rfeat, &d_n);
  return launch_apply_records(e, dev_records, pos, roff, rfeat, d_n, total, global_rows);
}

int fmx_owner_configure(fmx_engine* e, int32_t n_owners, int32_t rank) {
  FMX_CHECK(e != nullptr && !seq_mode(e), FMX_ERR_STATE, "the owner-sharded exchange exists only in FMX_MODE_MINIBATCH");
  FMX_CHECK(n_owners >= 1 && n_owners <= OWNERS_MAX && rank >= 0 && rank < n_owners, FMX_ERR_INVALID, "need 1 <= n_owners <= %d and 0 <= rank < n_owners", OWNERS_MAX);
  e->owner_parts = n_owners > 1 ? n_owners : 0;
  e->owner_rank = rank;
  return FMX_OK;
}

int fmx_owner_info(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t* counts, void** dev_ids) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(!seq_mode(e) && e->owner_parts > 1, FMX_ERR_STATE, "fmx_owner_info needs fmx_owner_configure(n_owners > 1) on a mini-batch engine");
  FMX_TRY(use_device(e->cfg.device));
  FMX_TRY(build_batch_csc(m, e->cfg.batch_rows, effective_tile_rows(e), e->stream));
  FMX_CHECK(batch >= 0 && batch < m->n_batches, FMX_ERR_INVALID, "batch %lld out of range", (long long)batch);
  const int64_t t0 = m->step_first_tile[(size_t)batch], t1 = m->step_first_tile[(size_t)batch + 1];
  FMX_CHECK(t1 - t0 == 1 && m->plans[(size_t)t0].feat, FMX_ERR_STATE, "step %lld is not one sparse tile", (long long)batch);
  FMX_TRY(ensure_owner_plan(e, m, t0));
  const auto& pl = m->plans[(size_t)t0];
  if (counts) for (int o = 0; o < e->owner_parts; ++o) counts[o] = pl.own_counts_h[o];
  if (dev_ids) *dev_ids = pl.own_ids;
  return FMX_OK;
}

int fmx_rows_pack(fmx_engine* e, const void* dev_ids_u32, int64_t n, void* dev_rows, int64_t* row_elems) {
  FMX_CHECK(e != nullptr && !seq_mode(e), FMX_ERR_STATE, "fmx_rows_pack needs a mini-batch engine");
  FMX_CHECK(n >= 0 && (n == 0 || (dev_ids_u32 && dev_rows)), FMX_ERR_INVALID, "bad argument");
  FMX_TRY(use_device(e->cfg.device));
  if (row_elems) *row_elems = mb_kp(e) + 4;
  return rows_pack(e, (const uint32_t*)dev_ids_u32, n, dev_rows, false);
}

int fmx_rows_unpack(fmx_engine* e, const void* dev_ids_u32, int64_t n, const void* dev_rows) {
  FMX_CHECK(e != nullptr && !seq_mode(e), FMX_ERR_STATE, "fmx_rows_unpack needs a mini-batch engine");
  FMX_CHECK(!group_outside(e), FMX_ERR_STATE, "this handle drives %d GPUs (cfg.n_gpus): train it with fmx_train", e->cfg.n_gpus);
  FMX_CHECK(n >= 0 && (n == 0 || (dev_ids_u32 && dev_rows)), FMX_ERR_INVALID, "bad argument");
  FMX_TRY(use_device(e->cfg.device));
  return rows_pack(e, (const uint32_t*)dev_ids_u32, n, const_cast<void*>(dev_rows), true);
}

int fmx_sync(fmx_engine* e) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  hipError_t err = hipStreamSynchronize(e->stream);
  FMX_CHECK(err == hipSuccess, FMX_ERR_HIP, "stream synchronize failed: %s", hipGetErrorString(err));
  FMX_TRY(seq_abort_check(e));
  return FMX_OK;
}

int fmx_stream(fmx_engine* e, void** stream) {
  FMX_CHECK(e != nullptr && stream != nullptr, FMX_ERR_INVALID, "NULL argument");
  *stream = (void*)e->stream;
  return FMX_OK;
}

static int vsweep_impl(fmx_engine* e, fmx_matrix* m, double* error, double alpha, const double* v_lambda, const double* v_mu,
                       const double* std_normals) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(seq_mode(e), FMX_ERR_STATE, "the ALS sweep runs on the fp64 tables: create the engine with FMX_MODE_SEQUENTIAL");
  FMX_CHECK(error != nullptr || m->n == 0, FMX_ERR_INVALID, "error is NULL");
  FMX_TRY(use_device(e->cfg.device));
  if (m->n == 0 || e->k == 0) return FMX_OK;
  double *d_err = nullptr, *d_z = nullptr;
  const size_t bytes = (size_t)m->n * sizeof(double);
  const size_t zbytes = (size_t)e->k * (size_t)e->p * sizeof(double);
  FMX_HIP(hipMalloc(&d_err, bytes));
  if (std_normals && hipMalloc(&d_z, zbytes) != hipSuccess) {
    (void)hipFree(d_err);
    set_error("out of device memory");
    return FMX_ERR_HIP;
  }
  int st = FMX_OK;
  if (hipMemcpy(d_err, error, bytes, hipMemcpyHostToDevice) != hipSuccess) { set_error("upload of the residual failed"); st = FMX_ERR_HIP; }
  if (st == FMX_OK && std_normals && hipMemcpy(d_z, std_normals, zbytes, hipMemcpyHostToDevice) != hipSuccess) { set_error("upload of the normal draws failed"); st = FMX_ERR_HIP; }
  if (st == FMX_OK) st = launch_als_vsweep_device(e, m, d_err, alpha, v_lambda, v_mu, d_z);  // the (q, e) pairs live in the engine
  if (st == FMX_OK && hipMemcpy(error, d_err, bytes, hipMemcpyDeviceToHost) != hipSuccess) { set_error("download of the residual failed"); st = FMX_ERR_HIP; }
  (void)hipFree(d_err); (void)hipFree(d_z);
  return st;
}

int fmx_als_vsweep(fmx_engine* e, fmx_matrix* m, double* error, double alpha, const double* v_lambda, const double* v_mu) {
  return vsweep_impl(e, m, error, alpha, v_lambda, v_mu, nullptr);
}

int fmx_mcmc_vsweep(fmx_engine* e, fmx_matrix* m, double* error, double alpha, const double* v_lambda, const double* v_mu,
                    const double* std_normals) {
  FMX_CHECK(std_normals != nullptr, FMX_ERR_INVALID, "std_normals is NULL (use fmx_als_vsweep for the ALS form)");
  return vsweep_impl(e, m, error, alpha, v_lambda, v_mu, std_normals);
}

int fmx_vsweep_device(fmx_engine* e, fmx_matrix* m, void* dev_error_f64, double alpha, const double* v_lambda, const double* v_mu, const void* dev_std_normals_f64) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(seq_mode(e), FMX_ERR_STATE, "the ALS sweep runs on the fp64 tables: create the engine with FMX_MODE_SEQUENTIAL");
  FMX_CHECK(dev_error_f64 != nullptr || m->n == 0, FMX_ERR_INVALID, "dev_error_f64 is NULL");
  FMX_TRY(use_device(e->cfg.device));
  if (m->n == 0 || e->k == 0) return FMX_OK;
  return launch_als_vsweep_device(e, m, (double*)dev_error_f64, alpha, v_lambda, v_mu, (const double*)dev_std_normals_f64);
}

int fmx_als_plan_info(fmx_engine* e, fmx_matrix* m, int64_t* levels, int64_t* largest_level, int32_t* approximate, int32_t* level_of_feature) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(seq_mode(e), FMX_ERR_STATE, "the ALS sweeps run on the fp64 tables: create the engine with FMX_MODE_SEQUENTIAL");
  FMX_TRY(use_device(e->cfg.device));
  return als_plan_info(e, m, levels, largest_level, approximate, level_of_feature);
}

int fmx_als_tiled_info(fmx_engine* e, fmx_matrix* m, int32_t* levels_tiled, int64_t* tile_rows, int32_t* n_tiles) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(seq_mode(e), FMX_ERR_STATE, "the ALS sweeps run on the fp64 tables: create the engine with FMX_MODE_SEQUENTIAL");
  FMX_TRY(use_device(e->cfg.device));
  FMX_TRY(als_plan_info(e, m, nullptr, nullptr, nullptr, nullptr));
  return als_tiled_info(m, levels_tiled, tile_rows, n_tiles);
}

int fmx_als_order_info(fmx_engine* e, fmx_matrix* m, int32_t* level_order) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(seq_mode(e), FMX_ERR_STATE, "the ALS sweeps run on the fp64 tables: create the engine with FMX_MODE_SEQUENTIAL");
  FMX_CHECK(level_order != nullptr, FMX_ERR_INVALID, "NULL argument");
  FMX_TRY(use_device(e->cfg.device));
  FMX_TRY(als_plan_info(e, m, nullptr, nullptr, nullptr, nullptr));
  *level_order = als_order_form(m);   // 0: none, 1: the tile form, 2: the block form (fm_als_blocks.hip)
  return FMX_OK;
}

int fmx_als_carry_q(fmx_engine* e, int32_t on) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  e->als_carry_q = on ? 1 : 0;
  e->als_q_have = 0;
  return FMX_OK;
}

int fmx_als_train(fmx_engine* e, fmx_matrix* m, int32_t max_iter, int32_t with_v) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(seq_mode(e), FMX_ERR_STATE, "ALS runs on the fp64 tables: create the engine with FMX_MODE_SEQUENTIAL");
  FMX_CHECK(max_iter >= 0, FMX_ERR_INVALID, "max_iter must be >= 0");
  FMX_CHECK(m->has_labels, FMX_ERR_STATE, "there are no labels in data");
  FMX_TRY(use_device(e->cfg.device));
  if (m->n == 0 || max_iter == 0) return FMX_OK;
  return launch_als_train(e, m, max_iter, with_v);
}

int fmx_mcmc_train(fmx_engine* e, fmx_matrix* m, int32_t max_iter, const double* std_gammas, const double* std_normals, double* state_out) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(seq_mode(e), FMX_ERR_STATE, "MCMC runs on the fp64 tables: create the engine with FMX_MODE_SEQUENTIAL");
  FMX_CHECK(max_iter >= 0, FMX_ERR_INVALID, "max_iter must be >= 0");
  FMX_CHECK(m->has_labels, FMX_ERR_STATE, "there are no labels in data");
  FMX_CHECK(max_iter == 0 || (std_gammas != nullptr && std_normals != nullptr), FMX_ERR_INVALID, "the pre-drawn variates are NULL");
  FMX_TRY(use_device(e->cfg.device));
  if (m->n == 0 || max_iter == 0) return FMX_OK;
  return launch_mcmc_train(e, m, max_iter, std_gammas, std_normals, state_out);
}

int fmx_mcmc_train_from(fmx_engine* e, fmx_matrix* m, int32_t max_iter, const double* std_gammas, const double* std_normals, double* state_io) {
  FMX_TRY(check_pair(e, m));
  FMX_CHECK(seq_mode(e), FMX_ERR_STATE, "MCMC runs on the fp64 tables: create the engine with FMX_MODE_SEQUENTIAL");
  FMX_CHECK(state_io != nullptr, FMX_ERR_INVALID, "state_io is NULL (use fmx_mcmc_train to start a chain)");
  FMX_CHECK(max_iter >= 0, FMX_ERR_INVALID, "max_iter must be >= 0");
  FMX_CHECK(m->has_labels, FMX_ERR_STATE, "there are no labels in data");
  FMX_CHECK(max_iter == 0 || (std_gammas != nullptr && std_normals != nullptr), FMX_ERR_INVALID, "the pre-drawn variates are NULL");
  FMX_TRY(use_device(e->cfg.device));
  if (m->n == 0 || max_iter == 0) return FMX_OK;
  const double in[3] = {state_io[0], state_io[1], state_io[2]};
  return launch_mcmc_train(e, m, max_iter, std_gammas, std_normals, state_io, in);
}

int fmx_mcmc_v_hyper(fmx_engine* e, const double* std_gammas, const double* std_normals, double* v_lambda, double* v_mu, int32_t sample) {
  FMX_CHECK(e != nullptr && v_lambda != nullptr && v_mu != nullptr, FMX_ERR_INVALID, "NULL argument");
  FMX_CHECK(seq_mode(e), FMX_ERR_STATE, "MCMC / ALS run on the fp64 tables: create the engine with FMX_MODE_SEQUENTIAL");
  FMX_CHECK(!sample || (std_gammas != nullptr && std_normals != nullptr), FMX_ERR_INVALID, "the pre-drawn variates are NULL");
  FMX_TRY(use_device(e->cfg.device));
  return launch_mcmc_v_hyper(e, std_gammas, std_normals, v_lambda, v_mu, sample);
}

int fmx_rccl_selftest(int32_t n, double* max_err) { return group_rccl_selftest(n, max_err); }

int fmx_debug_fail_next_plan_build(void) { debug_fail_next_plan_build(); return FMX_OK; }
int fmx_debug_fail_next_comm_init(void) { debug_fail_next_comm_init(); return FMX_OK; }
int fmx_debug_lose_next_seq_multiplier(void) { debug_lose_next_seq_multiplier(); return FMX_OK; }
int fmx_debug_stall_next_persistent_sweep(void) { debug_stall_next_persistent_sweep(); return FMX_OK; }
int fmx_group_info(fmx_engine* e, int32_t* n_replicas, int32_t* share_device, int32_t* peer_pairs, int32_t* peer_pairs_direct, int32_t* sparse_exchange) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  return group_info(e, n_replicas, share_device, peer_pairs, peer_pairs_direct, sparse_exchange);
}

int fmx_profile_enable(fmx_engine* e, int on) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  FMX_TRY(prof_collect(e));
  e->profile = on > 0 ? on : 0;
  for (int i = 0; i < FMX_KERNEL_COUNT; ++i) e->prof_seen[i] = 0;
  return FMX_OK;
}

int fmx_profile_get(fmx_engine* e, int kernel, double* total_ms, int64_t* launches) {
  FMX_CHECK(e != nullptr && kernel >= 0 && kernel < FMX_KERNEL_COUNT, FMX_ERR_INVALID, "bad kernel id");
  FMX_TRY(prof_collect(e));
  if (total_ms) *total_ms = e->prof_ms[kernel];
  if (launches) *launches = e->prof_n[kernel];
  return FMX_OK;
}

int fmx_rows_tune_info(fmx_engine* e, int32_t* serial, double* ms_serial, double* ms_pipelined) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  if (serial) *serial = e->rows_tune.decided;
  if (ms_serial) *ms_serial = e->rows_tune.ms[1];
  if (ms_pipelined) *ms_pipelined = e->rows_tune.ms[0];
  return FMX_OK;
}

int fmx_matrix_rows_form(const fmx_matrix* m, int32_t* form) {
  FMX_CHECK(m != nullptr && form != nullptr, FMX_ERR_INVALID, "NULL matrix or output");
  *form = rows_ragged(m) ? 2 : (rows_flat(m) ? 1 : 0);
  return FMX_OK;
}

int fmx_layout_info(fmx_engine* e, int32_t* v_row_stride, int32_t* w_in_row) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  if (v_row_stride) *v_row_stride = wide_state(e) ? e->kp64 : e->vstride32;
  if (w_in_row) *w_in_row = (!wide_state(e) && e->w_in_row) ? 1 : 0;
  return FMX_OK;
}

int fmx_profile_reset(fmx_engine* e) {
  FMX_CHECK(e != nullptr, FMX_ERR_INVALID, "NULL engine");
  FMX_TRY(prof_collect(e));
  for (int i = 0; i < FMX_KERNEL_COUNT; ++i) { e->prof_ms[i] = 0; e->prof_n[i] = 0; }
  return FMX_OK;
}

}  // extern "C"
