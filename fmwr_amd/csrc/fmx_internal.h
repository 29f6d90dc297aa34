// Internal declarations shared by the libfmx.so translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fmx.h"

namespace fmx {

void set_error(const char* fmt, ...);

#define FMX_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      fmx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
      return FMX_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

#define FMX_CHECK(cond, code, ...)  \
  do {                              \
    if (!(cond)) {                  \
      fmx::set_error(__VA_ARGS__);  \
      return (code);                \
    }                               \
  } while (0)

#define FMX_TRY(call)             \
  do {                            \
    int _s = (call);              \
    if (_s != FMX_OK) return _s;  \
  } while (0)

// regularisation flavour of the update, fixed at engine creation
enum UpdateKind : int {
  UPD_SGD_L2 = 0,  // solver/SGD_Learner.h:119,135 (lazy L2 on touched coordinates)
  UPD_SGD_L1 = 1,  // solver/SGD_Learner.h:195-204 (cumulative penalty)
  UPD_FTRL = 2,    // solver/FTRL_Learner.h:158-202
  UPD_TDAP = 3,    // solver/TDAP_Learner.h:79-233
};

// hyper-parameters as the kernels consume them (passed by value)
struct Hyper {
  int task, k0, k1;
  int kind;  // UpdateKind
  int mean;  // FMX_REDUCE_MEAN: per-coordinate mean gradient, one update per batch
  double lr, reg0, regw, regv;  // SGD: regw/regv are the L1 OR L2 rates (SGD_Learner.h:44-59)
  double l1w, l1v, l2w, l2v;    // FTRL prox
  double alpha_w, alpha_v, beta_w, beta_v;
  double min_t, max_t;
  double egamma;                    // TDAP: exp(-gamma)
  double gamma;                     // TDAP: the decay rate itself (mini-batch: exp(-gamma * occurrences))
  double decay_w, decay_v;          // 1 - lr*reg (SGD lazy L2, one touch)
  double log_decay_w, log_decay_v;  // log of the above, for c touches: exp(c * log)
};

// device scalars (double[SC_COUNT]) living next to the parameters
enum Scalar : int {
  SC_W0 = 0,
  SC_Z0 = 1,  // FTRL z_w0
  SC_N0 = 2,  // FTRL n_w0
  SC_UW = 3,  // SGD-L1 cumulative penalty u_w
  SC_UV = 4,  // SGD-L1 u_v
  SC_G0 = 5,  // last batch: sum of grad multipliers
  SC_Q0 = 6,  // last batch: sum of squared grad multipliers
  SC_T_NU = 7,     // TDAP nu_w0 (u_w0 lives in SC_N0, z_w0 in SC_Z0)
  SC_T_DELTA = 8,  // TDAP delta_w0
  SC_T_H = 9,      // TDAP h_w0
  SC_SEQ_ABORT = 10,  // the reassociated learner's bounded waits gave up (never seen): non-zero until the next fmx_set_params; fmx_get_params / fmx_sync report it
  SC_COUNT = 12
};

constexpr int WG_THREADS = 256;
constexpr int OWNERS_MAX = 16;       // ranks of the owner-sharded exchange (one node: 8)
constexpr int STAGE_ENTRIES = 2048;  // (id, x) pairs staged in LDS per chunk: 16 KiB

inline int pad_factor(int k, int vec) {
  int kp = vec;
  while (kp < k) kp <<= 1;
  return kp;
}

}  // namespace fmx

namespace fmx { struct MergeWs; struct Group; }

struct fmx_matrix {
  uint64_t uid = 0;  // process-unique, never reused (caches keyed on a matrix compare this, not its address)
  int device = 0;
  int64_t n = 0;
  uint32_t p = 0;
  int64_t nnz = 0;
  int64_t* row_ptr = nullptr;  // [n+1]
  uint32_t* col = nullptr;     // [nnz]
  float* val = nullptr;        // [nnz]
  float* y = nullptr;          // [n] or null
  int has_labels = 0;
  int rows_sorted = 0;  // every row strictly ascending in col (=> no duplicate column inside a row)
  int fixed_row_len = 0; // > 0: every row holds exactly this many entries (row of an entry = a division, not a search)
  int unit_values = 0;  // every stored value is exactly 1.0f (one-hot data): the kernels then never read the value arrays
  int dense_prefix = 0; // > 0 (with fixed_row_len): every row STARTS with the columns 0 .. dense_prefix-1 (always-present features with real
                        // values) and every other stored value is exactly 1.0f -- Criteo-shaped rows.  The plan builder then sorts only the
                        // one-hot part, as (column, row) pairs: the dense columns' lists are the rows in order (plan_build)
  std::vector<uint32_t> field_base;  // the generators only: entry dense_prefix + c of every row is an id of field c, in [field_base[c], field_base[c + 1]);
                                     // the last element is p (the uniform generator: dense_prefix = 0, field c = stratum c).  The plan builder then sorts field by field.
  int max_row_len = 0;  // entries of the longest row
  // Per-tile inverted index ("plan"), built lazily on the device for one (batch_rows, tile_rows) pair (fm_ingest.hip).
  // A step covers batch_rows consecutive rows and is cut into tiles of at most tile_rows rows.
  int64_t batch_rows = 0;
  int64_t tile_rows = 0;
  int64_t n_batches = 0;                 // steps
  std::vector<int64_t> step_first_tile;  // [n_batches+1]
  // One tile: its entries sorted by feature (stable: rows ascending inside a feature's list), the lists' offsets, and the
  // plan for lists too long for one lane group.  Two list directories exist:
  //   dense : off[p + 1], list i belongs to feature i            (tiles with at least as many entries as features, and
  //                                                                on demand for the launches that walk the dense exchange buffer)
  //   sparse: feat[n_lists] ascending ids of the features that occur + soff[n_lists + 1]   (tiles with fewer entries than features)
  struct TilePlan {
    int64_t r0 = 0, nrows = 0;     // rows [r0, r0 + nrows) of the matrix
    int64_t base = 0, cnt = 0;     // entries [base, base + cnt) of the CSR; brow / bval of the tile start at base
    uint32_t* off = nullptr;       // dense directory [p + 1] (entry offsets relative to base), or null
    uint32_t* feat = nullptr;      // sparse directory: ids [n_lists] ...
    uint32_t* soff = nullptr;      // ... and offsets [n_lists + 1], or null
    uint32_t* row0 = nullptr;      // ... and each list's FIRST entry inline (row, value bits): a one-entry list needs no trip to brow/bval
    uint32_t* val0 = nullptr;
    uint32_t n_lists = 0;          // occurring features (sparse directory)
    uint32_t cap_lists = 0;        // capacity of feat / soff - 1
    // long lists (more than list_long_min() entries): cut into segments of LIST_SEG entries
    uint32_t* lfeat = nullptr;     // [n_long] feature ids, ascending
    uint32_t* lpos = nullptr;      // [n_long] their index in the sparse directory (== lfeat without one)
    uint32_t* lseg_ptr = nullptr;  // [n_long + 1] segments of each long list
    uint32_t* seg_list = nullptr;  // [n_seg] index into lfeat
    uint32_t* seg_begin = nullptr; // [n_seg] entry range of the segment (offsets relative to base)
    uint32_t* seg_end = nullptr;
    uint32_t n_long = 0, n_seg = 0, cap_long = 0, cap_seg = 0;
    uint32_t* dcounts = nullptr;   // device copy of {n_lists, n_long, n_seg} as the builder wrote them
    void* pool = nullptr;          // one allocation behind all the arrays above
    int off_in_pool = 0;           // the dense directory is the tile's primary one (part of pool); 0 with off != null: added on demand
    // owner-sharded exchange (sparse tiles): the tile's lists in OWNER-MAJOR order, owner of feature j = j mod own_n, ids ascending
    // inside an owner's segment (a stable partition of the ascending directory).  Built by plan_owner_build.
    int own_n = 0;                 // owners the permutation was built for (0: not built)
    uint32_t* own_pos = nullptr;   // [n_lists] record slot of list i (its position in owner-major order)
    uint32_t* own_ids = nullptr;   // [n_lists] feature ids in owner-major order: segment o holds the ids this rank asks owner o for
    uint32_t* own_counts = nullptr;// device [OWNERS_MAX + 1] lists per owner (the last slot: the sentinel padding)
    int64_t own_counts_h[17] = {0};// the same on the host once read back
    void* own_pool = nullptr;      // one allocation behind the three arrays
    uint32_t own_cap = 0;          // lists the arrays have room for
  };
  std::vector<TilePlan> plans;   // [n_tiles]
  uint64_t value_generation = 0; // bumped when the stored values change (scales / normalize): cached copies elsewhere are stale
  uint64_t plan_generation = 0;  // bumped whenever the plans are rebuilt or dropped: engines holding an open step compare it
  int64_t max_long_seg = 0;      // largest n_seg over the tiles (sizes the partial buffer)
  int64_t max_tile_cnt = 0;      // most entries in one tile
  uint32_t* brow = nullptr;      // [nnz] row index local to the tile
  float* bval = nullptr;         // [nnz]
  // CSC of the whole matrix (ALS sweep), built lazily
  int64_t* col_ptr = nullptr;  // [p+1]
  uint32_t* crow = nullptr;    // [nnz]
  float* cval = nullptr;       // [nnz]
  // ALS level plan (features ordered by (level, index)), built lazily
  uint32_t* als_feats = nullptr;        // light features (one wave each), ordered by (level, index)
  std::vector<int64_t> als_level_ptr;
  uint32_t* als_heavy = nullptr;        // features with long columns (one workgroup each), same order
  std::vector<int64_t> als_heavy_ptr;
  // exact plan: columns of more than ALS_SPLIT entries are cut into segments walked by one workgroup each (fm_als_kernels.hip)
  uint32_t* als_vh = nullptr;           // such features, ordered by (level, index); vh_seg0[i]..vh_seg0[i+1]: their segments
  uint32_t* als_vh_seg0 = nullptr;
  uint32_t* als_vseg_feat = nullptr;    // per segment: index into als_vh, entry range inside the column
  int64_t* als_vseg_b = nullptr;
  int64_t* als_vseg_e = nullptr;
  double* als_vh_work = nullptr;        // [2 * n_vseg] partial sums | [n_vh] old value | [n_vh] difference
  int64_t als_n_vh = 0, als_n_vseg = 0;
  std::vector<int64_t> als_vh_ptr, als_vseg_ptr;   // per level
  std::vector<int32_t> als_level_of;    // [p] level (exact plan) or group (approximate plan) of every feature
  int als_approx = 0;                   // the plan holds the groups of the approximate sweep, not exact levels
  int als_coloured = 0;                 // the plan's levels are the colours of a colouring: exact steps, the engine's own feature order (cfg.als_max_levels < 0)
  std::vector<int64_t> als_level_maxlen; // per level: the longest light column (sizes the LDS of the feature-major kernel)
  int als_force_exact = 0;              // an approximate sweep raised the residual on this matrix: only exact plans from now on
  int als_plan_cap = -1;                // cfg.als_max_levels the plan was built for
  int64_t* als_level_ptr_dev = nullptr; // als_level_ptr on the device (the persistent form of a deep exact sweep walks the levels inside ONE launch), uploaded on first use
  uint16_t* als_rank = nullptr;         // rank of every column-major entry inside its row (als_exact_flow_k's expected tags), built on first use
  int als_rank_state = 0;               // 0 not built, 1 built, -1 does not apply (a row of more than 65 535 entries)
  void* als_tiled = nullptr;            // row-tiled form of the wide levels of an exact plan (fm_als_tiled.hip: AlsTiled), or null
  int als_tiled_tried = 0;              // the tiled plan was built, or found not to apply, for the current plan and values
};

// Phase 1 of a LARGE step has two schedules that compute the same bits: `serial` (one entry's V row and w outstanding per lane
// group, the next entry's requests go out when they have landed) and pipelined (four entries' requests out together).
// Which is faster is a property of the DATA: serial wins where the gathers miss (uniform columns: 0.146 against 0.196 ms per
// tile at configs[1], and still ahead at p = 16M), pipelined where they hit (Criteo-shaped skew: 0.137 against 0.196).  So
// the engine times both on its own first large launches -- TRIALS launches alternate, the first two are thrown away -- and
// keeps the faster one for that tile shape; FMX_ROWS_SERIAL=0/1 pins it.  Results do not depend on the choice.
struct RowsTune {
  static constexpr int TRIALS = 14;
  static constexpr int64_t MIN_ROWS = 65536;  // launches this large are timed; smaller wide launches follow the decision
  int decided = -1;          // -1: still measuring
  int launches = 0;
  int64_t key = -1;          // one-hot data or not: the other kind measures again
  hipEvent_t ev[2 * TRIALS] = {};
  bool events = false;
  double ms[2] = {0, 0};     // kept for fmx_rows_tune_info
};

struct fmx_engine {
  fmx_config cfg{};
  fmx::Hyper hyper{};
  uint64_t p = 0;
  int k = 0;
  int kp32 = 0;  // padded factor count of the fp32 tables (multiple of 4)
  // fp32 mini-batch tables, "w in the row" layout: V rows lie vstride32 floats apart; with w_in_row = 1 the stride is 2 * kp32 and
  // the feature's linear weight sits in slot kp32 of its own row (V[16] | w | ... in ONE 128-byte line for k = 16), e->w is null.
  // Out of the caches a nonzero then costs one memory request instead of two (V row + w: measured at p = 16 M, 14.8 M of the
  // 15.7 M fabric reads of a phase-1 launch were those two misses per nonzero, profiles/r03_pmc_p16m_separate_tables.json), and phase 2's
  // sparse walk touches one line per feature.  Chosen at engine creation (p >= 3 M and kp32 <= 16; FMX_W_IN_ROW=0/1 overrides):
  // cache-resident tables gain nothing and a dense phase-2 sweep would stream twice the bytes.
  int vstride32 = 0;
  int w_in_row = 0;
  int kp64 = 0;  // padded factor count of the fp64 tables (multiple of 2)
  hipStream_t stream = nullptr;
  // phase 2's long lists on a stream of their own, beside the short lists' kernel (fm_batch_kernels.hip: launch_cols_kind); made on first use
  hipStream_t side = nullptr;
  hipEvent_t side_fork = nullptr, side_join = nullptr;
  double* scal = nullptr;       // [SC_COUNT] current scalars (one half of scal_base)
  double* scal_next = nullptr;  // the other half: written by the step's last kernel, then swapped in
  double* scal_base = nullptr;  // [2][SC_COUNT] allocation
  // mini-batch fp32 state (cfg.state_fp64 == 0): tables are [p][kp32]
  float *V = nullptr, *w = nullptr;
  float *sV = nullptr, *sw = nullptr;    // q (SGD-L1) or z (FTRL)
  float *nV = nullptr, *nw = nullptr;    // n (FTRL), u (TDAP)
  float *t1V = nullptr, *t1w = nullptr;  // TDAP nu
  float *t2V = nullptr, *t2w = nullptr;  // TDAP delta
  float *t3V = nullptr, *t3w = nullptr;  // TDAP h
  // fp64 state (sequential mode, or mini-batch mode with cfg.state_fp64): tables are [p][kp64]
  double *dV = nullptr, *dw = nullptr;
  double *dsV = nullptr, *dsw = nullptr;
  double *dnV = nullptr, *dnw = nullptr;
  double *dt1V = nullptr, *dt1w = nullptr;  // TDAP nu   (u in dnV/dnw, z in dsV/dsw)
  double *dt2V = nullptr, *dt2w = nullptr;  // TDAP delta
  double *dt3V = nullptr, *dt3w = nullptr;  // TDAP h
  // workspace of the sequential learner: per-example (first entry, length, label) in visiting order
  int64_t* seq_b = nullptr;
  int* seq_len = nullptr;
  float* seq_y = nullptr;
  int64_t seq_cap = 0;
  // windowed sequential learner: per example the padded (id, x) entries and the last earlier example sharing a feature
  void* seq_packed = nullptr;     // uint2 [seq_wcap][32]
  int32_t* seq_conf = nullptr;    // [seq_wcap]
  uint32_t* seq_keys = nullptr;   // [4][seq_wcap * 32] sort buffers (keys in/out, values in/out)
  void* seq_sort_tmp = nullptr;
  size_t seq_sort_tmp_bytes = 0;
  int64_t seq_wcap = 0;
  // workspaces (mini-batch)
  int64_t ws_rows = 0;
  int64_t tile_rows = 0;      // rows per tile (<= cfg.batch_rows)
  void* S = nullptr;          // [ws_rows][kp] per-row factor sums (element type = state type, see mb_wide)
  void* amul = nullptr;       // [ws_rows] per-row gradient multiplier
  double* partials = nullptr; // [ws_partials][2]
  int64_t ws_partials = 0;
  double* probit = nullptr;        // [PN_POINTS + 1 | DP_POINTS + 1] probit tables (fm_probit.h), uploaded on first use
  double* long_partial = nullptr;  // segment sums of the long lists
  int64_t long_partial_cap = 0;
  // compact exchange (steps of one sparse tile): records of the occurring features instead of a p-sized buffer
  void* crec = nullptr;       // [crec_cap][rec_elems] this rank's records (element type = state type)
  int64_t crec_cap = 0;
  int rec_elems = 0;          // kp * (1 + has_q) + 4
  void* ctail = nullptr;      // [4] {sum mult, sum mult^2, rows hi, rows lo}
  uint32_t* crec_count = nullptr;  // device: records written by the last fmx_grad_compact (points into the tile plan)
  int64_t crec_n = 0;              // the same count on the host (the plan builder read it back)
  int owner_parts = 0;             // > 1: fmx_grad_compact writes its records owner-major for this many owners (fmx_owner_configure)
  int owner_rank = 0;
  fmx::MergeWs* merge = nullptr;   // scratch of fmx_apply_compact
  void* als_qe_new = nullptr;      // second (q, e) array of the approximate ALS sweep
  int64_t als_qe_new_rows = 0;
  double* als_Q = nullptr;         // q of every factor, [n][kp64], made by one forward pass per V sweep (grow-only)
  size_t als_Q_elems = 0;
  void* als_qe = nullptr;          // (q, e) pairs of the learners' loops and of fmx_vsweep_device (grow-only)
  int64_t als_qe_rows = 0;
  void* als_dyn = nullptr;         // device struct the sweep kernels read factor / alpha / lambda / mu / normals from (fm_als_kernels.hip)
  void* als_tile_ws = nullptr;     // tiled sweep: per (tile, feature) sums, the level's coordinates and steps (fm_als_tiled.hip; grow-only)
  size_t als_tile_ws_bytes = 0;
  void* als_lo[2] = {nullptr, nullptr};  // level-order form of the V sweep (fm_als_tiled.hip): the two (q, e) arrays it alternates between (grow-only)
  int64_t als_lo_rows = 0;
  int als_lo_cur = 0;              // which of the two holds the pairs now
  int als_q_level0 = 0;            // the q table of the sweep under way is in level 0's array order (block form: built on the permuted CSR)
  // q carried from sweep to sweep (fmx_als_carry_q; block form only): after a V sweep the table holds X v_f for the NEW V of every factor (each factor's final q is
  // written back as it leaves the pairs), so the next sweep needs no forward pass -- if V is still what the sweep left (a 64-bit fingerprint of the table) and the
  // plan is the same one.  Rebuilt every ALS_CARRY_REFRESH sweeps against rounding drift.
  int als_carry_q = 0;
  int als_q_have = 0, als_q_age = 0;
  uint64_t als_q_trusted = 0;      // plan uid: the table was filled a moment ago by the learner's own forward pass (launch_als_train) -- used once, no fingerprint needed
  uint64_t als_q_hash = 0, als_q_plan = 0;
  void* als_hash_word = nullptr;
  double* als_lam_mu = nullptr;    // (lambda_f, mu_f) of every factor for the feature-major sweep (cfg.als_max_levels = -2)
  // every writer of the fp64 V table other than the sweeps themselves calls this: a q table carried from an earlier sweep (fmx_als_carry_q) or left by the learner's own
  // forward pass no longer describes V
  void als_q_invalidate() { als_q_have = 0; als_q_trusted = 0; }
  uint32_t* als_rec = nullptr;               // 32-byte tagged (q, e) records of als_exact_flow_k, als_rec_rows of them
  int64_t als_rec_rows = 0;
  unsigned int* als_persist_ctl = nullptr;   // 64 bytes: {features done, abort, ...} of the persistent deep sweep (als_exact_persist_k), zeroed before every launch
  const double* als_qnext = nullptr;  // V sweep: q of the NEXT factor (one double per row), which the last level's correction pass stores in place of the
                                      // finished factor's q when that level is a tiled one (it then clears this pointer: the pick kernel is not needed)
  int als_vf_slot = -1, als_vf_buf = 0;  // tiled sweep: the tiled level whose coordinates the previous level's step kernel already gathered, and into which half
  void *als_graph_w = nullptr, *als_graph_v = nullptr;  // deep exact plans replayed as HIP graphs (AlsGraph)
  double* als_backup = nullptr;    // what an approximate sweep is rolled back to when it raises the residual
  size_t als_backup_elems = 0;
  fmx::Group* group = nullptr;     // cfg.n_gpus > 1: the other replicas and the exchange between them (fm_group.hip)
  void* gbuf = nullptr;       // multi-GPU exchange buffer (element type = state type)
  int64_t gbuf_floats = 0;    // its element count
  // Layout of the exchange buffer: blocks of gb_feats features, each block GV[F][kp] | GW[F] | CNT[F] (| QV[F][kp] | QW[F]),
  // then the tail {G0, Q0, rows, 0}.  One block (F >= p) unless cfg.exchange_chunks > 1: a block is then the unit of the
  // pipelined all-reduce (fmwr_amd/distributed.py).
  int64_t gb_feats = 0, gb_blocks = 0, gb_block_elems = 0;
  // the step opened by fmx_grad_begin (chunked exchange): its tiles and row count
  struct OpenTile { int64_t tile, r0, nrows, s_row0; };
  std::vector<OpenTile> open_tiles;
  int64_t open_rows = 0;
  int64_t open_partials = 0;
  fmx_matrix* open_matrix = nullptr;
  uint64_t open_generation = 0;  // the matrix's plan_generation at fmx_grad_begin
  // tracker (core/Tracker.h): records of the last fmx_train_tracked
  struct Snapshot { double w0; std::vector<double> w, v; };
  std::vector<int64_t> trace_iters;
  std::vector<double> trace_evals;
  std::vector<Snapshot> trace_params;
  // measurement
  int profile = 0;      // 0 off, n > 0: time every n-th launch of each kernel
  int64_t prof_seen[FMX_KERNEL_COUNT] = {0};
  int prof_open = 0;    // a begin without its end is pending
  double prof_ms[FMX_KERNEL_COUNT] = {0};
  int64_t prof_n[FMX_KERNEL_COUNT] = {0};
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_pending;
  RowsTune rows_tune;   // phase 1's request cadence for large steps, measured on this engine's own first launches
};

namespace fmx {

// mini-batch mode keeps its state in fp32 tables unless cfg.state_fp64 asks for the sequential mode's fp64 tables
inline bool mb_wide(const fmx_engine* e) { return e->cfg.mode == FMX_MODE_MINIBATCH && e->cfg.state_fp64 != 0; }
inline bool wide_state(const fmx_engine* e) { return e->cfg.mode == FMX_MODE_SEQUENTIAL || e->cfg.state_fp64 != 0; }
inline int mb_kp(const fmx_engine* e) { return mb_wide(e) ? e->kp64 : e->kp32; }        // padded factor count of the mini-batch tables
inline int mb_lpr(const fmx_engine* e) { return mb_wide(e) ? e->kp64 / 2 : e->kp32 / 4; } // lanes per row / per feature list (16 B each)
inline size_t mb_elem(const fmx_engine* e) { return mb_wide(e) ? sizeof(double) : sizeof(float); }
// where the mini-batch kernels find V rows and w: element stride between consecutive features' V rows, the address of feature 0's w,
// and the element stride between consecutive features' w
inline int mb_vstride(const fmx_engine* e) { return mb_wide(e) ? e->kp64 : e->vstride32; }
inline void* mb_vbase(const fmx_engine* e) { return mb_wide(e) ? (void*)e->dV : (void*)e->V; }
inline void* mb_wbase(const fmx_engine* e) { return mb_wide(e) ? (void*)e->dw : (e->w_in_row ? (void*)(e->V + e->kp32) : (void*)e->w); }
inline int mb_wstride(const fmx_engine* e) { return (!mb_wide(e) && e->w_in_row) ? e->vstride32 : 1; }
// the exchange carries the sums of squared gradients too: solvers with an accumulated-square state, when the batch is SUMMED
inline bool exchange_has_q(const fmx_engine* e) { return (e->hyper.kind == UPD_FTRL || e->hyper.kind == UPD_TDAP) && !e->hyper.mean; }

// ---- launchers implemented in the kernel translation units -------------------------------------------------
struct RowsArgs {
  const int64_t* row_ptr;
  const uint32_t* col;
  const float* val;
  const float* y;       // may be null when !train
  int64_t r0;           // first row (global index into the matrix)
  int64_t nrows;        // rows to process
  const void* V;        // [p][vs] float or double: the first kp elements of a row are the factors
  const void* w;        // feature j's linear weight at w[j * ws]
  int vs, ws;           // element strides (kp and 1, or the w-in-row layout's 2 kp and 2 kp): powers of two
  int vsh, wsh;         // their log2 (set by launch_rows_forward: the kernels shift instead of multiplying)
  const double* scal;
  void* S;              // [nrows][kp]   (train) element type of the tables
  void* amul;           // [nrows]       (train)
  double* partials;     // [grid][2]     (train)
  double* yhat;         // [nrows]       (predict) -- indexed from 0; may be null when only qout is wanted
  double* qout;         // [nrows][kp64] (predict, fp64 tables) per-row factor sums, or null
  int64_t qout_t;       // > 0: qout is FACTOR-major, factor f of row r at qout[f * qout_t + r] (the ALS sweeps pick one factor at a time)
  const double* pn_y;   // fast_pnorm table (FMX_LINK_PROBIT)
  int link;
  int unit;             // every value is 1.0f: a.val is not read
  int no_w;             // the launch's y_hat is not wanted (qout only): the w gathers collapse onto one word (fm_rows_forward_k)
  int embed;            // EmbedMode: how the multiplier is folded into the S row (set by the launcher)
  int wg_threads;       // 64: one-wave workgroups (rows_wg_threads); anything else: WG_THREADS
  int split;            // 4: four lane groups share a row (rows_split; one-wave workgroups only); anything else: one
  int serial;           // large steps: one entry's requests outstanding per lane group at a time (set by the launcher)
  int flat;             // 1: wide launches take fm_rows_forward_flat_k (rows_flat: FMX_ROWS_FLAT=1 on a matrix whose rows differ in length); nmat must be set
  int64_t nmat;         // rows of the matrix (the flat form cuts the MATRIX's blocks of rows, whole)
  int sort_rows;        // FMX_ROWS_PULL=1 on a matrix of differing row lengths: wide launches take fm_rows_forward_dyn_k (lane groups pull rows; same bits, measured slower)
};
// Small steps.  A CU sustains about 240 random 64-byte rows per microsecond whatever runs on it (its miss queue; the
// gather probe shows the same rate at 2 and at 8 workgroups per CU), so a phase 1 of fewer than one 256-thread workgroup
// per CU is bound by the FEW CUs it occupies: in-kernel stamps at batch_rows = 1024 show 17 of its 20 us in the four
// dependent gather rounds of 16 workgroups (profiles/r02_small_batch.txt).  Such a step goes out as one-WAVE workgroups, four times
// the CUs.  The choice is made from the STEP's row count, never the tile's: a workgroup's partial sum for the w0 step
// covers wg_threads / lpr rows, so the association of that sum -- and with it the last bits of w0 -- stays a function of
// the step alone.
constexpr size_t FMX_SERIAL_TABLE_MAX = 1ull << 30;
inline int rows_wg_threads(int64_t step_rows, int lpr) { return step_rows * lpr < 512LL * WG_THREADS ? 64 : WG_THREADS; }
// ... and the smallest steps (fewer than 1024 one-wave workgroups) spread every row over four lane groups: four times the waves again,
// and a 30-entry row is one gather round per group instead of four.  Also decided from the STEP's row count.
inline int rows_split(int64_t step_rows, int lpr) { return (lpr <= 16 && step_rows * lpr < 1024LL * 64) ? 4 : 1; }
// rows of differing lengths: FMX_ROWS_PULL=1 sends wide launches through fm_rows_forward_dyn_k (lane groups pull rows instead of owning one each).  Same bits as
// the static kernel; measured SLOWER on SURVEY 8(d)'s Poisson law (0.221 against 0.194 ms per 262 144-row tile, profiles/r04_ragged_probe2.txt), hence opt-in.
inline int rows_ragged(const fmx_matrix* m) {
  const char* s = getenv("FMX_ROWS_PULL");   // (read per call: the tests compare the two forms)
  return (s && s[0] == '1' && m->fixed_row_len == 0 && m->n > 0) ? 1 : 0;
}
// The flat form of phase 1 (fm_rows_forward_flat_k) on matrices whose rows differ in length: opt-in, FMX_ROWS_FLAT=1 (read per call: the tests compare the
// forms).  It changes the association of a row's sums (not the bars they are held to), so it could only ever be chosen by a rule on the MATRIX, never by a
// measurement -- and it is MEASURED SLOWER than one lane group per row on every row-length law tried (profiles/r04_ragged_forms.txt: Poisson(30) in [1, 64]
// 0.218 against 0.193 ms per 262 144-row tile, lengths 25..35 0.217 against 0.172), although it is perfectly balanced: what phase 1 lives on is that the lane
// groups of a wave sit at the SAME position of their column-sorted rows, so that the rows one gather instruction fetches lie in one narrow band of the table
// (one column per stratum 0.149 ms, i.i.d. sorted columns 0.162, ragged rows 0.193, the flat walk -- no common position at all -- 0.217).  The pull kernel
// (FMX_ROWS_PULL=1) goes first.
inline int rows_flat(const fmx_matrix* m) {
  if (m->fixed_row_len != 0 || m->n <= 0 || m->nnz <= 0) return 0;
  if ((int64_t)m->max_row_len * WG_THREADS >= (1LL << 31)) return 0;  // a block's entries are counted in 32 bits
  const char* s = getenv("FMX_ROWS_FLAT");
  return (s && s[0] == '1') ? 1 : 0;
}
// workgroups (= w0 partial sums) of a flat launch: the matrix's blocks of `g` rows that rows [r0, r0 + nrows) touch
inline int64_t rows_flat_blocks(int64_t r0, int64_t nrows, int g) { return nrows > 0 ? (r0 + nrows - 1) / g - r0 / g + 1 : 0; }
int launch_rows_forward(fmx_engine* e, const RowsArgs& a, bool train, bool fp64_tables);
int ensure_probit(fmx_engine* e);  // builds and uploads the probit tables (fm_probit.h) on first use

enum ScalarMode : int { SCALAR_NONE = 0, SCALAR_FUSED = 1, SCALAR_PUBLISH = 2, SCALAR_FROM_TAIL = 3 };

// One launch of fm_cols_update_k.  A step is one or more tiles (each with its own CSC); the flags say what this launch
// does with the coordinate sums:  walk the tile's lists -> [+ load_gbuf] -> [store_gbuf] -> [apply the update].
struct ColsArgs {
  const uint32_t* bptr;  // [p+1] for this tile
  const uint32_t* brow;  // based at the tile's first entry
  const float* bval;
  uint32_t rows_active;  // tile rows taking part (a truncated step cuts the last tile)
  uint32_t long_min;     // lists longer than this are handled by the long-list kernels (0: none in this tile)
  const uint32_t* tfeat; // ids of the features occurring in the tile (ascending), or null: walk all p features
  const uint32_t* toff;  // [n_tfeat+1] entry offsets of those features (compact copy of bptr)
  uint32_t n_tfeat;
  const uint32_t* trow0; // [n_tfeat] first entry of each list, inline (row; value bits)
  const uint32_t* tval0;
  int inline0;           // the lean form gathers a list's first S row beside its V row (set by the launcher)
  int walk;              // accumulate this tile's sums from S / amul
  int load_gbuf;         // add the sums already in the exchange buffer (earlier tiles, or the all-reduced global sums)
  int store_gbuf;        // write the sums back to the exchange buffer
  int apply;             // apply the update to V / w / optimizer state
  int scalar;            // ScalarMode: what workgroup 0 does for w0
  int64_t n_partials;    // phase 1's per-workgroup partial sums to reduce (SCALAR_FUSED / SCALAR_PUBLISH)
  double global_rows;    // rows of the whole step (all tiles; all ranks when known), <= 0: take it from the buffer tail
  uint32_t f0, f1;       // dense walk over the features [f0, f1) only (f1 == 0: all p): one chunk of the chunked exchange
  int64_t s_row0;        // row of the S / multiplier workspace where this tile's rows start (0 unless a whole step is resident)
  int store_compact;     // compact exchange: write one record per occurring feature (sparse walk only)
  const uint32_t* rec_pos; // ... at slot rec_pos[list] instead of slot `list` (owner-major records), or null
  int compact_tail;      // the scalar tail is the compact exchange's own 4 elements, not the end of the dense buffer
  int unit;              // every value of the tile is 1.0f: bval is not read
  int direct;            // set by the launcher: every group reads its list's entries itself (sparse tiles), no LDS staging
  int64_t list_entries;  // entries of the tile (sparse walk: decides `direct`)
  int embed;             // EmbedMode of the S rows (set by the launcher; must match phase 1's)
  int buf_gather;        // set by the launcher: S rows / multipliers are gathered through buffer descriptors (padding slots issue no request)
};
// long lists of one tile (heavy-hitter features): cut into segments, see fm_batch_kernels.hip
struct LongArgs {
  const uint32_t* lfeat;      // [n_long] feature ids, ascending
  const uint32_t* lpos;       // [n_long] index of each in the tile's sparse directory (compact records)
  const uint32_t* lseg_ptr;   // [n_long+1] segments of each long feature
  const uint32_t* seg_feat;   // [n_seg] index into lfeat
  const uint32_t* seg_begin;  // [n_seg] entry range of the segment (offsets like bptr)
  const uint32_t* seg_end;
  double* partial;            // [n_seg][2*kp+4] segment sums
  int64_t n_long, n_seg;
};
// a list of more than this many entries is a long list (FMX_LONG_MIN in the environment overrides it: tuning only)
inline uint32_t list_long_min() {
  static const uint32_t v = [] { const char* s = getenv("FMX_LONG_MIN"); const long x = s ? atol(s) : 0; return x > 0 ? (uint32_t)x : 64u; }();
  return v;
}
// widest fp32 row (padded factors) that can carry its w (w_in_row); the kernels of wider rows have their strides compiled in.  Tried at 32 (a 256-byte
// row = two lines, profiles/r03_wir_ab.txt): slower everywhere -- Criteo shape 523 -> 481 M examples/s, p = 16 M, k = 32: 245 -> 214 M.
constexpr int WIR_MAX_KP = 16;
constexpr uint32_t LIST_SEG = 1024;  // entries per segment (one wave)
int launch_cols_update(fmx_engine* e, const ColsArgs& a, const LongArgs& la);
// compact exchange (fm_batch_kernels.hip; the merge itself is in fm_ingest.hip)
// the parts of a record buffer: part r holds prefix[r + 1] - prefix[r] records starting at record start[r] (kernel argument, by value)
constexpr int REC_PARTS_MAX = 64;
struct RecParts { int n; int64_t prefix[REC_PARTS_MAX + 1]; int64_t start[REC_PARTS_MAX]; };
int launch_record_keys(fmx_engine* e, const void* recs, const RecParts& parts, int64_t total, uint32_t* keys, uint32_t* pos);
int launch_apply_records(fmx_engine* e, const void* recs, const uint32_t* pos, const uint32_t* roff, const uint32_t* rfeat, const uint32_t* d_n,
                         int64_t max_lists, int64_t global_rows);
struct MergeWs;
// part r: counts[r] records starting at record starts[r] (starts == nullptr: at r * stride)
int merge_records(fmx_engine* e, const void* recs, const int64_t* counts, const int64_t* starts, int n_parts, int64_t stride, int64_t* total_out);
void merge_ws_free(MergeWs* w);
void merge_result(const fmx_engine* e, const uint32_t** pos, const uint32_t** roff, const uint32_t** rfeat, const uint32_t** d_n);

int launch_seq_learn(fmx_engine* e, const fmx_matrix* m, const int64_t* d_order, int64_t count);
int launch_seq_learn_grid(fmx_engine* const* es, int n, const fmx_matrix* m, const int64_t* d_order, int64_t count);   // n models, one order, one launch per chunk

// ingest
// scratch of the plan builder, sized for tiles of up to max_cnt entries over p features (grow-only)
struct PlanWorkspace {
  int64_t max_cnt = 0;
  uint32_t p = 0;
  int bits = 0;
  uint32_t* keys_out = nullptr;
  uint64_t *vals_in = nullptr, *vals_out = nullptr;
  void* sort_temp = nullptr;
  size_t sort_bytes = 0;
  uint8_t* flags = nullptr;
  uint32_t* nseg = nullptr;
  uint32_t* blk = nullptr;   // block counts of the ordered compactions (fm_ingest.hip: compact_indices)
  void* prim_temp = nullptr;
  size_t prim_bytes = 0;
  uint32_t *fq_counts = nullptr, *fq_totals = nullptr;   // per-field sort: block histograms, digit totals
  int reserve(int64_t cnt, uint32_t p, hipStream_t stream);
  PlanWorkspace() = default;
  PlanWorkspace(const PlanWorkspace&) = delete;
  PlanWorkspace& operator=(const PlanWorkspace&) = delete;
  ~PlanWorkspace();
};
int plan_alloc(fmx_matrix::TilePlan& t, uint32_t p, int64_t cap_cnt, bool dense);
void plan_free(fmx_matrix::TilePlan& t);
int plan_build(fmx_matrix::TilePlan& t, PlanWorkspace& ws, uint32_t p, const int64_t* row_ptr, const uint32_t* col, const float* val,
               uint32_t* brow, float* bval, hipStream_t stream, int unit_values, int fixed_row_len, int dense_prefix = 0,
               const std::vector<uint32_t>* field_base = nullptr, bool presplit = false);
bool plan_fields_split_applies(const PlanWorkspace& ws, int unit_values, int fixed_row_len, int dense_prefix, const std::vector<uint32_t>* field_base);
void plan_set_counts(fmx_matrix::TilePlan& t, uint32_t p, const uint32_t* h);
int plan_ensure_dense(fmx_matrix* m, int64_t tile, hipStream_t stream);
// scratch of the owner partition (a 4-bit radix sort of the directory), grow-only
struct OwnerWorkspace {
  uint32_t cap = 0;
  uint32_t *keys_in = nullptr, *keys_out = nullptr, *idx_in = nullptr, *idx_out = nullptr;
  void* temp = nullptr;
  size_t temp_bytes = 0;
  int reserve(uint32_t n, hipStream_t stream);
  OwnerWorkspace() = default;
  OwnerWorkspace(const OwnerWorkspace&) = delete;
  OwnerWorkspace& operator=(const OwnerWorkspace&) = delete;
  ~OwnerWorkspace();
};
// Owner-major order of tile t's sparse directory for n_owners owners; n_sort lists are sorted (the host's n_lists, or the
// directory's capacity while the count is still on the device: slots beyond dcounts[0] sort last).  Enqueues only.
int plan_owner_alloc(fmx_matrix::TilePlan& t, uint32_t cap_lists);
int plan_owner_build(fmx_matrix::TilePlan& t, OwnerWorkspace& ws, int n_owners, uint32_t n_sort, hipStream_t stream);
// rows (V row | w | 0 0 0) of the listed features <-> a packed buffer [n][kp + 4] in the state's element type
int rows_pack(fmx_engine* e, const uint32_t* d_ids, int64_t n, void* d_rows, bool unpack);
void drop_plans(fmx_matrix* m);
int build_batch_csc(fmx_matrix* m, int64_t batch_rows, int64_t tile_rows, hipStream_t stream);
void debug_fail_next_plan_build();
void debug_lose_next_seq_multiplier();        // fm_seq_kernels.hip
void debug_stall_next_persistent_sweep();     // fm_als_kernels.hip
int build_full_csc(fmx_matrix* m, hipStream_t stream);
int generate_synthetic(fmx_matrix* m, int32_t nnz_per_row, uint64_t seed, int64_t row_offset);
int generate_synthetic_async(fmx_matrix* m, int64_t n, int32_t z, uint64_t seed, int64_t row_offset, hipStream_t stream);
void strata_bounds(uint32_t p, int32_t z, std::vector<uint32_t>* out);
int generate_ragged(int device, int64_t n, uint32_t p, double mean, int lo, int hi, uint64_t seed, int64_t row_offset, fmx_matrix** out);
int matrix_values_uniform(fmx_matrix* m, uint64_t seed, int64_t row_offset);
int generate_iid_async(fmx_matrix* m, int64_t n, int32_t z, uint64_t seed, int64_t row_offset, int kind, double s_exp, hipStream_t stream);
constexpr int FMX_MAX_FIELDS = 64;
struct FieldSpec {  // Criteo-shaped generator (passed to the kernel by value)
  int n_dense, n_fields;
  double skew;
  uint32_t base[FMX_MAX_FIELDS], vocab[FMX_MAX_FIELDS];
};
int generate_fields_async(fmx_matrix* m, int64_t n, const FieldSpec& fs, uint64_t seed, int64_t row_offset, hipStream_t stream);
int generate_fields_split_async(fmx_matrix* m, int64_t n, const FieldSpec& fs, uint64_t seed, int64_t row_offset, hipStream_t stream, uint32_t* keys_sorted, uint32_t* brow,
                                float* bval, uint32_t* keys_in);
int check_rows_sorted(fmx_matrix* m);
int params_to_device(fmx_engine* e, const double* w, const double* v);
int params_from_device(fmx_engine* e, double* w, double* v);
int ingest_host_arrays(fmx_matrix* m, const void* values, bool values_f64, const void* cols, bool cols_signed, const int32_t* row_size, const int64_t* row_ptr,
                       const void* labels, bool labels_f64, uint64_t bad[2], int64_t* total);
int matrix_set_fields(fmx_matrix* m, int n_dense, int n_fields, const uint32_t* base);
int init_normal(fmx_engine* e, uint64_t seed, double mean, double stdev);
int rows_copy(fmx_engine* e, const uint32_t* d_ids, int64_t n, double* d_w, double* d_v, bool set);
int matrix_scales(fmx_matrix* m, const uint8_t* h_listed, double* h_mean, double* h_std);
int matrix_normalize(fmx_matrix* m, const double* h_mean, const double* h_std);

int launch_als_train(fmx_engine* e, fmx_matrix* m, int max_iter, int with_v);
int als_plan_info(fmx_engine* e, fmx_matrix* m, int64_t* levels, int64_t* largest, int32_t* approx, int32_t* level_of);
int launch_mcmc_train(fmx_engine* e, fmx_matrix* m, int max_iter, const double* h_gammas, const double* h_normals, double* h_state, const double* h_state_in = nullptr);
int launch_mcmc_v_hyper(fmx_engine* e, const double* h_gammas, const double* h_normals, double* v_lambda, double* v_mu, int sample);
int launch_als_vsweep(fmx_engine* e, fmx_matrix* m, double* d_error, double* d_q, double alpha, const double* h_lambda, const double* h_mu,
                      const double* d_znorm);
void als_graph_free(void* g);
// What changes from factor to factor (and from call to call) in a sweep: it lives in device memory and every sweep kernel reads it from
// there, so that the launches of one factor's sweep are IDENTICAL for every factor and every call (a deep plan can then be replayed as a graph).
struct SweepDyn {
  int f, pad;
  double alpha, lambda, mu;
  const double* znorm;   // this factor's (or w's) standard normals, or null: the ALS mean
};
// the row-tiled form of the wide levels (fm_als_tiled.hip)
int als_tiled_build(fmx_matrix* m, hipStream_t stream);
void als_tiled_free(fmx_matrix* m);
int als_tiled_info(const fmx_matrix* m, int32_t* levels_tiled, int64_t* tile_rows, int32_t* n_tiles);
// last: no further level of this factor's sweep follows (the fold of the next factor's q, e->als_qnext, happens there)
template <bool W>
int als_tiled_level(fmx_engine* e, fmx_matrix* m, int level, bool last, double2* d_qe, const SweepDyn* dyn, bool* done);
// the level-order form of the V sweep on complete plans (fm_als_tiled.hip)
bool als_order_ready(const fmx_matrix* m);
int als_order_levels(const fmx_matrix* m);
int als_order_enter(fmx_engine* e, fmx_matrix* m, const double2* d_qe, const double* d_Q0, bool* ok);
// d_qnext: the tile form folds the next factor's q (row order) into a factor's LAST level; the block form takes a factor's q (level 0's array order) in at its FIRST
int als_order_level(fmx_engine* e, fmx_matrix* m, int slot, const SweepDyn* dyn, const double* d_qnext, double* d_qprev_out = nullptr);
// the block form builds q for all factors on a copy of the CSR whose rows are in level 0's array order: the copy (and the pair buffers, allocated here), or *colP =
// null where the sweep will not take the block form (no such plan, no memory)
int als_order_prepare(fmx_engine* e, fmx_matrix* m, const uint32_t** colP, const float** valP, const uint32_t** row0);   // row0 (optional): the row at every position of that order
int als_order_exit(fmx_engine* e, fmx_matrix* m, double2* d_qe, double* d_qlast_out = nullptr);
uint64_t als_order_plan_uid(const fmx_matrix* m);   // the block form's plan (0: none)
// the w sweep (update_w, :208-256) through the block form: one kernel per level; *done = false: no such plan (or no memory), the caller takes the other forms
int als_order_w_sweep(fmx_engine* e, fmx_matrix* m, double2* d_qe, const SweepDyn* dyn, bool* done);
// the BLOCK form of the level-order sweep (fm_als_blocks.hip): the level's array feature-block-major, ONE kernel per level.  Built by als_tiled_build on
// complete plans whose lists all fit a block; *out stays null where it does not apply.
struct AlsBlocksIn {
  int64_t n; int n_slots;
  const uint32_t *lvl0, *cnt;      // [n_slots] (host) first feature of every level in the feats arrays, features per level
  const uint32_t* h_feats;         // (host) feature ids by (level, index)
  const uint32_t* d_feats;         // (device) the same
  int unit;                        // every stored value is 1.0f
};
int als_blocks_build(fmx_matrix* m, const AlsBlocksIn& in, void** out, hipStream_t stream);
void als_blocks_free(void* b);
int als_blocks_info(const void* b, int32_t* block_rows, int32_t* blocks_level0);
int als_blocks_enter(fmx_engine* e, const void* b, const double2* d_qe, const double* d_Q0, double2* dst);
int als_blocks_level(fmx_engine* e, const void* b, int s, const double2* src, double2* dst, const uint32_t* d_feats, const SweepDyn* dyn, const double* d_qin, double* d_qprev_out, bool w_sweep = false);
uint64_t als_blocks_uid(const void* b);
int als_vhash(fmx_engine* e, uint64_t* out);
void als_blocks_csr(const void* b, const uint32_t** colP, const float** valP, const uint32_t** row0);
int als_blocks_exit(fmx_engine* e, const void* b, const double2* src, double2* d_qe, double* d_qlast_out, bool e_only = false);
int als_order_form(const fmx_matrix* m);   // 0: none, 1: the tile form, 2: the block form
int launch_als_vsweep_device(fmx_engine* e, fmx_matrix* m, double* d_error, double alpha, const double* h_lambda, const double* h_mu, const double* d_znorm);

int evaluate_device(fmx_engine* e, const double* d_yhat, const float* d_y, int64_t n, int metric, double* result);

// N GPUs behind one handle (fm_group.hip)
int group_create(fmx_engine* e);
void group_destroy(fmx_engine* e);
int group_set_params(fmx_engine* e, double w0, const double* w, const double* v);
int group_init_normal(fmx_engine* e, uint64_t seed, double mean, double stdev);
int group_set_rows(fmx_engine* e, const uint32_t* ids, int64_t n, const double* w, const double* v);
bool group_outside(const fmx_engine* e);  // a cfg.n_gpus > 1 handle called from outside fmx_train
int group_load(fmx_engine* e, const char* path);
// after_step (may be null): called after every global step with the example indices [first, last] it covered; *stop = true ends the training
// (the tracker of fmx_train_tracked: core/Tracker.h)
using GroupStepHook = std::function<int(int64_t first, int64_t last, bool* stop)>;
int group_train(fmx_engine* e, fmx_matrix* m, int64_t max_iter, int64_t* examples_done, const GroupStepHook* after_step = nullptr);
int group_train_stream(fmx_engine* e, const fmx_fields_spec* spec, int32_t nnz_per_row, uint64_t seed, int64_t row_offset, int64_t total_rows,
                       int64_t* examples_done, double* ingest_wait_s);
void free_matrix(fmx_matrix* m);
void engine_tables(fmx_engine* e, std::vector<std::pair<void*, size_t>>* out, bool params_only);
int group_make_replicated(fmx_engine* e);  // every replica's copy of every table current again (no-op unless owner-sharded steps ran since)
int group_grad_empty(fmx_engine* e, fmx_matrix* m, int64_t batch);
int group_grad_compact(fmx_engine* e, fmx_matrix* m, int64_t batch, int64_t rows);
int group_rccl_selftest(int n, double* max_err);
int group_info(const fmx_engine* e, int32_t* n, int32_t* shared, int32_t* peer_pairs, int32_t* peer_direct, int32_t* sparse_exchange);
void debug_fail_next_comm_init();
int use_device_public(int device);
int alloc_matrix_public(int device, int64_t n, uint32_t p, int64_t nnz, bool labels, fmx_matrix** out);

// profiling helpers
void prof_begin(fmx_engine* e, int kernel);
void prof_end(fmx_engine* e);

}  // namespace fmx
