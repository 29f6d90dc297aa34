// N GPUs behind the C ABI: cfg.n_gpus > 1 makes fmx_train() a synchronous data-parallel run over N replicas driven by ONE host
// thread, so that the reference's single R process (src/FM.cpp:59,97 plumb `nthreads`; here it is `n_gpus`) scales without any
// other runtime.  The reference has no counterpart; BASELINE.json's north_star prescribes the scheme:
// rows shard contiguously by rank, every replica holds the full (w0, w, V), one all-reduce(sum) of the gradient-sum buffer per
// step, every replica applies the identical update.
//
//   exchange on distinct devices : RCCL (ncclCommInitAll, one grouped ncclAllReduce per step, enqueued on each engine's own
//                                  stream: no host synchronisation inside a step).  librccl is loaded on first use with
//                                  dlopen -- libfmx.so has no link-time dependency on it, and a process that brought its own
//                                  copy (a PyTorch wheel does) keeps using that one.
//   exchange on ONE device       : cfg.gpus_share_device = 1 places all replicas on cfg.device (rehearsal / tests on a one-GPU
//                                  box): a kernel adds the N buffers in rank order and writes the sum back to each.
//
// Shards are cut from the caller's matrix on the device (peer copies + a row_ptr rebase), cached per matrix.
#include <dlfcn.h>

#include <atomic>

#include <hip/hip_runtime.h>

#include "fmx_internal.h"

namespace fmx {

// ---- the handful of RCCL entry points, resolved at run time ---------------------------------------------------------------
typedef void* rcclComm_t;
struct Rccl {
  void* lib = nullptr;
  int (*CommInitAll)(rcclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
constexpr int RCCL_SUM = 0, RCCL_FLOAT32 = 7, RCCL_FLOAT64 = 8;  // ncclRedOp_t / ncclDataType_t values of nccl.h (rccl.h)

static Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.lib ? &r : nullptr;
  tried = true;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.lib) break;
  }
  if (!r.lib) return nullptr;
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
  r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
  r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  if (!r.CommInitAll || !r.CommDestroy || !r.AllReduce || !r.AllGather || !r.GroupStart || !r.GroupEnd) { dlclose(r.lib); r.lib = nullptr; return nullptr; }
  return &r;
}

#define FMX_RCCL(call)                                                                                               \
  do {                                                                                                               \
    const int _r = (call);                                                                                           \
    if (_r != 0) {                                                                                                   \
      Rccl* _l = rccl();                                                                                             \
      fmx::set_error("%s failed: %s", #call, _l && _l->GetErrorString ? _l->GetErrorString(_r) : "rccl error");       \
      return FMX_ERR_HIP;                                                                                            \
    }                                                                                                                \
  } while (0)

// ---- exchange among replicas that share one device: sum in rank order, written back to every buffer ------------------------
constexpr int GROUP_MAX = 16;
struct BufList { void* b[GROUP_MAX]; int n; };

template <typename ST>
__global__ void sum_buffers_k(BufList bl, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  ST acc = reinterpret_cast<const ST*>(bl.b[0])[i];
  for (int r = 1; r < bl.n; ++r) acc = acc + reinterpret_cast<const ST*>(bl.b[r])[i];  // rank order, in the buffer's own type (what an all-reduce does)
  for (int r = 0; r < bl.n; ++r) reinterpret_cast<ST*>(bl.b[r])[i] = acc;
}

__global__ void rebase_rows_k(const int64_t* __restrict__ src, int64_t r0, int64_t n, int64_t base, int64_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n) dst[i] = src[r0 + i] - base;
}

struct Group {
  int n = 0;
  std::vector<fmx_engine*> rep;   // rep[0] is the caller's handle
  std::vector<int> dev;
  bool shared = false;
  std::vector<rcclComm_t> comm;   // distinct devices
  hipStream_t xs = nullptr;       // shared device: the exchange stream
  std::vector<hipEvent_t> ready;  // per replica: gradient sums written
  hipEvent_t summed = nullptr;
  // shards of the last matrix trained on
  uint64_t src_uid = 0;  // fmx_matrix::uid of the matrix the shards were cut from (0: none); an address can be reused, a uid is not
  uint64_t src_values = 0;
  std::vector<fmx_matrix*> shard;
  // compact exchange (steps of one sparse tile): every replica's gather buffer [N][stride][record]
  std::vector<void*> gath;
  int64_t gath_records = 0;
  bool busy = false;  // group_train is driving the replicas: the step-level entry points are its own calls, not a caller's
  // owner-sharded exchange (exchange_owner below): per replica, grow-only
  struct OwnerBuf {
    void *req = nullptr, *rows_out = nullptr, *rows_in = nullptr, *parts = nullptr;   // ids asked of me, the rows I answer with, the rows I asked for, record slices sent to me
    int64_t req_cap = 0, out_cap = 0, in_cap = 0, parts_cap = 0;                      // in ids / rows / rows / records
  };
  std::vector<OwnerBuf> ox;
  std::vector<hipEvent_t> ev_ids, ev_packed, ev_grad, ev_done;   // per replica, recorded on its stream
  // peer access between the replicas' devices, asked for at creation (distinct devices): ordered pairs (a, b), a != b, for which device a may address device b's memory
  // directly -- hipMemcpyPeerAsync (the shards, the owner-sharded exchange's slices, set_params) then goes over xGMI without a bounce through the host.
  int peer_pairs = 0, peer_direct = 0;
  bool owners_on = false;     // the replicas are configured as owners (records and ids come out owner-major)
  bool owner_dirty = false;   // owner-sharded steps ran: a feature's tables are current at its owner (V and w also on replica 0) and nowhere else
};

// A group handle reached from outside group_train: the step-level mutators would change replica 0 alone
bool group_outside(const fmx_engine* e) { return e->group != nullptr && !e->group->busy; }

static void free_shards(Group* g) {
  for (size_t r = 0; r < g->shard.size(); ++r)
    if (g->shard[r]) { (void)hipSetDevice(g->dev[r]); fmx_matrix_destroy(g->shard[r]); }
  g->shard.clear();
  g->src_uid = 0;
}

void group_destroy(fmx_engine* e) {
  Group* g = e->group;
  if (!g) return;
  free_shards(g);
  for (size_t r = 0; r < g->gath.size(); ++r) if (g->gath[r]) { (void)hipSetDevice(g->dev[r]); (void)hipFree(g->gath[r]); }
  for (size_t r = 0; r < g->ox.size(); ++r) {
    (void)hipSetDevice(g->dev[r]);
    (void)hipFree(g->ox[r].req); (void)hipFree(g->ox[r].rows_out); (void)hipFree(g->ox[r].rows_in); (void)hipFree(g->ox[r].parts);
  }
  for (auto* v : {&g->ev_ids, &g->ev_packed, &g->ev_grad, &g->ev_done})
    for (size_t r = 0; r < v->size(); ++r) if ((*v)[r]) { (void)hipSetDevice(g->dev[r]); (void)hipEventDestroy((*v)[r]); }
  Rccl* l = rccl();
  for (rcclComm_t c : g->comm) if (c && l) (void)l->CommDestroy(c);
  (void)hipSetDevice(e->cfg.device);
  for (hipEvent_t ev : g->ready) if (ev) (void)hipEventDestroy(ev);
  if (g->summed) (void)hipEventDestroy(g->summed);
  if (g->xs) (void)hipStreamDestroy(g->xs);
  for (int r = 1; r < g->n; ++r) fmx_engine_destroy(g->rep[(size_t)r]);
  delete g;
  e->group = nullptr;
}

// test hook (fmx_test_hooks.h): the next group creation fails where RCCL is initialised, after the other replicas exist -- what a failing
// ncclCommInitAll on some device i > 0 leaves behind must be torn down by the caller's error path (fmx_engine_create's deleter -> group_destroy)
static std::atomic<int> g_fail_next_comm_init{0};
void debug_fail_next_comm_init() { g_fail_next_comm_init.store(1); }

int group_create(fmx_engine* e) {
  const int n = e->cfg.n_gpus;
  FMX_CHECK(n >= 2 && n <= GROUP_MAX, FMX_ERR_INVALID, "n_gpus must be in 1..%d", GROUP_MAX);
  FMX_CHECK(e->cfg.mode == FMX_MODE_MINIBATCH, FMX_ERR_INVALID,
            "n_gpus > 1 needs FMX_MODE_MINIBATCH: the reference's per-example algorithm does not shard (every example depends on the one before)");
  FMX_CHECK(e->cfg.solver == FMX_SOLVER_SGD || e->cfg.solver == FMX_SOLVER_FTRL || e->cfg.solver == FMX_SOLVER_TDAP, FMX_ERR_INVALID,
            "n_gpus > 1 trains SGD / FTRL / TDAP engines (the ALS / MCMC sweeps run as replicas only)");
  Group* g = new Group();
  e->group = g;
  g->n = n;
  g->shared = e->cfg.gpus_share_device != 0;
  g->rep.assign((size_t)n, nullptr);
  g->rep[0] = e;
  int count = 0;
  FMX_HIP(hipGetDeviceCount(&count));
  for (int r = 0; r < n; ++r) g->dev.push_back(g->shared ? e->cfg.device : e->cfg.device + r);
  FMX_CHECK(g->dev.back() < count, FMX_ERR_INVALID, "n_gpus = %d starting at device %d, but only %d devices are visible", n, e->cfg.device, count);
  for (int r = 1; r < n; ++r) {
    fmx_config c = e->cfg;
    c.n_gpus = 1;
    c.device = g->dev[(size_t)r];
    FMX_TRY(fmx_engine_create(&c, e->p, &g->rep[(size_t)r]));
  }
  g->ready.assign((size_t)n, nullptr);
  if (g_fail_next_comm_init.exchange(0) > 0) {
    set_error("ncclCommInitAll failed on device %d (forced: fmx_debug_fail_next_comm_init)", g->dev[(size_t)n - 1]);
    return FMX_ERR_HIP;   // (the caller destroys the engine: group_destroy frees the replicas made above)
  }
  if (g->shared) {
    FMX_HIP(hipSetDevice(e->cfg.device));
    FMX_HIP(hipStreamCreateWithFlags(&g->xs, hipStreamNonBlocking));
    FMX_HIP(hipEventCreateWithFlags(&g->summed, hipEventDisableTiming));
    for (int r = 0; r < n; ++r) FMX_HIP(hipEventCreateWithFlags(&g->ready[(size_t)r], hipEventDisableTiming));
  } else {
    Rccl* l = rccl();
    FMX_CHECK(l != nullptr, FMX_ERR_STATE, "n_gpus > 1 on distinct devices needs librccl.so (not found by dlopen)");
    g->comm.assign((size_t)n, nullptr);
    FMX_RCCL(l->CommInitAll(g->comm.data(), n, g->dev.data()));
    // direct peer access wherever the devices allow it (all pairs of one xGMI hive); a pair that refuses keeps working through staged copies
    for (int a = 0; a < n; ++a) {
      FMX_HIP(hipSetDevice(g->dev[(size_t)a]));
      for (int b = 0; b < n; ++b) {
        if (a == b) continue;
        ++g->peer_pairs;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, g->dev[(size_t)a], g->dev[(size_t)b]) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        if (!can) continue;
        const hipError_t pe = hipDeviceEnablePeerAccess(g->dev[(size_t)b], 0);
        if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) ++g->peer_direct;
        (void)hipGetLastError();
      }
    }
  }
  FMX_HIP(hipSetDevice(e->cfg.device));
  return FMX_OK;
}

int group_load(fmx_engine* e, const char* path) {
  Group* g = e->group;
  for (int r = 1; r < g->n; ++r) FMX_TRY(fmx_engine_load(g->rep[(size_t)r], path));
  g->owner_dirty = false;   // every replica holds every table again (the file's): nothing left for an owner refresh to move
  return use_device_public(e->cfg.device);
}

// the handle (replica 0) holds the caller's model already: the other replicas take their copy from IT, device to device (one trip over PCIe for
// the job instead of N: 8.4 GB of doubles at configs[3]'s shape), after the same reset of optimizer state and traces that fmx_set_params does
int group_set_params(fmx_engine* e, double w0, const double* w, const double* v) {
  (void)w; (void)v;
  Group* g = e->group;
  std::vector<std::pair<void*, size_t>> src;
  engine_tables(e, &src, true);
  FMX_HIP(hipSetDevice(e->cfg.device));
  FMX_HIP(hipStreamSynchronize(e->stream));
  for (int r = 1; r < g->n; ++r) {
    FMX_TRY(fmx_set_params(g->rep[(size_t)r], w0, nullptr, nullptr));   // zeros + the reset; then the tables themselves
    std::vector<std::pair<void*, size_t>> dst;
    engine_tables(g->rep[(size_t)r], &dst, true);
    FMX_CHECK(dst.size() == src.size(), FMX_ERR_STATE, "replica %d has another table layout", r);
    for (size_t t = 0; t < src.size(); ++t) {
      FMX_CHECK(dst[t].second == src[t].second, FMX_ERR_STATE, "replica %d has another table layout", r);
      const size_t bytes = src[t].second * (size_t)e->p;
      if (g->dev[(size_t)r] == g->dev[0]) FMX_HIP(hipMemcpy(dst[t].first, src[t].first, bytes, hipMemcpyDeviceToDevice));
      else FMX_HIP(hipMemcpyPeer(dst[t].first, g->dev[(size_t)r], src[t].first, g->dev[0], bytes));
    }
  }
  FMX_HIP(hipDeviceSynchronize());
  g->owner_dirty = false;   // every replica holds every table again
  return use_device_public(e->cfg.device);
}

// the same draw (seed, feature, factor pair) on every other replica: V0 must be identical everywhere or the replicas never agree
int group_init_normal(fmx_engine* e, uint64_t seed, double mean, double stdev) {
  Group* g = e->group;
  for (int r = 1; r < g->n; ++r) FMX_TRY(fmx_init_normal(g->rep[(size_t)r], seed, mean, stdev));
  return use_device_public(e->cfg.device);
}

int group_set_rows(fmx_engine* e, const uint32_t* ids, int64_t n, const double* w, const double* v) {
  Group* g = e->group;
  for (int r = 1; r < g->n; ++r) FMX_TRY(fmx_set_rows(g->rep[(size_t)r], ids, n, w, v));
  return use_device_public(e->cfg.device);
}

// rows [r0, r1) of src as a matrix of its own on device `dev`
static int cut_shard(const fmx_matrix* src, int64_t r0, int64_t r1, int dev, fmx_matrix** out) {
  FMX_HIP(hipSetDevice(src->device));
  int64_t ends[2] = {0, 0};
  FMX_HIP(hipMemcpy(&ends[0], src->row_ptr + r0, sizeof(int64_t), hipMemcpyDeviceToHost));
  FMX_HIP(hipMemcpy(&ends[1], src->row_ptr + r1, sizeof(int64_t), hipMemcpyDeviceToHost));
  const int64_t base = ends[0], cnt = ends[1] - ends[0], n = r1 - r0;
  fmx_matrix* m = nullptr;
  FMX_TRY(alloc_matrix_public(dev, n, src->p, cnt, src->has_labels != 0, &m));
  auto body = [&]() -> int {
    FMX_HIP(hipSetDevice(dev));
    if (dev == src->device) {
      hipLaunchKernelGGL(rebase_rows_k, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, nullptr, src->row_ptr, r0, n, base, m->row_ptr);
    } else {  // the source's row_ptr is on another device: stage the slice, then rebase in place
      FMX_HIP(hipMemcpyPeer(m->row_ptr, dev, src->row_ptr + r0, src->device, ((size_t)n + 1) * sizeof(int64_t)));
      hipLaunchKernelGGL(rebase_rows_k, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, nullptr, m->row_ptr, (int64_t)0, n, base, m->row_ptr);
    }
    if (cnt > 0) {
      FMX_HIP(hipMemcpyPeer(m->col, dev, src->col + base, src->device, (size_t)cnt * sizeof(uint32_t)));
      FMX_HIP(hipMemcpyPeer(m->val, dev, src->val + base, src->device, (size_t)cnt * sizeof(float)));
    }
    if (n > 0) FMX_HIP(hipMemcpyPeer(m->y, dev, src->y + r0, src->device, (size_t)n * sizeof(float)));
    FMX_HIP(hipGetLastError());
    FMX_HIP(hipDeviceSynchronize());
    return FMX_OK;
  };
  const int st = body();
  if (st != FMX_OK) { fmx_matrix_destroy(m); return st; }
  m->rows_sorted = src->rows_sorted;
  m->unit_values = src->unit_values;
  m->fixed_row_len = src->fixed_row_len;
  m->dense_prefix = src->dense_prefix;
  m->field_base = src->field_base;
  m->max_row_len = src->max_row_len;
  *out = m;
  return FMX_OK;
}

static int ensure_shards(Group* g, const fmx_matrix* m) {
  if (g->src_uid == m->uid && g->src_values == m->value_generation && !g->shard.empty()) return FMX_OK;
  free_shards(g);
  g->shard.assign((size_t)g->n, nullptr);
  for (int r = 0; r < g->n; ++r) {
    const int64_t r0 = (m->n * r) / g->n, r1 = (m->n * (r + 1)) / g->n;  // rank r gets rows [r n / N, (r + 1) n / N)
    FMX_TRY(cut_shard(m, r0, r1, g->dev[(size_t)r], &g->shard[(size_t)r]));
  }
  g->src_uid = m->uid; g->src_values = m->value_generation;
  return FMX_OK;
}

// one all-reduce(sum) of every replica's exchange buffer, ordered after its gradient kernels and before its update
static int exchange(Group* g) {
  void* buf[GROUP_MAX];
  int64_t count = 0;
  for (int r = 0; r < g->n; ++r) {
    FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
    FMX_TRY(fmx_grad_buffer(g->rep[(size_t)r], &buf[r], &count));
  }
  const bool wide = mb_wide(g->rep[0]);
  if (!g->shared) {
    Rccl* l = rccl();
    FMX_RCCL(l->GroupStart());
    for (int r = 0; r < g->n; ++r)
      FMX_RCCL(l->AllReduce(buf[r], buf[r], (size_t)count, wide ? RCCL_FLOAT64 : RCCL_FLOAT32, RCCL_SUM, g->comm[(size_t)r], g->rep[(size_t)r]->stream));
    FMX_RCCL(l->GroupEnd());
    return FMX_OK;
  }
  FMX_HIP(hipSetDevice(g->dev[0]));
  BufList bl{};
  bl.n = g->n;
  for (int r = 0; r < g->n; ++r) {
    bl.b[r] = buf[r];
    FMX_HIP(hipEventRecord(g->ready[(size_t)r], g->rep[(size_t)r]->stream));
    FMX_HIP(hipStreamWaitEvent(g->xs, g->ready[(size_t)r], 0));
  }
  const dim3 grid((unsigned)((count + 255) / 256)), block(256);
  if (wide) hipLaunchKernelGGL((sum_buffers_k<double>), grid, block, 0, g->xs, bl, count);
  else hipLaunchKernelGGL((sum_buffers_k<float>), grid, block, 0, g->xs, bl, count);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipEventRecord(g->summed, g->xs));
  for (int r = 0; r < g->n; ++r) FMX_HIP(hipStreamWaitEvent(g->rep[(size_t)r]->stream, g->summed, 0));
  return FMX_OK;
}

// ---- compact exchange inside the group (steps of one sparse tile: p >> entries per step, BASELINE.json configs[3]) -------------
// all-reduce of the 4-element tails, all-gather of the records (padded to the step's largest count), fmx_apply_compact on every
// replica: the same protocol fmwr_amd/distributed.py runs between processes
static int exchange_compact(Group* g, const std::vector<int64_t>& counts, int64_t stride) {
  const bool wide = mb_wide(g->rep[0]);
  const size_t eb = wide ? 8 : 4;
  const int64_t rec = g->rep[0]->rec_elems;
  void *recs[GROUP_MAX], *tails[GROUP_MAX];
  for (int r = 0; r < g->n; ++r) {
    int64_t n = 0;
    FMX_TRY(fmx_compact_records(g->rep[(size_t)r], &recs[r], &n, &tails[r]));
  }
  if (!g->shared) {
    Rccl* l = rccl();
    FMX_RCCL(l->GroupStart());
    for (int r = 0; r < g->n; ++r) {
      FMX_RCCL(l->AllReduce(tails[r], tails[r], 4, wide ? RCCL_FLOAT64 : RCCL_FLOAT32, RCCL_SUM, g->comm[(size_t)r], g->rep[(size_t)r]->stream));
      if (stride > 0)
        FMX_RCCL(l->AllGather(recs[r], g->gath[(size_t)r], (size_t)(stride * rec), wide ? RCCL_FLOAT64 : RCCL_FLOAT32, g->comm[(size_t)r], g->rep[(size_t)r]->stream));
    }
    FMX_RCCL(l->GroupEnd());
  } else {
    FMX_HIP(hipSetDevice(g->dev[0]));
    BufList bl{};
    bl.n = g->n;
    for (int r = 0; r < g->n; ++r) {
      bl.b[r] = tails[r];
      FMX_HIP(hipEventRecord(g->ready[(size_t)r], g->rep[(size_t)r]->stream));
      FMX_HIP(hipStreamWaitEvent(g->xs, g->ready[(size_t)r], 0));
    }
    if (wide) hipLaunchKernelGGL((sum_buffers_k<double>), dim3(1), dim3(256), 0, g->xs, bl, (int64_t)4);
    else hipLaunchKernelGGL((sum_buffers_k<float>), dim3(1), dim3(256), 0, g->xs, bl, (int64_t)4);
    FMX_HIP(hipGetLastError());
    for (int dst = 0; dst < g->n && stride > 0; ++dst)
      for (int src = 0; src < g->n; ++src)
        if (counts[(size_t)src] > 0)
          FMX_HIP(hipMemcpyAsync((char*)g->gath[(size_t)dst] + (size_t)src * stride * rec * eb, recs[src], (size_t)counts[(size_t)src] * rec * eb, hipMemcpyDeviceToDevice, g->xs));
    FMX_HIP(hipEventRecord(g->summed, g->xs));
    for (int r = 0; r < g->n; ++r) FMX_HIP(hipStreamWaitEvent(g->rep[(size_t)r]->stream, g->summed, 0));
  }
  for (int r = 0; r < g->n; ++r) {
    FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
    FMX_TRY(fmx_apply_compact(g->rep[(size_t)r], g->gath[(size_t)r], counts.data(), g->n, stride, 0));
  }
  return FMX_OK;
}

// ---- owner-sharded exchange inside the group (SURVEY 8(e) option (ii); include/fmx.h "owner-sharded exchange") -------------------------
// Feature j belongs to replica j mod N.  One process sees every device, so the three all-to-alls of the protocol are plain peer copies of
// contiguous slices (the plans and the records are owner-major), each enqueued on the RECEIVING replica's stream behind an event of the
// sending one: xGMI is point to point, and so is this.  No host synchronisation inside a step.
//   ids -> owners, rows back (pull) | fmx_grad_compact | record slices -> owners, tails all-reduced | fmx_apply_compact_parts at the owners
// The same additions in the same order as the all-gather form: bitwise equal to it (tests/test_gpu_group.py).
static int owner_setup(Group* g, bool on) {
  if (on && g->ev_ids.empty()) {
    g->ox.assign((size_t)g->n, Group::OwnerBuf());
    for (auto* v : {&g->ev_ids, &g->ev_packed, &g->ev_grad, &g->ev_done}) {
      v->assign((size_t)g->n, nullptr);
      for (int r = 0; r < g->n; ++r) { FMX_HIP(hipSetDevice(g->dev[(size_t)r])); FMX_HIP(hipEventCreateWithFlags(&(*v)[(size_t)r], hipEventDisableTiming)); }
    }
  }
  if (on != g->owners_on) {
    for (int r = 0; r < g->n; ++r) { FMX_HIP(hipSetDevice(g->dev[(size_t)r])); FMX_TRY(fmx_owner_configure(g->rep[(size_t)r], on ? g->n : 1, on ? r : 0)); }
    g->owners_on = on;
  }
  return FMX_OK;
}

static int owner_grow(Group* g, int r, void** buf, int64_t* cap, int64_t need, size_t unit) {
  if (need <= *cap && *buf) return FMX_OK;
  FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
  for (int q = 0; q < g->n; ++q) { FMX_HIP(hipSetDevice(g->dev[(size_t)q])); FMX_HIP(hipStreamSynchronize(g->rep[(size_t)q]->stream)); }  // (rare: copies in flight may read the old buffer)
  FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
  (void)hipFree(*buf); *buf = nullptr;
  const int64_t want = need + need / 8 + 1;
  FMX_HIP(hipMalloc(buf, (size_t)want * unit));
  *cap = want;
  return FMX_OK;
}

// one global step: replica r trains rows_limit[r] rows of step batch[r] of mats[r] (0 rows: an empty share)
static int exchange_owner(Group* g, fmx_matrix* const* mats, const int64_t* batch, const int64_t* rows) {
  const int N = g->n;
  const bool wide = mb_wide(g->rep[0]);
  const size_t eb = wide ? 8 : 4;
  const int64_t rowe = mb_kp(g->rep[0]) + 4;
  int64_t rec = 0;
  {  // (the engine's own rec_elems is set by its first fmx_grad_compact: a fresh streamed job has none yet)
    int64_t cap = 0; int32_t ok = 0;
    FMX_HIP(hipSetDevice(g->dev[0]));
    FMX_TRY(fmx_compact_info(g->rep[0], mats[0], &rec, &cap, &ok));
    FMX_CHECK(ok && rec > 0, FMX_ERR_STATE, "the step is not one sparse tile");
  }
  int64_t cnt[GROUP_MAX][GROUP_MAX];      // cnt[r][o]: lists of r's step owned by o
  int64_t soff[GROUP_MAX][GROUP_MAX + 1];  // where owner o's slice starts in r's owner-major arrays
  int64_t roff[GROUP_MAX][GROUP_MAX + 1];  // where r's slice starts in what owner o receives (rank order)
  void* ids[GROUP_MAX];
  for (int r = 0; r < N; ++r) {
    FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
    FMX_TRY(fmx_owner_info(g->rep[(size_t)r], mats[r], batch[r], cnt[r], &ids[r]));
    soff[r][0] = 0;
    for (int o = 0; o < N; ++o) soff[r][o + 1] = soff[r][o] + cnt[r][o];
  }
  for (int o = 0; o < N; ++o) {
    roff[o][0] = 0;
    for (int r = 0; r < N; ++r) roff[o][r + 1] = roff[o][r] + cnt[r][o];
  }
  for (int r = 0; r < N; ++r) {
    Group::OwnerBuf& b = g->ox[(size_t)r];
    FMX_TRY(owner_grow(g, r, &b.req, &b.req_cap, roff[r][N], sizeof(uint32_t)));
    FMX_TRY(owner_grow(g, r, &b.rows_out, &b.out_cap, roff[r][N], (size_t)rowe * eb));
    FMX_TRY(owner_grow(g, r, &b.rows_in, &b.in_cap, soff[r][N], (size_t)rowe * eb));
    FMX_TRY(owner_grow(g, r, &b.parts, &b.parts_cap, roff[r][N], (size_t)rec * eb));
  }
  auto copy = [&](void* dst, int ddev, const void* src, int sdev, size_t bytes, hipStream_t st) -> int {
    if (bytes == 0) return FMX_OK;
    if (g->dev[(size_t)ddev] == g->dev[(size_t)sdev]) FMX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    else FMX_HIP(hipMemcpyPeerAsync(dst, g->dev[(size_t)ddev], src, g->dev[(size_t)sdev], bytes, st));
    return FMX_OK;
  };
  // pull: ids to their owners ...
  for (int r = 0; r < N; ++r) { FMX_HIP(hipSetDevice(g->dev[(size_t)r])); FMX_HIP(hipEventRecord(g->ev_ids[(size_t)r], g->rep[(size_t)r]->stream)); }
  for (int o = 0; o < N; ++o) {
    FMX_HIP(hipSetDevice(g->dev[(size_t)o]));
    hipStream_t st = g->rep[(size_t)o]->stream;
    for (int r = 0; r < N; ++r) {
      if (r == o) continue;   // (a replica's own features are current where they are: nothing to ask, pack, send or store)
      FMX_HIP(hipStreamWaitEvent(st, g->ev_ids[(size_t)r], 0));
      FMX_TRY(copy((uint32_t*)g->ox[(size_t)o].req + roff[o][r], o, (const uint32_t*)ids[r] + soff[r][o], r, (size_t)cnt[r][o] * sizeof(uint32_t), st));
    }
    int64_t re = rowe;
    // the rows asked for by the ranks before and after this one (its own slice lies between them and stays unpacked)
    if (roff[o][o] > 0) FMX_TRY(fmx_rows_pack(g->rep[(size_t)o], g->ox[(size_t)o].req, roff[o][o], g->ox[(size_t)o].rows_out, &re));
    if (roff[o][N] > roff[o][o + 1])
      FMX_TRY(fmx_rows_pack(g->rep[(size_t)o], (const uint32_t*)g->ox[(size_t)o].req + roff[o][o + 1], roff[o][N] - roff[o][o + 1],
                            (char*)g->ox[(size_t)o].rows_out + (size_t)roff[o][o + 1] * rowe * eb, &re));
    FMX_CHECK(re == rowe, FMX_ERR_STATE, "row width changed");
    FMX_HIP(hipEventRecord(g->ev_packed[(size_t)o], st));
  }
  // ... the owners' current rows back, stored; then this replica's sums
  void *recs[GROUP_MAX], *tails[GROUP_MAX];
  for (int r = 0; r < N; ++r) {
    FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
    hipStream_t st = g->rep[(size_t)r]->stream;
    for (int o = 0; o < N; ++o) {
      if (o == r) continue;
      FMX_HIP(hipStreamWaitEvent(st, g->ev_packed[(size_t)o], 0));
      FMX_TRY(copy((char*)g->ox[(size_t)r].rows_in + (size_t)soff[r][o] * rowe * eb, r, (const char*)g->ox[(size_t)o].rows_out + (size_t)roff[o][r] * rowe * eb, o,
                   (size_t)cnt[r][o] * rowe * eb, st));
    }
    if (soff[r][r] > 0) FMX_TRY(fmx_rows_unpack(g->rep[(size_t)r], ids[r], soff[r][r], g->ox[(size_t)r].rows_in));
    if (soff[r][N] > soff[r][r + 1])
      FMX_TRY(fmx_rows_unpack(g->rep[(size_t)r], (const uint32_t*)ids[r] + soff[r][r + 1], soff[r][N] - soff[r][r + 1],
                              (const char*)g->ox[(size_t)r].rows_in + (size_t)soff[r][r + 1] * rowe * eb));
    FMX_TRY(group_grad_compact(g->rep[(size_t)r], mats[r], batch[r], rows[r]));
    int64_t n = 0;
    FMX_TRY(fmx_compact_records(g->rep[(size_t)r], &recs[r], &n, &tails[r]));
    FMX_CHECK(n == soff[r][N], FMX_ERR_STATE, "replica %d published %lld records, its plan lists %lld", r, (long long)n, (long long)soff[r][N]);
    FMX_HIP(hipEventRecord(g->ev_grad[(size_t)r], st));
  }
  // the 4-element tails: summed over the replicas (w0's sums and the global row count)
  if (!g->shared) {
    Rccl* l = rccl();
    FMX_RCCL(l->GroupStart());
    for (int r = 0; r < N; ++r)
      FMX_RCCL(l->AllReduce(tails[r], tails[r], 4, wide ? RCCL_FLOAT64 : RCCL_FLOAT32, RCCL_SUM, g->comm[(size_t)r], g->rep[(size_t)r]->stream));
    FMX_RCCL(l->GroupEnd());
  } else {
    FMX_HIP(hipSetDevice(g->dev[0]));
    BufList bl{};
    bl.n = N;
    for (int r = 0; r < N; ++r) { bl.b[r] = tails[r]; FMX_HIP(hipStreamWaitEvent(g->xs, g->ev_grad[(size_t)r], 0)); }
    if (wide) hipLaunchKernelGGL((sum_buffers_k<double>), dim3(1), dim3(256), 0, g->xs, bl, (int64_t)4);
    else hipLaunchKernelGGL((sum_buffers_k<float>), dim3(1), dim3(256), 0, g->xs, bl, (int64_t)4);
    FMX_HIP(hipGetLastError());
    FMX_HIP(hipEventRecord(g->summed, g->xs));
    for (int r = 0; r < N; ++r) FMX_HIP(hipStreamWaitEvent(g->rep[(size_t)r]->stream, g->summed, 0));
  }
  // push: record slices to their owners, which add a feature's parts in rank order and update it
  for (int o = 0; o < N; ++o) {
    FMX_HIP(hipSetDevice(g->dev[(size_t)o]));
    hipStream_t st = g->rep[(size_t)o]->stream;
    int64_t counts[GROUP_MAX], starts[GROUP_MAX];
    for (int r = 0; r < N; ++r) {
      if (r != o) FMX_HIP(hipStreamWaitEvent(st, g->ev_grad[(size_t)r], 0));
      FMX_TRY(copy((char*)g->ox[(size_t)o].parts + (size_t)roff[o][r] * rec * eb, o, (const char*)recs[r] + (size_t)soff[r][o] * rec * eb, r, (size_t)cnt[r][o] * rec * eb, st));
      counts[r] = cnt[r][o]; starts[r] = roff[o][r];
    }
    FMX_TRY(fmx_apply_compact_parts(g->rep[(size_t)o], g->ox[(size_t)o].parts, counts, starts, N, 0));
    FMX_HIP(hipEventRecord(g->ev_done[(size_t)o], st));
  }
  // nobody overwrites what a peer may still be reading (ids, rows, records of this step): every stream waits for every update
  for (int r = 0; r < N; ++r) {
    FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
    for (int o = 0; o < N; ++o) if (o != r) FMX_HIP(hipStreamWaitEvent(g->rep[(size_t)r]->stream, g->ev_done[(size_t)o], 0));
  }
  g->owner_dirty = true;
  return FMX_OK;
}

// rows j = first, first + step, ... of a table of `row_words` 4-byte words per feature <-> a packed buffer
__global__ void strided_rows_k(uint32_t* __restrict__ table, int64_t row_words, uint64_t first, uint64_t step, int64_t n_rows, uint32_t* __restrict__ packed, int scatter) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * row_words) return;
  const int64_t i = idx / row_words, t = idx - i * row_words;
  const size_t at = (size_t)(first + step * (uint64_t)i) * (size_t)row_words + (size_t)t;
  if (scatter) table[at] = packed[idx]; else packed[idx] = table[at];
}

// After owner-sharded steps: the owners' rows of the listed tables to replica `to` (-1: to every replica).  params_only: V and w.
static int owner_refresh(Group* g, int to, bool params_only) {
  const int N = g->n;
  const uint64_t p = g->rep[0]->p;
  const int64_t CHUNK_BYTES = 64LL << 20;
  std::vector<void*> stage((size_t)N, nullptr);
  std::vector<hipEvent_t> packed((size_t)N, nullptr), taken((size_t)N, nullptr);
  auto body = [&]() -> int {
    for (int r = 0; r < N; ++r) {
      FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
      FMX_HIP(hipMalloc(&stage[(size_t)r], (size_t)CHUNK_BYTES * 2));   // first half: what I pack for a peer; second half: what a peer packed for me
      FMX_HIP(hipEventCreateWithFlags(&packed[(size_t)r], hipEventDisableTiming));
      FMX_HIP(hipEventCreateWithFlags(&taken[(size_t)r], hipEventDisableTiming));
    }
    std::vector<std::vector<std::pair<void*, size_t>>> tabs((size_t)N);
    for (int r = 0; r < N; ++r) engine_tables(g->rep[(size_t)r], &tabs[(size_t)r], params_only);
    for (int dst = 0; dst < N; ++dst) {
      if (to >= 0 && dst != to) continue;
      for (int o = 0; o < N; ++o) {
        if (o == dst) continue;
        const int64_t n_own = p > (uint64_t)o ? (int64_t)((p - (uint64_t)o + (uint64_t)N - 1) / (uint64_t)N) : 0;
        for (size_t t = 0; t < tabs[(size_t)o].size(); ++t) {
          const int64_t words = (int64_t)(tabs[(size_t)o][t].second / 4);
          const int64_t per = CHUNK_BYTES / (words * 4) > 0 ? CHUNK_BYTES / (words * 4) : 1;
          for (int64_t i0 = 0; i0 < n_own; i0 += per) {
            const int64_t n = n_own - i0 < per ? n_own - i0 : per;
            const dim3 grid((unsigned)((n * words + 255) / 256)), blk(256);
            hipStream_t so = g->rep[(size_t)o]->stream, sd = g->rep[(size_t)dst]->stream;
            FMX_HIP(hipSetDevice(g->dev[(size_t)o]));
            for (int q = 0; q < N; ++q) FMX_HIP(hipStreamWaitEvent(so, taken[(size_t)q], 0));   // whatever was packed here before has left (copies run on the taker's stream)
            hipLaunchKernelGGL(strided_rows_k, grid, blk, 0, so, (uint32_t*)tabs[(size_t)o][t].first, words, (uint64_t)o + (uint64_t)N * (uint64_t)i0, (uint64_t)N, n,
                               (uint32_t*)stage[(size_t)o], 0);
            FMX_HIP(hipEventRecord(packed[(size_t)o], so));
            FMX_HIP(hipSetDevice(g->dev[(size_t)dst]));
            FMX_HIP(hipStreamWaitEvent(sd, packed[(size_t)o], 0));
            void* in = (char*)stage[(size_t)dst] + CHUNK_BYTES;
            if (g->dev[(size_t)dst] == g->dev[(size_t)o]) FMX_HIP(hipMemcpyAsync(in, stage[(size_t)o], (size_t)n * words * 4, hipMemcpyDeviceToDevice, sd));
            else FMX_HIP(hipMemcpyPeerAsync(in, g->dev[(size_t)dst], stage[(size_t)o], g->dev[(size_t)o], (size_t)n * words * 4, sd));
            FMX_HIP(hipEventRecord(taken[(size_t)dst], sd));
            hipLaunchKernelGGL(strided_rows_k, grid, blk, 0, sd, (uint32_t*)tabs[(size_t)dst][t].first, words, (uint64_t)o + (uint64_t)N * (uint64_t)i0, (uint64_t)N, n,
                               (uint32_t*)in, 1);
            FMX_HIP(hipGetLastError());
            // (one chunk in flight per pair: the next pack into stage[o] waits for `taken`, the next copy into `in` is behind this scatter on sd)
          }
        }
      }
    }
    for (int r = 0; r < N; ++r) { FMX_HIP(hipSetDevice(g->dev[(size_t)r])); FMX_HIP(hipStreamSynchronize(g->rep[(size_t)r]->stream)); }
    return FMX_OK;
  };
  const int st = body();
  for (int r = 0; r < N; ++r) {
    (void)hipSetDevice(g->dev[(size_t)r]);
    if (st != FMX_OK) (void)hipDeviceSynchronize();
    (void)hipFree(stage[(size_t)r]);
    if (packed[(size_t)r]) (void)hipEventDestroy(packed[(size_t)r]);
    if (taken[(size_t)r]) (void)hipEventDestroy(taken[(size_t)r]);
  }
  (void)hipSetDevice(g->rep[0]->cfg.device);
  return st;
}

int group_make_replicated(fmx_engine* e) {
  Group* g = e->group;
  if (!g || !g->owner_dirty) return FMX_OK;
  FMX_TRY(owner_refresh(g, -1, false));
  g->owner_dirty = false;
  return FMX_OK;
}

// which exchange a training call uses: FMX_GROUP_EXCHANGE = dense | compact | owner (read at every call).  Steps of one sparse tile default to the
// all-gather of records (compact).  The owner-sharded form moves fewer bytes, but between DISTINCT devices it is a protocol of peer copies ordered by
// cross-device events in which only the owner holds a feature's current optimizer state -- a missed edge would diverge the replicas silently -- and
// no box with two devices has run it yet (SCALE_r01..r03 skipped): there it is opt-in (FMX_GROUP_EXCHANGE=owner) until tests/test_gpu_group.py has
// passed on 2+ GPUs.  Replicas that share one device (cfg.gpus_share_device: the rehearsal every GPU suite runs, bitwise the all-gather form) take it
// by default, as before.  Nobody may look at the model between owner-sharded steps (`watched`: the tracker).
enum GroupMode { GM_DENSE = 0, GM_COMPACT = 1, GM_OWNER = 2 };
static GroupMode group_mode(const Group* g, bool compact_usable, bool watched) {
  const char* v = getenv("FMX_GROUP_EXCHANGE");
  if (!compact_usable || (v && v[0] == 'd')) return GM_DENSE;
  if ((v && v[0] == 'c') || watched) return GM_COMPACT;
  if (v && v[0] == 'o') return GM_OWNER;
  return g->shared ? GM_OWNER : GM_COMPACT;
}

// n replicas, sharing one device or not, ordered device pairs and how many of them have direct peer access, and the exchange steps of one sparse tile
// take by default (1: all-gather of records, 2: owner-sharded)
int group_info(const fmx_engine* e, int32_t* n, int32_t* shared, int32_t* peer_pairs, int32_t* peer_direct, int32_t* sparse_exchange) {
  const Group* g = e->group;
  if (n) *n = g ? g->n : 1;
  if (shared) *shared = g && g->shared ? 1 : 0;
  if (peer_pairs) *peer_pairs = g ? g->peer_pairs : 0;
  if (peer_direct) *peer_direct = g ? g->peer_direct : 0;
  if (sparse_exchange) *sparse_exchange = g ? (int32_t)group_mode(g, true, false) : 0;
  return FMX_OK;
}

// every replica's gather buffer holds N parts of `stride` records
static int ensure_gath(Group* g, int64_t stride) {
  const size_t eb = mb_wide(g->rep[0]) ? 8 : 4;
  const int64_t rec = g->rep[0]->rec_elems;
  if (g->gath.empty()) g->gath.assign((size_t)g->n, nullptr);
  if (stride <= g->gath_records && g->gath[0]) return FMX_OK;
  const int64_t want = stride + stride / 8 + 1;  // some headroom: the count moves a little from step to step
  for (int r = 0; r < g->n; ++r) {
    FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
    FMX_HIP(hipStreamSynchronize(g->rep[(size_t)r]->stream));
    (void)hipFree(g->gath[(size_t)r]); g->gath[(size_t)r] = nullptr;
    FMX_HIP(hipMalloc(&g->gath[(size_t)r], (size_t)g->n * (size_t)want * (size_t)rec * eb));
  }
  g->gath_records = want;
  return FMX_OK;
}

// every replica's steps are single sparse tiles?  then reserve the record buffers and the gather buffers once
static int prepare_compact(Group* g, bool* usable) {
  const char* v = getenv("FMX_GROUP_EXCHANGE");  // "dense": tests and A/B runs (read at every fmx_train)
  const bool allow = !(v && v[0] == 'd');
  *usable = false;
  if (!allow) return FMX_OK;
  int64_t max_n = 0, rec = 0;
  for (int r = 0; r < g->n; ++r) {
    int64_t cap = 0; int32_t ok = 0;
    FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
    FMX_TRY(fmx_compact_info(g->rep[(size_t)r], g->shard[(size_t)r], &rec, &cap, &ok));
    if (!ok) return FMX_OK;
    int64_t nb = 0;
    FMX_TRY(fmx_num_batches(g->rep[(size_t)r], g->shard[(size_t)r], &nb));
    for (int64_t b = 0; b < nb; ++b) {
      int64_t n = 0;
      FMX_TRY(fmx_compact_count(g->rep[(size_t)r], g->shard[(size_t)r], b, &n));
      if (n > max_n) max_n = n;
    }
  }
  const size_t eb = mb_wide(g->rep[0]) ? 8 : 4;
  if (g->gath.empty()) g->gath.assign((size_t)g->n, nullptr);
  for (int r = 0; r < g->n; ++r) {
    FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
    FMX_TRY(fmx_compact_reserve(g->rep[(size_t)r], max_n > 0 ? max_n : 1));  // slices of the step's largest count are sent from every replica
    if (max_n > g->gath_records || !g->gath[(size_t)r]) {
      (void)hipFree(g->gath[(size_t)r]); g->gath[(size_t)r] = nullptr;
      FMX_HIP(hipMalloc(&g->gath[(size_t)r], (size_t)g->n * (size_t)(max_n > 0 ? max_n : 1) * (size_t)rec * eb));
    }
  }
  if (max_n > g->gath_records) g->gath_records = max_n;
  *usable = true;
  return FMX_OK;
}

// Learner::learn over N replicas: global step s = local batch (s mod nb) of every shard; max_iter counts examples of the
// whole job, the last step is truncated rank by rank (lower ranks first).
int group_train(fmx_engine* e, fmx_matrix* m, int64_t max_iter, int64_t* examples_done, const GroupStepHook* after_step) {
  Group* g = e->group;
  struct Busy { Group* g; explicit Busy(Group* g_) : g(g_) { g->busy = true; } ~Busy() { g->busy = false; } } busy(g);
  FMX_TRY(ensure_shards(g, m));
  // fp32 exchange: counts travel as floats -- exact while every per-feature occurrence count of a global batch stays below 2^24
  FMX_CHECK(mb_wide(e) || e->cfg.batch_rows * g->n < (1LL << 24), FMX_ERR_INVALID,
            "batch_rows * n_gpus must stay below 2^24 with fp32 state (occurrence counts are exchanged as floats); use state_fp64 or smaller batches");
  std::vector<int64_t> nb((size_t)g->n, 0);
  int64_t nb_min = -1;
  for (int r = 0; r < g->n; ++r) {
    FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
    FMX_TRY(fmx_num_batches(g->rep[(size_t)r], g->shard[(size_t)r], &nb[(size_t)r]));
    if (nb_min < 0 || nb[(size_t)r] < nb_min) nb_min = nb[(size_t)r];
  }
  FMX_CHECK(nb_min >= 1, FMX_ERR_INVALID, "a shard is empty: fewer rows than GPUs");
  bool compact = false;
  FMX_TRY(prepare_compact(g, &compact));
  const GroupMode mode = group_mode(g, compact, after_step != nullptr);
  compact = mode != GM_DENSE;
  if (mode != GM_OWNER) FMX_TRY(group_make_replicated(e));   // the other forms apply every update on every replica: all copies must be current
  FMX_TRY(owner_setup(g, mode == GM_OWNER));
  std::vector<int64_t> counts((size_t)g->n, 0);
  int64_t done = 0;
  for (int64_t s = 0; done < max_iter; ++s) {
    int64_t left = max_iter - done;
    int64_t stride = 0;
    const int64_t step_first = done;
    int64_t o_batch[GROUP_MAX], o_rows[GROUP_MAX];
    for (int r = 0; r < g->n; ++r) {
      const fmx_matrix* sh = g->shard[(size_t)r];
      const int64_t b = s % nb[(size_t)r];  // shards differ by at most one row: their batch counts agree except for a ragged tail
      const int64_t b0 = b * e->cfg.batch_rows;
      int64_t rows = b0 + e->cfg.batch_rows <= sh->n ? e->cfg.batch_rows : sh->n - b0;
      if (rows > left) rows = left;
      FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
      o_batch[r] = b; o_rows[r] = rows;
      if (mode == GM_OWNER) {
        // (the step's kernels are enqueued by exchange_owner, behind the pull of the rows they read)
      } else if (compact) {
        // an empty share (the truncated last step) still publishes its records with zero counts, and its tail
        FMX_TRY(group_grad_compact(g->rep[(size_t)r], g->shard[(size_t)r], b, rows));
        FMX_TRY(fmx_compact_count(g->rep[(size_t)r], g->shard[(size_t)r], b, &counts[(size_t)r]));
        if (counts[(size_t)r] > stride) stride = counts[(size_t)r];
      } else if (rows > 0) {
        FMX_TRY(fmx_grad(g->rep[(size_t)r], g->shard[(size_t)r], b, rows));
      } else {
        FMX_TRY(group_grad_empty(g->rep[(size_t)r], g->shard[(size_t)r], b));  // publishes zeros
      }
      left -= rows;
      done += rows;
    }
    if (mode == GM_OWNER) {
      FMX_TRY(exchange_owner(g, g->shard.data(), o_batch, o_rows));
    } else if (compact) {
      FMX_TRY(exchange_compact(g, counts, stride));
    } else {
      FMX_TRY(exchange(g));
      for (int r = 0; r < g->n; ++r) {
        FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
        FMX_TRY(fmx_apply(g->rep[(size_t)r], 0));  // the global row count travelled in the buffer's tail
      }
    }
    if (after_step) {   // (replica 0 is the caller's handle: its tables hold the step's result, on its own stream)
      bool stop = false;
      FMX_HIP(hipSetDevice(e->cfg.device));
      FMX_TRY((*after_step)(step_first, done - 1, &stop));
      if (stop) break;
    }
  }
  if (mode == GM_OWNER && g->owner_dirty) FMX_TRY(owner_refresh(g, 0, true));   // the handle answers fmx_get_params / fmx_predict: its V and w current again
  for (int r = 0; r < g->n; ++r) {
    FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
    FMX_TRY(fmx_sync(g->rep[(size_t)r]));
  }
  FMX_HIP(hipSetDevice(e->cfg.device));
  if (examples_done) *examples_done = done;
  return FMX_OK;
}

// fmx_train_stream over N replicas: replica r streams rows [r T / N, (r + 1) T / N) of the caller's range -- the generators are keyed
// by the global row id, so a shard is just another row_offset -- plans each of its steps on its own device and the replicas
// exchange per step as in group_train: the records of the occurring features when the steps are sparse tiles (configs[3]:
// 33 M features against 10 M entries per step), the dense buffer otherwise.  A replica whose range ends a step early publishes
// an empty share.
int group_train_stream(fmx_engine* e, const fmx_fields_spec* spec, int32_t nnz_per_row, uint64_t seed, int64_t row_offset, int64_t total_rows,
                       int64_t* examples_done, double* ingest_wait_s) {
  Group* g = e->group;
  struct Busy { Group* g; explicit Busy(Group* g_) : g(g_) { g->busy = true; } ~Busy() { g->busy = false; } } busy(g);
  FMX_CHECK(total_rows >= 0, FMX_ERR_INVALID, "total_rows must be >= 0");
  FMX_CHECK(mb_wide(e) || e->cfg.batch_rows * g->n < (1LL << 24), FMX_ERR_INVALID,
            "batch_rows * n_gpus must stay below 2^24 with fp32 state (occurrence counts are exchanged as floats); use state_fp64 or smaller batches");
  if (total_rows == 0) return FMX_OK;
  const int N = g->n;
  std::vector<fmx_source*> S((size_t)N, nullptr);
  std::vector<fmx_matrix*> last((size_t)N, nullptr);
  std::vector<int64_t> counts((size_t)N, 0);
  int64_t done = 0;
  double waited = 0.0;
  // a streamed step is one tile: sparse (records / owners) when it holds fewer entries than there are features
  const int64_t z_row = spec ? (int64_t)spec->n_dense + spec->n_fields : (int64_t)nnz_per_row;
  const GroupMode mode = group_mode(g, e->cfg.batch_rows * z_row < (int64_t)e->p, false);
  if (mode != GM_OWNER) FMX_TRY(group_make_replicated(e));
  FMX_TRY(owner_setup(g, mode == GM_OWNER));   // (before the sources open: the ingest builds the owner-major order with the plan)
  auto body = [&]() -> int {
    int64_t steps = 0;
    for (int r = 0; r < N; ++r) {
      const int64_t r0 = (total_rows * r) / N, r1 = (total_rows * (r + 1)) / N;
      FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
      FMX_TRY(fmx_source_open(g->rep[(size_t)r], spec, nnz_per_row, seed, row_offset + r0, r1 - r0, &S[(size_t)r]));
      const int64_t st = (r1 - r0 + e->cfg.batch_rows - 1) / e->cfg.batch_rows;
      if (st > steps) steps = st;
    }
    FMX_CHECK(total_rows >= N, FMX_ERR_INVALID, "a shard is empty: fewer rows than GPUs");
    for (int64_t s = 0; s < steps; ++s) {
      int64_t stride = 0;
      bool compact = false;
      fmx_matrix* s_mats[GROUP_MAX];
      int64_t s_rows[GROUP_MAX];
      for (int r = 0; r < N; ++r) {
        fmx_engine* rep = g->rep[(size_t)r];
        fmx_matrix* m = nullptr;
        int64_t rows = 0;
        FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
        FMX_TRY(fmx_source_next(S[(size_t)r], &m, &rows));
        if (m) last[(size_t)r] = m; else m = last[(size_t)r];
        FMX_CHECK(m != nullptr, FMX_ERR_STATE, "a replica has no step to share");
        compact = m->plans[0].feat != nullptr && mode != GM_DENSE;
        s_mats[r] = m; s_rows[r] = rows;
        if (compact && mode == GM_OWNER) {
          // (enqueued by exchange_owner)
        } else if (compact) {
          FMX_TRY(group_grad_compact(rep, m, 0, rows));  // rows == 0: the shard ended a step early -- zero counts, an empty tail
          counts[(size_t)r] = (int64_t)m->plans[0].n_lists;
          if (counts[(size_t)r] > stride) stride = counts[(size_t)r];
        } else if (rows > 0) {
          FMX_TRY(fmx_grad(rep, m, 0, 0));
        } else {
          FMX_TRY(group_grad_empty(rep, m, 0));
        }
        done += rows;
      }
      if (compact && mode == GM_OWNER) {
        int64_t zero[GROUP_MAX] = {0};
        FMX_TRY(exchange_owner(g, s_mats, zero, s_rows));
      } else if (compact) {
        FMX_TRY(ensure_gath(g, stride > 0 ? stride : 1));
        FMX_TRY(exchange_compact(g, counts, stride));
      } else {
        FMX_TRY(exchange(g));
        for (int r = 0; r < N; ++r) {
          FMX_HIP(hipSetDevice(g->dev[(size_t)r]));
          FMX_TRY(fmx_apply(g->rep[(size_t)r], 0));
        }
      }
    }
    return FMX_OK;
  };
  int st = body();
  if (st == FMX_OK && mode == GM_OWNER && g->owner_dirty) st = owner_refresh(g, 0, true);
  for (int r = 0; r < N; ++r) {
    (void)hipSetDevice(g->dev[(size_t)r]);
    double w = 0.0;
    const int st2 = fmx_source_close(S[(size_t)r], &w);
    if (st == FMX_OK) st = st2;
    if (w > waited) waited = w;
  }
  (void)hipSetDevice(e->cfg.device);
  if (examples_done) *examples_done = done;
  if (ingest_wait_s) *ingest_wait_s = waited;
  return st;
}

// RCCL smoke test on the devices this process sees (n ranks on devices 0..n-1): all-reduce of a small buffer, checked.  Lets a
// one-GPU box prove that librccl loads, that the entry points have the signatures assumed above and that the enum values are
// right (n = 1: the collective is a copy).
int group_rccl_selftest(int n, double* max_err) {
  Rccl* l = rccl();
  FMX_CHECK(l != nullptr, FMX_ERR_STATE, "librccl.so not found by dlopen");
  int count = 0;
  FMX_HIP(hipGetDeviceCount(&count));
  FMX_CHECK(n >= 1 && n <= count && n <= GROUP_MAX, FMX_ERR_INVALID, "need 1 <= n <= visible devices (%d)", count);
  std::vector<int> dev((size_t)n);
  std::vector<rcclComm_t> comm((size_t)n, nullptr);
  std::vector<float*> f((size_t)n, nullptr);
  std::vector<double*> d((size_t)n, nullptr);
  std::vector<hipStream_t> st((size_t)n, nullptr);
  for (int r = 0; r < n; ++r) dev[(size_t)r] = r;
  const int N = 1000;
  int rc = FMX_OK;
  auto body = [&]() -> int {
    FMX_RCCL(l->CommInitAll(comm.data(), n, dev.data()));
    std::vector<float> hf(N);
    std::vector<double> hd(N);
    for (int r = 0; r < n; ++r) {
      FMX_HIP(hipSetDevice(r));
      FMX_HIP(hipStreamCreateWithFlags(&st[(size_t)r], hipStreamNonBlocking));
      FMX_HIP(hipMalloc(&f[(size_t)r], N * sizeof(float)));
      FMX_HIP(hipMalloc(&d[(size_t)r], N * sizeof(double)));
      for (int i = 0; i < N; ++i) { hf[(size_t)i] = (float)(i + 1) * (float)(r + 1); hd[(size_t)i] = (double)(i + 1) * 1e-3 * (double)(r + 1); }
      FMX_HIP(hipMemcpy(f[(size_t)r], hf.data(), N * sizeof(float), hipMemcpyHostToDevice));
      FMX_HIP(hipMemcpy(d[(size_t)r], hd.data(), N * sizeof(double), hipMemcpyHostToDevice));
    }
    FMX_RCCL(l->GroupStart());
    for (int r = 0; r < n; ++r) {
      FMX_RCCL(l->AllReduce(f[(size_t)r], f[(size_t)r], (size_t)N, RCCL_FLOAT32, RCCL_SUM, comm[(size_t)r], st[(size_t)r]));
      FMX_RCCL(l->AllReduce(d[(size_t)r], d[(size_t)r], (size_t)N, RCCL_FLOAT64, RCCL_SUM, comm[(size_t)r], st[(size_t)r]));
    }
    FMX_RCCL(l->GroupEnd());
    double err = 0.0;
    const double tri = (double)n * (n + 1) / 2.0;
    for (int r = 0; r < n; ++r) {
      FMX_HIP(hipSetDevice(r));
      FMX_HIP(hipStreamSynchronize(st[(size_t)r]));
      FMX_HIP(hipMemcpy(hf.data(), f[(size_t)r], N * sizeof(float), hipMemcpyDeviceToHost));
      FMX_HIP(hipMemcpy(hd.data(), d[(size_t)r], N * sizeof(double), hipMemcpyDeviceToHost));
      for (int i = 0; i < N; ++i) {
        const double e1 = fabs((double)hf[(size_t)i] - (double)(i + 1) * tri), e2 = fabs(hd[(size_t)i] - (double)(i + 1) * 1e-3 * tri) * 1e6;
        if (e1 > err) err = e1;
        if (e2 > err) err = e2;
      }
    }
    if (max_err) *max_err = err;
    return FMX_OK;
  };
  rc = body();
  for (int r = 0; r < n; ++r) {
    (void)hipSetDevice(r);
    (void)hipFree(f[(size_t)r]); (void)hipFree(d[(size_t)r]);
    if (st[(size_t)r]) (void)hipStreamDestroy(st[(size_t)r]);
    if (comm[(size_t)r]) (void)l->CommDestroy(comm[(size_t)r]);
  }
  (void)hipSetDevice(0);
  return rc;
}

}  // namespace fmx
