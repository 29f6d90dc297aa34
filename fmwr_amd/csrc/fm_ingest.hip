// Ingest side of the path (one-off per matrix, not the hot loop):
//   * per-batch CSC ("inverted index": for every feature the (local row, x) pairs of one batch in row order),
//     which phase 2 of the mini-batch step walks; built with a stable device radix sort by column;
//   * CSC of the whole matrix for the ALS sweep -- the reference builds it with an O(p*n) scan
//     (util/Smatrix.h:155-185, called at src/FM.cpp:148-152); result layout is the same (rows ascending per feature);
//   * the synthetic workload generator, Philox4x32-10 keyed by (seed, global row id);
//   * the strictly-ascending-rows check the sequential learner uses.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>
#include <cstring>  // rocprim's texture_cache_iterator.hpp uses memset without including it

#include <rocprim/rocprim.hpp>

#include "fmx_internal.h"

namespace fmx {

// fault injection for tests/test_gpu_api.py: the next build_batch_csc fails once (no environment access from inside the library)
static std::atomic<int> g_fail_next_plan_build{0};
void debug_fail_next_plan_build() { g_fail_next_plan_build.store(1); }

// ------------------------------------------------------------------------------------------------ CSC builders
// row of entry t: last r in [r0, r0+nrows) with row_ptr[r] <= t (rows of one fixed length: a division)
__device__ __forceinline__ uint32_t entry_row(const int64_t* __restrict__ row_ptr, int64_t r0, int64_t nrows, int64_t base, int64_t t, int fixed_len) {
  if (fixed_len > 0) return (uint32_t)((t - base) / fixed_len);
  int64_t lo = r0, hi = r0 + nrows;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (row_ptr[mid] <= t) lo = mid; else hi = mid;
  }
  return (uint32_t)(lo - r0);
}

__global__ void pack_entries_k(const int64_t* __restrict__ row_ptr, const float* __restrict__ val, int64_t r0, int64_t nrows,
                               int64_t base, int64_t cnt, uint64_t* __restrict__ packed, int fixed_len) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const int64_t t = base + i;
  packed[i] = ((uint64_t)entry_row(row_ptr, r0, nrows, base, t, fixed_len) << 32) | (uint64_t)__float_as_uint(val[t]);
}

// one-hot matrices: the payload is the row alone (the values are never read)
__global__ void pack_rows_k(const int64_t* __restrict__ row_ptr, int64_t r0, int64_t nrows, int64_t base, int64_t cnt, uint32_t* __restrict__ rows, int fixed_len) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cnt) rows[i] = entry_row(row_ptr, r0, nrows, base, base + i, fixed_len);
}

// sorted positions (parked, as raw bits, in the value array) -> the entries' rows and values
__global__ void gather_sorted_k(const int64_t* __restrict__ row_ptr, const float* __restrict__ val, int64_t r0, int64_t nrows, int64_t base, int64_t cnt,
                                uint32_t* __restrict__ rows, float* vals, int fixed_len) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const int64_t t = base + (int64_t)__float_as_uint(vals[i]);
  rows[i] = entry_row(row_ptr, r0, nrows, base, t, fixed_len);
  vals[i] = val[t];
}

__global__ void unpack_entries_k(const uint64_t* __restrict__ packed, int64_t cnt, uint32_t* __restrict__ rows, float* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const uint64_t v = packed[i];
  rows[i] = (uint32_t)(v >> 32);
  vals[i] = __uint_as_float((uint32_t)v);
}

// Offsets of the per-feature lists from the sorted columns: the last entry of a run of equal columns knows where the run
// ends (ptr[col + 1] = its index + 1, everything else 0); a running maximum then carries each end over the features that
// have no entry.  Two passes over the p + 1 offsets instead of a binary search per feature (33 M searches in a 10 M-entry
// tile cost 0.64 ms at configs[3]'s shape; this costs 0.1 ms).
template <typename OffT>
__global__ void run_ends_k(const uint32_t* __restrict__ sorted_cols, int64_t cnt, OffT* __restrict__ ptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  if (i == cnt - 1 || sorted_cols[i + 1] != sorted_cols[i]) ptr[(size_t)sorted_cols[i] + 1] = (OffT)(i + 1);
}

__global__ void gather_i64_k(const int64_t* __restrict__ src, int64_t stride, int64_t n, int64_t count, int64_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int64_t at = i * stride;
  dst[i] = src[at < n ? at : n];
}

// The (column, row) pair sort of one-hot tiles: rocprim's tuned default for gfx950 sorts 8 bits per pass -- 25 bits of column id
// (33 M features) are then FOUR passes, the last one for a single bit.  Nine bits per pass make it three.
// The tuned default also sorts 16 384 items per workgroup: a 6.8 M-entry tile is then 415 workgroups on 256 CUs.  Smaller workgroups
// (FMX_SORT_CFG = 1: 512 x 8, 2: 256 x 12 items) are compiled beside it; the default is the one measured fastest at the streamed
// tile's size (profiles/r03_sort_cfg.txt).
template <int BS, int IPT>
using PairSort9 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                             rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, 9,
                                                                                 rocprim::block_radix_rank_algorithm::match>>;
static bool nine_bit_passes(int bits) {
  static const bool ok = [] { const char* v = getenv("FMX_SORT9"); return !(v && v[0] == '0'); }();
  return ok && (bits + 8) / 9 < (bits + 7) / 8;
}
static int sort_cfg() { static const int c = [] { const char* v = getenv("FMX_SORT_CFG"); return v ? atoi(v) : 0; }(); return c; }
static hipError_t sort_pairs_u32(void* temp, size_t& bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, int bits, hipStream_t stream) {
  if (nine_bit_passes(bits)) {
    switch (sort_cfg()) {
      case 1: return rocprim::radix_sort_pairs<PairSort9<512, 8>>(temp, bytes, kin, kout, vin, vout, n, 0, bits, stream);
      case 2: return rocprim::radix_sort_pairs<PairSort9<256, 12>>(temp, bytes, kin, kout, vin, vout, n, 0, bits, stream);
      default: return rocprim::radix_sort_pairs<PairSort9<1024, 16>>(temp, bytes, kin, kout, vin, vout, n, 0, bits, stream);
    }
  }
  return rocprim::radix_sort_pairs(temp, bytes, kin, kout, vin, vout, n, 0, bits, stream);
}

static int col_bits(uint32_t p) {
  int bits = 1;
  while (bits < 32 && (1ull << bits) < (uint64_t)p) ++bits;
  return bits;
}

struct SortScratch {
  uint32_t* keys_out = nullptr;
  uint64_t *vals_in = nullptr, *vals_out = nullptr;
  void* temp = nullptr;
  size_t temp_bytes = 0;
  void* scan_temp = nullptr;  // running-maximum pass over the p + 1 offsets
  size_t scan_bytes = 0;
  ~SortScratch() {
    (void)hipFree(keys_out); (void)hipFree(vals_in); (void)hipFree(vals_out); (void)hipFree(temp); (void)hipFree(scan_temp);
  }
};

static int sort_scratch_alloc(SortScratch& s, int64_t max_cnt, int bits, hipStream_t stream) {
  const size_t m = (size_t)(max_cnt > 0 ? max_cnt : 1);
  FMX_HIP(hipMalloc(&s.keys_out, m * sizeof(uint32_t)));
  FMX_HIP(hipMalloc(&s.vals_in, m * sizeof(uint64_t)));
  FMX_HIP(hipMalloc(&s.vals_out, m * sizeof(uint64_t)));
  FMX_HIP(rocprim::radix_sort_pairs(nullptr, s.temp_bytes, (const uint32_t*)nullptr, s.keys_out, s.vals_in, s.vals_out, m, 0, bits, stream));
  FMX_HIP(hipMalloc(&s.temp, s.temp_bytes ? s.temp_bytes : 16));
  return FMX_OK;
}

// sort entries [base, base+cnt) of rows [r0, r0+nrows) by column (stable => rows stay ascending per column), then the
// per-feature offsets of the whole range (the full CSC of the ALS sweep)
template <typename OffT>
static int csc_of_range(const fmx_matrix* m, SortScratch& s, int bits, int64_t r0, int64_t nrows, int64_t base, int64_t cnt,
                        uint32_t* out_rows, float* out_vals, OffT* out_ptr, hipStream_t stream) {
  const int T = 256;
  if (cnt > 0) {
    hipLaunchKernelGGL(pack_entries_k, dim3((unsigned)((cnt + T - 1) / T)), dim3(T), 0, stream, m->row_ptr, m->val, r0, nrows, base, cnt, s.vals_in, 0);
    FMX_HIP(rocprim::radix_sort_pairs(s.temp, s.temp_bytes, m->col + base, s.keys_out, s.vals_in, s.vals_out, (size_t)cnt, 0, bits, stream));
    hipLaunchKernelGGL(unpack_entries_k, dim3((unsigned)((cnt + T - 1) / T)), dim3(T), 0, stream, s.vals_out, cnt, out_rows, out_vals);
  }
  const size_t np1 = (size_t)m->p + 1;
  FMX_HIP(hipMemsetAsync(out_ptr, 0, np1 * sizeof(OffT), stream));
  if (cnt > 0) hipLaunchKernelGGL((run_ends_k<OffT>), dim3((unsigned)((cnt + T - 1) / T)), dim3(T), 0, stream, s.keys_out, cnt, out_ptr);
  size_t need = 0;
  FMX_HIP(rocprim::inclusive_scan(nullptr, need, out_ptr, out_ptr, np1, rocprim::maximum<OffT>(), stream));
  if (need < 16) need = 16;  // a null scratch pointer would mean "size query"
  if (need > s.scan_bytes) {
    FMX_HIP(hipStreamSynchronize(stream));
    (void)hipFree(s.scan_temp); s.scan_temp = nullptr; s.scan_bytes = 0;
    FMX_HIP(hipMalloc(&s.scan_temp, need));
    s.scan_bytes = need;
  }
  need = s.scan_bytes;
  FMX_HIP(rocprim::inclusive_scan(s.scan_temp, need, out_ptr, out_ptr, np1, rocprim::maximum<OffT>(), stream));
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// ------------------------------------------------------------------------------------------------ tile plans
// Everything below runs on the device without a host round trip, so that a streamed tile can be planned on one stream
// while the previous tile trains on another (fmx_train_stream); the three counts are read back once per tile.

// head[i] = entry i starts a new feature's run in the sorted columns
__global__ void head_flags_k(const uint32_t* __restrict__ keys, int64_t cnt, uint8_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cnt) flags[i] = (i == 0) || (keys[i] != keys[i - 1]);
}

// ---- ordered compaction without a flag array: which indices of [0, n) satisfy a predicate, in ascending order -------------------------
// (rocprim::select over a byte-flag array cost 86 us per call at 10.2 M items -- two calls per streamed tile, a seventh of the step:
// profiles/r03_stream_trace.txt.)  Three small launches: every block counts the hits among its CP_CHUNK consecutive indices, one block
// scans the block counts, every block writes its hits behind its offset.  The predicate is evaluated twice instead of being stored.
// n may live on the device (n_dev): blocks beyond it leave at once.
constexpr int CP_THREADS = 256, CP_PER = 16, CP_CHUNK = CP_THREADS * CP_PER;
struct HeadPred {   // index i starts a run of equal keys
  const uint32_t* keys;
  __device__ bool operator()(uint32_t i) const { return i == 0 || keys[i] != keys[i - 1]; }
};
struct LongPred {   // list i holds more than long_min entries
  const uint32_t* off;
  uint32_t long_min;
  __device__ bool operator()(uint32_t i) const { return off[i + 1] - off[i] > long_min; }
};
template <typename Pred>
__global__ __launch_bounds__(CP_THREADS) void compact_count_k(Pred pred, uint32_t n_fixed, const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ blk) {
  __shared__ uint32_t red[CP_THREADS];
  const uint32_t n = n_dev ? *n_dev : n_fixed;
  const uint32_t b0 = blockIdx.x * CP_CHUNK;
  uint32_t c = 0;
  if (b0 < n) {
    const uint32_t i0 = b0 + threadIdx.x * CP_PER;
#pragma unroll
    for (int u = 0; u < CP_PER; ++u) { const uint32_t i = i0 + u; if (i < n && pred(i)) ++c; }
  }
  red[threadIdx.x] = c;
  __syncthreads();
  for (int off = CP_THREADS / 2; off > 0; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) blk[blockIdx.x] = red[0];
}
// exclusive scan of the block counts in place (one block), total -> *total
__global__ __launch_bounds__(1024) void compact_scan_k(uint32_t* __restrict__ blk, uint32_t n_blocks, uint32_t* __restrict__ total, uint32_t add = 0) {
  __shared__ uint32_t part[1024];
  const uint32_t per = (n_blocks + 1023) / 1024;
  const uint32_t b = threadIdx.x * per, e = b + per < n_blocks ? b + per : n_blocks;
  uint32_t s = 0;
  for (uint32_t i = b; i < e; ++i) s += blk[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan of the thread sums
    const uint32_t v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = (threadIdx.x ? part[threadIdx.x - 1] : 0u) + add;   // (add: hits known beforehand that take the first positions)
  for (uint32_t i = b; i < e; ++i) { const uint32_t c = blk[i]; blk[i] = run; run += c; }
  if (threadIdx.x == 1023) *total = part[1023] + add;
}
// Emit(position, index) is called for every hit, positions ascending with the index
template <typename Pred, typename Emit>
__global__ __launch_bounds__(CP_THREADS) void compact_write_k(Pred pred, Emit emit, uint32_t n_fixed, const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ blk) {
  __shared__ uint32_t sc[CP_THREADS];
  const uint32_t n = n_dev ? *n_dev : n_fixed;
  const uint32_t b0 = blockIdx.x * CP_CHUNK;
  if (b0 >= n) return;
  const uint32_t i0 = b0 + threadIdx.x * CP_PER;
  uint32_t hit = 0, c = 0;
#pragma unroll
  for (int u = 0; u < CP_PER; ++u) { const uint32_t i = i0 + u; if (i < n && pred(i)) { hit |= 1u << u; ++c; } }
  sc[threadIdx.x] = c;
  __syncthreads();
  for (int off = 1; off < CP_THREADS; off <<= 1) {
    const uint32_t v = (int)threadIdx.x >= off ? sc[threadIdx.x - off] : 0u;
    __syncthreads();
    sc[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t pos = blk[blockIdx.x] + sc[threadIdx.x] - c;
#pragma unroll
  for (int u = 0; u < CP_PER; ++u) if (hit & (1u << u)) emit(pos++, i0 + u);
}
// The run heads of the sorted columns, specialised: the chunk's keys are staged through LDS with coalesced loads (a thread reading its
// 16 consecutive keys straight from memory strides the wave over 4 KiB: 143 us per 10.2 M-entry tile against 25 us for the count).
__global__ __launch_bounds__(CP_THREADS) void heads_count_k(const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ blk) {
  __shared__ uint32_t red[CP_THREADS];
  const uint32_t b0 = blockIdx.x * CP_CHUNK;
  uint32_t c = 0;
#pragma unroll
  for (int u = 0; u < CP_PER; ++u) {   // any order will do for a count: consecutive threads read consecutive keys
    const uint32_t i = b0 + u * CP_THREADS + threadIdx.x;
    if (i < n && (i == 0 || keys[i] != keys[i - 1])) ++c;
  }
  red[threadIdx.x] = c;
  __syncthreads();
  for (int off = CP_THREADS / 2; off > 0; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) blk[blockIdx.x] = red[0];
}
struct HeadEmit;
// The writer: thread t of the block looks at entries t, t + 256, ... of the chunk (coalesced loads, the key before an entry comes from the
// neighbouring lane), a head's position is the block's offset + the heads of the chunk's earlier 256-entry slices and waves (a 64-value
// scan) + the heads in lower lanes (a ballot); the head's first row and value are read only where there is a head.  (Round 3's first
// version staged the chunk's keys, rows and values through 52 KB of LDS to give every thread 16 CONSECUTIVE entries: three workgroups per
// CU, 60 us per 6.8 M entries where the count takes 8.)
__global__ __launch_bounds__(CP_THREADS) void heads_write_k(const uint32_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ blk, uint32_t* __restrict__ soff,
                                                            uint32_t* __restrict__ feat, const uint32_t* __restrict__ brow, const float* __restrict__ bval,
                                                            uint32_t* __restrict__ row0, uint32_t* __restrict__ val0, uint32_t first = 0) {
  // (first: keys / brow / bval point at entry `first` of the tile -- the entries before it have their heads already)
  constexpr int WAVES = CP_THREADS / 64;
  static_assert(CP_PER * WAVES == 64, "the (slice, wave) counts are scanned by one wave");
  __shared__ uint32_t wc[CP_PER * WAVES];   // heads per (slice, wave), then their exclusive prefix
  const uint32_t b0 = blockIdx.x * CP_CHUNK;
  if (b0 >= n) return;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t key[CP_PER], rank[CP_PER], hit = 0;
#pragma unroll
  for (int u = 0; u < CP_PER; ++u) {
    const uint32_t i = b0 + u * CP_THREADS + threadIdx.x;
    key[u] = i < n ? keys[i] : 0u;
  }
#pragma unroll
  for (int u = 0; u < CP_PER; ++u) {
    const uint32_t i = b0 + u * CP_THREADS + threadIdx.x;
    uint32_t prev = __shfl_up(key[u], 1);
    if (lane == 0 && i > 0 && i < n) prev = keys[i - 1];
    const bool head = i < n && (i == 0 || key[u] != prev);
    const uint64_t bal = __ballot(head);
    if (head) hit |= 1u << u;
    rank[u] = (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wc[u * WAVES + wave] = (uint32_t)__builtin_popcountll(bal);
  }
  __syncthreads();
  if (threadIdx.x < 64) {   // exclusive scan of the CP_PER * WAVES = 64 counts, in (slice, wave) order
    const uint32_t v = wc[threadIdx.x];
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if ((int)lane >= off) x += y; }
    wc[threadIdx.x] = x - v;
  }
  __syncthreads();
  const uint32_t base = blk[blockIdx.x];
#pragma unroll
  for (int u = 0; u < CP_PER; ++u) {
    if (hit & (1u << u)) {
      const uint32_t i = b0 + u * CP_THREADS + threadIdx.x;
      const uint32_t pos = base + wc[u * WAVES + wave] + rank[u];
      soff[pos] = first + i;
      feat[pos] = key[u];
      if (row0) { row0[pos] = brow[i]; val0[pos] = bval ? __float_as_uint(bval[i]) : 0x3f800000u; }
    }
  }
}

struct HeadEmit {   // a run head: the list's start, its feature id and (tile plans) its first entry inline; the last hit also closes the directory
  const uint32_t* keys; uint32_t* soff; uint32_t* feat; const uint32_t* brow; const float* bval; uint32_t* row0; uint32_t* val0;
  __device__ void operator()(uint32_t pos, uint32_t i) const {
    soff[pos] = i;
    feat[pos] = keys[i];
    if (row0) { row0[pos] = brow[i]; val0[pos] = bval ? __float_as_uint(bval[i]) : 0x3f800000u; }
  }
};
struct LongEmit {
  uint32_t* lpos;
  __device__ void operator()(uint32_t pos, uint32_t i) const { lpos[pos] = i; }
};
__global__ void close_directory_k(uint32_t* __restrict__ soff, const uint32_t* __restrict__ n_lists, uint32_t cnt) { soff[*n_lists] = cnt; }

template <typename Pred, typename Emit>
static int compact_indices(Pred pred, Emit emit, uint32_t n_max, const uint32_t* n_dev, uint32_t* blk, uint32_t* total, hipStream_t stream) {
  if (n_max == 0) { FMX_HIP(hipMemsetAsync(total, 0, sizeof(uint32_t), stream)); return FMX_OK; }
  const uint32_t nb = (n_max + CP_CHUNK - 1) / CP_CHUNK;
  hipLaunchKernelGGL((compact_count_k<Pred>), dim3(nb), dim3(CP_THREADS), 0, stream, pred, n_max, n_dev, blk);
  hipLaunchKernelGGL(compact_scan_k, dim3(1), dim3(1024), 0, stream, blk, nb, total);
  hipLaunchKernelGGL((compact_write_k<Pred, Emit>), dim3(nb), dim3(CP_THREADS), 0, stream, pred, emit, n_max, n_dev, (const uint32_t*)blk);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// sparse directory: the run starts are in soff[0 .. n); add the ids and the end marker
// (a tile plan also gets each list's first entry inline: row0 / val0, from the sorted entries; the record merge passes nulls)
__global__ void lists_finish_k(const uint32_t* __restrict__ keys, uint32_t* __restrict__ soff, const uint32_t* __restrict__ dcounts,
                               uint32_t cnt, uint32_t* __restrict__ feat, const uint32_t* __restrict__ brow = nullptr,
                               const float* __restrict__ bval = nullptr, uint32_t* __restrict__ row0 = nullptr, uint32_t* __restrict__ val0 = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = dcounts[0];
  if (i < (int64_t)n) {
    const uint32_t o = soff[i];
    feat[i] = keys[o];
    if (row0) { row0[i] = brow[o]; val0[i] = bval ? __float_as_uint(bval[o]) : 0x3f800000u; }
  } else if (i == (int64_t)n) soff[i] = cnt;
}

// flags[i] = list i holds more than long_min entries (lists beyond the directory's end: 0)
__global__ void long_flags_k(const uint32_t* __restrict__ off, const uint32_t* __restrict__ dcount, uint32_t n_fixed, uint32_t n_max,
                             uint32_t long_min, uint8_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_max) return;
  const uint32_t n = dcount ? *dcount : n_fixed;
  flags[i] = (i < (int64_t)n) && (off[i + 1] - off[i] > long_min);
}

__global__ void long_nseg_k(const uint32_t* __restrict__ off, const uint32_t* __restrict__ lpos, const uint32_t* __restrict__ dcounts,
                            uint32_t cap_long, uint32_t* __restrict__ nseg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > (int64_t)cap_long) return;
  uint32_t v = 0;
  if (i < (int64_t)dcounts[1]) { const uint32_t l = lpos[i]; v = (off[l + 1] - off[l] + LIST_SEG - 1) / LIST_SEG; }
  nseg[i] = v;
}

// one thread per long list: its feature id and its segments
__global__ void long_segments_k(const uint32_t* __restrict__ off, const uint32_t* __restrict__ feat, const uint32_t* __restrict__ lpos,
                                const uint32_t* __restrict__ lseg_ptr, uint32_t* __restrict__ dcounts, uint32_t* __restrict__ lfeat,
                                uint32_t* __restrict__ seg_list, uint32_t* __restrict__ seg_begin, uint32_t* __restrict__ seg_end) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_long = dcounts[1];
  if (i == 0) dcounts[2] = lseg_ptr[n_long];
  if (i >= (int64_t)n_long) return;
  const uint32_t l = lpos[i];
  lfeat[i] = feat ? feat[l] : l;
  uint32_t sg = lseg_ptr[i];
  for (uint32_t b = off[l]; b < off[l + 1]; b += LIST_SEG, ++sg) {
    seg_list[sg] = (uint32_t)i;
    seg_begin[sg] = b;
    seg_end[sg] = b + LIST_SEG < off[l + 1] ? b + LIST_SEG : off[l + 1];
  }
}

// dense directory from a sparse one (no re-sort): off[feat[i] + 1] = soff[i + 1], then the running maximum
__global__ void scatter_ends_k(const uint32_t* __restrict__ feat, const uint32_t* __restrict__ soff, uint32_t n, uint32_t* __restrict__ off) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)n) off[(size_t)feat[i] + 1] = soff[i + 1];
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

void plan_free(fmx_matrix::TilePlan& t) {
  (void)hipFree(t.pool);
  (void)hipFree(t.own_pool);
  if (t.off && !t.off_in_pool) (void)hipFree(t.off);
  t = fmx_matrix::TilePlan{};
}

// one allocation holding every array of a plan for a tile of at most cap_cnt entries
int plan_alloc(fmx_matrix::TilePlan& t, uint32_t p, int64_t cap_cnt, bool dense) {
  const uint32_t lm = list_long_min();
  t.cap_lists = dense ? 0u : (uint32_t)((int64_t)p < cap_cnt ? (int64_t)p : cap_cnt);
  t.cap_long = (uint32_t)(cap_cnt / ((int64_t)lm + 1) + 1);
  t.cap_seg = (uint32_t)(cap_cnt / LIST_SEG + t.cap_long + 1);
  size_t at = 0;
  auto take = [&](size_t count) { const size_t o = at; at = align_up(at + count * sizeof(uint32_t), 256); return o; };
  const size_t o_cnt = take(4);
  const size_t o_off = dense ? take((size_t)p + 1) : 0;
  const size_t o_feat = dense ? 0 : take(t.cap_lists ? t.cap_lists : 1);
  const size_t o_soff = dense ? 0 : take((size_t)t.cap_lists + 1);
  const size_t o_row0 = dense ? 0 : take(t.cap_lists ? t.cap_lists : 1), o_val0 = dense ? 0 : take(t.cap_lists ? t.cap_lists : 1);
  const size_t o_lfeat = take(t.cap_long), o_lpos = take(t.cap_long), o_lseg = take((size_t)t.cap_long + 2);
  const size_t o_sl = take(t.cap_seg), o_sb = take(t.cap_seg), o_se = take(t.cap_seg);
  FMX_HIP(hipMalloc(&t.pool, at));
  char* b = (char*)t.pool;
  t.dcounts = (uint32_t*)(b + o_cnt);
  t.off = dense ? (uint32_t*)(b + o_off) : nullptr;
  t.off_in_pool = dense ? 1 : 0;
  t.feat = dense ? nullptr : (uint32_t*)(b + o_feat);
  t.soff = dense ? nullptr : (uint32_t*)(b + o_soff);
  t.row0 = dense ? nullptr : (uint32_t*)(b + o_row0);
  t.val0 = dense ? nullptr : (uint32_t*)(b + o_val0);
  t.lfeat = (uint32_t*)(b + o_lfeat); t.lpos = (uint32_t*)(b + o_lpos); t.lseg_ptr = (uint32_t*)(b + o_lseg);
  t.seg_list = (uint32_t*)(b + o_sl); t.seg_begin = (uint32_t*)(b + o_sb); t.seg_end = (uint32_t*)(b + o_se);
  return FMX_OK;
}

// per-field sort (further down): the workspace fits blocks of at least this many entries and digit histograms of at most this size
constexpr int FQ_TILE_MIN = 2048, FQ_NBMAX = 512;
PlanWorkspace::~PlanWorkspace() {
  (void)hipFree(keys_out); (void)hipFree(vals_in); (void)hipFree(vals_out); (void)hipFree(sort_temp); (void)hipFree(flags);
  (void)hipFree(nseg); (void)hipFree(prim_temp); (void)hipFree(blk); (void)hipFree(fq_counts); (void)hipFree(fq_totals);
}

int PlanWorkspace::reserve(int64_t cnt, uint32_t p_, hipStream_t stream) {
  if (cnt <= max_cnt && p_ == p && keys_out) return FMX_OK;
  FMX_HIP(hipStreamSynchronize(stream));
  (void)hipFree(keys_out); (void)hipFree(vals_in); (void)hipFree(vals_out); (void)hipFree(sort_temp); (void)hipFree(flags);
  (void)hipFree(nseg); (void)hipFree(prim_temp); (void)hipFree(blk); (void)hipFree(fq_counts); (void)hipFree(fq_totals);
  keys_out = nullptr; vals_in = vals_out = nullptr; sort_temp = nullptr; flags = nullptr; nseg = nullptr; prim_temp = nullptr; blk = nullptr;
  fq_counts = fq_totals = nullptr;
  max_cnt = 0;
  const size_t m = (size_t)(cnt > 0 ? cnt : 1);
  bits = col_bits(p_);
  FMX_HIP(hipMalloc(&keys_out, m * sizeof(uint32_t)));
  FMX_HIP(hipMalloc(&vals_in, m * sizeof(uint64_t)));
  FMX_HIP(hipMalloc(&vals_out, m * sizeof(uint64_t)));
  FMX_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint32_t*)nullptr, keys_out, vals_in, vals_out, m, 0, bits, stream));
  {
    size_t b32 = 0;  // the (u32, u32) sort of one-hot matrices
    FMX_HIP(sort_pairs_u32(nullptr, b32, (const uint32_t*)nullptr, keys_out, (const uint32_t*)nullptr, (uint32_t*)nullptr, m, bits, stream));
    if (b32 > sort_bytes) sort_bytes = b32;
  }
  FMX_HIP(hipMalloc(&sort_temp, sort_bytes ? sort_bytes : 16));
  const size_t nflag = m > (size_t)p_ ? m : (size_t)p_;
  FMX_HIP(hipMalloc(&flags, nflag));
  const size_t cap_long = m / ((size_t)list_long_min() + 1) + 1;
  FMX_HIP(hipMalloc(&nseg, (cap_long + 2) * sizeof(uint32_t)));
  FMX_HIP(hipMalloc(&blk, (nflag / CP_CHUNK + 2) * sizeof(uint32_t)));
  // scratch of the device primitives: the largest of select over nflag items, max-scan over p + 1, sum-scan over cap_long + 1
  size_t b1 = 0, b2 = 0, b3 = 0;
  rocprim::counting_iterator<uint32_t> ids(0);
  FMX_HIP(rocprim::select(nullptr, b1, ids, flags, (uint32_t*)nullptr, (uint32_t*)nullptr, nflag, stream));
  FMX_HIP(rocprim::inclusive_scan(nullptr, b2, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)p_ + 1, rocprim::maximum<uint32_t>(), stream));
  FMX_HIP(rocprim::exclusive_scan(nullptr, b3, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, cap_long + 1, rocprim::plus<uint32_t>(), stream));
  prim_bytes = b1 > b2 ? b1 : b2;
  if (b3 > prim_bytes) prim_bytes = b3;
  if (prim_bytes < 16) prim_bytes = 16;
  FMX_HIP(hipMalloc(&prim_temp, prim_bytes));
  // per-field sort: a digit histogram per 8192-entry block of every field (at most cnt / 8192 + one ragged block per field) and its
  // running sums over the field's blocks (second half), digit totals per field
  FMX_HIP(hipMalloc(&fq_counts, 2 * (m / FQ_TILE_MIN + FMX_MAX_FIELDS) * (size_t)FQ_NBMAX * sizeof(uint32_t)));
  FMX_HIP(hipMalloc(&fq_totals, (size_t)FMX_MAX_FIELDS * FQ_NBMAX * sizeof(uint32_t)));
  max_cnt = (int64_t)m;
  p = p_;
  return FMX_OK;
}

// ---- field-structured rows (fmx_matrix::dense_prefix): split instead of sort ------------------------------------------------------
// Every row is [dense columns 0..d-1 with real values | one one-hot entry per categorical field].  The sorted order of the tile then
// starts with the d dense columns' lists, each simply the rows in order -- written directly, no sort -- and only the one-hot part
// (z - d of z entries per row, values all 1) goes through the radix sort, as (column, row) u32 pairs: a third fewer entries at
// 8 instead of 12 bytes per entry and pass.  One workgroup transposes FS_ROWS rows through LDS so that every output stream is
// written in contiguous runs.
constexpr int FS_ROWS = 64;
__global__ __launch_bounds__(256) void fields_split_k(const uint32_t* __restrict__ col, const float* __restrict__ val, int64_t nrows, int z, int d,
                                                      uint32_t* __restrict__ keys_sorted, uint32_t* __restrict__ brow, float* __restrict__ bval,
                                                      uint32_t* __restrict__ keys_in, uint32_t* __restrict__ rows_in) {
  __shared__ uint32_t s_col[FS_ROWS * 64];
  __shared__ float s_val[FS_ROWS * 64];
  const int64_t R0 = (int64_t)blockIdx.x * FS_ROWS;
  const int rows = (int)(nrows - R0 < FS_ROWS ? nrows - R0 : FS_ROWS);
  const int cnt = rows * z;
  for (int i = threadIdx.x; i < cnt; i += 256) { s_col[i] = col[R0 * z + i]; s_val[i] = val[R0 * z + i]; }
  __syncthreads();
  for (int i = threadIdx.x; i < d * rows; i += 256) {   // dense column c: entries [c * nrows, (c + 1) * nrows) of the sorted order
    const int c = i / rows, r = i - c * rows;
    const int64_t at = (int64_t)c * nrows + R0 + r;
    keys_sorted[at] = s_col[r * z + c];
    brow[at] = (uint32_t)(R0 + r);
    bval[at] = s_val[r * z + c];
  }
  const int zc = z - d;
  for (int i = threadIdx.x; i < zc * rows; i += 256) {  // the one-hot part, row-major, compacted
    const int r = i / zc, c = i - r * zc;
    const int64_t at = R0 * zc + i;
    keys_in[at] = s_col[r * z + d + c];
    rows_in[at] = (uint32_t)(R0 + r);
    bval[(int64_t)d * nrows + at] = 1.0f;   // the one-hot part's values, wherever the sort puts its entries
  }
}

// ---- per-field sort (field-structured rows whose field ranges are known: fmx_matrix::field_base) -----------------------------------------
// Entry d + c of every row is the one value of categorical field c, an id in [base[c], base[c + 1]).  The tile's sorted order is then the
// fields one after another, each field its n = nrows entries stably sorted by the id INSIDE the field -- 26 independent sorts of (local id,
// row) pairs whose keys have ceil(log2 vocab_c) bits: 24 for a field of ten million values, 2 for a field of three.  One global radix sort
// of the 25-bit column ids drags every entry through every pass (78 field-passes at the Criteo shape); sorted field by field the same tile
// needs 52 (digits of at most 8 bits), no lookback chains, no buffer fills between passes.  An LSD pass is three launches over the fields
// still active: count (digit histogram of every 4096-entry block), scan (a digit's blocks in order), scatter.  The scatter ranks its entries
// with wave ballots and per-wave histograms, orders the block by digit in LDS and writes each digit's run contiguously (16 entries on
// average when the digit is uniform: a first version that stored every entry straight from its lane, with 11-bit digits, spent its time
// in 4-byte stores to 64 different lines per instruction -- 51 us per pass against 13 for the count that reads the same keys).  Pass 0
// takes the row from the entry's position; a field's last pass adds base[c] back and writes straight into the plan.  Blocks of one field
// run on ONE XCD (consecutive workgroup ids go round the eight XCDs) so that the next pass could re-read what this one wrote from that XCD's L2 --
// measured against plain field-major order (FMX_FQ_PLAIN=1): no difference at the Criteo shape (383 / 386 M examples/s streamed); matrices of fewer than
// eight fields (user id / item id pairs) take the plain order, which keeps the whole chip busy.
struct FieldPass {  // one LSD pass (by value)
  int n_active;                    // fields taking part
  int plain;                       // blocks in field-major order over the whole chip instead of one field per XCD at a time
  uint32_t n, tiles;               // entries per field, blocks per field
  uint8_t field[FMX_MAX_FIELDS];   // active slot -> field
  uint8_t shift[FMX_MAX_FIELDS], db[FMX_MAX_FIELDS], last[FMX_MAX_FIELDS];  // per slot: digit position and width, final pass of the field
  uint32_t base[FMX_MAX_FIELDS];   // per slot: first column id of the field
};
// geometry: THREADS per block, PER entries per thread (TILE = THREADS * PER entries per block), digits of at most DBMAX bits
template <int THREADS_, int PER_, int DBMAX_>
struct FqCfg {
  static constexpr int THREADS = THREADS_, PER = PER_, DBMAX = DBMAX_, TILE = THREADS_ * PER_, NBMAX = 1 << DBMAX_, WAVES = THREADS_ / 64;
  static constexpr int DPT = NBMAX > THREADS ? NBMAX / THREADS : 1;   // digits a thread looks after
  static_assert(TILE >= FQ_TILE_MIN && NBMAX <= FQ_NBMAX, "the workspace is sized for blocks of at least FQ_TILE_MIN entries and FQ_NBMAX digits");
};
__device__ __forceinline__ bool fq_block(const FieldPass& P, int* slot, uint32_t* tile) {
  if (P.plain) {   // fewer fields than XCDs (user id / item id pairs): field after field over the whole chip
    *slot = (int)(blockIdx.x / P.tiles);
    *tile = blockIdx.x % P.tiles;
    return *slot < P.n_active;
  }
  const uint32_t x = blockIdx.x & 7u, s = blockIdx.x >> 3;   // XCD, position in the XCD's queue
  *slot = (int)(x + 8u * (s / P.tiles));
  *tile = s % P.tiles;
  return *slot < P.n_active;
}
// lanes of the wave holding the same digit (valid lanes only)
__device__ __forceinline__ uint64_t fq_match(uint32_t d, int db, bool valid) {
  uint64_t m = __ballot(valid);
  for (int b = 0; b < db; ++b) {
    const bool bit = (d >> b) & 1u;
    const uint64_t bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return m;
}
// exclusive scan over the block's threads; ends with a barrier
template <int WAVES>
__device__ __forceinline__ uint32_t fq_block_scan(uint32_t v, uint32_t* wsum) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off); if ((int)lane >= off) x += y; }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) before += (uint32_t)w < wave ? wsum[w] : 0u;
  __syncthreads();
  return before + x - v;
}
template <typename G>
__global__ __launch_bounds__(G::THREADS) void fq_count_k(FieldPass P, const uint32_t* __restrict__ src_keys, uint32_t* __restrict__ counts) {
  __shared__ uint32_t hist[G::NBMAX];
  int slot; uint32_t tile;
  if (!fq_block(P, &slot, &tile)) return;
  const int db = P.db[slot], shift = P.shift[slot];
  const uint32_t NB = 1u << db;
  for (uint32_t d = threadIdx.x; d < NB; d += G::THREADS) hist[d] = 0;
  __syncthreads();
  const uint32_t* __restrict__ k = src_keys + (size_t)P.field[slot] * P.n;
  const uint32_t i0 = tile * G::TILE + threadIdx.x;
  uint32_t key[G::PER];
#pragma unroll
  for (int c = 0; c < G::PER; ++c) { const uint32_t i = i0 + c * G::THREADS; key[c] = i < P.n ? k[i] : 0u; }
#pragma unroll
  for (int c = 0; c < G::PER; ++c)   // (a histogram: any order; equal digits in one instruction serialise in the LDS, a few hundred cycles at worst)
    if (i0 + c * G::THREADS < P.n) atomicAdd(&hist[(key[c] >> shift) & (NB - 1)], 1u);
  __syncthreads();
  uint32_t* __restrict__ out = counts + ((size_t)slot * P.tiles + tile) * G::NBMAX;
  for (uint32_t d = threadIdx.x; d < NB; d += G::THREADS) out[d] = hist[d];
}
// pre[slot][tile][d] = entries of digit d in the field's earlier blocks; totals[slot][d] = the digit's entries in the whole field
template <typename G>
__global__ __launch_bounds__(256) void fq_scan_k(FieldPass P, const uint32_t* __restrict__ counts, uint32_t* __restrict__ pre, uint32_t* __restrict__ totals) {
  const int slot = blockIdx.y;
  const uint32_t d = blockIdx.x * 256 + threadIdx.x;
  if (d >= (1u << P.db[slot])) return;
  const size_t at = (size_t)slot * P.tiles * G::NBMAX + d;
  uint32_t run = 0;
  for (uint32_t t0 = 0; t0 < P.tiles; t0 += 8) {   // eight independent loads at a time (the output is another array: nothing to wait for)
    uint32_t v[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) v[u] = t0 + u < P.tiles ? counts[at + (size_t)(t0 + u) * G::NBMAX] : 0u;
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) { if (t0 + u < P.tiles) pre[at + (size_t)(t0 + u) * G::NBMAX] = run; run += v[u]; }
  }
  totals[(size_t)slot * G::NBMAX + d] = run;
}
template <typename G>
__global__ __launch_bounds__(G::THREADS) void fq_scatter_k(FieldPass P, const uint32_t* __restrict__ src_keys, const uint32_t* __restrict__ src_rows,
                                                           uint32_t* __restrict__ dst_keys, uint32_t* __restrict__ dst_rows, uint32_t* __restrict__ fin_keys,
                                                           uint32_t* __restrict__ fin_rows, const uint32_t* __restrict__ pre, const uint32_t* __restrict__ totals) {
  constexpr int WAVES = G::WAVES, NBMAX = G::NBMAX, PER = G::PER, DPT = G::DPT, THREADS = G::THREADS;
  __shared__ uint32_t whist[WAVES][NBMAX];   // entries of a digit in each wave's part of the block; then where that part starts in the ordered block
  __shared__ uint32_t delta[NBMAX];          // ordered-block position -> position in the field's output, per digit (mod 2^32)
  __shared__ uint32_t stage[G::TILE];        // the block ordered by digit
  __shared__ uint32_t wsum[WAVES];
  int slot; uint32_t tile;
  if (!fq_block(P, &slot, &tile)) return;
  const int db = P.db[slot], shift = P.shift[slot];
  const uint32_t NB = 1u << db;
  const uint32_t d0 = threadIdx.x * DPT;   // the DPT digits this thread looks after: d0 .. d0 + DPT - 1
  uint32_t tot[DPT], pr[DPT];
#pragma unroll
  for (int u = 0; u < DPT; ++u) {   // (asked for before anything else: both are needed only after the ranks)
    const bool in = d0 + u < NB;
    tot[u] = in ? totals[(size_t)slot * NBMAX + d0 + u] : 0u;
    pr[u] = in ? pre[((size_t)slot * P.tiles + tile) * NBMAX + d0 + u] : 0u;
    if (in) {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) whist[w][d0 + u] = 0;
    }
  }
  const size_t f0 = (size_t)P.field[slot] * P.n;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t i0 = tile * G::TILE + wave * (G::TILE / WAVES) + lane;
  uint32_t key[PER], row[PER], lp[PER];
#pragma unroll
  for (int c = 0; c < PER; ++c) {
    const uint32_t i = i0 + c * 64;
    const bool valid = i < P.n;
    key[c] = valid ? src_keys[f0 + i] : 0u;
    row[c] = src_rows ? (valid ? src_rows[f0 + i] : 0u) : i;   // pass 0: the entry's position IS its row
  }
  uint32_t tsum = 0;
#pragma unroll
  for (int u = 0; u < DPT; ++u) tsum += tot[u];
  uint32_t dbase = fq_block_scan<WAVES>(tsum, wsum);   // where digit d0 starts in the field's output
  // ranks inside the wave's part, chunk after chunk: the lanes of a digit read its counter, the first of them moves it on
  volatile uint32_t* wh = whist[wave];
#pragma unroll
  for (int c = 0; c < PER; ++c) {
    const bool valid = i0 + c * 64 < P.n;
    const uint32_t d = (key[c] >> shift) & (NB - 1);
    const uint64_t m = fq_match(d, db, valid);
    const uint32_t old = wh[valid ? d : 0u];
    __builtin_amdgcn_wave_barrier();
    if (valid && lane == (uint32_t)__builtin_ctzll(m)) wh[d] = old + (uint32_t)__builtin_popcountll(m);
    __builtin_amdgcn_wave_barrier();
    lp[c] = old + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  uint32_t cw[DPT][WAVES], bc[DPT], bsum = 0;
#pragma unroll
  for (int u = 0; u < DPT; ++u) {
    bc[u] = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { cw[u][w] = d0 + u < NB ? whist[w][d0 + u] : 0u; bc[u] += cw[u][w]; }
    bsum += bc[u];
  }
  uint32_t lbase = fq_block_scan<WAVES>(bsum, wsum);   // where digit d0 starts in the ordered block
#pragma unroll
  for (int u = 0; u < DPT; ++u) {
    if (d0 + u < NB) {
      uint32_t g = lbase;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) { whist[w][d0 + u] = g; g += cw[u][w]; }
      delta[d0 + u] = dbase + pr[u] - lbase;
    }
    lbase += bc[u]; dbase += tot[u];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < PER; ++c) {
    if (i0 + c * 64 < P.n) {
      lp[c] += whist[wave][(key[c] >> shift) & (NB - 1)];
      stage[lp[c]] = key[c];
    }
  }
  __syncthreads();
  const bool last = P.last[slot] != 0;
  uint32_t* __restrict__ ok = (last ? fin_keys : dst_keys) + f0;
  uint32_t* __restrict__ orow = (last ? fin_rows : dst_rows) + f0;
  const uint32_t add = last ? P.base[slot] : 0u;
  const uint32_t have = P.n - tile * G::TILE < (uint32_t)G::TILE ? P.n - tile * G::TILE : (uint32_t)G::TILE;
  uint32_t dest[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const uint32_t l = threadIdx.x + j * THREADS;
    if (l < have) {
      const uint32_t kk = stage[l];
      dest[j] = l + delta[(kk >> shift) & (NB - 1)];
      ok[dest[j]] = kk + add;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < PER; ++c)
    if (i0 + c * 64 < P.n) stage[lp[c]] = row[c];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const uint32_t l = threadIdx.x + j * THREADS;
    if (l < have) orow[dest[j]] = stage[l];
  }
}
// split for the per-field sort: the dense columns as in fields_split_k, the one-hot part FIELD-major with the field's base taken off
struct FieldBases { uint32_t base[FMX_MAX_FIELDS]; };
__global__ __launch_bounds__(256) void fields_split_local_k(const uint32_t* __restrict__ col, const float* __restrict__ val, int64_t nrows, int z, int d, FieldBases fb,
                                                            uint32_t* __restrict__ keys_sorted, uint32_t* __restrict__ brow, float* __restrict__ bval,
                                                            uint32_t* __restrict__ keys_in, int unit) {
  // (unit: a one-hot matrix -- d = 0, nobody reads its value arrays)
  __shared__ uint32_t s_col[FS_ROWS * 64];
  __shared__ float s_val[FS_ROWS * 64];
  const int64_t R0 = (int64_t)blockIdx.x * FS_ROWS;
  const int rows = (int)(nrows - R0 < FS_ROWS ? nrows - R0 : FS_ROWS);
  const int cnt = rows * z;
  for (int i = threadIdx.x; i < cnt; i += 256) { s_col[i] = col[R0 * z + i]; if (!unit) s_val[i] = val[R0 * z + i]; }
  __syncthreads();
  for (int i = threadIdx.x; i < d * rows; i += 256) {
    const int c = i / rows, r = i - c * rows;
    const int64_t at = (int64_t)c * nrows + R0 + r;
    keys_sorted[at] = s_col[r * z + c];
    brow[at] = (uint32_t)(R0 + r);
    bval[at] = s_val[r * z + c];
  }
  const int zc = z - d;
  for (int i = threadIdx.x; i < zc * rows; i += 256) {
    const int c = i / rows, r = i - c * rows;
    const int64_t at = (int64_t)c * nrows + R0 + r;
    keys_in[at] = s_col[r * z + d + c] - fb.base[c];
    if (!unit) bval[(int64_t)d * nrows + at] = 1.0f;
  }
}
// the dense columns' list heads (column c: entries [c * nrows, (c + 1) * nrows), first row 0)
__global__ void dense_heads_k(int d, uint32_t nrows, const float* __restrict__ bval, uint32_t* __restrict__ soff, uint32_t* __restrict__ feat,
                              uint32_t* __restrict__ row0, uint32_t* __restrict__ val0) {
  const int c = threadIdx.x;
  if (c < d) { soff[c] = (uint32_t)c * nrows; feat[c] = (uint32_t)c; row0[c] = 0u; val0[c] = __float_as_uint(bval[(size_t)c * nrows]); }
}
// A, B: ping-pong buffers of n_cat keys + n_cat rows each; fin_*: the plan's arrays at the one-hot part
// keys0 / rows0 (optional): where the FIRST pass reads its keys and rows (a read-only source -- the matrix's own column array -- and rows that are not the
// entries' positions: the one-"field" sort of a tile without a field layout); the ping-pong buffers are then only written from pass 0 on
template <typename G>
static int field_sort_g(const std::vector<uint32_t>& fbase, uint32_t n, uint32_t* a_keys, uint32_t* a_rows, uint32_t* b_keys, uint32_t* b_rows, uint32_t* fin_keys,
                        uint32_t* fin_rows, uint32_t* counts, uint32_t* totals, hipStream_t stream, const uint32_t* keys0 = nullptr, const uint32_t* rows0 = nullptr) {
  const int C = (int)fbase.size() - 1;
  int passes[FMX_MAX_FIELDS], dbq[FMX_MAX_FIELDS], bits[FMX_MAX_FIELDS], max_passes = 0;
  for (int c = 0; c < C; ++c) {
    bits[c] = col_bits(fbase[(size_t)c + 1] - fbase[(size_t)c]);
    passes[c] = (bits[c] + G::DBMAX - 1) / G::DBMAX;
    dbq[c] = (bits[c] + passes[c] - 1) / passes[c];   // balanced digits: 17 bits = 9 + 8, not 9 + 8 + 0 or 11 + 6
    if (passes[c] > max_passes) max_passes = passes[c];
  }
  const uint32_t tiles = (n + G::TILE - 1) / G::TILE;
  uint32_t* pre = counts + ((size_t)C * tiles) * G::NBMAX;   // (the workspace holds twice the histograms' size)
  for (int q = 0; q < max_passes; ++q) {
    FieldPass P{};
    P.n = n; P.tiles = tiles;
    int db_max = 0;
    for (int c = 0; c < C; ++c) {
      if (passes[c] <= q) continue;
      const int s = P.n_active++;
      P.field[s] = (uint8_t)c; P.shift[s] = (uint8_t)(q * dbq[c]);
      P.db[s] = (uint8_t)((bits[c] - q * dbq[c]) < dbq[c] ? (bits[c] - q * dbq[c]) : dbq[c]);
      P.last[s] = (uint8_t)(q + 1 == passes[c]);
      P.base[s] = fbase[(size_t)c];
      if (P.db[s] > db_max) db_max = P.db[s];
    }
    const uint32_t* sk = (q == 0 && keys0) ? keys0 : ((q & 1) ? b_keys : a_keys);
    const uint32_t* sr = q == 0 ? rows0 : ((q & 1) ? b_rows : a_rows);
    uint32_t* dk = (q & 1) ? a_keys : b_keys;
    uint32_t* dr = (q & 1) ? a_rows : b_rows;
    static const int plain_env = [] { const char* v = getenv("FMX_FQ_PLAIN"); return v ? atoi(v) : -1; }();
    P.plain = plain_env >= 0 ? plain_env : (P.n_active < 8 ? 1 : 0);
    if (P.n_active < 8) P.plain = 1;
    const dim3 grid(P.plain ? tiles * (unsigned)P.n_active : 8u * tiles * (unsigned)((P.n_active + 7) / 8));
    hipLaunchKernelGGL((fq_count_k<G>), grid, dim3(G::THREADS), 0, stream, P, sk, counts);
    hipLaunchKernelGGL((fq_scan_k<G>), dim3((unsigned)(((1u << db_max) + 255) / 256), (unsigned)P.n_active), dim3(256), 0, stream, P, (const uint32_t*)counts, pre, totals);
    hipLaunchKernelGGL((fq_scatter_k<G>), grid, dim3(G::THREADS), 0, stream, P, sk, sr, dk, dr, fin_keys, fin_rows, (const uint32_t*)pre, (const uint32_t*)totals);
  }
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}
static int field_sort(const std::vector<uint32_t>& fbase, uint32_t n, uint32_t* a_keys, uint32_t* a_rows, uint32_t* b_keys, uint32_t* b_rows, uint32_t* fin_keys,
                      uint32_t* fin_rows, uint32_t* counts, uint32_t* totals, hipStream_t stream, const uint32_t* keys0 = nullptr, const uint32_t* rows0 = nullptr) {
  // block geometry (profiles/r03_field_sort.txt: streamed Criteo shape, M examples/s): 256 x 16 entries with 8-bit digits 367, 256 x 8 / 8 bits 368,
  // 128 x 16 / 8 bits 365, 256 x 16 / 7 bits 362, 256 x 16 / 9 bits 355, 128 x 32 / 8 bits 348, 512 x 16 / 9 bits 319, 1024 x 8 / 9 bits 314,
  // 256 x 32 / 9 bits 303 -- small blocks win: a block is a chain of barriers, and the chip wants many of them in flight
  static const int cfg = [] { const char* v = getenv("FMX_FQ_CFG"); return v ? atoi(v) : 1; }();
  switch (cfg) {
    case 0: return field_sort_g<FqCfg<512, 16, 9>>(fbase, n, a_keys, a_rows, b_keys, b_rows, fin_keys, fin_rows, counts, totals, stream, keys0, rows0);
    case 6: return field_sort_g<FqCfg<256, 8, 8>>(fbase, n, a_keys, a_rows, b_keys, b_rows, fin_keys, fin_rows, counts, totals, stream, keys0, rows0);
    case 7: return field_sort_g<FqCfg<128, 16, 8>>(fbase, n, a_keys, a_rows, b_keys, b_rows, fin_keys, fin_rows, counts, totals, stream, keys0, rows0);
    default: return field_sort_g<FqCfg<256, 16, 8>>(fbase, n, a_keys, a_rows, b_keys, b_rows, fin_keys, fin_rows, counts, totals, stream, keys0, rows0);
  }
}

__global__ void fill_ones_k(float* __restrict__ x, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = 1.0f;
}

// Plan one tile: entries [t.base, t.base + t.cnt) of rows [t.r0, t.r0 + t.nrows), CSR arrays given explicitly (a streamed
// tile has its own).  Enqueues on `stream`, never waits for it; t must come from plan_alloc with room for t.cnt entries.
// Does a tile of field-structured rows with a dense prefix take the split + per-field sort path of plan_build?  (The streamed generator then writes the split's
// outputs itself -- generate_fields_split_async -- and plan_build is told so: `presplit`.)
bool plan_fields_split_applies(const PlanWorkspace& ws, int unit_values, int fixed_row_len, int dense_prefix, const std::vector<uint32_t>* field_base) {
  const char* split_env = getenv("FMX_FIELDS_SPLIT");
  const char* fq_env = getenv("FMX_FIELD_SORT");
  return !(split_env && split_env[0] == '0') && !(fq_env && fq_env[0] == '0') && !unit_values && dense_prefix > 0 && fixed_row_len > dense_prefix && fixed_row_len <= 64 &&
         field_base && (int)field_base->size() == fixed_row_len - dense_prefix + 1 && ws.fq_counts != nullptr;
}

int plan_build(fmx_matrix::TilePlan& t, PlanWorkspace& ws, uint32_t p, const int64_t* row_ptr, const uint32_t* col, const float* val,
               uint32_t* brow, float* bval, hipStream_t stream, int unit_values, int fixed_row_len, int dense_prefix, const std::vector<uint32_t>* field_base, bool presplit) {
  const int T = 256;
  const int64_t cnt = t.cnt;
  FMX_CHECK(cnt <= ws.max_cnt && p == ws.p, FMX_ERR_STATE, "plan workspace too small");
  FMX_CHECK(cnt < (1LL << 32), FMX_ERR_INVALID, "a tile holds %lld nonzeros; at most 2^32-1 are supported (lower tile_rows)", (long long)cnt);
  auto grid = [&](int64_t n) { return dim3((unsigned)((n + T - 1) / T)); };
  bool field_sorted = false;
  const char* split_env = getenv("FMX_FIELDS_SPLIT");  // read per build: the tests compare both forms
  const bool split_ok = !(split_env && split_env[0] == '0');
  if (cnt > 0 && split_ok && !unit_values && dense_prefix > 0 && fixed_row_len > dense_prefix && fixed_row_len <= 64 &&
      cnt == t.nrows * (int64_t)fixed_row_len) {
    const int z = fixed_row_len, d = dense_prefix;
    const int64_t n_dense = t.nrows * d, n_cat = cnt - n_dense;
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(ws.vals_in);          // the u64 payload buffer holds both u32 inputs of the pair sort
    uint32_t* rows_in = keys_in + n_cat;
    const char* fq_env = getenv("FMX_FIELD_SORT");  // read per build: the tests compare the forms
    field_sorted = field_base && (int)field_base->size() == z - d + 1 && !(fq_env && fq_env[0] == '0') && ws.fq_counts != nullptr;
    if (field_sorted) {   // the fields' id ranges are known: 26 short sorts instead of one long one (field_sort above)
      FieldBases fb{};
      for (int c = 0; c < z - d; ++c) fb.base[c] = (*field_base)[(size_t)c];
      if (!presplit)   // (a streamed tile's generator has written these arrays already)
        hipLaunchKernelGGL(fields_split_local_k, dim3((unsigned)((t.nrows + FS_ROWS - 1) / FS_ROWS)), dim3(256), 0, stream, col + t.base, val + t.base, t.nrows, z, d, fb,
                           ws.keys_out, brow + t.base, bval + t.base, keys_in, 0);
      uint32_t* b_keys = reinterpret_cast<uint32_t*>(ws.vals_out);
      FMX_TRY(field_sort(*field_base, (uint32_t)t.nrows, keys_in, rows_in, b_keys, b_keys + n_cat, ws.keys_out + n_dense, brow + t.base + n_dense, ws.fq_counts,
                         ws.fq_totals, stream));
    } else {
      hipLaunchKernelGGL(fields_split_k, dim3((unsigned)((t.nrows + FS_ROWS - 1) / FS_ROWS)), dim3(256), 0, stream, col + t.base, val + t.base, t.nrows, z, d,
                         ws.keys_out, brow + t.base, bval + t.base, keys_in, rows_in);
      size_t tb32 = ws.sort_bytes;
      FMX_HIP(sort_pairs_u32(ws.sort_temp, tb32, keys_in, ws.keys_out + n_dense, rows_in, brow + t.base + n_dense, (size_t)n_cat, ws.bits, stream));
    }
  } else if (cnt > 0 && unit_values && field_base && fixed_row_len > 0 && fixed_row_len <= FMX_MAX_FIELDS && (int)field_base->size() == fixed_row_len + 1 &&
             cnt == t.nrows * (int64_t)fixed_row_len && !(getenv("FMX_FIELD_SORT") && getenv("FMX_FIELD_SORT")[0] == '0') && ws.fq_counts != nullptr) {
    // one-hot rows whose entry i is an id of field i (the uniform generator: one column per stratum of [0, p)): the per-field sort with no dense part
    const int z = fixed_row_len;
    field_sorted = true;
    FieldBases fb{};
    for (int c = 0; c < z; ++c) fb.base[c] = (*field_base)[(size_t)c];
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(ws.vals_in);
    uint32_t* rows_in = keys_in + cnt;
    hipLaunchKernelGGL(fields_split_local_k, dim3((unsigned)((t.nrows + FS_ROWS - 1) / FS_ROWS)), dim3(256), 0, stream, col + t.base, (const float*)nullptr, t.nrows, z, 0, fb,
                       ws.keys_out, brow + t.base, bval + t.base, keys_in, 1);
    uint32_t* b_keys = reinterpret_cast<uint32_t*>(ws.vals_out);
    FMX_TRY(field_sort(*field_base, (uint32_t)t.nrows, keys_in, rows_in, b_keys, b_keys + cnt, ws.keys_out, brow + t.base, ws.fq_counts, ws.fq_totals, stream));
  } else if (cnt > 0 && unit_values) {
    // one-hot values: sort (column, row) pairs straight into brow -- 8 bytes per entry and pass instead of 12, no unpack pass;
    // bval is never read for such a matrix
    uint32_t* rows32 = reinterpret_cast<uint32_t*>(ws.vals_in);
    hipLaunchKernelGGL(pack_rows_k, grid(cnt), dim3(T), 0, stream, row_ptr, t.r0, t.nrows, t.base, cnt, rows32, fixed_row_len);
    const char* ps_env = getenv("FMX_PAIR_SORT");  // read per build (the tests compare the two): "hand" takes the hand-written sort below
    if (ps_env && ps_env[0] == 'h' && ws.fq_counts != nullptr) {
      // [r4] the hand-written LSD sort of the per-field plans, run as ONE "field" [0, p) over the tile's cnt entries (block histograms, a digit's blocks scanned in
      // order, blocks ordered by digit in LDS and written as runs): the first pass reads the matrix's own column array and the rows just packed, the last one
      // writes the sorted columns and rows straight into the plan.  Stable, so rows stay ascending inside a list: the same plan, bit for bit, as the library sort
      // (tests/test_gpu_api.py).  MEASURED (profiles/r04_pair_sort.txt): 24.2 against 12.9 ms for the 39 tiles of a 10 M x 1 M matrix of i.i.d. columns, 28.0 against
      // 16.4 ms on ragged rows -- with one field there is nothing to run side by side, every pass reads the keys twice (count, then scatter) where the library's
      // onesweep reads them once and chains its blocks by lookback; the streamed rate does not move (144.7 M examples/s either way: phase 2 bounds it).  So the
      // library's sort stays the default for tiles WITHOUT a field layout, and this form is opt-in (FMX_PAIR_SORT=hand); tiles with one keep the per-field sort.
      const std::vector<uint32_t> one{0u, p};
      uint32_t* a_keys = rows32 + cnt;                                  // (the u64 payload buffers hold two u32 arrays each)
      uint32_t* b_keys = reinterpret_cast<uint32_t*>(ws.vals_out);
      FMX_TRY(field_sort(one, (uint32_t)cnt, a_keys, rows32, b_keys, b_keys + cnt, ws.keys_out, brow + t.base, ws.fq_counts, ws.fq_totals, stream, col + t.base, rows32));
    } else {
      size_t tb32 = ws.sort_bytes;  // sized for the (u32, u64) sort of the same length: the (u32, u32) one needs no more
      FMX_HIP(sort_pairs_u32(ws.sort_temp, tb32, col + t.base, ws.keys_out, rows32, brow + t.base, (size_t)cnt, ws.bits, stream));
    }
  } else if (cnt > 0) {
    const char* ps_env = getenv("FMX_PAIR_SORT");
    if (ps_env && ps_env[0] == 'h' && ws.fq_counts != nullptr) {
      // [r4] real values (opt-in like the one-hot form above): the same hand-written sort on (column, POSITION in the tile) -- pass 0 takes the position for the row by itself -- with the sorted
      // positions parked in the plan's value array; one gather then turns a position into the entry's row and value (12 bytes per entry and pass before: a u64 payload)
      const std::vector<uint32_t> one{0u, p};
      uint32_t* a_keys = reinterpret_cast<uint32_t*>(ws.vals_in);
      uint32_t* b_keys = reinterpret_cast<uint32_t*>(ws.vals_out);
      uint32_t* pos = reinterpret_cast<uint32_t*>(bval + t.base);
      FMX_TRY(field_sort(one, (uint32_t)cnt, a_keys, a_keys + cnt, b_keys, b_keys + cnt, ws.keys_out, pos, ws.fq_counts, ws.fq_totals, stream, col + t.base, nullptr));
      hipLaunchKernelGGL(gather_sorted_k, grid(cnt), dim3(T), 0, stream, row_ptr, val, t.r0, t.nrows, t.base, cnt, brow + t.base, bval + t.base, fixed_row_len);
    } else {
      hipLaunchKernelGGL(pack_entries_k, grid(cnt), dim3(T), 0, stream, row_ptr, val, t.r0, t.nrows, t.base, cnt, ws.vals_in, fixed_row_len);
      FMX_HIP(rocprim::radix_sort_pairs(ws.sort_temp, ws.sort_bytes, col + t.base, ws.keys_out, ws.vals_in, ws.vals_out, (size_t)cnt, 0, ws.bits, stream));
      hipLaunchKernelGGL(unpack_entries_k, grid(cnt), dim3(T), 0, stream, ws.vals_out, cnt, brow + t.base, bval + t.base);
    }
  }
  FMX_HIP(hipMemsetAsync(t.dcounts, 0, 4 * sizeof(uint32_t), stream));
  rocprim::counting_iterator<uint32_t> ids(0);
  size_t tb = ws.prim_bytes;
  const uint32_t* off;
  uint32_t n_max;
  if (t.off_in_pool) {  // dense directory
    FMX_HIP(hipMemsetAsync(t.off, 0, ((size_t)p + 1) * sizeof(uint32_t), stream));
    if (cnt > 0) hipLaunchKernelGGL((run_ends_k<uint32_t>), grid(cnt), dim3(T), 0, stream, ws.keys_out, cnt, t.off);
    FMX_HIP(rocprim::inclusive_scan(ws.prim_temp, tb, t.off, t.off, (size_t)p + 1, rocprim::maximum<uint32_t>(), stream));
    off = t.off; n_max = p;
  } else {              // sparse directory: run starts -> soff, ids -> feat
    // run heads of the sorted columns -> list starts, ids, first entries (one ordered compaction; dcounts[0] = the number of lists)
    if (cnt > 0 && field_sorted) {
      // the dense columns' heads are known (column c starts at c * nrows with row 0); only the one-hot part is searched, and its values are all 1
      const uint32_t d = (uint32_t)dense_prefix, first = d * (uint32_t)t.nrows, n_cat = (uint32_t)cnt - first;
      const uint32_t nb = (n_cat + CP_CHUNK - 1) / CP_CHUNK;
      hipLaunchKernelGGL(dense_heads_k, dim3(1), dim3(64), 0, stream, (int)d, (uint32_t)t.nrows, (const float*)(bval + t.base), t.soff, t.feat, t.row0, t.val0);
      hipLaunchKernelGGL(heads_count_k, dim3(nb), dim3(CP_THREADS), 0, stream, ws.keys_out + first, n_cat, ws.blk);
      hipLaunchKernelGGL(compact_scan_k, dim3(1), dim3(1024), 0, stream, ws.blk, nb, t.dcounts, d);
      hipLaunchKernelGGL(heads_write_k, dim3(nb), dim3(CP_THREADS), 0, stream, ws.keys_out + first, n_cat, (const uint32_t*)ws.blk, t.soff, t.feat,
                         (const uint32_t*)(brow + t.base + first), (const float*)nullptr, t.row0, t.val0, first);
    } else if (cnt > 0) {
      const uint32_t nb = ((uint32_t)cnt + CP_CHUNK - 1) / CP_CHUNK;
      hipLaunchKernelGGL(heads_count_k, dim3(nb), dim3(CP_THREADS), 0, stream, ws.keys_out, (uint32_t)cnt, ws.blk);
      hipLaunchKernelGGL(compact_scan_k, dim3(1), dim3(1024), 0, stream, ws.blk, nb, t.dcounts, 0u);
      hipLaunchKernelGGL(heads_write_k, dim3(nb), dim3(CP_THREADS), 0, stream, ws.keys_out, (uint32_t)cnt, (const uint32_t*)ws.blk, t.soff, t.feat,
                         (const uint32_t*)(brow + t.base), unit_values ? (const float*)nullptr : (const float*)(bval + t.base), t.row0, t.val0, 0u);
    }
    hipLaunchKernelGGL(close_directory_k, dim3(1), dim3(1), 0, stream, t.soff, (const uint32_t*)t.dcounts, (uint32_t)cnt);
    off = t.soff; n_max = t.cap_lists;
  }
  // long lists
  if (cnt > (int64_t)list_long_min() && n_max > 0) {
    // lists longer than long_min, ascending (dense directory: among all p features; sparse: among the dcounts[0] lists)
    FMX_TRY(compact_indices(LongPred{off, list_long_min()}, LongEmit{t.lpos}, t.off_in_pool ? p : n_max, t.off_in_pool ? (const uint32_t*)nullptr : (const uint32_t*)t.dcounts,
                            ws.blk, t.dcounts + 1, stream));
    hipLaunchKernelGGL(long_nseg_k, grid((int64_t)t.cap_long + 1), dim3(T), 0, stream, off, t.lpos, t.dcounts, t.cap_long, ws.nseg);
    tb = ws.prim_bytes;
    FMX_HIP(rocprim::exclusive_scan(ws.prim_temp, tb, ws.nseg, t.lseg_ptr, 0u, (size_t)t.cap_long + 1, rocprim::plus<uint32_t>(), stream));
    hipLaunchKernelGGL(long_segments_k, grid(t.cap_long), dim3(T), 0, stream, off, t.off_in_pool ? (const uint32_t*)nullptr : t.feat, t.lpos, t.lseg_ptr, t.dcounts,
                       t.lfeat, t.seg_list, t.seg_begin, t.seg_end);
  }
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// the counts the builder left on the device -> host fields (the caller has waited for the stream)
void plan_set_counts(fmx_matrix::TilePlan& t, uint32_t p, const uint32_t* h) {
  t.n_lists = t.off_in_pool ? p : h[0];
  t.n_long = h[1];
  t.n_seg = h[2];
}

// a dense directory for a tile planned with the sparse one (launches that walk the dense exchange buffer visit every feature)
int plan_ensure_dense(fmx_matrix* m, int64_t tile, hipStream_t stream) {
  fmx_matrix::TilePlan& t = m->plans[(size_t)tile];
  if (t.off) return FMX_OK;
  const size_t np1 = (size_t)m->p + 1;
  uint32_t* off = nullptr;
  FMX_HIP(hipMalloc(&off, np1 * sizeof(uint32_t)));
  void* tmp = nullptr;
  size_t need = 0;
  auto body = [&]() -> int {
    FMX_HIP(hipMemsetAsync(off, 0, np1 * sizeof(uint32_t), stream));
    if (t.n_lists > 0) hipLaunchKernelGGL(scatter_ends_k, dim3((t.n_lists + 255) / 256), dim3(256), 0, stream, t.feat, t.soff, t.n_lists, off);
    FMX_HIP(rocprim::inclusive_scan(nullptr, need, off, off, np1, rocprim::maximum<uint32_t>(), stream));
    FMX_HIP(hipMalloc(&tmp, need ? need : 16));
    FMX_HIP(rocprim::inclusive_scan(tmp, need, off, off, np1, rocprim::maximum<uint32_t>(), stream));
    FMX_HIP(hipStreamSynchronize(stream));
    return FMX_OK;
  };
  const int st = body();
  (void)hipFree(tmp);
  if (st != FMX_OK) { (void)hipFree(off); return st; }
  t.off = off;
  t.off_in_pool = 0;
  return FMX_OK;
}

// ------------------------------------------------------------------------------------------------ owner-major order of a directory
// Owner-sharded exchange: feature j belongs to rank j mod N.  A rank's records (and the ids it asks the owners
// for) travel in owner-major order, so that the part for owner o is ONE contiguous slice -- no packing pass per step.  The order
// is a stable partition of the tile's ascending directory by (id mod N): ids stay ascending inside an owner's slice, which is
// what keeps the owner's merge (stable sort by id over the parts in rank order) equal to the all-gather form's.
OwnerWorkspace::~OwnerWorkspace() { (void)hipFree(keys_in); (void)hipFree(keys_out); (void)hipFree(idx_in); (void)hipFree(idx_out); (void)hipFree(temp); }

int OwnerWorkspace::reserve(uint32_t n, hipStream_t stream) {
  if (n <= cap && keys_in) return FMX_OK;
  FMX_HIP(hipStreamSynchronize(stream));
  (void)hipFree(keys_in); (void)hipFree(keys_out); (void)hipFree(idx_in); (void)hipFree(idx_out); (void)hipFree(temp);
  keys_in = keys_out = idx_in = idx_out = nullptr; temp = nullptr; cap = 0;
  const size_t m = n ? n : 1;
  FMX_HIP(hipMalloc(&keys_in, m * 4)); FMX_HIP(hipMalloc(&keys_out, m * 4)); FMX_HIP(hipMalloc(&idx_in, m * 4)); FMX_HIP(hipMalloc(&idx_out, m * 4));
  FMX_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, idx_in, idx_out, m, 0, 5, stream));
  FMX_HIP(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
  cap = (uint32_t)m;
  return FMX_OK;
}

int plan_owner_alloc(fmx_matrix::TilePlan& t, uint32_t cap_lists) {
  if (t.own_pool && t.own_cap >= cap_lists) return FMX_OK;
  (void)hipFree(t.own_pool); t.own_pool = nullptr; t.own_cap = 0; t.own_n = 0;
  const size_t c = cap_lists ? cap_lists : 1;
  const size_t b_counts = align_up((OWNERS_MAX + 2) * sizeof(uint32_t), 256), b_arr = align_up(c * sizeof(uint32_t), 256);
  FMX_HIP(hipMalloc(&t.own_pool, b_counts + 2 * b_arr));
  char* b = (char*)t.own_pool;
  t.own_counts = (uint32_t*)b;
  t.own_pos = (uint32_t*)(b + b_counts);
  t.own_ids = (uint32_t*)(b + b_counts + b_arr);
  t.own_cap = (uint32_t)c;
  return FMX_OK;
}

__global__ void owner_keys_k(const uint32_t* __restrict__ feat, const uint32_t* __restrict__ dcounts, uint32_t n_sort, uint32_t n_owners,
                             uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n_sort) return;
  keys[i] = i < (int64_t)dcounts[0] ? feat[i] % n_owners : n_owners;  // slots beyond the directory's end sort behind every owner
  idx[i] = (uint32_t)i;
}

// sorted keys -> counts per owner (n_owners + 1 threads: the first position of each key value), positions and ids
__global__ void owner_finish_k(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ order, const uint32_t* __restrict__ feat,
                               const uint32_t* __restrict__ dcounts, uint32_t n_sort, uint32_t n_owners, uint32_t* __restrict__ counts,
                               uint32_t* __restrict__ pos, uint32_t* __restrict__ ids) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = dcounts[0] < n_sort ? dcounts[0] : n_sort;
  if (j < (int64_t)n) {
    const uint32_t i = order[j];
    pos[i] = (uint32_t)j;
    ids[j] = feat[i];
  }
  if (j <= (int64_t)n_owners) {  // owner j's slice starts at the first sorted key >= j; counts[j] = start(j + 1) - start(j)
    auto lower = [&](uint32_t key) {
      uint32_t lo = 0, hi = n;
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
      return lo;
    };
    if (j < (int64_t)n_owners) counts[j] = lower((uint32_t)j + 1) - lower((uint32_t)j);
    else counts[j] = n;  // total, behind the per-owner counts
  }
}

int plan_owner_build(fmx_matrix::TilePlan& t, OwnerWorkspace& ws, int n_owners, uint32_t n_sort, hipStream_t stream) {
  FMX_CHECK(n_owners >= 1 && n_owners <= OWNERS_MAX, FMX_ERR_INVALID, "1..%d owners are supported (got %d)", OWNERS_MAX, n_owners);
  FMX_CHECK(t.feat != nullptr, FMX_ERR_STATE, "the owner-sharded exchange needs a sparse tile (fewer entries than features)");
  FMX_CHECK(t.own_pool != nullptr && n_sort <= t.own_cap && n_sort <= ws.cap, FMX_ERR_STATE, "owner plan: arrays too small");
  const int T = 256;
  FMX_HIP(hipMemsetAsync(t.own_counts, 0, (OWNERS_MAX + 2) * sizeof(uint32_t), stream));
  if (n_sort > 0) {
    hipLaunchKernelGGL(owner_keys_k, dim3((n_sort + T - 1) / T), dim3(T), 0, stream, t.feat, t.dcounts, n_sort, (uint32_t)n_owners, ws.keys_in, ws.idx_in);
    size_t tb = ws.temp_bytes;
    int bits = 1;
    while ((1 << bits) <= n_owners) ++bits;  // keys 0 .. n_owners
    FMX_HIP(rocprim::radix_sort_pairs(ws.temp, tb, ws.keys_in, ws.keys_out, ws.idx_in, ws.idx_out, (size_t)n_sort, 0, bits, stream));
  }
  const uint32_t g = (n_sort > (uint32_t)n_owners + 1 ? n_sort : (uint32_t)n_owners + 1);
  hipLaunchKernelGGL(owner_finish_k, dim3((g + T - 1) / T), dim3(T), 0, stream, ws.keys_out, ws.idx_out, t.feat, t.dcounts, n_sort, (uint32_t)n_owners, t.own_counts,
                     t.own_pos, t.own_ids);
  FMX_HIP(hipGetLastError());
  t.own_n = n_owners;
  return FMX_OK;
}

// ------------------------------------------------------------------------------------------------ rows <-> packed buffer
// What an owner sends back for a pulled feature, and what the asking rank stores: the V row and w in the STATE's element type,
// [kp | w 0 0 0] per feature (16-byte aligned rows).  Optimizer state never travels: it lives with the owner.
template <typename T, int VEC, bool UNPACK>
__global__ void rows_pack_k(T* __restrict__ V, T* __restrict__ w, int kp, int vs, int ws, const uint32_t* __restrict__ ids, int64_t n, T* __restrict__ rows) {
  const int lpr = kp / VEC + 1;  // the row's slices, then the w slice
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * lpr) return;
  const int64_t i = t / lpr;
  const int s = (int)(t - i * lpr);
  const size_t j = ids[i];
  T* row = rows + (size_t)i * (kp + 4);
  using vec_t = typename std::conditional<sizeof(T) == 4, float4, double2>::type;
  if (s < kp / VEC) {
    vec_t* a = reinterpret_cast<vec_t*>(V + j * vs + s * VEC);
    vec_t* b = reinterpret_cast<vec_t*>(row + s * VEC);
    if (UNPACK) *a = *b; else *b = *a;
  } else if (UNPACK) {
    w[j * ws] = row[kp];
  } else {
    row[kp] = w[j * ws];
    for (int q = 1; q < 4; ++q) row[kp + q] = (T)0;
  }
}

int rows_pack(fmx_engine* e, const uint32_t* d_ids, int64_t n, void* d_rows, bool unpack) {
  if (n <= 0) return FMX_OK;
  const int kp = mb_kp(e);
  if (mb_wide(e)) {
    const int64_t total = n * (kp / 2 + 1);
    const dim3 g((unsigned)((total + 255) / 256)), b(256);
    if (unpack) hipLaunchKernelGGL((rows_pack_k<double, 2, true>), g, b, 0, e->stream, e->dV, e->dw, kp, kp, 1, d_ids, n, (double*)d_rows);
    else hipLaunchKernelGGL((rows_pack_k<double, 2, false>), g, b, 0, e->stream, e->dV, e->dw, kp, kp, 1, d_ids, n, (double*)d_rows);
  } else {
    const int64_t total = n * (kp / 4 + 1);
    const dim3 g((unsigned)((total + 255) / 256)), b(256);
    float* wb = (float*)mb_wbase(e);
    if (unpack) hipLaunchKernelGGL((rows_pack_k<float, 4, true>), g, b, 0, e->stream, e->V, wb, kp, mb_vstride(e), mb_wstride(e), d_ids, n, (float*)d_rows);
    else hipLaunchKernelGGL((rows_pack_k<float, 4, false>), g, b, 0, e->stream, e->V, wb, kp, mb_vstride(e), mb_wstride(e), d_ids, n, (float*)d_rows);
  }
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

void drop_plans(fmx_matrix* m) {
  for (auto& t : m->plans) plan_free(t);
  m->plans.clear();
  m->step_first_tile.clear();
  (void)hipFree(m->brow); (void)hipFree(m->bval);
  m->brow = nullptr; m->bval = nullptr;
  m->batch_rows = 0; m->tile_rows = 0; m->n_batches = 0; m->max_long_seg = 0; m->max_tile_cnt = 0;
  m->plan_generation++;
}

__global__ void gather_rows_k(const int64_t* __restrict__ src, const int64_t* __restrict__ rows, int64_t count, int64_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = src[rows[i]];
}

// Plans of every tile of the matrix for steps of batch_rows rows cut into tiles of at most tile_rows rows.  Built into
// locals and committed to the matrix only when everything succeeded: a failure (out of memory is the expected one: the
// sorted copy is as large as the matrix) leaves the matrix without plans, and the next call starts over.
int build_batch_csc(fmx_matrix* m, int64_t batch_rows, int64_t tile_rows, hipStream_t stream) {
  FMX_CHECK(batch_rows > 0 && tile_rows > 0, FMX_ERR_INVALID, "batch_rows and tile_rows must be positive");
  if (m->batch_rows == batch_rows && m->tile_rows == tile_rows && !m->plans.empty()) return FMX_OK;
  FMX_HIP(hipSetDevice(m->device));
  drop_plans(m);
  // steps of batch_rows rows, each cut into tiles of at most tile_rows rows
  const int64_t nb = (m->n + batch_rows - 1) / batch_rows;
  std::vector<int64_t> tile_start, step_first((size_t)nb + 1, 0);
  for (int64_t s = 0; s < nb; ++s) {
    step_first[(size_t)s] = (int64_t)tile_start.size();
    const int64_t end = (s + 1) * batch_rows < m->n ? (s + 1) * batch_rows : m->n;
    for (int64_t r = s * batch_rows; r < end; r += tile_rows) tile_start.push_back(r);
  }
  step_first[(size_t)nb] = (int64_t)tile_start.size();
  tile_start.push_back(m->n);
  const int64_t nt = (int64_t)tile_start.size() - 1;

  struct Locals {  // released unless committed
    std::vector<fmx_matrix::TilePlan> plans;
    uint32_t* brow = nullptr; float* bval = nullptr;
    int64_t *d_rows = nullptr, *d_ptr = nullptr;
    uint32_t* h_counts = nullptr;
    ~Locals() {
      for (auto& t : plans) plan_free(t);
      (void)hipFree(brow); (void)hipFree(bval); (void)hipFree(d_rows); (void)hipFree(d_ptr);
      if (h_counts) (void)hipHostFree(h_counts);
    }
  } L;
  // row_ptr at the tile boundaries -> host
  std::vector<int64_t> h_ptr((size_t)nt + 1, 0);
  FMX_HIP(hipMalloc(&L.d_rows, ((size_t)nt + 1) * sizeof(int64_t)));
  FMX_HIP(hipMalloc(&L.d_ptr, ((size_t)nt + 1) * sizeof(int64_t)));
  FMX_HIP(hipMemcpyAsync(L.d_rows, tile_start.data(), ((size_t)nt + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(gather_rows_k, dim3((unsigned)((nt + 1 + 255) / 256)), dim3(256), 0, stream, m->row_ptr, L.d_rows, nt + 1, L.d_ptr);
  FMX_HIP(hipMemcpyAsync(h_ptr.data(), L.d_ptr, ((size_t)nt + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
  FMX_HIP(hipStreamSynchronize(stream));
  int64_t max_cnt = 0;
  for (int64_t t = 0; t < nt; ++t) {
    const int64_t c = h_ptr[(size_t)t + 1] - h_ptr[(size_t)t];
    FMX_CHECK(c < (1LL << 32), FMX_ERR_INVALID, "a tile holds %lld nonzeros; at most 2^32-1 are supported (lower tile_rows)", (long long)c);
    if (c > max_cnt) max_cnt = c;
  }
  // tests: an allocation failure halfway must leave no half-built cache (armed by fmx_debug_fail_next_plan_build; one shot)
  if (g_fail_next_plan_build.exchange(0) > 0) { set_error("plan build failed (fmx_debug_fail_next_plan_build)"); return FMX_ERR_HIP; }
  FMX_HIP(hipMalloc(&L.brow, (size_t)(m->nnz > 0 ? m->nnz : 1) * sizeof(uint32_t)));
  FMX_HIP(hipMalloc(&L.bval, (size_t)(m->nnz > 0 ? m->nnz : 1) * sizeof(float)));
  FMX_HIP(hipHostMalloc(&L.h_counts, (size_t)(nt > 0 ? nt : 1) * 4 * sizeof(uint32_t)));
  PlanWorkspace ws;
  FMX_TRY(ws.reserve(max_cnt, m->p, stream));
  L.plans.resize((size_t)nt);
  for (int64_t t = 0; t < nt; ++t) {
    fmx_matrix::TilePlan& pl = L.plans[(size_t)t];
    const int64_t base = h_ptr[(size_t)t], cnt = h_ptr[(size_t)t + 1] - base;
    // a tile with at least as many entries as features touches most of them: one list per feature; otherwise only the
    // occurring features get a list (p = 33 M against 10 M entries per tile at configs[3]: no p-sized array per tile)
    FMX_TRY(plan_alloc(pl, m->p, cnt, cnt >= (int64_t)m->p));
    pl.r0 = tile_start[(size_t)t]; pl.nrows = tile_start[(size_t)t + 1] - pl.r0; pl.base = base; pl.cnt = cnt;
    FMX_TRY(plan_build(pl, ws, m->p, m->row_ptr, m->col, m->val, L.brow, L.bval, stream, m->unit_values, m->fixed_row_len, m->dense_prefix,
                       m->field_base.empty() ? nullptr : &m->field_base));
    FMX_HIP(hipMemcpyAsync(L.h_counts + 4 * t, pl.dcounts, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  }
  FMX_HIP(hipStreamSynchronize(stream));
  int64_t max_seg = 0;
  for (int64_t t = 0; t < nt; ++t) {
    plan_set_counts(L.plans[(size_t)t], m->p, L.h_counts + 4 * t);
    if ((int64_t)L.plans[(size_t)t].n_seg > max_seg) max_seg = L.plans[(size_t)t].n_seg;
  }
  // commit
  m->plans.swap(L.plans);
  m->brow = L.brow; L.brow = nullptr;
  m->bval = L.bval; L.bval = nullptr;
  m->step_first_tile.swap(step_first);
  m->batch_rows = batch_rows; m->tile_rows = tile_rows; m->n_batches = nb;
  m->max_long_seg = max_seg; m->max_tile_cnt = max_cnt;
  m->plan_generation++;
  return FMX_OK;
}

// ------------------------------------------------------------------------------------------------ compact exchange: merge
// The records of all ranks (part r = counts[r] records starting at record r * stride) are put in feature order without
// moving them: a stable sort of (feature id, position) pairs keeps the parts of one feature in rank order, a select of the
// run heads gives one list per feature.  Results stay in the engine's MergeWs for launch_apply_records().
struct MergeWs {
  int64_t cap = 0;
  int n_parts_cap = 0;
  uint32_t *keys_in = nullptr, *keys_out = nullptr, *pos_in = nullptr, *pos_out = nullptr, *roff = nullptr, *rfeat = nullptr, *dcount = nullptr;
  uint32_t* blk = nullptr;   // block counts of the ordered compaction (heads)
  void* sort_temp = nullptr; size_t sort_bytes = 0;
};

void merge_ws_free(MergeWs* w) {
  if (!w) return;
  (void)hipFree(w->keys_in); (void)hipFree(w->keys_out); (void)hipFree(w->pos_in); (void)hipFree(w->pos_out); (void)hipFree(w->roff);
  (void)hipFree(w->rfeat); (void)hipFree(w->dcount); (void)hipFree(w->blk); (void)hipFree(w->sort_temp);
  delete w;
}

static int merge_ws_reserve(fmx_engine* e, int64_t total, int n_parts) {
  if (!e->merge) e->merge = new MergeWs();
  MergeWs& w = *e->merge;
  if (total <= w.cap && n_parts <= w.n_parts_cap) return FMX_OK;
  FMX_HIP(hipStreamSynchronize(e->stream));
  MergeWs* old = e->merge;
  e->merge = new MergeWs();
  merge_ws_free(old);
  MergeWs& n = *e->merge;
  const size_t m = (size_t)(total > 0 ? total : 1) * 5 / 4 + 64;  // some headroom: the total changes a little from step to step
  FMX_HIP(hipMalloc(&n.keys_in, m * 4)); FMX_HIP(hipMalloc(&n.keys_out, m * 4));
  FMX_HIP(hipMalloc(&n.pos_in, m * 4)); FMX_HIP(hipMalloc(&n.pos_out, m * 4));
  FMX_HIP(hipMalloc(&n.roff, (m + 1) * 4)); FMX_HIP(hipMalloc(&n.rfeat, m * 4));
  FMX_HIP(hipMalloc(&n.dcount, 16)); FMX_HIP(hipMalloc(&n.blk, (m / CP_CHUNK + 2) * 4));
  {  // scratch of the pair sort: the larger of the tuned nine-bit form and rocprim's default (whichever sort_pairs_u32 picks for the bit count)
    size_t a = 0, b = 0;
    FMX_HIP(rocprim::radix_sort_pairs(nullptr, a, n.keys_in, n.keys_out, n.pos_in, n.pos_out, m, 0, 32, e->stream));
    FMX_HIP(sort_pairs_u32(nullptr, b, n.keys_in, n.keys_out, n.pos_in, n.pos_out, m, 25, e->stream));
    n.sort_bytes = a > b ? a : b;
  }
  FMX_HIP(hipMalloc(&n.sort_temp, n.sort_bytes ? n.sort_bytes : 16));
  n.cap = (int64_t)m;
  n.n_parts_cap = n_parts;
  return FMX_OK;
}

int merge_records(fmx_engine* e, const void* recs, const int64_t* counts, const int64_t* starts, int n_parts, int64_t stride, int64_t* total_out) {
  FMX_CHECK(n_parts >= 1 && n_parts <= REC_PARTS_MAX, FMX_ERR_INVALID, "1..%d record parts are supported (got %d)", REC_PARTS_MAX, n_parts);
  RecParts parts{};
  parts.n = n_parts;
  int64_t top = 0;
  for (int r = 0; r < n_parts; ++r) {
    FMX_CHECK(counts[r] >= 0 && (starts != nullptr || counts[r] <= stride), FMX_ERR_INVALID, "part %d holds %lld records, stride is %lld", r, (long long)counts[r], (long long)stride);
    parts.prefix[r + 1] = parts.prefix[r] + counts[r];
    parts.start[r] = starts ? starts[r] : (int64_t)r * stride;
    FMX_CHECK(parts.start[r] >= 0, FMX_ERR_INVALID, "part %d starts at a negative record", r);
    if (parts.start[r] + counts[r] > top) top = parts.start[r] + counts[r];
  }
  const int64_t total = parts.prefix[n_parts];
  FMX_CHECK(top < (1LL << 32), FMX_ERR_INVALID, "more than 2^32 record slots");
  FMX_TRY(merge_ws_reserve(e, total, n_parts));
  MergeWs& w = *e->merge;
  *total_out = total;
  FMX_HIP(hipMemsetAsync(w.dcount, 0, 16, e->stream));
  if (total == 0) return FMX_OK;
  FMX_TRY(launch_record_keys(e, recs, parts, total, w.keys_in, w.pos_in));
  // (feature id, slot) pairs sorted by id -- stable: a feature's parts stay in rank order -- with the plan builder's pair sort (nine-bit passes where
  // they save one) and its ordered compaction for the run heads (list starts and ids in one pass) instead of a flag array + rocprim::select + a fix-up pass
  const int bits = col_bits((uint32_t)e->p);
  size_t tb = w.sort_bytes;
  FMX_HIP(sort_pairs_u32(w.sort_temp, tb, w.keys_in, w.keys_out, w.pos_in, w.pos_out, (size_t)total, bits, e->stream));
  const uint32_t nb = ((uint32_t)total + CP_CHUNK - 1) / CP_CHUNK;
  hipLaunchKernelGGL(heads_count_k, dim3(nb), dim3(CP_THREADS), 0, e->stream, w.keys_out, (uint32_t)total, w.blk);
  hipLaunchKernelGGL(compact_scan_k, dim3(1), dim3(1024), 0, e->stream, w.blk, nb, w.dcount, 0u);
  hipLaunchKernelGGL(heads_write_k, dim3(nb), dim3(CP_THREADS), 0, e->stream, w.keys_out, (uint32_t)total, (const uint32_t*)w.blk, w.roff, w.rfeat,
                     (const uint32_t*)nullptr, (const float*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u);
  hipLaunchKernelGGL(close_directory_k, dim3(1), dim3(1), 0, e->stream, w.roff, (const uint32_t*)w.dcount, (uint32_t)total);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// what launch_apply_records needs from the merge
void merge_result(const fmx_engine* e, const uint32_t** pos, const uint32_t** roff, const uint32_t** rfeat, const uint32_t** d_n) {
  const MergeWs& w = *e->merge;
  *pos = w.pos_out; *roff = w.roff; *rfeat = w.rfeat; *d_n = w.dcount;
}

int build_full_csc(fmx_matrix* m, hipStream_t stream) {
  if (m->col_ptr) return FMX_OK;
  FMX_CHECK(m->n < (1LL << 32), FMX_ERR_INVALID, "ALS sweep supports fewer than 2^32 rows");
  // one sort over ALL entries and u32 row ids per column list: verified up to 3e8 entries; beyond 2^32 the whole-matrix CSC is refused, not built on trust
  // (the mini-batch path has no whole-matrix structure: its plans are per tile -- tests/test_gpu_nnz_2p32.py)
  FMX_CHECK(m->nnz < (1LL << 32), FMX_ERR_INVALID, "the ALS / MCMC sweeps and column scaling need the CSC of the whole matrix, which is limited to fewer than 2^32 stored entries (this one: %lld)",
            (long long)m->nnz);
  FMX_HIP(hipSetDevice(m->device));
  FMX_HIP(hipMalloc(&m->col_ptr, ((size_t)m->p + 1) * sizeof(int64_t)));
  FMX_HIP(hipMalloc(&m->crow, (size_t)(m->nnz > 0 ? m->nnz : 1) * sizeof(uint32_t)));
  FMX_HIP(hipMalloc(&m->cval, (size_t)(m->nnz > 0 ? m->nnz : 1) * sizeof(float)));
  const int bits = col_bits(m->p);
  SortScratch s;
  FMX_TRY(sort_scratch_alloc(s, m->nnz, bits, stream));
  FMX_TRY(csc_of_range<int64_t>(m, s, bits, 0, m->n, 0, m->nnz, m->crow, m->cval, m->col_ptr, stream));
  FMX_HIP(hipStreamSynchronize(stream));
  return FMX_OK;
}

// ------------------------------------------------------------------------------------------------ synthetic data
struct Philox {
  uint32_t c[4];
};
__host__ __device__ inline Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox{{c0, c1, c2, c3}};
}

// entry i of global row g: stratum i of nnz equal-width strata of [0,p); position inside it from Philox(seed; g, i/4)[i%4]
__global__ void synth_entries_k(int64_t n, uint32_t p, int32_t z, uint64_t seed, int64_t row_offset, uint32_t* __restrict__ col,
                                float* __restrict__ val) {
  // one thread per Philox block (entries 4 q .. 4 q + 3 of a row)
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int gq = (z + 3) / 4;
  if (t >= n * gq) return;
  const int64_t r = t / gq;
  const uint32_t q = (uint32_t)(t - r * gq);
  const uint64_t g = (uint64_t)(row_offset + r);
  const Philox ph = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), q, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t i = 4 * q + (uint32_t)u;
    if ((int)i >= z) break;
    const uint32_t lo = (uint32_t)(((uint64_t)i * p) / (uint32_t)z);
    const uint32_t hi = (uint32_t)(((uint64_t)(i + 1) * p) / (uint32_t)z);
    col[r * z + i] = lo + (uint32_t)(((uint64_t)ph.c[u] * (hi - lo)) >> 32);
    val[r * z + i] = 1.0f;
  }
}

__global__ void synth_rows_k(int64_t n, int32_t z, uint64_t seed, int64_t row_offset, int64_t* __restrict__ row_ptr, float* __restrict__ y) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  row_ptr[r] = r * z;
  if (r < n) {
    const uint64_t g = (uint64_t)(row_offset + r);
    const Philox ph = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), 0xFFFFFFFFu, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    y[r] = (ph.c[0] & 1u) ? 1.0f : -1.0f;
  }
}

// entry i of every row of the uniform generator is a column of stratum i = [i p / z, (i + 1) p / z): field-structured rows (fmx_matrix::field_base)
void strata_bounds(uint32_t p, int32_t z, std::vector<uint32_t>* out) {
  out->clear();
  if (z < 1 || z > FMX_MAX_FIELDS) return;
  for (int i = 0; i <= z; ++i) out->push_back((uint32_t)(((uint64_t)i * p) / (uint32_t)z));
}

// enqueue the generator for rows [row_offset, row_offset + n) on `stream` (no wait)
int generate_synthetic_async(fmx_matrix* m, int64_t n, int32_t z, uint64_t seed, int64_t row_offset, hipStream_t stream) {
  const int T = 256;
  const int64_t total = n * ((z + 3) / 4);
  if (total > 0)
    hipLaunchKernelGGL(synth_entries_k, dim3((unsigned)((total + T - 1) / T)), dim3(T), 0, stream, n, m->p, z, seed, row_offset, m->col, m->val);
  hipLaunchKernelGGL(synth_rows_k, dim3((unsigned)((n + 1 + T - 1) / T)), dim3(T), 0, stream, n, z, seed, row_offset, m->row_ptr, m->y);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int generate_synthetic(fmx_matrix* m, int32_t z, uint64_t seed, int64_t row_offset) {
  FMX_TRY(generate_synthetic_async(m, m->n, z, seed, row_offset, nullptr));
  FMX_HIP(hipDeviceSynchronize());
  m->rows_sorted = 1;  // strata are disjoint and ascending
  m->max_row_len = z;
  m->fixed_row_len = z;
  { const char* v = getenv("FMX_UNIT_VALUES"); m->unit_values = !(v && v[0] == '0'); }  // the generator writes 1.0f everywhere
  strata_bounds(m->p, z, &m->field_base);
  return FMX_OK;
}

// SURVEY 8(d)'s other column laws: nnz columns drawn i.i.d. over [0, p) -- uniform (kind 1), or Zipf-like with exponent s (kind 2:
// rank = ((p^(1-s) - 1) u + 1)^(1/(1-s)) - 1, the inverse CDF of the continuous power law on [1, p]; "s = 1.05 for conflict
// stress") -- then sorted inside the row; a repeated column is bumped to the next free id so that rows stay strictly
// ascending (what a dgCMatrix row is).  One thread per row (z <= 64), Philox keyed by (seed; global row, entry / 4).
constexpr int SYNTH_MAX_Z = 64;
__global__ void synth_iid_rows_k(int64_t n, uint32_t p, int32_t z, uint64_t seed, int64_t row_offset, int kind, double s_exp, uint32_t* __restrict__ col,
                                 float* __restrict__ val) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint64_t g = (uint64_t)(row_offset + r);
  uint32_t c[SYNTH_MAX_Z];
  const double a = 1.0 - s_exp, top = pow((double)p, a) - 1.0;
  for (int i = 0; i < z; i += 4) {
    const Philox ph = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(i >> 2), 0x11Du, (uint32_t)seed, (uint32_t)(seed >> 32));
    for (int q = 0; q < 4 && i + q < z; ++q) {
      const double u = (double)ph.c[q] / 4294967296.0;
      double x;
      if (kind == 1) x = u * (double)p;
      else x = pow(top * u + 1.0, 1.0 / a) - 1.0;
      uint32_t id = (uint32_t)x;
      if (id >= p) id = p - 1;
      // insertion into the sorted prefix
      int j = i + q;
      while (j > 0 && c[j - 1] > id) { c[j] = c[j - 1]; --j; }
      c[j] = id;
    }
  }
  // strictly ascending: bump repeats upwards, then pull an overflow at the top back down
  for (int i = 1; i < z; ++i) if (c[i] <= c[i - 1]) c[i] = c[i - 1] + 1;
  if (c[z - 1] >= p) { c[z - 1] = p - 1; for (int i = z - 2; i >= 0 && c[i] >= c[i + 1]; --i) c[i] = c[i + 1] - 1; }
  for (int i = 0; i < z; ++i) { col[r * z + i] = c[i]; val[r * z + i] = 1.0f; }
}

int generate_iid_async(fmx_matrix* m, int64_t n, int32_t z, uint64_t seed, int64_t row_offset, int kind, double s_exp, hipStream_t stream) {
  const int T = 128;
  if (n > 0) hipLaunchKernelGGL(synth_iid_rows_k, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, stream, n, m->p, z, seed, row_offset, kind, s_exp, m->col, m->val);
  hipLaunchKernelGGL(synth_rows_k, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, stream, n, z, seed, row_offset, m->row_ptr, m->y);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

static void drop_value_caches(fmx_matrix* m);
// SURVEY 8(d)'s value variant ("val = 1.0f (variant: U(0,1))"): the stored values of a resident matrix redrawn uniform in (0, 1), entry i of global
// row g from Philox(seed; g, i / 4 | stream 0x7A1)[i % 4] -- keyed like the column generators, so a shard draws what the whole matrix would.  One thread per
// row; the kernels then read the value arrays (util/Smatrix.h:44-61: the reference's values are real floats).
__global__ void synth_values_k(int64_t n, uint64_t seed, int64_t row_offset, const int64_t* __restrict__ row_ptr, float* __restrict__ val) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint64_t g = (uint64_t)(row_offset + r);
  const int64_t b = row_ptr[r], z = row_ptr[r + 1] - b;
  for (int64_t i = 0; i < z; i += 4) {
    const Philox ph = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(i >> 2), 0x7A1u, (uint32_t)seed, (uint32_t)(seed >> 32));
    for (int q = 0; q < 4 && i + q < z; ++q) val[b + i + q] = ((float)(ph.c[q] >> 9) + 0.5f) * (1.0f / 8388608.0f);   // 23 bits + the half: exact in fp32, never 0 or 1
  }
}

int matrix_values_uniform(fmx_matrix* m, uint64_t seed, int64_t row_offset) {
  if (m->n > 0) hipLaunchKernelGGL(synth_values_k, dim3((unsigned)((m->n + 127) / 128)), dim3(128), 0, nullptr, m->n, seed, row_offset, (const int64_t*)m->row_ptr, m->val);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipDeviceSynchronize());
  drop_value_caches(m);
  return check_rows_sorted(m);   // no longer one-hot: unit_values and the field layout go
}

// SURVEY 8(d)'s ragged variant: "nnz/row = Poisson(30) clipped to [1, 64]", columns i.i.d. uniform over [0, p), sorted inside the row (repeats bumped).
// Lengths by inversion of the Poisson CDF on one Philox word keyed (seed; global row): shard independent like every generator here.
__global__ void synth_ragged_len_k(int64_t n, double mean, int lo, int hi, uint64_t seed, int64_t row_offset, int64_t* __restrict__ lens) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  if (r == n) { lens[r] = 0; return; }
  const uint64_t g = (uint64_t)(row_offset + r);
  const Philox ph = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), 0x4A66EDu, 0x11Du, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double u = ((double)ph.c[0] + 0.5) / 4294967296.0;
  int kq = 0;
  double term = exp(-mean), cdf = term;
  while (u > cdf && kq < 4 * hi + 64) { ++kq; term *= mean / (double)kq; cdf += term; }
  lens[r] = kq < lo ? lo : (kq > hi ? hi : kq);
}
__global__ void synth_ragged_rows_k(int64_t n, uint32_t p, uint64_t seed, int64_t row_offset, const int64_t* __restrict__ row_ptr, uint32_t* __restrict__ col,
                                    float* __restrict__ val, float* __restrict__ y) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint64_t g = (uint64_t)(row_offset + r);
  const int64_t b = row_ptr[r];
  const int z = (int)(row_ptr[r + 1] - b);
  uint32_t c[SYNTH_MAX_Z];
  for (int i = 0; i < z; i += 4) {
    const Philox ph = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(i >> 2), 0x11Du, (uint32_t)seed, (uint32_t)(seed >> 32));
    for (int q = 0; q < 4 && i + q < z; ++q) {
      uint32_t id = (uint32_t)(((uint64_t)ph.c[q] * p) >> 32);
      int j = i + q;
      while (j > 0 && c[j - 1] > id) { c[j] = c[j - 1]; --j; }
      c[j] = id;
    }
  }
  for (int i = 1; i < z; ++i) if (c[i] <= c[i - 1]) c[i] = c[i - 1] + 1;
  if (z > 0 && c[z - 1] >= p) { c[z - 1] = p - 1; for (int i = z - 2; i >= 0 && c[i] >= c[i + 1]; --i) c[i] = c[i + 1] - 1; }
  for (int i = 0; i < z; ++i) { col[b + i] = c[i]; val[b + i] = 1.0f; }
  const Philox pl = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), 0xFFFFFFFFu, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  y[r] = (pl.c[0] & 1u) ? 1.0f : -1.0f;
}

int generate_ragged(int device, int64_t n, uint32_t p, double mean, int lo, int hi, uint64_t seed, int64_t row_offset, fmx_matrix** out) {
  *out = nullptr;
  FMX_TRY(use_device_public(device));
  int64_t *d_len = nullptr, *d_ptr = nullptr;
  void* d_tmp = nullptr;
  fmx_matrix* m = nullptr;
  auto body = [&]() -> int {
    FMX_HIP(hipMalloc(&d_len, ((size_t)n + 1) * sizeof(int64_t)));
    FMX_HIP(hipMalloc(&d_ptr, ((size_t)n + 1) * sizeof(int64_t)));
    hipLaunchKernelGGL(synth_ragged_len_k, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, nullptr, n, mean, lo, hi, seed, row_offset, d_len);
    size_t bytes = 0;
    FMX_HIP(rocprim::exclusive_scan(nullptr, bytes, d_len, d_ptr, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), nullptr));
    FMX_HIP(hipMalloc(&d_tmp, bytes ? bytes : 16));
    FMX_HIP(rocprim::exclusive_scan(d_tmp, bytes, d_len, d_ptr, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), nullptr));
    int64_t total = 0;
    FMX_HIP(hipMemcpy(&total, d_ptr + n, sizeof(int64_t), hipMemcpyDeviceToHost));
    FMX_TRY(alloc_matrix_public(device, n, p, total, true, &m));
    FMX_HIP(hipMemcpy(m->row_ptr, d_ptr, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice));
    if (n > 0) hipLaunchKernelGGL(synth_ragged_rows_k, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, nullptr, n, p, seed, row_offset, (const int64_t*)m->row_ptr, m->col, m->val, m->y);
    FMX_HIP(hipGetLastError());
    FMX_HIP(hipDeviceSynchronize());
    m->rows_sorted = 1;
    m->max_row_len = hi;
    m->fixed_row_len = 0;
    { const char* v = getenv("FMX_UNIT_VALUES"); m->unit_values = !(v && v[0] == '0'); }
    return FMX_OK;
  };
  const int st = body();
  (void)hipFree(d_len); (void)hipFree(d_ptr); (void)hipFree(d_tmp);
  if (st != FMX_OK) { free_matrix(m); return st; }
  *out = m;
  return FMX_OK;
}

// Criteo-shaped rows (SURVEY 8(d), configs[3]: "13 dense-ish + 26 categorical"): entry i < n_dense is feature i with a value
// in [0, 1); entry n_dense + f is one feature of categorical field f, whose ids occupy [base[f], base[f] + vocab[f]): the id
// inside the field is floor(vocab * u^skew), u uniform -- skew = 1 is uniform, larger values pile the mass on a field's first
// ids (a power-law head like real click logs: a few values of a field occur in most rows, most values almost never).
// One-hot value 1.  Philox keyed by (seed; global row, entry / 4) like the uniform generator: shard independent.
__global__ void synth_fields_k(int64_t n, FieldSpec fs, uint64_t seed, int64_t row_offset, uint32_t* __restrict__ col, float* __restrict__ val) {
  // one thread per Philox block: entries 4 q .. 4 q + 3 of a row share the block keyed (row, q) -- four threads used to compute it each for one word
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int z = fs.n_dense + fs.n_fields;
  const int gq = (z + 3) / 4;
  if (t >= n * gq) return;
  const int64_t r = t / gq;
  const uint32_t q = (uint32_t)(t - r * gq);
  const uint64_t g = (uint64_t)(row_offset + r);
  const Philox ph = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), q, 0xF1E1D5u, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t i = 4 * q + (uint32_t)u;
    if ((int)i >= z) break;
    const int64_t at = r * z + i;
    const double uu = (double)ph.c[u] / 4294967296.0;
    if ((int)i < fs.n_dense) {
      col[at] = i;
      val[at] = (float)uu;
    } else {
      const int f = (int)i - fs.n_dense;
      // (small integer exponents by multiplication: pow() in fp64 was most of this kernel's 98 us per 262 144-row tile)
      const double x = fs.skew == 1.0 ? uu : fs.skew == 2.0 ? uu * uu : fs.skew == 3.0 ? uu * uu * uu : pow(uu, fs.skew);
      uint32_t id = (uint32_t)(x * (double)fs.vocab[f]);
      if (id >= fs.vocab[f]) id = fs.vocab[f] - 1;
      col[at] = fs.base[f] + id;
      val[at] = 1.0f;
    }
  }
}

// The streamed form: the same rows (the same Philox words), generated FS_ROWS rows at a time into LDS and written out THREE ways from there -- row-major
// (col, val: what phase 1 reads), and the two outputs of fields_split_local_k (the dense columns' lists; the one-hot part field-major with the field's base taken
// off: what the per-field sort reads) -- instead of writing the rows, reading them back and splitting them in a second kernel (82 MB written, 82 MB read and a
// launch per 262 144-row step: 197 us of the ingest stream's 680, profiles/r05_kernel_stats_stream.csv).
__global__ __launch_bounds__(256) void synth_fields_split_k(int64_t n, FieldSpec fs, uint64_t seed, int64_t row_offset, uint32_t* __restrict__ col, float* __restrict__ val,
                                                            uint32_t* __restrict__ keys_sorted, uint32_t* __restrict__ brow, float* __restrict__ bval,
                                                            uint32_t* __restrict__ keys_in) {
  __shared__ uint32_t s_col[FS_ROWS * 64];
  __shared__ float s_val[FS_ROWS * 64];
  const int z = fs.n_dense + fs.n_fields, d = fs.n_dense;
  const int gq = (z + 3) / 4;
  const int64_t R0 = (int64_t)blockIdx.x * FS_ROWS;
  const int rows = (int)(n - R0 < FS_ROWS ? n - R0 : FS_ROWS);
  for (int it = threadIdx.x; it < rows * gq; it += 256) {   // one Philox block = entries 4 q .. 4 q + 3 of a row (as synth_fields_k)
    const int r = it / gq;
    const uint32_t q = (uint32_t)(it - r * gq);
    const uint64_t g = (uint64_t)(row_offset + R0 + r);
    const Philox ph = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), q, 0xF1E1D5u, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = 4 * (int)q + u;
      if (i >= z) break;
      const double uu = (double)ph.c[u] / 4294967296.0;
      if (i < d) { s_col[r * z + i] = (uint32_t)i; s_val[r * z + i] = (float)uu; }
      else {
        const int f = i - d;
        const double x = fs.skew == 1.0 ? uu : fs.skew == 2.0 ? uu * uu : fs.skew == 3.0 ? uu * uu * uu : pow(uu, fs.skew);
        uint32_t id = (uint32_t)(x * (double)fs.vocab[f]);
        if (id >= fs.vocab[f]) id = fs.vocab[f] - 1;
        s_col[r * z + i] = fs.base[f] + id;
        s_val[r * z + i] = 1.0f;
      }
    }
  }
  __syncthreads();
  const int cnt = rows * z;
  for (int i = threadIdx.x; i < cnt; i += 256) { col[R0 * z + i] = s_col[i]; val[R0 * z + i] = s_val[i]; }
  for (int i = threadIdx.x; i < d * rows; i += 256) {   // the dense columns' lists (fields_split_local_k)
    const int c = i / rows, r = i - c * rows;
    const int64_t at = (int64_t)c * n + R0 + r;
    keys_sorted[at] = s_col[r * z + c];
    brow[at] = (uint32_t)(R0 + r);
    bval[at] = s_val[r * z + c];
  }
  const int zc = z - d;
  for (int i = threadIdx.x; i < zc * rows; i += 256) {   // the one-hot part, field-major, local ids
    const int c = i / rows, r = i - c * rows;
    const int64_t at = (int64_t)c * n + R0 + r;
    keys_in[at] = s_col[r * z + d + c] - fs.base[c];
    bval[(int64_t)d * n + at] = 1.0f;
  }
}

int generate_fields_split_async(fmx_matrix* m, int64_t n, const FieldSpec& fs, uint64_t seed, int64_t row_offset, hipStream_t stream, uint32_t* keys_sorted, uint32_t* brow,
                                float* bval, uint32_t* keys_in) {
  const int z = fs.n_dense + fs.n_fields;
  if (n > 0) hipLaunchKernelGGL(synth_fields_split_k, dim3((unsigned)((n + FS_ROWS - 1) / FS_ROWS)), dim3(256), 0, stream, n, fs, seed, row_offset, m->col, m->val, keys_sorted, brow, bval,
                                keys_in);
  hipLaunchKernelGGL(synth_rows_k, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, stream, n, z, seed, row_offset, m->row_ptr, m->y);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

int generate_fields_async(fmx_matrix* m, int64_t n, const FieldSpec& fs, uint64_t seed, int64_t row_offset, hipStream_t stream) {
  const int T = 256;
  const int z = fs.n_dense + fs.n_fields;
  const int64_t total = n * ((z + 3) / 4);
  if (total > 0) hipLaunchKernelGGL(synth_fields_k, dim3((unsigned)((total + T - 1) / T)), dim3(T), 0, stream, n, fs, seed, row_offset, m->col, m->val);
  hipLaunchKernelGGL(synth_rows_k, dim3((unsigned)((n + 1 + T - 1) / T)), dim3(T), 0, stream, n, z, seed, row_offset, m->row_ptr, m->y);
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// ------------------------------------------------------------------------------------------------ parameter tables
// V0 ~ N(mean, stdev) drawn on the device (Philox4x32-10 keyed by (seed; feature, factor / 2), Box-Muller in fp64): for
// synthetic workloads whose V does not fit a host round trip comfortably (p = 33 M, k = 32: 8.4 GB of doubles).  It is NOT the
// reference's generator (Rf_rnorm, util/Dmatrix.h:143-146): parity runs pass V0 through fmx_set_params.
template <typename T>
__global__ void init_normal_k(T* __restrict__ V, uint64_t p, int k, int kp, uint64_t seed, double mean, double stdev) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int half = (k + 1) / 2;
  if (t >= (int64_t)p * half) return;
  const uint64_t j = (uint64_t)(t / half);
  const int f = (int)(t - (int64_t)j * half) * 2;
  const Philox ph = philox4x32_10((uint32_t)j, (uint32_t)(j >> 32), (uint32_t)(f >> 1), 0x56u, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double u1 = ((double)ph.c[0] + 1.0) / 4294967296.0, u2 = (double)ph.c[1] / 4294967296.0;
  const double r = sqrt(-2.0 * log(u1)), a = 6.283185307179586476925 * u2;
  V[j * kp + f] = (T)(mean + stdev * r * cos(a));
  if (f + 1 < k) V[j * kp + f + 1] = (T)(mean + stdev * r * sin(a));
}

int init_normal(fmx_engine* e, uint64_t seed, double mean, double stdev) {
  const int64_t total = (int64_t)e->p * ((e->k + 1) / 2);
  if (total == 0) return FMX_OK;
  const dim3 g((unsigned)((total + 255) / 256)), b(256);
  if (wide_state(e)) hipLaunchKernelGGL((init_normal_k<double>), g, b, 0, e->stream, e->dV, e->p, e->k, e->kp64, seed, mean, stdev);
  else hipLaunchKernelGGL((init_normal_k<float>), g, b, 0, e->stream, e->V, e->p, e->k, e->vstride32, seed, mean, stdev);  // (kp = the row stride)
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// rows of (w, V) picked by feature id <-> a dense [n][k] / [n] pair of double buffers
template <typename T, bool SET>
__global__ void rows_copy_k(T* __restrict__ V, T* __restrict__ w, int k, int kp, int ws, const uint32_t* __restrict__ ids, int64_t n, double* __restrict__ bw,
                            double* __restrict__ bv) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kk = k + 1;
  if (t >= n * kk) return;
  const int64_t i = t / kk;
  const int f = (int)(t - i * kk);
  const size_t j = ids[i];
  if (f == k) { if (SET) w[j * ws] = (T)bw[i]; else bw[i] = (double)w[j * ws]; }
  else if (SET) V[j * kp + f] = (T)bv[i * k + f];
  else bv[i * k + f] = (double)V[j * kp + f];
}

int rows_copy(fmx_engine* e, const uint32_t* d_ids, int64_t n, double* d_w, double* d_v, bool set) {
  const int64_t total = n * (e->k + 1);
  if (total == 0) return FMX_OK;
  const dim3 g((unsigned)((total + 255) / 256)), b(256);
  if (wide_state(e)) {
    if (set) hipLaunchKernelGGL((rows_copy_k<double, true>), g, b, 0, e->stream, e->dV, e->dw, e->k, e->kp64, 1, d_ids, n, d_w, d_v);
    else hipLaunchKernelGGL((rows_copy_k<double, false>), g, b, 0, e->stream, e->dV, e->dw, e->k, e->kp64, 1, d_ids, n, d_w, d_v);
  } else {
    float* wb = (float*)mb_wbase(e);   // (kp = the row stride)
    if (set) hipLaunchKernelGGL((rows_copy_k<float, true>), g, b, 0, e->stream, e->V, wb, e->k, e->vstride32, mb_wstride(e), d_ids, n, d_w, d_v);
    else hipLaunchKernelGGL((rows_copy_k<float, false>), g, b, 0, e->stream, e->V, wb, e->k, e->vstride32, mb_wstride(e), d_ids, n, d_w, d_v);
  }
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// ------------------------------------------------------------------------------------------------ scales / normalize
// Column sums in the reference run over the entries in storage order, i.e. ascending row inside a column: the full CSC
// gives exactly that order, so the sums (hence every scaled float) are bit-identical to util/Smatrix.h:104-111.
__global__ void col_moments_k(const int64_t* __restrict__ col_ptr, const float* __restrict__ cval, uint32_t p, double* __restrict__ sum,
                              double* __restrict__ sumsq) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= (int64_t)p) return;
  double s = 0.0, q = 0.0;
  for (int64_t t = col_ptr[c]; t < col_ptr[c + 1]; ++t) {
    const double v = cval[t];
    s += v;
    q += v * v;
  }
  sum[c] = s;
  sumsq[c] = q;
}

__global__ void scales_finish_k(uint32_t p, int64_t n, const uint8_t* __restrict__ listed, double* __restrict__ mean, double* __restrict__ std) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= (int64_t)p) return;
  if (listed[c]) {  // util/Smatrix.h:116-118
    const double mult_dim = (double)n * ((double)n - 1);
    std[c] = sqrt(std[c] / (double)(n - 1) - mean[c] * mean[c] / mult_dim);
    mean[c] /= (double)n;
  } else {
    std[c] = 1.0;
    mean[c] = 0.0;
  }
}

// (mean, std) of a column travel as ONE 16-byte gather per entry (two 8-byte gathers from two tables were 8 ms for 3e8 entries: the request rate again)
__global__ void pair_up_k(const double* __restrict__ mean, const double* __restrict__ std, uint32_t p, double2* __restrict__ ms) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < p) ms[j] = make_double2(mean[j], std[j]);
}
__global__ void scales_apply_k(int64_t nnz, const uint32_t* __restrict__ col, float* __restrict__ val, const double2* __restrict__ ms) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nnz) return;
  const double2 c = ms[col[t]];
  float v = val[t];
  v = (float)((double)v - c.x);            // value[p] -= colSum[idx]        (:127)
  v = (float)((double)v / (c.y + 1e-30));  // value[p] /= (colSumSqr + 1e-30) (:128)
  val[t] = v;
}

__global__ void normalize_apply_k(int64_t nnz, const uint32_t* __restrict__ col, float* __restrict__ val, const double2* __restrict__ ms) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nnz) return;
  const double2 c = ms[col[t]];
  if (c.y != 0) val[t] = (float)(((double)val[t] - c.x) / c.y);  // util/Smatrix.h:148-150
}

// the cached inverted indices hold copies of the values: drop them so they are rebuilt from the new values
static void drop_value_caches(fmx_matrix* m) {
  drop_plans(m);
  m->value_generation++;
  (void)hipFree(m->col_ptr); (void)hipFree(m->crow); (void)hipFree(m->cval);
  m->col_ptr = nullptr; m->crow = nullptr; m->cval = nullptr;
  als_tiled_free(m); m->als_tiled_tried = 0;   // (its lists hold values too)
}

int matrix_scales(fmx_matrix* m, const uint8_t* h_listed, double* h_mean, double* h_std) {
  FMX_CHECK(m->n >= 2, FMX_ERR_INVALID, "scales needs at least two rows");
  FMX_TRY(build_full_csc(m, nullptr));
  const uint32_t p = m->p;
  double *d_mean = nullptr, *d_std = nullptr;
  uint8_t* d_listed = nullptr;
  FMX_HIP(hipMalloc(&d_mean, (size_t)p * sizeof(double)));
  FMX_HIP(hipMalloc(&d_std, (size_t)p * sizeof(double)));
  FMX_HIP(hipMalloc(&d_listed, (size_t)p));
  FMX_HIP(hipMemcpy(d_listed, h_listed, (size_t)p, hipMemcpyHostToDevice));
  const unsigned gp = (unsigned)(((int64_t)p + 255) / 256);
  hipLaunchKernelGGL(col_moments_k, dim3(gp), dim3(256), 0, nullptr, m->col_ptr, m->cval, p, d_mean, d_std);
  hipLaunchKernelGGL(scales_finish_k, dim3(gp), dim3(256), 0, nullptr, p, m->n, d_listed, d_mean, d_std);
  double2* d_ms = nullptr;
  FMX_HIP(hipMalloc(&d_ms, (size_t)p * sizeof(double2)));
  hipLaunchKernelGGL(pair_up_k, dim3(gp), dim3(256), 0, nullptr, (const double*)d_mean, (const double*)d_std, p, d_ms);
  if (m->nnz > 0) hipLaunchKernelGGL(scales_apply_k, dim3((unsigned)((m->nnz + 255) / 256)), dim3(256), 0, nullptr, m->nnz, m->col, m->val, (const double2*)d_ms);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipMemcpy(h_mean, d_mean, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
  FMX_HIP(hipMemcpy(h_std, d_std, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
  (void)hipFree(d_mean); (void)hipFree(d_std); (void)hipFree(d_listed); (void)hipFree(d_ms);
  drop_value_caches(m);
  return check_rows_sorted(m);  // the values changed: is the matrix still one-hot?
}

int matrix_normalize(fmx_matrix* m, const double* h_mean, const double* h_std) {
  const uint32_t p = m->p;
  double *d_mean = nullptr, *d_std = nullptr;
  FMX_HIP(hipMalloc(&d_mean, (size_t)p * sizeof(double)));
  FMX_HIP(hipMalloc(&d_std, (size_t)p * sizeof(double)));
  FMX_HIP(hipMemcpy(d_mean, h_mean, (size_t)p * sizeof(double), hipMemcpyHostToDevice));
  FMX_HIP(hipMemcpy(d_std, h_std, (size_t)p * sizeof(double), hipMemcpyHostToDevice));
  double2* d_ms = nullptr;
  FMX_HIP(hipMalloc(&d_ms, (size_t)p * sizeof(double2)));
  hipLaunchKernelGGL(pair_up_k, dim3((unsigned)(((int64_t)p + 255) / 256)), dim3(256), 0, nullptr, (const double*)d_mean, (const double*)d_std, p, d_ms);
  if (m->nnz > 0) hipLaunchKernelGGL(normalize_apply_k, dim3((unsigned)((m->nnz + 255) / 256)), dim3(256), 0, nullptr, m->nnz, m->col, m->val, (const double2*)d_ms);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipDeviceSynchronize());
  (void)hipFree(d_mean); (void)hipFree(d_std); (void)hipFree(d_ms);
  drop_value_caches(m);
  return check_rows_sorted(m);
}

// ------------------------------------------------------------------------------------------------ host hand-over
// fm.matrix's arrays live in pageable host memory (R vectors): value f64, col_idx i32, row_size i32, labels f64.  They are copied as they are --
// 64 MB pieces, a few host threads filling one pinned buffer while the other is on the wire and the piece before it is being converted -- and
// narrowed, range-checked and prefix-summed ON THE DEVICE (util/Smatrix.h:53-60 does this in a host loop; so did round 1-3's first version:
// 0.96 s for the 3.7 GB of the 10 M x 1 M matrix, 3.9 GB/s, thirty-five times the epoch that follows).
namespace {
struct Stager {
  size_t PIECE = 64u << 20;   // (smaller for small matrices: two pinned and two device buffers of this size are made per hand-over)
  void* pin[2] = {nullptr, nullptr};
  void* dev[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  hipStream_t st = nullptr;
  int threads = 1;
  bool used[2] = {false, false};
  size_t cap = 0;             // bytes of each buffer
  int open(size_t largest_array_bytes) {
    PIECE = 64u << 20;
    while (PIECE > (1u << 20) && PIECE / 2 >= largest_array_bytes) PIECE /= 2;
    used[0] = used[1] = false;
    if (cap >= PIECE) return FMX_OK;   // (a kept stager: buffers, events and stream are there)
    release();
    unsigned hc = std::thread::hardware_concurrency();
    threads = hc >= 16 ? 8 : (hc >= 4 ? 4 : 1);
    FMX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      FMX_HIP(hipHostMalloc(&pin[i], PIECE, hipHostMallocDefault));
      FMX_HIP(hipMalloc(&dev[i], PIECE));
      FMX_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    }
    cap = PIECE;
    return FMX_OK;
  }
  void release() {
    if (st) (void)hipStreamSynchronize(st);
    for (int i = 0; i < 2; ++i) {
      if (pin[i]) (void)hipHostFree(pin[i]);
      (void)hipFree(dev[i]);
      if (done[i]) (void)hipEventDestroy(done[i]);
      pin[i] = dev[i] = nullptr; done[i] = nullptr;
    }
    if (st) (void)hipStreamDestroy(st);
    st = nullptr; cap = 0;
  }
  // `count` elements of `elem` bytes from host memory; consume(device piece, index of its first element, elements in it) enqueues on st
  template <typename F>
  int run(const void* host, size_t elem, int64_t count, F consume) {
    const int64_t per = (int64_t)(PIECE / elem);
    int64_t piece = 0;
    for (int64_t i0 = 0; i0 < count; i0 += per, ++piece) {
      const int b = (int)(piece & 1);
      const int64_t n = count - i0 < per ? count - i0 : per;
      if (used[b]) FMX_HIP(hipEventSynchronize(done[b]));   // the piece that last went through this buffer has been consumed
      const char* src = (const char*)host + (size_t)i0 * elem;
      const size_t bytes = (size_t)n * elem;
      if (threads > 1 && bytes >= (4u << 20)) {
        std::vector<std::thread> pool;
        const size_t slice = (bytes / (size_t)threads + 4095) & ~(size_t)4095;
        for (int t = 0; t < threads; ++t) {
          const size_t a = (size_t)t * slice;
          if (a >= bytes) break;
          const size_t len = a + slice < bytes ? slice : bytes - a;
          try { pool.emplace_back([=] { memcpy((char*)pin[b] + a, src + a, len); }); }
          catch (...) { memcpy((char*)pin[b] + a, src + a, len); }   // (no thread to be had: this slice is copied here)
        }
        for (auto& th : pool) th.join();
      } else {
        memcpy(pin[b], src, bytes);
      }
      FMX_HIP(hipMemcpyAsync(dev[b], pin[b], bytes, hipMemcpyHostToDevice, st));
      FMX_TRY(consume(dev[b], i0, n));
      FMX_HIP(hipEventRecord(done[b], st));
      used[b] = true;
    }
    return FMX_OK;
  }
  // the other way: produce(device piece, first element, elements) enqueues on st; the piece then comes down and is copied into `host`
  template <typename F>
  int run_down(void* host, size_t elem, int64_t count, F produce) {
    const int64_t per = (int64_t)(PIECE / elem);
    int64_t prev_i0 = 0, prev_n = 0;
    int prev_b = -1;
    auto drain = [&]() -> int {   // the previous piece: wait for its copy, hand it to the caller's array
      if (prev_b < 0) return FMX_OK;
      FMX_HIP(hipEventSynchronize(done[prev_b]));
      const size_t bytes = (size_t)prev_n * elem;
      char* dst = (char*)host + (size_t)prev_i0 * elem;
      if (threads > 1 && bytes >= (4u << 20)) {
        std::vector<std::thread> pool;
        const size_t slice = (bytes / (size_t)threads + 4095) & ~(size_t)4095;
        const char* src = (const char*)pin[prev_b];
        for (int t = 0; t < threads; ++t) {
          const size_t a = (size_t)t * slice;
          if (a >= bytes) break;
          const size_t len = a + slice < bytes ? slice : bytes - a;
          try { pool.emplace_back([=] { memcpy(dst + a, src + a, len); }); }
          catch (...) { memcpy(dst + a, src + a, len); }
        }
        for (auto& th : pool) th.join();
      } else {
        memcpy(dst, pin[prev_b], bytes);
      }
      return FMX_OK;
    };
    int64_t piece = 0;
    for (int64_t i0 = 0; i0 < count; i0 += per, ++piece) {
      const int b = (int)(piece & 1);
      const int64_t n = count - i0 < per ? count - i0 : per;
      FMX_TRY(produce(dev[b], i0, n));                       // (buffer b was drained two pieces ago)
      FMX_HIP(hipMemcpyAsync(pin[b], dev[b], (size_t)n * elem, hipMemcpyDeviceToHost, st));
      FMX_HIP(hipEventRecord(done[b], st));
      FMX_TRY(drain());                                      // while this piece is produced and copied
      prev_b = b; prev_i0 = i0; prev_n = n;
    }
    return drain();
  }
};
// One stager per device is KEPT between calls (two pinned and two device buffers of up to 64 MB, a stream): making them costs 20-30 ms, more than
// the hand-over of a model's parameters takes.  Never destroyed (the HIP runtime may be gone before a static's destructor runs).
struct StagerLease {
  Stager* s = nullptr;
  bool kept = false;
  explicit StagerLease(int device) {
    static std::mutex mu;
    static Stager* keep[64] = {};
    static bool busy[64] = {};
    std::lock_guard<std::mutex> g(mu);
    if (device >= 0 && device < 64 && !busy[device]) {
      if (!keep[device]) keep[device] = new Stager();
      s = keep[device]; kept = true; busy[device] = true; dev_ = device; busy_ = busy;
    } else {
      s = new Stager();
    }
  }
  ~StagerLease() {
    if (kept) { if (s->st) (void)hipStreamSynchronize(s->st); busy_[dev_] = false; }
    else { s->release(); delete s; }
  }
  int dev_ = 0;
  bool* busy_ = nullptr;
};
}  // namespace

__global__ void narrow_values_k(const double* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}
template <typename IN>
__global__ void check_cols_k(const IN* __restrict__ in, uint32_t* __restrict__ out, int64_t n, uint32_t p, int64_t first, unsigned long long* __restrict__ first_bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const IN c = in[i];
  if (c < (IN)0 || (uint64_t)c >= (uint64_t)p) atomicMin(first_bad, (unsigned long long)(first + i));
  out[i] = (uint32_t)c;
}
__global__ void copy_words_k(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void sizes_to_i64_k(const int32_t* __restrict__ in, int64_t* __restrict__ out, int64_t n, int64_t first, unsigned long long* __restrict__ first_bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (in[i] < 0) atomicMin(first_bad, (unsigned long long)(first + i));
  out[i] = (int64_t)in[i];
}
__global__ void rowptr_check_k(const int64_t* __restrict__ rp, int64_t n, unsigned long long* __restrict__ first_bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && rp[i + 1] < rp[i]) atomicMin(first_bad, (unsigned long long)i);
}

// (w0, w, V) in R's layout -- V a k x p column-major matrix of doubles, i.e. p rows of k -- <-> the engine's tables (rows of `stride` elements)
template <typename ST>
__global__ void rows_from_f64_k(const double* __restrict__ in, ST* __restrict__ table, int k, int64_t stride, int64_t j0, int64_t n_elems) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_elems) return;
  const int64_t j = i / k;
  table[(size_t)(j0 + j) * stride + (i - j * k)] = (ST)in[i];
}
template <typename ST>
__global__ void rows_to_f64_k(double* __restrict__ out, const ST* __restrict__ table, int k, int64_t stride, int64_t j0, int64_t n_elems) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_elems) return;
  const int64_t j = i / k;
  out[i] = (double)table[(size_t)(j0 + j) * stride + (i - j * k)];
}

// v (may be null: zeros) and w (may be null: zeros) into the tables; the padding of the rows is zeroed
int params_to_device(fmx_engine* e, const double* w, const double* v) {
  const int64_t p = (int64_t)e->p;
  const int k = e->k;
  const bool wide = wide_state(e);
  void* V = wide ? (void*)e->dV : (void*)e->V;
  const int64_t vs = wide ? e->kp64 : e->vstride32;
  const size_t eb = wide ? 8 : 4;
  FMX_HIP(hipMemset(V, 0, (size_t)p * vs * eb));   // (the w slot of a w-in-row table included)
  void* wt = wide ? (void*)e->dw : mb_wbase(e);
  const int64_t ws = wide ? 1 : mb_wstride(e);
  if (!(!wide && e->w_in_row)) FMX_HIP(hipMemset(wt, 0, (size_t)p * eb));
  FMX_HIP(hipDeviceSynchronize());
  if (!w && !(v && k > 0)) return FMX_OK;
  StagerLease lease(e->cfg.device);
  Stager& S = *lease.s;
  FMX_TRY(S.open((size_t)p * (size_t)(k > 1 ? k : 1) * 8));
  const int T = 256;
  auto grid = [&](int64_t n) { return dim3((unsigned)((n + T - 1) / T)); };
  hipStream_t st = S.st;
  if (v && k > 0) {
    const int64_t rows_per = (int64_t)(S.PIECE / 8) / k;   // whole rows per piece
    S.PIECE = (size_t)rows_per * k * 8;
    FMX_TRY(S.run(v, 8, p * k, [&](void* d, int64_t i0, int64_t n) {
      if (wide) hipLaunchKernelGGL((rows_from_f64_k<double>), grid(n), dim3(T), 0, st, (const double*)d, (double*)V, k, vs, i0 / k, n);
      else hipLaunchKernelGGL((rows_from_f64_k<float>), grid(n), dim3(T), 0, st, (const double*)d, (float*)V, k, vs, i0 / k, n);
      return FMX_OK; }));
  }
  if (w) {
    FMX_TRY(S.run(w, 8, p, [&](void* d, int64_t i0, int64_t n) {
      if (wide) hipLaunchKernelGGL((rows_from_f64_k<double>), grid(n), dim3(T), 0, st, (const double*)d, (double*)wt, 1, ws, i0, n);
      else hipLaunchKernelGGL((rows_from_f64_k<float>), grid(n), dim3(T), 0, st, (const double*)d, (float*)wt, 1, ws, i0, n);
      return FMX_OK; }));
  }
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipStreamSynchronize(st));
  return FMX_OK;
}

int params_from_device(fmx_engine* e, double* w, double* v) {
  const int64_t p = (int64_t)e->p;
  const int k = e->k;
  if (!w && !(v && k > 0)) return FMX_OK;
  const bool wide = wide_state(e);
  const void* V = wide ? (const void*)e->dV : (const void*)e->V;
  const int64_t vs = wide ? e->kp64 : e->vstride32;
  const void* wt = wide ? (const void*)e->dw : (const void*)mb_wbase(e);
  const int64_t ws = wide ? 1 : mb_wstride(e);
  StagerLease lease(e->cfg.device);
  Stager& S = *lease.s;
  FMX_TRY(S.open((size_t)p * (size_t)(k > 1 ? k : 1) * 8));
  const int T = 256;
  auto grid = [&](int64_t n) { return dim3((unsigned)((n + T - 1) / T)); };
  hipStream_t st = S.st;
  if (v && k > 0) {
    const int64_t rows_per = (int64_t)(S.PIECE / 8) / k;
    S.PIECE = (size_t)rows_per * k * 8;
    FMX_TRY(S.run_down(v, 8, p * k, [&](void* d, int64_t i0, int64_t n) {
      if (wide) hipLaunchKernelGGL((rows_to_f64_k<double>), grid(n), dim3(T), 0, st, (double*)d, (const double*)V, k, vs, i0 / k, n);
      else hipLaunchKernelGGL((rows_to_f64_k<float>), grid(n), dim3(T), 0, st, (double*)d, (const float*)V, k, vs, i0 / k, n);
      return FMX_OK; }));
  }
  if (w) {
    FMX_TRY(S.run_down(w, 8, p, [&](void* d, int64_t i0, int64_t n) {
      if (wide) hipLaunchKernelGGL((rows_to_f64_k<double>), grid(n), dim3(T), 0, st, (double*)d, (const double*)wt, 1, ws, i0, n);
      else hipLaunchKernelGGL((rows_to_f64_k<float>), grid(n), dim3(T), 0, st, (double*)d, (const float*)wt, 1, ws, i0, n);
      return FMX_OK; }));
  }
  FMX_HIP(hipGetLastError());
  return FMX_OK;
}

// the device arrays of m from host arrays: values f64 or f32, columns i32 or u32, row sizes (i32) or row offsets (i64), labels f64 or f32.
// bad[0] = first column out of range, bad[1] = first negative row size / decreasing offset, total = what the offsets end in
int ingest_host_arrays(fmx_matrix* m, const void* values, bool values_f64, const void* cols, bool cols_signed, const int32_t* row_size, const int64_t* row_ptr,
                       const void* labels, bool labels_f64, uint64_t bad[2], int64_t* total) {
  StagerLease lease(m->device);
  Stager& S = *lease.s;
  FMX_HIP(hipSetDevice(m->device));
  FMX_HIP(hipDeviceSynchronize());   // (the matrix was allocated and zeroed through the null stream; the pieces below run on a stream of their own)
  FMX_TRY(S.open((size_t)(m->nnz > m->n ? m->nnz : m->n + 1) * 8));
  unsigned long long* d_bad = nullptr;
  FMX_HIP(hipMalloc(&d_bad, 2 * sizeof(unsigned long long)));
  struct Free { void* p; ~Free() { (void)hipFree(p); } } free_bad{d_bad};
  FMX_HIP(hipMemsetAsync(d_bad, 0xFF, 2 * sizeof(unsigned long long), S.st));
  const int T = 256;
  auto grid = [&](int64_t n) { return dim3((unsigned)((n + T - 1) / T)); };
  if (m->nnz > 0) {
    float* val = m->val; uint32_t* col = m->col; const uint32_t p = m->p;
    hipStream_t st = S.st;
    if (values_f64) FMX_TRY(S.run(values, 8, m->nnz, [&](void* d, int64_t i0, int64_t n) { hipLaunchKernelGGL(narrow_values_k, grid(n), dim3(T), 0, st, (const double*)d, val + i0, n); return FMX_OK; }));
    else FMX_TRY(S.run(values, 4, m->nnz, [&](void* d, int64_t i0, int64_t n) { hipLaunchKernelGGL(copy_words_k, grid(n), dim3(T), 0, st, (const uint32_t*)d, (uint32_t*)val + i0, n); return FMX_OK; }));
    if (cols_signed) FMX_TRY(S.run(cols, 4, m->nnz, [&](void* d, int64_t i0, int64_t n) { hipLaunchKernelGGL((check_cols_k<int32_t>), grid(n), dim3(T), 0, st, (const int32_t*)d, col + i0, n, p, i0, d_bad); return FMX_OK; }));
    else FMX_TRY(S.run(cols, 4, m->nnz, [&](void* d, int64_t i0, int64_t n) { hipLaunchKernelGGL((check_cols_k<uint32_t>), grid(n), dim3(T), 0, st, (const uint32_t*)d, col + i0, n, p, i0, d_bad); return FMX_OK; }));
  }
  if (row_size) {   // sizes -> offsets: an exclusive scan in place over [0, n], entry n = the total
    int64_t* rp = m->row_ptr;
    hipStream_t st = S.st;
    FMX_HIP(hipMemsetAsync(rp + m->n, 0, sizeof(int64_t), st));
    if (m->n > 0) FMX_TRY(S.run(row_size, 4, m->n, [&](void* d, int64_t i0, int64_t n) { hipLaunchKernelGGL(sizes_to_i64_k, grid(n), dim3(T), 0, st, (const int32_t*)d, rp + i0, n, i0, d_bad + 1); return FMX_OK; }));
    size_t tb = 0;
    FMX_HIP(rocprim::exclusive_scan(nullptr, tb, rp, rp, (int64_t)0, (size_t)m->n + 1, rocprim::plus<int64_t>(), st));
    void* tmp = nullptr;
    FMX_HIP(hipMalloc(&tmp, tb ? tb : 16));
    Free free_tmp{tmp};
    FMX_HIP(rocprim::exclusive_scan(tmp, tb, rp, rp, (int64_t)0, (size_t)m->n + 1, rocprim::plus<int64_t>(), st));
    FMX_HIP(hipStreamSynchronize(st));
  } else {
    int64_t* rp = m->row_ptr;
    hipStream_t st = S.st;
    FMX_TRY(S.run(row_ptr, 8, m->n + 1, [&](void* d, int64_t i0, int64_t n) { hipLaunchKernelGGL(copy_words_k, grid(2 * n), dim3(T), 0, st, (const uint32_t*)d, (uint32_t*)(rp + i0), 2 * n); return FMX_OK; }));
    if (m->n > 0) hipLaunchKernelGGL(rowptr_check_k, grid(m->n), dim3(T), 0, st, (const int64_t*)rp, m->n, d_bad + 1);
  }
  if (labels && m->n > 0) {
    float* y = m->y;
    hipStream_t st = S.st;
    if (labels_f64) FMX_TRY(S.run(labels, 8, m->n, [&](void* d, int64_t i0, int64_t n) { hipLaunchKernelGGL(narrow_values_k, grid(n), dim3(T), 0, st, (const double*)d, y + i0, n); return FMX_OK; }));
    else FMX_TRY(S.run(labels, 4, m->n, [&](void* d, int64_t i0, int64_t n) { hipLaunchKernelGGL(copy_words_k, grid(n), dim3(T), 0, st, (const uint32_t*)d, (uint32_t*)y + i0, n); return FMX_OK; }));
  }
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipStreamSynchronize(S.st));
  unsigned long long h_bad[2];
  FMX_HIP(hipMemcpy(h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost));
  bad[0] = h_bad[0]; bad[1] = h_bad[1];
  FMX_HIP(hipMemcpy(total, m->row_ptr + m->n, sizeof(int64_t), hipMemcpyDeviceToHost));
  return FMX_OK;
}

// ------------------------------------------------------------------------------------------------ sortedness
// Are the rows strictly ascending, are all values 1, how long are the rows?  One thread per ROW walking its entries was 12 ms for the 10 M x 30
// matrix (every lane on its own 120-byte stride) -- of every hand-over, of every normalize.  Now the entries are read once, coalesced: a descent
// col[t] >= col[t + 1] is counted wherever it occurs, the descents that are merely the seam between two rows are counted from the row offsets,
// and the rows are sorted exactly when the two counts agree.
constexpr int SCAN_SLOTS = 256;   // counters a scan spreads its atomics over (summed on the host)
constexpr int ES_PER = 16;   // entries per thread: one atomic per wave and 1024 entries (every wave sees a row seam: one atomic per 64 entries on ONE address took 90 ms)
__global__ __launch_bounds__(256) void entries_scan_k(const uint32_t* __restrict__ col, const float* __restrict__ val, int64_t nnz, unsigned long long* __restrict__ descents,
                                                      int* __restrict__ out) {
  const int64_t base = (int64_t)blockIdx.x * (256 * ES_PER) + threadIdx.x;
  unsigned down = 0;
  bool other = false;
#pragma unroll
  for (int i = 0; i < ES_PER; ++i) {
    const int64_t t = base + (int64_t)i * 256;
    if (t < nnz) {
      other |= val[t] != 1.0f;
      if (t + 1 < nnz) down += col[t] >= col[t + 1];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) down += __shfl_xor(down, off);
  // (one address for every wave is a queue, not a counter: 293 K stores of `1` to out[2] after a normalize took 12 ms.  The counts go to one of
  // SCAN_SLOTS slots, the flag is written only while it still reads 0)
  if ((threadIdx.x & 63) == 0 && down) atomicAdd(descents + (blockIdx.x & (SCAN_SLOTS - 1)), (unsigned long long)down);
  if (__ballot(other) && (threadIdx.x & 63) == 0 && __atomic_load_n(out + 2, __ATOMIC_RELAXED) == 0) out[2] = 1;
}
__global__ __launch_bounds__(256) void rows_scan_k(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, int64_t n, int64_t nnz,
                                                   unsigned long long* __restrict__ seams, int* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool seam_down = false;
  int longest = 0, shortest_c = 0;   // (the shortest row as a maximum of the complement)
  if (r < n) {
    const int64_t a = row_ptr[r], b = row_ptr[r + 1];
    const int len = (int)(b - a > 0x7fffffff ? 0x7fffffff : b - a);
    longest = len; shortest_c = 0x7fffffff - len;
    if (b > a && b < nnz) seam_down = col[b - 1] >= col[b];   // this row's last entry against the next stored entry (the next non-empty row's first)
  }
  const unsigned long long d = __ballot(seam_down);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int a2 = __shfl_xor(longest, off), b2 = __shfl_xor(shortest_c, off);
    longest = a2 > longest ? a2 : longest;
    shortest_c = b2 > shortest_c ? b2 : shortest_c;
  }
  if ((threadIdx.x & 63) == 0) {
    if (d) atomicAdd(seams + (blockIdx.x & (SCAN_SLOTS - 1)), (unsigned long long)__builtin_popcountll(d));
    if (longest > __atomic_load_n(out + 1, __ATOMIC_RELAXED)) atomicMax(out + 1, longest);
    if (shortest_c > __atomic_load_n(out + 3, __ATOMIC_RELAXED)) atomicMax(out + 3, shortest_c);
  }
}

// Rows of one length z: the smallest and largest column and "some value is not 1" per POSITION in the row (z <= 64).  Positions whose ranges are
// disjoint and ascending are fields (check_rows_sorted below).  Reduced per block in LDS; a global atomic only where it still improves the value.
__global__ __launch_bounds__(256) void positions_scan_k(const uint32_t* __restrict__ col, const float* __restrict__ val, int64_t nnz, int z,
                                                        uint32_t* __restrict__ pos_min, uint32_t* __restrict__ pos_max, uint32_t* __restrict__ pos_other) {
  __shared__ uint32_t lo[64], hi[64], ot[64];
  if (threadIdx.x < 64) { lo[threadIdx.x] = 0xFFFFFFFFu; hi[threadIdx.x] = 0u; ot[threadIdx.x] = 0u; }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (256 * ES_PER) + threadIdx.x;
#pragma unroll
  for (int i = 0; i < ES_PER; ++i) {
    const int64_t t = base + (int64_t)i * 256;
    if (t < nnz) {
      const int pos = (int)(t % z);
      const uint32_t c = col[t];
      if (c < lo[pos]) atomicMin(&lo[pos], c);
      if (c > hi[pos]) atomicMax(&hi[pos], c);
      if (val[t] != 1.0f && !ot[pos]) ot[pos] = 1u;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < z) {
    const int i = threadIdx.x;
    if (lo[i] < __atomic_load_n(pos_min + i, __ATOMIC_RELAXED)) atomicMin(pos_min + i, lo[i]);
    if (hi[i] > __atomic_load_n(pos_max + i, __ATOMIC_RELAXED)) atomicMax(pos_max + i, hi[i]);
    if (ot[i] && !__atomic_load_n(pos_other + i, __ATOMIC_RELAXED)) pos_other[i] = 1u;
  }
}

// rows of one length whose entry positions have disjoint ascending column ranges ARE field-structured (fmx_matrix::field_base): the first d positions
// that always hold column i are the dense prefix, every other position is a field and must hold value 1.  Found here for any uploaded matrix (a one-hot
// encoded data frame, user / item ids ...), so that the per-field plan builder needs no hint; fmx_matrix_set_fields remains for callers who know the
// vocabularies (its ranges may be wider than the ids that occur).  FMX_DETECT_FIELDS=0 switches the detection off.
static int detect_fields(fmx_matrix* m) {
  static const bool on = [] { const char* v = getenv("FMX_DETECT_FIELDS"); return !(v && v[0] == '0'); }();
  const int z = m->fixed_row_len;
  if (!on || !m->rows_sorted || z < 1 || z > 64 || m->nnz != m->n * (int64_t)z || m->nnz == 0) return FMX_OK;
  uint32_t* d = nullptr;
  FMX_HIP(hipMalloc(&d, 3 * 64 * sizeof(uint32_t)));
  struct Free { void* p; ~Free() { (void)hipFree(p); } } fr{d};
  FMX_HIP(hipMemset(d, 0xFF, 64 * sizeof(uint32_t)));
  FMX_HIP(hipMemset(d + 64, 0, 2 * 64 * sizeof(uint32_t)));
  hipLaunchKernelGGL(positions_scan_k, dim3((unsigned)((m->nnz + 256 * ES_PER - 1) / (256 * ES_PER))), dim3(256), 0, nullptr, m->col, m->val, m->nnz, z, d, d + 64, d + 128);
  uint32_t h[3 * 64];
  FMX_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  const uint32_t *lo = h, *hi = h + 64, *other = h + 128;
  int dn = 0;
  while (dn < z && lo[dn] == (uint32_t)dn && hi[dn] == (uint32_t)dn) ++dn;   // always-present columns 0 .. dn-1 (any values)
  // (a leading position that always holds column i WITH value 1 may as well be a one-value field: keep the matrix one-hot if it is)
  if (m->unit_values) dn = 0;
  const int C = z - dn;
  if (C < 1 || C > FMX_MAX_FIELDS) return FMX_OK;
  for (int c = dn; c < z; ++c) {
    if (other[c]) return FMX_OK;                                 // a field entry with a value other than 1
    if (c > dn && lo[c] <= hi[c - 1]) return FMX_OK;             // ranges overlap: positions are not fields
  }
  if (lo[dn] < (uint32_t)dn) return FMX_OK;
  std::vector<uint32_t> base((size_t)C + 1);
  base[0] = (uint32_t)dn;
  for (int c = 1; c < C; ++c) base[(size_t)c] = lo[dn + c];
  base[(size_t)C] = m->p;
  if (hi[z - 1] >= m->p) return FMX_OK;
  if (dn > 0 && m->unit_values) return FMX_OK;
  m->dense_prefix = dn;
  m->field_base = base;
  return FMX_OK;
}

// Does every row read [columns 0 .. d-1 | one id of field c in [base[c], base[c + 1]) for c = 0 .. C-1, value 1]?
struct FieldCheck { int d, C; uint32_t base[FMX_MAX_FIELDS + 1]; };
__global__ void rows_fields_k(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, const float* __restrict__ val, int64_t n, FieldCheck fc,
                              int* __restrict__ bad) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t a = row_ptr[r];
  const int z = fc.d + fc.C;
  int wrong = row_ptr[r + 1] - a != z;
  for (int i = 0; i < z && !wrong; ++i) {
    const uint32_t c = col[a + i];
    if (i < fc.d) wrong |= c != (uint32_t)i;
    else wrong |= c < fc.base[i - fc.d] || c >= fc.base[i - fc.d + 1] || val[a + i] != 1.0f;
  }
  if (wrong) *bad = 1;
}

// the caller vouches for a field layout (a one-hot encoded data frame: every factor column is a contiguous range of dummy columns); checked here
int matrix_set_fields(fmx_matrix* m, int n_dense, int n_fields, const uint32_t* base) {
  FieldCheck fc{};
  fc.d = n_dense; fc.C = n_fields;
  for (int c = 0; c <= n_fields; ++c) fc.base[c] = base[c];
  FMX_CHECK(base[0] == (uint32_t)n_dense && base[n_fields] == m->p, FMX_ERR_INVALID, "field_base must start at n_dense (%d) and end at the feature count (%u)", n_dense, m->p);
  for (int c = 0; c < n_fields; ++c) FMX_CHECK(base[c] < base[c + 1], FMX_ERR_INVALID, "field %d is empty or out of order", c);
  int* d = nullptr;
  int h = 0;
  FMX_HIP(hipSetDevice(m->device));
  FMX_HIP(hipMalloc(&d, sizeof(int)));
  FMX_HIP(hipMemset(d, 0, sizeof(int)));
  if (m->n > 0) hipLaunchKernelGGL(rows_fields_k, dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, nullptr, m->row_ptr, m->col, m->val, m->n, fc, d);
  FMX_HIP(hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost));
  FMX_HIP(hipFree(d));
  FMX_CHECK(!h, FMX_ERR_INVALID, "the rows do not have this layout: every row must hold the %d dense columns 0..%d, then exactly one id of every field in its range, with value 1",
            n_dense, n_dense - 1);
  drop_plans(m);   // plans built on the general path stay valid, but the point of the call is the other builder
  m->value_generation++;   // (shards cut from this matrix before the call do not know the layout)
  m->dense_prefix = n_dense;
  m->fixed_row_len = n_dense + n_fields;
  m->field_base.assign(base, base + n_fields + 1);
  return FMX_OK;
}

int check_rows_sorted(fmx_matrix* m) {
  int* d = nullptr;
  int h[4] = {0, 0, 0, 0};
  std::vector<unsigned long long> slots(2 * SCAN_SLOTS, 0);
  const size_t bytes = 4 * sizeof(int) + 2 * SCAN_SLOTS * sizeof(unsigned long long);
  FMX_HIP(hipMalloc(&d, bytes));
  FMX_HIP(hipMemset(d, 0, bytes));
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(d + 4);
  if (m->nnz > 0) hipLaunchKernelGGL(entries_scan_k, dim3((unsigned)((m->nnz + 256 * ES_PER - 1) / (256 * ES_PER))), dim3(256), 0, nullptr, m->col, m->val, m->nnz, cnt, d);
  if (m->n > 0) hipLaunchKernelGGL(rows_scan_k, dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, nullptr, m->row_ptr, m->col, m->n, m->nnz, cnt + SCAN_SLOTS, d);
  FMX_HIP(hipMemcpy(h, d, 4 * sizeof(int), hipMemcpyDeviceToHost));
  FMX_HIP(hipMemcpy(slots.data(), cnt, 2 * SCAN_SLOTS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  FMX_HIP(hipFree(d));
  unsigned long long h_cnt[2] = {0, 0};
  for (int i = 0; i < SCAN_SLOTS; ++i) { h_cnt[0] += slots[(size_t)i]; h_cnt[1] += slots[(size_t)SCAN_SLOTS + i]; }
  h[0] = h_cnt[0] != h_cnt[1];   // a descent inside a row
  m->rows_sorted = !h[0];
  m->max_row_len = h[1];
  m->dense_prefix = 0;  // (only the field generator vouches for it; changed values may have broken it)
  m->field_base.clear();
  // FMX_UNIT_VALUES=0 in the environment keeps the general path (tuning / A-B runs only)
  static const bool allow = [] { const char* v = getenv("FMX_UNIT_VALUES"); return !(v && v[0] == '0'); }();
  m->unit_values = (allow && !h[2]) ? 1 : 0;
  const int shortest = 0x7fffffff - h[3];
  m->fixed_row_len = (m->n > 0 && shortest == h[1] && h[1] > 0) ? h[1] : 0;  // every row holds the same number of entries
  return detect_fields(m);
}

}  // namespace fmx
