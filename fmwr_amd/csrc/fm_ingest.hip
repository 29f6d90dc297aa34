// Ingest side of the path (one-off per matrix, not the hot loop):
//   * per-batch CSC ("inverted index": for every feature the (local row, x) pairs of one batch in row order),
//     which phase 2 of the mini-batch step walks; built with a stable device radix sort by column;
//   * CSC of the whole matrix for the ALS sweep -- the reference builds it with an O(p*n) scan
//     (util/Smatrix.h:155-185, called at src/FM.cpp:148-152); result layout is the same (rows ascending per feature);
//   * the synthetic workload generator, Philox4x32-10 keyed by (seed, global row id);
//   * the strictly-ascending-rows check the sequential learner uses.
#include <hip/hip_runtime.h>

#include <cstring>  // rocprim's texture_cache_iterator.hpp uses memset without including it

#include <rocprim/rocprim.hpp>

#include "fmx_internal.h"

namespace fmx {

// ------------------------------------------------------------------------------------------------ CSC builders
__global__ void pack_entries_k(const int64_t* __restrict__ row_ptr, const float* __restrict__ val, int64_t r0, int64_t nrows,
                               int64_t base, int64_t cnt, uint64_t* __restrict__ packed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const int64_t t = base + i;
  // row of entry t: last r in [r0, r0+nrows) with row_ptr[r] <= t
  int64_t lo = r0, hi = r0 + nrows;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (row_ptr[mid] <= t) lo = mid; else hi = mid;
  }
  packed[i] = ((uint64_t)(uint32_t)(lo - r0) << 32) | (uint64_t)__float_as_uint(val[t]);
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

// sort entries [base, base+cnt) of rows [r0, r0+nrows) by column (stable => rows stay ascending per column)
template <typename OffT>
static int csc_of_range(const fmx_matrix* m, SortScratch& s, int bits, int64_t r0, int64_t nrows, int64_t base, int64_t cnt,
                        uint32_t* out_rows, float* out_vals, OffT* out_ptr, hipStream_t stream) {
  const int T = 256;
  if (cnt > 0) {
    hipLaunchKernelGGL(pack_entries_k, dim3((unsigned)((cnt + T - 1) / T)), dim3(T), 0, stream, m->row_ptr, m->val, r0, nrows, base, cnt, s.vals_in);
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

// flags[j] = feature j has at least one entry in the tile
__global__ void touched_flags_k(const uint32_t* __restrict__ bptr, uint32_t p, uint8_t* __restrict__ flags) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < (int64_t)p) flags[j] = bptr[j + 1] > bptr[j];
}

__global__ void max_list_len_k(const uint32_t* __restrict__ bptr, uint32_t p, uint32_t long_min, uint32_t* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < (int64_t)p) {
    const uint32_t len = bptr[j + 1] - bptr[j];
    if (len > long_min) atomicMax(out, len);
  }
}

__global__ void touched_offsets_k(const uint32_t* __restrict__ bptr, const uint32_t* __restrict__ tfeat, uint32_t n, uint32_t end,
                                  uint32_t* __restrict__ toff) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)n) toff[i] = bptr[tfeat[i]];
  else if (i == (int64_t)n) toff[i] = end;
}

__global__ void gather_rows_k(const int64_t* __restrict__ src, const int64_t* __restrict__ rows, int64_t count, int64_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = src[rows[i]];
}

int build_batch_csc(fmx_matrix* m, int64_t batch_rows, int64_t tile_rows, hipStream_t stream) {
  FMX_CHECK(batch_rows > 0 && tile_rows > 0, FMX_ERR_INVALID, "batch_rows and tile_rows must be positive");
  if (m->batch_rows == batch_rows && m->tile_rows == tile_rows && m->bptr) return FMX_OK;
  FMX_HIP(hipSetDevice(m->device));
  (void)hipFree(m->bptr); (void)hipFree(m->brow); (void)hipFree(m->bval); (void)hipFree(m->tfeat); (void)hipFree(m->toff);
  m->bptr = nullptr; m->brow = nullptr; m->bval = nullptr; m->tfeat = nullptr; m->toff = nullptr;
  // steps of batch_rows rows, each cut into tiles of at most tile_rows rows
  const int64_t nb = (m->n + batch_rows - 1) / batch_rows;
  m->batch_rows = batch_rows;
  m->tile_rows = tile_rows;
  m->n_batches = nb;
  m->tile_start.clear();
  m->step_first_tile.assign((size_t)nb + 1, 0);
  for (int64_t s = 0; s < nb; ++s) {
    m->step_first_tile[(size_t)s] = (int64_t)m->tile_start.size();
    const int64_t end = (s + 1) * batch_rows < m->n ? (s + 1) * batch_rows : m->n;
    for (int64_t r = s * batch_rows; r < end; r += tile_rows) m->tile_start.push_back(r);
  }
  m->step_first_tile[(size_t)nb] = (int64_t)m->tile_start.size();
  m->tile_start.push_back(m->n);
  const int64_t nt = (int64_t)m->tile_start.size() - 1;
  // row_ptr at the tile boundaries -> host
  m->h_row_ptr_batches.assign((size_t)nt + 1, 0);
  {
    int64_t *d_rows = nullptr, *d = nullptr;
    FMX_HIP(hipMalloc(&d_rows, ((size_t)nt + 1) * sizeof(int64_t)));
    FMX_HIP(hipMalloc(&d, ((size_t)nt + 1) * sizeof(int64_t)));
    FMX_HIP(hipMemcpyAsync(d_rows, m->tile_start.data(), ((size_t)nt + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(gather_rows_k, dim3((unsigned)((nt + 1 + 255) / 256)), dim3(256), 0, stream, m->row_ptr, d_rows, nt + 1, d);
    FMX_HIP(hipMemcpyAsync(m->h_row_ptr_batches.data(), d, ((size_t)nt + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    FMX_HIP(hipStreamSynchronize(stream));
    FMX_HIP(hipFree(d)); FMX_HIP(hipFree(d_rows));
  }
  int64_t max_cnt = 0;
  for (int64_t t = 0; t < nt; ++t) {
    const int64_t c = m->h_row_ptr_batches[t + 1] - m->h_row_ptr_batches[t];
    FMX_CHECK(c < (1LL << 32), FMX_ERR_INVALID, "a tile holds %lld nonzeros; at most 2^32-1 are supported (lower tile_rows)", (long long)c);
    if (c > max_cnt) max_cnt = c;
  }
  FMX_HIP(hipMalloc(&m->bptr, (size_t)(nt > 0 ? nt : 1) * ((size_t)m->p + 1) * sizeof(uint32_t)));
  FMX_HIP(hipMalloc(&m->brow, (size_t)(m->nnz > 0 ? m->nnz : 1) * sizeof(uint32_t)));
  FMX_HIP(hipMalloc(&m->bval, (size_t)(m->nnz > 0 ? m->nnz : 1) * sizeof(float)));
  const int bits = col_bits(m->p);
  SortScratch s;
  FMX_TRY(sort_scratch_alloc(s, max_cnt, bits, stream));
  for (int64_t t = 0; t < nt; ++t) {
    const int64_t r0 = m->tile_start[(size_t)t];
    const int64_t nrows = m->tile_start[(size_t)t + 1] - r0;
    const int64_t base = m->h_row_ptr_batches[t], cnt = m->h_row_ptr_batches[t + 1] - base;
    FMX_TRY(csc_of_range<uint32_t>(m, s, bits, r0, nrows, base, cnt, m->brow + base, m->bval + base,
                                   m->bptr + (size_t)t * ((size_t)m->p + 1), stream));
  }
  // Long lists (heavy hitters of a skewed feature distribution): per tile the features whose list exceeds list_long_min() entries,
  // each cut into segments of LIST_SEG entries (fm_batch_kernels.hip walks a segment with one wave).
  (void)hipFree(m->lplan); m->lplan = nullptr;
  m->long_tiles.assign((size_t)nt, fmx_matrix::LongTile{0, 0, 0, 0, 0, 0, 0});
  m->max_long_seg = 0;
  {
    uint32_t* d_max = nullptr;
    FMX_HIP(hipMalloc(&d_max, sizeof(uint32_t)));
    std::vector<uint32_t> plan;  // all tiles: lfeat | lseg_ptr | seg_feat | seg_begin | seg_end
    std::vector<uint32_t> hb((size_t)m->p + 1);
    for (int64_t t = 0; t < nt; ++t) {
      const uint32_t* tb = m->bptr + (size_t)t * ((size_t)m->p + 1);
      uint32_t h = 0;
      FMX_HIP(hipMemsetAsync(d_max, 0, sizeof(uint32_t), stream));
      hipLaunchKernelGGL(max_list_len_k, dim3((unsigned)(((int64_t)m->p + 255) / 256)), dim3(256), 0, stream, tb, m->p, list_long_min(), d_max);
      FMX_HIP(hipMemcpyAsync(&h, d_max, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      FMX_HIP(hipStreamSynchronize(stream));
      if (h == 0) continue;  // no list above list_long_min() in this tile
      FMX_HIP(hipMemcpy(hb.data(), tb, hb.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
      std::vector<uint32_t> lfeat, lseg{0}, sfeat, sbeg, send;
      for (uint32_t j = 0; j < m->p; ++j) {
        const uint32_t len = hb[j + 1] - hb[j];
        if (len <= list_long_min()) continue;
        const uint32_t li = (uint32_t)lfeat.size();
        lfeat.push_back(j);
        for (uint32_t b = hb[j]; b < hb[j + 1]; b += LIST_SEG) {
          sfeat.push_back(li);
          sbeg.push_back(b);
          send.push_back(b + LIST_SEG < hb[j + 1] ? b + LIST_SEG : hb[j + 1]);
        }
        lseg.push_back((uint32_t)sfeat.size());
      }
      auto& lt = m->long_tiles[(size_t)t];
      lt.n_long = (int64_t)lfeat.size(); lt.n_seg = (int64_t)sfeat.size();
      lt.off_lfeat = (int64_t)plan.size(); plan.insert(plan.end(), lfeat.begin(), lfeat.end());
      lt.off_lseg = (int64_t)plan.size(); plan.insert(plan.end(), lseg.begin(), lseg.end());
      lt.off_sfeat = (int64_t)plan.size(); plan.insert(plan.end(), sfeat.begin(), sfeat.end());
      lt.off_sbeg = (int64_t)plan.size(); plan.insert(plan.end(), sbeg.begin(), sbeg.end());
      lt.off_send = (int64_t)plan.size(); plan.insert(plan.end(), send.begin(), send.end());
      if (lt.n_seg > m->max_long_seg) m->max_long_seg = lt.n_seg;
    }
    (void)hipFree(d_max);
    if (!plan.empty()) {
      FMX_HIP(hipMalloc(&m->lplan, plan.size() * sizeof(uint32_t)));
      FMX_HIP(hipMemcpy(m->lplan, plan.data(), plan.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
  }

  // Sparse tiles: when fewer than half the features occur in a tile (few entries, or a skewed feature distribution whose
  // entries pile up on few features), the per-feature walk should skip the rest: keep the ascending list of the features
  // that do occur, with a compact copy of their entry offsets.
  m->tfeat_ptr.assign((size_t)nt + 1, 0);
  {
    uint8_t* d_flags = nullptr;
    uint32_t* d_count = nullptr;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0, tb2 = 0;
    FMX_HIP(hipMalloc(&d_flags, (size_t)m->p));
    FMX_HIP(hipMalloc(&d_count, sizeof(uint32_t)));
    rocprim::counting_iterator<uint32_t> ids(0);
    auto flags32 = rocprim::make_transform_iterator(d_flags, [] __device__(uint8_t v) { return (uint32_t)v; });
    FMX_HIP(rocprim::select(nullptr, tmp_bytes, ids, d_flags, (uint32_t*)nullptr, d_count, (size_t)m->p, stream));
    FMX_HIP(rocprim::reduce(nullptr, tb2, flags32, d_count, (uint32_t)0, (size_t)m->p, rocprim::plus<uint32_t>(), stream));
    if (tb2 > tmp_bytes) tmp_bytes = tb2;
    FMX_HIP(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
    // pass 1: how many features occur in each tile
    std::vector<uint32_t> occ((size_t)nt, 0);
    int64_t total = 0;
    for (int64_t t = 0; t < nt; ++t) {
      const int64_t cnt = m->h_row_ptr_batches[t + 1] - m->h_row_ptr_batches[t];
      if (cnt >= (int64_t)m->p * 4) continue;  // dense enough that (almost) every feature occurs: not worth counting
      hipLaunchKernelGGL(touched_flags_k, dim3((unsigned)(((int64_t)m->p + 255) / 256)), dim3(256), 0, stream,
                         m->bptr + (size_t)t * ((size_t)m->p + 1), m->p, d_flags);
      FMX_HIP(rocprim::reduce(d_tmp, tmp_bytes, flags32, d_count, (uint32_t)0, (size_t)m->p, rocprim::plus<uint32_t>(), stream));
      uint32_t h = 0;
      FMX_HIP(hipMemcpyAsync(&h, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      FMX_HIP(hipStreamSynchronize(stream));
      if ((int64_t)h * 2 < (int64_t)m->p) { occ[(size_t)t] = h; total += h; }
    }
    // pass 2: the lists
    if (total > 0) {
      FMX_HIP(hipMalloc(&m->tfeat, (size_t)total * sizeof(uint32_t)));
      FMX_HIP(hipMalloc(&m->toff, ((size_t)total + (size_t)nt) * sizeof(uint32_t)));
    }
    int64_t at = 0;
    for (int64_t t = 0; t < nt; ++t) {
      m->tfeat_ptr[(size_t)t] = at;
      if (occ[(size_t)t] == 0) continue;
      hipLaunchKernelGGL(touched_flags_k, dim3((unsigned)(((int64_t)m->p + 255) / 256)), dim3(256), 0, stream,
                         m->bptr + (size_t)t * ((size_t)m->p + 1), m->p, d_flags);
      FMX_HIP(rocprim::select(d_tmp, tmp_bytes, ids, d_flags, m->tfeat + at, d_count, (size_t)m->p, stream));
      const uint32_t h = occ[(size_t)t];
      const uint32_t entries = (uint32_t)(m->h_row_ptr_batches[t + 1] - m->h_row_ptr_batches[t]);
      hipLaunchKernelGGL(touched_offsets_k, dim3((unsigned)((h + 1 + 255) / 256)), dim3(256), 0, stream,
                         m->bptr + (size_t)t * ((size_t)m->p + 1), m->tfeat + at, h, entries, m->toff + at + t);
      at += h;
    }
    m->tfeat_ptr[(size_t)nt] = at;
    FMX_HIP(hipStreamSynchronize(stream));
    (void)hipFree(d_flags); (void)hipFree(d_count); (void)hipFree(d_tmp);
  }
  FMX_HIP(hipStreamSynchronize(stream));
  return FMX_OK;
}

int build_full_csc(fmx_matrix* m, hipStream_t stream) {
  if (m->col_ptr) return FMX_OK;
  FMX_CHECK(m->n < (1LL << 32), FMX_ERR_INVALID, "ALS sweep supports fewer than 2^32 rows");
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
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * z) return;
  const int64_t r = t / z;
  const uint32_t i = (uint32_t)(t - r * z);
  const uint64_t g = (uint64_t)(row_offset + r);
  const Philox ph = philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), i >> 2, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t rnd = ph.c[i & 3];
  const uint32_t lo = (uint32_t)(((uint64_t)i * p) / (uint32_t)z);
  const uint32_t hi = (uint32_t)(((uint64_t)(i + 1) * p) / (uint32_t)z);
  col[t] = lo + (uint32_t)(((uint64_t)rnd * (hi - lo)) >> 32);
  val[t] = 1.0f;
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

int generate_synthetic(fmx_matrix* m, int32_t z, uint64_t seed, int64_t row_offset) {
  const int T = 256;
  const int64_t total = m->n * z;
  if (total > 0)
    hipLaunchKernelGGL(synth_entries_k, dim3((unsigned)((total + T - 1) / T)), dim3(T), 0, nullptr, m->n, m->p, z, seed, row_offset, m->col, m->val);
  hipLaunchKernelGGL(synth_rows_k, dim3((unsigned)((m->n + 1 + T - 1) / T)), dim3(T), 0, nullptr, m->n, z, seed, row_offset, m->row_ptr, m->y);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipDeviceSynchronize());
  m->rows_sorted = 1;  // strata are disjoint and ascending
  m->max_row_len = z;
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

__global__ void scales_apply_k(int64_t nnz, const uint32_t* __restrict__ col, float* __restrict__ val, const double* __restrict__ mean,
                               const double* __restrict__ std) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nnz) return;
  float v = val[t];
  v = (float)((double)v - mean[col[t]]);           // value[p] -= colSum[idx]        (:127)
  v = (float)((double)v / (std[col[t]] + 1e-30));  // value[p] /= (colSumSqr + 1e-30) (:128)
  val[t] = v;
}

__global__ void normalize_apply_k(int64_t nnz, const uint32_t* __restrict__ col, float* __restrict__ val, const double* __restrict__ mean,
                                  const double* __restrict__ std) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nnz) return;
  const uint32_t i = col[t];
  if (std[i] != 0) val[t] = (float)(((double)val[t] - mean[i]) / std[i]);  // util/Smatrix.h:148-150
}

// the cached inverted indices hold copies of the values: drop them so they are rebuilt from the new values
static void drop_value_caches(fmx_matrix* m) {
  (void)hipFree(m->bptr); (void)hipFree(m->brow); (void)hipFree(m->bval); (void)hipFree(m->tfeat); (void)hipFree(m->toff); (void)hipFree(m->lplan);
  (void)hipFree(m->col_ptr); (void)hipFree(m->crow); (void)hipFree(m->cval);
  m->bptr = nullptr; m->brow = nullptr; m->bval = nullptr; m->tfeat = nullptr; m->toff = nullptr; m->lplan = nullptr; m->col_ptr = nullptr; m->crow = nullptr; m->cval = nullptr;
  m->batch_rows = 0; m->tile_rows = 0;
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
  if (m->nnz > 0) hipLaunchKernelGGL(scales_apply_k, dim3((unsigned)((m->nnz + 255) / 256)), dim3(256), 0, nullptr, m->nnz, m->col, m->val, d_mean, d_std);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipMemcpy(h_mean, d_mean, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
  FMX_HIP(hipMemcpy(h_std, d_std, (size_t)p * sizeof(double), hipMemcpyDeviceToHost));
  (void)hipFree(d_mean); (void)hipFree(d_std); (void)hipFree(d_listed);
  drop_value_caches(m);
  return FMX_OK;
}

int matrix_normalize(fmx_matrix* m, const double* h_mean, const double* h_std) {
  const uint32_t p = m->p;
  double *d_mean = nullptr, *d_std = nullptr;
  FMX_HIP(hipMalloc(&d_mean, (size_t)p * sizeof(double)));
  FMX_HIP(hipMalloc(&d_std, (size_t)p * sizeof(double)));
  FMX_HIP(hipMemcpy(d_mean, h_mean, (size_t)p * sizeof(double), hipMemcpyHostToDevice));
  FMX_HIP(hipMemcpy(d_std, h_std, (size_t)p * sizeof(double), hipMemcpyHostToDevice));
  if (m->nnz > 0) hipLaunchKernelGGL(normalize_apply_k, dim3((unsigned)((m->nnz + 255) / 256)), dim3(256), 0, nullptr, m->nnz, m->col, m->val, d_mean, d_std);
  FMX_HIP(hipGetLastError());
  FMX_HIP(hipDeviceSynchronize());
  (void)hipFree(d_mean); (void)hipFree(d_std);
  drop_value_caches(m);
  return FMX_OK;
}

// ------------------------------------------------------------------------------------------------ sortedness
__global__ void rows_sorted_k(const int64_t* __restrict__ row_ptr, const uint32_t* __restrict__ col, int64_t n, int* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int bad = 0;
  for (int64_t t = row_ptr[r]; t + 1 < row_ptr[r + 1]; ++t) bad |= (col[t] >= col[t + 1]);
  if (bad) out[0] = 1;
  const int64_t len = row_ptr[r + 1] - row_ptr[r];
  atomicMax(out + 1, (int)(len > 0x7fffffff ? 0x7fffffff : len));  // longest row
}

int check_rows_sorted(fmx_matrix* m) {
  int* d = nullptr;
  int h[2] = {0, 0};
  FMX_HIP(hipMalloc(&d, 2 * sizeof(int)));
  FMX_HIP(hipMemset(d, 0, 2 * sizeof(int)));
  if (m->n > 0) hipLaunchKernelGGL(rows_sorted_k, dim3((unsigned)((m->n + 255) / 256)), dim3(256), 0, nullptr, m->row_ptr, m->col, m->n, d);
  FMX_HIP(hipMemcpy(h, d, 2 * sizeof(int), hipMemcpyDeviceToHost));
  FMX_HIP(hipFree(d));
  m->rows_sorted = !h[0];
  m->max_row_len = h[1];
  return FMX_OK;
}

}  // namespace fmx
