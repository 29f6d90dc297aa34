// Measurement aids behind the C ABI (include/fmx.h, "measurement" section).  Nothing here is on the product path.
//
// fmx_measure_gather: the rate at which the memory system serves uniformly random table rows -- the access pattern of
// fm_rows_forward (V rows by column id) and fm_cols_update (S rows by row id) with everything else stripped away: the row
// ids are generated in registers (an integer hash of the fetch number: no index stream, no side table), LPR = row_bytes / 16
// lanes fetch one row as one contiguous segment, `in_flight` rows are outstanding per lane, every group sums `per_group`
// rows and writes one row.  bench.py divides a kernel's rows/s by this figure at the kernel's own table size and row
// width ("ceiling_frac"): how much of what the hardware gives this access pattern the kernel really gets.
#include <hip/hip_runtime.h>

#include "fmx_internal.h"

namespace fmx {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // a full-avalanche integer hash (lowbias32)
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int AUX>
__device__ __forceinline__ float4 probe_load_aux(const float4* base, uint32_t bytes, uint32_t off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(base), 0, (int)bytes, 0x00020000);
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, AUX);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// mode 0: a plain load; 1: non-temporal (nt); 2: system scope (sc0 sc1); 3: both -- what a miss then fetches from the fabric is read off the TCC_EA0_RDREQ_* counters
__device__ __forceinline__ float4 probe_load(const float4* base, uint32_t bytes, uint32_t off, int mode) {
  if (mode == 1) return probe_load_aux<2>(base, bytes, off);
  if (mode == 2) return probe_load_aux<17>(base, bytes, off);
  if (mode == 3) return probe_load_aux<19>(base, bytes, off);
  return probe_load_aux<0>(base, bytes, off);
}

static int g_probe_mode = 0;
static const float* g_probe_side = nullptr;   // FMX_PROBE_LOAD (0..3, see probe_load); FMX_PROBE_UNCACHED=1: the table is allocated hipDeviceMallocUncached

template <int LPR, int U>
__global__ __launch_bounds__(WG_THREADS) void gather_probe_k(const float4* __restrict__ table, uint32_t rows, int per_group, int64_t n_groups, uint32_t salt,
                                                             float4* __restrict__ out, int mode = 0, const float* __restrict__ side = nullptr) {
  extern __shared__ char occupancy_pad[];  // dynamic LDS only limits how many workgroups share a CU (fmx_measure_gather_occ)
  (void)occupancy_pad;
  const int64_t g = ((int64_t)blockIdx.x * WG_THREADS + threadIdx.x) / LPR;
  const int lig = threadIdx.x % LPR;
  if (g >= n_groups) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t base = (uint32_t)g * (uint32_t)per_group + salt;
  for (int t = 0; t < per_group; t += U) {
    float4 v[U];
    float sv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // multiply-shift maps the hash onto [0, rows) without a division
      uint32_t r = (uint32_t)(((uint64_t)mix32(base + (uint32_t)(t + u)) * rows) >> 32);
      // FMX_PROBE_STRATA=1 (mode bit 2): fetch t + u of every group comes from stratum t + u of the table (rows / per_group rows wide) -- the order in which
      // phase 1 walks column-sorted rows of one column per stratum: every group of the chip is then in the same narrow band of the table at the same step
      if (mode & 4) { const uint32_t w = rows / (uint32_t)per_group; r = (uint32_t)(t + u) * w + (uint32_t)(((uint64_t)mix32(base + (uint32_t)(t + u)) * w) >> 32); }
      v[u] = (mode & 3) ? probe_load(table, rows * (uint32_t)(LPR * 16), (r * LPR + lig) * 16u, mode & 3) : table[(size_t)r * LPR + lig];
      sv[u] = side ? side[r] : 0.f;   // FMX_PROBE_SIDE=1: a 4-byte word of a second table under the same id, as phase 1 reads w beside the V row
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc.x += sv[u];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  out[g * LPR + lig] = acc;
}

// The same stripped-down gather, but of the rows a MATRIX asks for: lane group g walks row r0 + g of the matrix and fetches the table row
// of every column id in it (U outstanding per lane).  On skewed columns (the heads of Criteo-shaped fields) most of these fetches are
// served on-die, and uniformly random ids are no ceiling for the kernel that reads them; this is.
template <int LPR, int U>
__global__ __launch_bounds__(WG_THREADS) void gather_matrix_k(const float4* __restrict__ table, uint32_t table_rows, const int64_t* __restrict__ row_ptr,
                                                              const uint32_t* __restrict__ col, int64_t r0, int64_t nrows, float4* __restrict__ out) {
  const int64_t g = ((int64_t)blockIdx.x * WG_THREADS + threadIdx.x) / LPR;
  const int lig = threadIdx.x % LPR;
  if (g >= nrows) return;
  const int64_t a = row_ptr[r0 + g], b = row_ptr[r0 + g + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t t = a; t < b; t += U) {
    uint32_t id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = col[t + u < b ? t + u : t];
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = table[(size_t)(id[u] < table_rows ? id[u] : 0u) * LPR + lig];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (t + u < b) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  out[g * LPR + lig] = acc;
}

static int g_probe_lds = 0;  // bytes of dynamic LDS per workgroup (0: none): 160 KiB / this = workgroups per CU

template <int LPR>
static void launch_probe(int in_flight, dim3 g, dim3 b, hipStream_t s, const float4* table, uint32_t rows, int per_group, int64_t n_groups, uint32_t salt, float4* out) {
  if (in_flight >= 8) hipLaunchKernelGGL((gather_probe_k<LPR, 8>), g, b, g_probe_lds, s, table, rows, per_group, n_groups, salt, out, g_probe_mode, g_probe_side);
  else if (in_flight >= 4) hipLaunchKernelGGL((gather_probe_k<LPR, 4>), g, b, g_probe_lds, s, table, rows, per_group, n_groups, salt, out, g_probe_mode, g_probe_side);
  else if (in_flight >= 2) hipLaunchKernelGGL((gather_probe_k<LPR, 2>), g, b, g_probe_lds, s, table, rows, per_group, n_groups, salt, out, g_probe_mode, g_probe_side);
  else hipLaunchKernelGGL((gather_probe_k<LPR, 1>), g, b, g_probe_lds, s, table, rows, per_group, n_groups, salt, out, g_probe_mode, g_probe_side);
}

}  // namespace fmx

using namespace fmx;

// tuning aid: the same probe with `lds_bytes` of dynamic LDS per workgroup, i.e. at most 160 KiB / lds_bytes workgroups per CU --
// how many requests in flight the ceiling needs
extern "C" int fmx_measure_gather(int device, int64_t table_bytes, int32_t row_bytes, int64_t n_groups, int32_t per_group, int32_t in_flight, int32_t reps,
                                  double* rows_per_s);
extern "C" int fmx_measure_gather_occ(int device, int64_t table_bytes, int32_t row_bytes, int64_t n_groups, int32_t per_group, int32_t in_flight, int32_t reps,
                                      int32_t lds_bytes, double* rows_per_s) {
  FMX_CHECK(lds_bytes >= 0 && lds_bytes <= 64 * 1024, FMX_ERR_INVALID, "lds_bytes must be in 0..65536");
  g_probe_lds = lds_bytes;
  const int st = fmx_measure_gather(device, table_bytes, row_bytes, n_groups, per_group, in_flight, reps, rows_per_s);
  g_probe_lds = 0;
  return st;
}

extern "C" int fmx_measure_gather(int device, int64_t table_bytes, int32_t row_bytes, int64_t n_groups, int32_t per_group, int32_t in_flight, int32_t reps,
                                  double* rows_per_s) {
  FMX_CHECK(rows_per_s != nullptr, FMX_ERR_INVALID, "rows_per_s is NULL");
  FMX_CHECK(row_bytes == 16 || row_bytes == 32 || row_bytes == 64 || row_bytes == 128 || row_bytes == 256, FMX_ERR_INVALID, "row_bytes must be 16..256, a power of two");
  FMX_CHECK(table_bytes >= row_bytes && table_bytes / row_bytes < (1LL << 32) && n_groups > 0 && per_group > 0 && per_group % 8 == 0 && reps > 0, FMX_ERR_INVALID,
            "bad probe geometry (per_group must be a multiple of 8)");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { set_error("no HIP device available; libfmx has no CPU fallback"); return FMX_ERR_NOGPU; }
  FMX_CHECK(device >= 0 && device < count, FMX_ERR_INVALID, "device %d out of range", device);
  FMX_HIP(hipSetDevice(device));
  const int lpr = row_bytes / 16;
  const uint32_t rows = (uint32_t)(table_bytes / row_bytes);
  float4 *table = nullptr, *out = nullptr;
  float* side = nullptr;
  hipStream_t s = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  int st = FMX_OK;
  auto body = [&]() -> int {
    { const char* v = getenv("FMX_PROBE_LOAD"); g_probe_mode = v ? atoi(v) : 0; }
    { const char* v = getenv("FMX_PROBE_STRATA"); if (v && v[0] == '1') g_probe_mode |= 4; }
    { const char* v = getenv("FMX_PROBE_UNCACHED");
      if (v && v[0] == '1') FMX_HIP(hipExtMallocWithFlags((void**)&table, (size_t)rows * row_bytes, hipDeviceMallocUncached));
      else FMX_HIP(hipMalloc(&table, (size_t)rows * row_bytes)); }
    FMX_HIP(hipMalloc(&out, (size_t)n_groups * row_bytes));
    FMX_HIP(hipMemset(table, 0, (size_t)rows * row_bytes));
    { const char* v = getenv("FMX_PROBE_SIDE");
      if (v && v[0] == '1') { FMX_HIP(hipMalloc(&side, (size_t)rows * sizeof(float))); FMX_HIP(hipMemset(side, 0, (size_t)rows * sizeof(float))); } }
    g_probe_side = side;
    FMX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    FMX_HIP(hipEventCreate(&a)); FMX_HIP(hipEventCreate(&b));
    FMX_HIP(hipDeviceSynchronize());
    const int64_t threads = n_groups * lpr;
    const dim3 g((unsigned)((threads + WG_THREADS - 1) / WG_THREADS)), blk(WG_THREADS);
    auto launch = [&](uint32_t salt) {
      switch (lpr) {
        case 1: launch_probe<1>(in_flight, g, blk, s, table, rows, per_group, n_groups, salt, out); break;
        case 2: launch_probe<2>(in_flight, g, blk, s, table, rows, per_group, n_groups, salt, out); break;
        case 4: launch_probe<4>(in_flight, g, blk, s, table, rows, per_group, n_groups, salt, out); break;
        case 8: launch_probe<8>(in_flight, g, blk, s, table, rows, per_group, n_groups, salt, out); break;
        default: launch_probe<16>(in_flight, g, blk, s, table, rows, per_group, n_groups, salt, out); break;
      }
    };
    for (int i = 0; i < 3; ++i) launch(0x9e3779b9u * (uint32_t)i);
    FMX_HIP(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) launch(0x85ebca6bu * (uint32_t)(i + 7));  // other rows every launch
    FMX_HIP(hipEventRecord(b, s));
    FMX_HIP(hipEventSynchronize(b));
    FMX_HIP(hipGetLastError());
    float ms = 0.f;
    FMX_HIP(hipEventElapsedTime(&ms, a, b));
    *rows_per_s = (double)n_groups * per_group * reps / ((double)ms * 1e-3);
    return FMX_OK;
  };
  st = body();
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  if (s) (void)hipStreamDestroy(s);
  (void)hipFree(table); (void)hipFree(out); (void)hipFree(side);
  g_probe_side = nullptr;
  return st;
}

extern "C" int fmx_measure_gather_matrix(fmx_matrix* m, int64_t r0, int64_t nrows, int64_t table_rows, int32_t row_bytes, int32_t in_flight, int32_t reps,
                                         double* rows_per_s) {
  FMX_CHECK(m != nullptr && rows_per_s != nullptr, FMX_ERR_INVALID, "NULL argument");
  FMX_CHECK(row_bytes == 16 || row_bytes == 32 || row_bytes == 64 || row_bytes == 128 || row_bytes == 256, FMX_ERR_INVALID, "row_bytes must be 16..256, a power of two");
  FMX_CHECK(r0 >= 0 && nrows > 0 && r0 + nrows <= m->n && table_rows > 0 && table_rows < (1LL << 32) && reps > 0, FMX_ERR_INVALID, "bad probe geometry");
  FMX_HIP(hipSetDevice(m->device));
  const int lpr = row_bytes / 16;
  float4 *table = nullptr, *out = nullptr;
  hipStream_t s = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  auto body = [&]() -> int {
    FMX_HIP(hipMalloc(&table, (size_t)table_rows * row_bytes));
    FMX_HIP(hipMalloc(&out, (size_t)nrows * row_bytes));
    FMX_HIP(hipMemset(table, 0, (size_t)table_rows * row_bytes));
    FMX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    FMX_HIP(hipEventCreate(&a)); FMX_HIP(hipEventCreate(&b));
    FMX_HIP(hipDeviceSynchronize());
    const dim3 g((unsigned)((nrows * lpr + WG_THREADS - 1) / WG_THREADS)), blk(WG_THREADS);
    int64_t h_ab[2] = {0, 0};
    FMX_HIP(hipMemcpy(&h_ab[0], m->row_ptr + r0, sizeof(int64_t), hipMemcpyDeviceToHost));
    FMX_HIP(hipMemcpy(&h_ab[1], m->row_ptr + r0 + nrows, sizeof(int64_t), hipMemcpyDeviceToHost));
    auto launch = [&]() {
#define FMX_GM(L) \
  if (in_flight >= 8) hipLaunchKernelGGL((gather_matrix_k<L, 8>), g, blk, 0, s, table, (uint32_t)table_rows, m->row_ptr, m->col, r0, nrows, out); \
  else if (in_flight >= 4) hipLaunchKernelGGL((gather_matrix_k<L, 4>), g, blk, 0, s, table, (uint32_t)table_rows, m->row_ptr, m->col, r0, nrows, out); \
  else hipLaunchKernelGGL((gather_matrix_k<L, 1>), g, blk, 0, s, table, (uint32_t)table_rows, m->row_ptr, m->col, r0, nrows, out);
      switch (lpr) {
        case 1: FMX_GM(1) break;
        case 2: FMX_GM(2) break;
        case 4: FMX_GM(4) break;
        case 8: FMX_GM(8) break;
        default: FMX_GM(16) break;
      }
#undef FMX_GM
    };
    for (int i = 0; i < 3; ++i) launch();
    FMX_HIP(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) launch();
    FMX_HIP(hipEventRecord(b, s));
    FMX_HIP(hipEventSynchronize(b));
    FMX_HIP(hipGetLastError());
    float ms = 0.f;
    FMX_HIP(hipEventElapsedTime(&ms, a, b));
    *rows_per_s = (double)(h_ab[1] - h_ab[0]) * reps / ((double)ms * 1e-3);
    return FMX_OK;
  };
  const int st = body();
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  if (s) (void)hipStreamDestroy(s);
  (void)hipFree(table); (void)hipFree(out);
  return st;
}
