// Evaluation metrics on the device (core/Evaluation.h:20-115), used by the tracker (core/Tracker.h:65-94,
// solver/SGD_Learner.h:140-166) and by fmx_evaluate (what FMTrack computes per snapshot, src/FM.cpp:218-258).
// y_hat is the forward's output after the link (probability for CLASSIFICATION, clamped prediction for REGRESSION).
// Reductions run in a fixed two-level order (slab per workgroup, then one workgroup), so a metric is bitwise
// reproducible; the reference sums serially, which only changes the last bits.  Quirks kept (SURVEY A-15): mae() returns
// the SQUARE ROOT of the mean absolute error, MSE falls into the RMSE branch, AUC orders by |score| with the label
// folded into the sign and returns max(a, 1-a).
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "fmx_internal.h"

namespace fmx {

constexpr int EVAL_SLAB = 4096;  // elements per workgroup in the first reduction level

__device__ __forceinline__ double eval_term(int kind, double yh, float y) {
  switch (kind) {
    case 0:  // LL, Evaluation.h:80-89 ((1 + y) and (1 - y) are float expressions there; exact for y = +-1)
      return (double)(1 + y) * log(yh + 1e-20) + (double)(1 - y) * log(1 - yh - 1e-20);
    case 1:  // ACC, :43-53 (cutoff 0.5)
      return (((yh >= 0.5) && (y > 0)) || ((yh < 0.5) && (y < 0))) ? 1.0 : 0.0;
    case 2: { const double err = yh - (double)y; return err * err; }  // RMSE, :91-102
    default: { const double err = yh - (double)y; return fabs(err); } // MAE, :104-115
  }
}

__global__ __launch_bounds__(WG_THREADS) void eval_partial_k(const double* __restrict__ yhat, const float* __restrict__ y, int64_t n,
                                                            int kind, double* __restrict__ partials) {
  __shared__ double red[WG_THREADS];
  const int64_t base = (int64_t)blockIdx.x * EVAL_SLAB;
  double acc = 0.0;
  for (int i = threadIdx.x; i < EVAL_SLAB; i += WG_THREADS) {
    const int64_t r = base + i;
    if (r < n) acc += eval_term(kind, yhat[r], y[r]);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = WG_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(WG_THREADS) void eval_final_k(const double* __restrict__ partials, int64_t n_partials, double* __restrict__ out) {
  __shared__ double red[WG_THREADS];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n_partials; i += WG_THREADS) acc += partials[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = WG_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}

// AUC, Evaluation.h:55-78: tmp = y > 0 ? yh : -yh; order by |tmp|; positives are tmp > 0
__global__ void auc_keys_k(const double* __restrict__ yhat, const float* __restrict__ y, int64_t n, uint64_t* __restrict__ keys,
                           uint32_t* __restrict__ pos) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const double tmp = y[r] > 0 ? yhat[r] : -yhat[r];
  keys[r] = (uint64_t)__double_as_longlong(fabs(tmp));  // non-negative doubles order like their bit patterns
  pos[r] = tmp > 0 ? 1u : 0u;
}

struct AucPairs {
  __device__ uint64_t operator()(const rocprim::tuple<uint32_t, uint64_t>& t) const {
    return rocprim::get<0>(t) ? 0ull : rocprim::get<1>(t);  // a negative contributes the positives ranked before it
  }
};

static int auc_device(const double* d_yhat, const float* d_y, int64_t n, hipStream_t stream, double* result) {
  uint64_t *keys = nullptr, *keys_s = nullptr, *before = nullptr, *d_sums = nullptr;
  uint32_t *pos = nullptr, *pos_s = nullptr;
  void* temp = nullptr;
  size_t temp_bytes = 0, tb = 0;
  int st = FMX_OK;
  auto cleanup = [&]() {
    (void)hipFree(keys); (void)hipFree(keys_s); (void)hipFree(before); (void)hipFree(d_sums); (void)hipFree(pos); (void)hipFree(pos_s); (void)hipFree(temp);
  };
#define AUC_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(_e)); cleanup(); return FMX_ERR_HIP; } } while (0)
  const size_t m = (size_t)n;
  AUC_HIP(hipMalloc(&keys, m * 8)); AUC_HIP(hipMalloc(&keys_s, m * 8)); AUC_HIP(hipMalloc(&before, m * 8)); AUC_HIP(hipMalloc(&d_sums, 16));
  AUC_HIP(hipMalloc(&pos, m * 4)); AUC_HIP(hipMalloc(&pos_s, m * 4));
  hipLaunchKernelGGL(auc_keys_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_yhat, d_y, n, keys, pos);
  AUC_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys, keys_s, pos, pos_s, m, 0, 64, stream));
  auto pos64 = rocprim::make_transform_iterator(pos_s, [] __device__(uint32_t v) { return (uint64_t)v; });
  AUC_HIP(rocprim::exclusive_scan(nullptr, tb, pos64, before, (uint64_t)0, m, rocprim::plus<uint64_t>(), stream));
  if (tb > temp_bytes) temp_bytes = tb;
  AUC_HIP(rocprim::reduce(nullptr, tb, pos64, d_sums, (uint64_t)0, m, rocprim::plus<uint64_t>(), stream));
  if (tb > temp_bytes) temp_bytes = tb;
  AUC_HIP(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
  AUC_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_s, pos, pos_s, m, 0, 64, stream));  // stable: ties keep row order
  AUC_HIP(rocprim::exclusive_scan(temp, temp_bytes, pos64, before, (uint64_t)0, m, rocprim::plus<uint64_t>(), stream));
  AUC_HIP(rocprim::reduce(temp, temp_bytes, pos64, d_sums, (uint64_t)0, m, rocprim::plus<uint64_t>(), stream));
  auto pairs = rocprim::make_transform_iterator(rocprim::make_zip_iterator(rocprim::make_tuple(pos_s, before)), AucPairs());
  AUC_HIP(rocprim::reduce(temp, temp_bytes, pairs, d_sums + 1, (uint64_t)0, m, rocprim::plus<uint64_t>(), stream));
  uint64_t h[2] = {0, 0};
  AUC_HIP(hipMemcpyAsync(h, d_sums, 16, hipMemcpyDeviceToHost, stream));
  AUC_HIP(hipStreamSynchronize(stream));
#undef AUC_HIP
  cleanup();
  const double cum_tp = (double)h[0];
  if (cum_tp == 0 || cum_tp == (double)n) { *result = 1.0; return st; }
  double area = (double)h[1] / (cum_tp * ((double)n - cum_tp));
  *result = area < 0.5 ? 1 - area : area;
  return st;
}

// evaluates(), core/Evaluation.h:20-41
int evaluate_device(fmx_engine* e, const double* d_yhat, const float* d_y, int64_t n, int metric, double* result) {
  FMX_CHECK(n > 0, FMX_ERR_INVALID, "cannot evaluate an empty data set");
  int kind;
  bool root = false;
  if (e->cfg.task == FMX_TASK_REGRESSION) {
    if (metric <= FMX_EVAL_RMSE) { kind = 2; root = true; } else { kind = 3; root = true; }
  } else {
    if (metric >= FMX_EVAL_ACC) kind = 1;
    else if (metric == FMX_EVAL_LL) kind = 0;
    else return auc_device(d_yhat, d_y, n, e->stream, result);
  }
  const int64_t np = (n + EVAL_SLAB - 1) / EVAL_SLAB;
  double* d = nullptr;
  FMX_HIP(hipMalloc(&d, ((size_t)np + 1) * sizeof(double)));
  hipLaunchKernelGGL(eval_partial_k, dim3((unsigned)np), dim3(WG_THREADS), 0, e->stream, d_yhat, d_y, n, kind, d);
  hipLaunchKernelGGL(eval_final_k, dim3(1), dim3(WG_THREADS), 0, e->stream, d, np, d + np);
  double sum = 0.0;
  hipError_t err = hipMemcpyAsync(&sum, d + np, sizeof(double), hipMemcpyDeviceToHost, e->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  (void)hipFree(d);
  FMX_CHECK(err == hipSuccess, FMX_ERR_HIP, "metric reduction failed: %s", hipGetErrorString(err));
  if (kind == 0) *result = sum / 2.0;
  else if (kind == 1) *result = sum / (double)n;
  else *result = root ? sqrt(sum / (double)n) : sum / (double)n;
  return FMX_OK;
}

}  // namespace fmx
