// The reassociated reference-order learner's CHAIN WAVE alone (fm_seq_reassoc_k's wave 0; no workers: every example's row part is ready in LDS): clocks per example of its
// loop in several forms.  What does an example cost the chain when nothing else runs?
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o profiles/probes/bin/seq_chain_bench profiles/probes/seq_chain_bench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ double bcast(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ float bcast(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ double seq_exp_re(double x) {
  auto B = [](unsigned long long u) { return __longlong_as_double((long long)u); };
  const double n = __builtin_rint(x * B(0x3ff71547652b82feull));
  double r = __builtin_fma(n, B(0xbfe62e42fefa39efull), x);
  r = __builtin_fma(n, B(0xbc7abc9e3b39803full), r);
  const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
  const double p01 = 1.0 + r;
  const double p23 = __builtin_fma(B(0x3fc5555555555511ull), r, B(0x3fe000000000000bull));
  const double p45 = __builtin_fma(B(0x3f81111111122322ull), r, B(0x3fa55555555502a1ull));
  const double p67 = __builtin_fma(B(0x3f2a01a014761f6eull), r, B(0x3f56c16c1852b7b0ull));
  const double p89 = __builtin_fma(B(0x3ec71dee623fde64ull), r, B(0x3efa01997c89e6b0ull));
  const double pab = __builtin_fma(B(0x3e5ade156a5dcb37ull), r, B(0x3e928af3fca7ab0cull));
  const double q0 = __builtin_fma(p23, r2, p01), q1 = __builtin_fma(p67, r2, p45), q2 = __builtin_fma(pab, r2, p89);
  return ldexp(__builtin_fma(q2, r8, __builtin_fma(q1, r4, q0)), (int)n);
}
__device__ __forceinline__ double seq_rcp_re(double d) {
  double xr = __builtin_amdgcn_rcp(d);
  xr = __builtin_fma(xr, __builtin_fma(-d, xr, 1.0), xr);
  return __builtin_fma(xr, __builtin_fma(-d, xr, 1.0), xr);
}
__device__ __forceinline__ double seq_exp_small1(double dl) {
  constexpr double c2 = 1.0 / 2, c3 = 1.0 / 6, c4 = 1.0 / 24, c5 = 1.0 / 120, c6 = 1.0 / 720, c7 = 1.0 / 5040;
  const double d2 = dl * dl, d4 = d2 * d2;
  return __builtin_fma(__builtin_fma(__builtin_fma(c7, dl, c6), d2, __builtin_fma(c5, dl, c4)), d4, __builtin_fma(__builtin_fma(c3, dl, c2), d2, 1.0 + dl));
}
constexpr int R = 60, N = 1 << 16;
// FORM 0: the whole exponential per example (the first session's chain).  1: split, lane parity, scalar control (fm_seq_reassoc_k's batch_split).  2: 1 without the per-example
// hand-over to LDS.  3: 1 without the data-dependent branches (labels known +-1, steps known small, E recomputed every 32).  4: 3 with the hand-over by ALL lanes to the same word.
template <int FORM>
__global__ __launch_bounds__(64) void chain_k(const double* __restrict__ rs, const float* __restrict__ ys, double lr, double reg0, double* out, unsigned long long* ticks) {
  __shared__ double s_mult[R];
  __shared__ int f_m[R];
  const int lane = threadIdx.x;
  double w0 = 0.01, rw = reg0 * w0, E = 1.0;
  const bool odd = lane & 1;
  const double lr_l = odd ? lr : -lr;
  bool small_steps = false;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int e = 0; e < N; e += 16) {
    const int n = 16;
    const double r = lane < n ? rs[e + lane] : 0.0;
    const float y = lane < n ? ys[e + lane] : 0.f;
    if (FORM == 0) {
      for (int i = 0; i < n; ++i) {
        const double pred = w0 + bcast(r, i);
        const double yd = (double)bcast(y, i);
        const double a = yd * pred;
        const double xr = seq_rcp_re(1.0 + seq_exp_re(fmin(fmax(a, -750.0), 700.0)));
        const double mult = -yd * (a != a ? a : xr);
        w0 -= lr * (mult + reg0 * w0);
        if (lane == i) { s_mult[(e + i) % R] = mult; __atomic_signal_fence(__ATOMIC_SEQ_CST); __hip_atomic_store(&f_m[(e + i) % R], e + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      }
    } else {
      const double f = seq_exp_re(fmin(fmax((double)y * r, -350.0), 350.0));
      const unsigned long long pm1 = __ballot(lane < n && (y == 1.0f || y == -1.0f)), pos = __ballot(y > 0.f);
      int sl = e % R;
      for (int i = 0; i < n; ++i, sl = sl + 1 < R ? sl + 1 : 0) {
        if (((e + i) & 31) == 0) {
          const double wc = fmin(fmax(w0, -350.0), 350.0);
          E = seq_exp_re(odd ? -wc : wc);
          small_steps = lr * (1.0 + fabs(reg0) * (fabs(w0) + 1.0)) <= 0.03125;
        }
        double mult;
        if (FORM >= 3 || (small_steps && ((pm1 >> i) & 1ull))) {
          const bool up = (pos >> i) & 1ull;
          const double xr_l = seq_rcp_re(__builtin_fma(E, bcast(f, i), 1.0));
          const double xr = bcast(xr_l, up ? 0 : 1);
          mult = up ? -xr : xr;
          const double g = mult + rw;
          w0 -= lr * g;
          rw = reg0 * w0;
          E *= seq_exp_small1(lr_l * g);
        } else {
          const double yd = (double)bcast(y, i), a = yd * (w0 + bcast(r, i));
          const double xr = seq_rcp_re(1.0 + seq_exp_re(fmin(fmax(a, -750.0), 700.0)));
          mult = -yd * (a != a ? a : xr);
          w0 -= lr * (mult + reg0 * w0);
          rw = reg0 * w0;
          if (small_steps) { const double wc = fmin(fmax(w0, -350.0), 350.0); E = seq_exp_re(odd ? -wc : wc); }
        }
        if (FORM == 1 || FORM == 3) {
          if (lane == 0) { s_mult[sl] = mult; __atomic_signal_fence(__ATOMIC_SEQ_CST); __hip_atomic_store(&f_m[sl], e + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        } else if (FORM == 4) {
          s_mult[sl] = mult; __atomic_signal_fence(__ATOMIC_SEQ_CST); __hip_atomic_store(&f_m[sl], e + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[0] = w0; out[1] = s_mult[5] + f_m[7]; ticks[0] = t1 - t0; }
}
// FORM 5: as 4, nothing of the chain leaves the vector registers: the right sign's 1 / d comes from the neighbouring lane by DPP (quad_perm 1 0 3 2) and a select on a mask
// the scalar unit picked long before; the sign of the multiplier is a scalar operand picked the same way; F of the NEXT example is read (v_readlane) a step ahead.
template <int FORM, int NB>
__global__ __launch_bounds__(64) void chain5_k(const double* __restrict__ rs, const float* __restrict__ ys, double lr, double reg0, double* out, unsigned long long* ticks) {
  __shared__ double s_mult[R];
  __shared__ int f_m[R];
  const int lane = threadIdx.x;
  double w0 = 0.01, rw = reg0 * w0, E = 1.0;
  const bool odd = lane & 1;
  const double lr_l = odd ? lr : -lr;
  int small_steps = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int e = 0; e + NB <= N; e += NB) {
    const int n = NB;
    const double r = lane < n ? rs[e + lane] : 0.0;
    const float y = lane < n ? ys[e + lane] : 0.f;
    const double f = seq_exp_re(fmin(fmax((double)y * r, -350.0), 350.0));
    const unsigned long long pm1 = __ballot(lane < n && (y == 1.0f || y == -1.0f)), pos = __ballot(y > 0.f);
    int sl = e % R;
    double Fi = bcast(f, 0);
    for (int i = 0; i < n; ++i, sl = sl + 1 < R ? sl + 1 : 0) {
      if (((e + i) & 31) == 0) {
        const double wc = fmin(fmax(w0, -350.0), 350.0);
        E = seq_exp_re(odd ? -wc : wc);
        small_steps = __ballot(lr * (1.0 + fabs(reg0) * (fabs(w0) + 1.0)) <= 0.03125) != 0ull;
      }
      const double Fn = bcast(f, i + 1 < n ? i + 1 : i);
      double mult;
      if (FORM == 6 || (small_steps && ((pm1 >> i) & 1ull))) {
        const bool up = (pos >> i) & 1ull;
        const double xr_l = seq_rcp_re(__builtin_fma(E, Fi, 1.0));
        const double xr_n = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(xr_l), 0xB1, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, __double2loint(xr_l), 0xB1, 0xF, 0xF, false));
        const double xr = (up ? !odd : odd) ? xr_l : xr_n;
        const double sgn = up ? -1.0 : 1.0;
        mult = sgn * xr;
        const double g = __builtin_fma(sgn, xr, rw);
        w0 -= lr * g;
        rw = reg0 * w0;
        E *= seq_exp_small1(lr_l * g);
      } else {
        const double yd = (double)bcast(y, i), a = yd * (w0 + bcast(r, i));
        const double xr = seq_rcp_re(1.0 + seq_exp_re(fmin(fmax(a, -750.0), 700.0)));
        mult = -yd * (a != a ? a : xr);
        w0 -= lr * (mult + reg0 * w0);
        rw = reg0 * w0;
        if (small_steps) { const double wc = fmin(fmax(w0, -350.0), 350.0); E = seq_exp_re(odd ? -wc : wc); }
      }
      Fi = Fn;
      s_mult[sl] = mult; __atomic_signal_fence(__ATOMIC_SEQ_CST); __hip_atomic_store(&f_m[sl], e + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[0] = w0; out[1] = s_mult[5] + f_m[7]; ticks[0] = t1 - t0; }
}
template <int FORM, int NB> void run5(const char* what, const double* r, const float* y) {
  double* out; unsigned long long *t, h; double ho[2];
  CK(hipMalloc(&out, 16)); CK(hipMalloc(&t, 8));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((chain5_k<FORM, NB>), dim3(1), dim3(64), 0, 0, r, y, 0.01, 1e-4, out, t); CK(hipDeviceSynchronize()); }
  CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost));
  printf("%-100s %7.1f clocks per example   (w0 %.17g)\n", what, (double)h / N, ho[0]);
}
template <int FORM> void run(const char* what, const double* r, const float* y) {
  double* out; unsigned long long *t, h; double ho[2];
  CK(hipMalloc(&out, 16)); CK(hipMalloc(&t, 8));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((chain_k<FORM>), dim3(1), dim3(64), 0, 0, r, y, 0.01, 1e-4, out, t); CK(hipDeviceSynchronize()); }
  CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost));
  printf("%-100s %7.1f clocks per example   (w0 %.17g)\n", what, (double)h / N, ho[0]);
}
int main() {
  double* hr = (double*)malloc(N * 8); float* hy = (float*)malloc(N * 4);
  srand(7);
  for (int i = 0; i < N; ++i) { hr[i] = (rand() / (double)RAND_MAX - 0.5) * 2.0; hy[i] = rand() & 1 ? 1.f : -1.f; }
  double* r; float* y;
  CK(hipMalloc(&r, N * 8)); CK(hipMalloc(&y, N * 4));
  CK(hipMemcpy(r, hr, N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(y, hy, N * 4, hipMemcpyHostToDevice));
  run<0>("0: the whole exponential on the chain, multiplier handed over by lane i", r, y);
  run<1>("1: exponential split, the two signs in the even / odd lanes, scalar control, handed over by lane 0", r, y);
  run<2>("2: ... without the hand-over", r, y);
  run<3>("3: ... with the hand-over, without the data-dependent branches", r, y);
  run<4>("4: ... handed over by all lanes", r, y);
  run5<5, 16>("5: 1 with the neighbour's 1 / d by DPP, scalar picks made early, F read a step ahead, handed over by all lanes", r, y);
  run5<6, 16>("6: ... without the data-dependent branches", r, y);
  run5<6, 5>("6 in batches of 5 examples", r, y);
  run5<6, 2>("6 in batches of 2 examples", r, y);
  run5<6, 1>("6 in batches of 1 example", r, y);
  return 0;
}
