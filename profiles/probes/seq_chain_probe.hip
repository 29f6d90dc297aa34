// What bounds the reference-order learner (FMX_MODE_SEQUENTIAL)?  In fm_seq_pipe_k the gathers (A) and scatters (C) of conflict-free examples already run on
// other waves beside the scalar chain (S): pred = ((w0 + t_1) + t_2) + ... in the reference's association (core/Model.h:77-100: w0 enters FIRST, so no part of
// the sum can be formed before the previous example's w0 step), the gradient multiplier (exp, divide), the w0 step (solver/SGD_Learner.h:100-112).  This probe
// runs S ALONE: one wave, the terms of every example already in LDS, nothing to gather, nothing to wait for -- the rate no number of producer workgroups can
// beat while the results stay bitwise the one-wave kernel's.
//   build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o profiles/probes/bin/seq_chain_probe profiles/probes/seq_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int TERMS, int MODE>   // MODE 0: the whole chain; 1: the adds only; 2: adds + multiplier, no w0 step
__global__ __launch_bounds__(64) void chain_k(const double* __restrict__ t_in, int groups, int G, double lr, double reg0, double* __restrict__ out) {
  __shared__ double terms[8][TERMS];
  __shared__ float s_y[8];
  __shared__ double s_mult[8];
  for (int i = threadIdx.x; i < 8 * TERMS; i += 64) terms[i / TERMS][i % TERMS] = t_in[i];
  if (threadIdx.x < 8) s_y[threadIdx.x] = (threadIdx.x & 1) ? 1.0f : -1.0f;
  __syncthreads();
  double w0 = 0.01;
  for (int g = 0; g < groups; ++g) {
    double cb0[8], cb1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cb0[i] = terms[0][i];
    for (int e = 0; e < G; ++e) {
      const double* __restrict__ T = terms[e];
      const double* __restrict__ nextT = terms[e + 1 < G ? e + 1 : e];
      double pred = w0;
#pragma unroll
      for (int u = 0; u < TERMS; u += 16) {   // as fm_seq_pipe_k: the addends fetched in eights, one eight ahead
        const double* second = T + u + 8;
        const double* after = u + 16 < TERMS ? T + u + 16 : nextT;
#pragma unroll
        for (int i = 0; i < 8; ++i) cb1[i] = second[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) pred += cb0[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) cb0[i] = after[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) pred += cb1[i];
      }
      double mult = pred;
      if (MODE != 1) { const double y = (double)s_y[e]; mult = -y * (1.0 - 1.0 / (1.0 + exp(-y * pred))); }   // SGD_Learner.h:187-190
      if (MODE == 0) w0 -= lr * (mult + reg0 * w0); else w0 += 1e-9 * mult;
      if (threadIdx.x == 0) s_mult[e] = mult;
    }
  }
  if (threadIdx.x == 0) out[0] = w0 + s_mult[0];
}

int main() {
  constexpr int TERMS = 48;   // 32 padded w x terms + 16 factor terms: configs[1]'s shape (30 entries per row, k = 16)
  double h[8 * TERMS];
  for (int i = 0; i < 8 * TERMS; ++i) h[i] = 1e-3 * ((i * 7919) % 101 - 50);
  double *d_t, *d_o;
  CK(hipMalloc(&d_t, sizeof(h))); CK(hipMalloc(&d_o, 8)); CK(hipMemcpy(d_t, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int groups = 50000, G = 8;
  auto run = [&](const char* name, auto kern) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d_t, 1000, G, 0.01, 1e-4, d_o); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d_t, groups, G, 0.01, 1e-4, d_o);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double per = ms * 1e-3 / ((double)groups * G);
    printf("%-64s %7.1f ns per example = %6.2f M examples/s\n", name, per * 1e9, 1e-6 / per);
  };
  run("the chain alone (48 ordered adds, exp, divide, w0 step)", chain_k<TERMS, 0>);
  run("  its 48 ordered fp64 adds only", chain_k<TERMS, 1>);
  run("  adds + gradient multiplier (exp, divide), no w0 recurrence", chain_k<TERMS, 2>);
  return 0;
}
