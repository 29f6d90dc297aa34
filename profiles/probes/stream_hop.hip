// How long does a dependency between two HIP streams take?  Ping-pong: kernel on A, event, B waits, kernel on B, event, A waits ...
// hipcc --offload-arch=gfx950 -O2 profiles/probes/stream_hop.hip -o /tmp/stream_hop && /tmp/stream_hop
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(int* p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
__global__ void busy(int* p, int iters) { int x = 0; for (int i = 0; i < iters; ++i) x += __builtin_amdgcn_s_memtime() & 1; if (threadIdx.x == 0 && x == -1) *p = x; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int* d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
  for (int flags : {0, 1}) {            // event flags: 0 default, 1 disable timing
    for (int sflag : {0, 1}) {          // streams: 0 default flags, 1 non-blocking
      hipStream_t A, B;
      CK(hipStreamCreateWithFlags(&A, sflag ? hipStreamNonBlocking : hipStreamDefault));
      CK(hipStreamCreateWithFlags(&B, sflag ? hipStreamNonBlocking : hipStreamDefault));
      const int N = 200;
      std::vector<hipEvent_t> ev(2 * N);
      for (auto& e : ev) CK(hipEventCreateWithFlags(&e, flags ? hipEventDisableTiming : hipEventDefault));
      // baseline: 2N tiny kernels on ONE stream
      CK(hipDeviceSynchronize());
      double t0 = now();
      for (int i = 0; i < 2 * N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, A, d);
      CK(hipStreamSynchronize(A));
      const double one = (now() - t0) / (2 * N);
      // ping-pong
      t0 = now();
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, A, d);
        CK(hipEventRecord(ev[2 * i], A));
        CK(hipStreamWaitEvent(B, ev[2 * i], 0));
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, B, d);
        CK(hipEventRecord(ev[2 * i + 1], B));
        CK(hipStreamWaitEvent(A, ev[2 * i + 1], 0));
      }
      CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
      const double hop = (now() - t0) / (2 * N);
      // fork/join around work: A: busy(100us) ; fork B: tiny ; join ; repeated
      t0 = now();
      for (int i = 0; i < N; ++i) {
        CK(hipEventRecord(ev[2 * i], A));
        CK(hipStreamWaitEvent(B, ev[2 * i], 0));
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, B, d);
        CK(hipEventRecord(ev[2 * i + 1], B));
        hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, A, d, 2000);
        CK(hipStreamWaitEvent(A, ev[2 * i + 1], 0));
      }
      CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
      const double fj = (now() - t0) / N;
      t0 = now();
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, A, d, 2000);
      CK(hipStreamSynchronize(A));
      const double alone = (now() - t0) / N;
      // the same fork/join with the SAME two events re-recorded every iteration
      t0 = now();
      for (int i = 0; i < N; ++i) {
        CK(hipEventRecord(ev[0], A));
        CK(hipStreamWaitEvent(B, ev[0], 0));
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, B, d);
        CK(hipEventRecord(ev[1], B));
        hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, A, d, 2000);
        CK(hipStreamWaitEvent(A, ev[1], 0));
      }
      CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
      const double fj2 = (now() - t0) / N;
      printf("   (two events reused: %.1f us) ", fj2 * 1e6);
      printf("event flags %d stream flags %d: kernel on one stream %.1f us, per hop between streams %.1f us, busy kernel alone %.1f us, with a fork+join beside it %.1f us\n",
             flags, sflag, one * 1e6, hop * 1e6, alone * 1e6, fj * 1e6);
      for (auto& e : ev) CK(hipEventDestroy(e));
      CK(hipStreamDestroy(A)); CK(hipStreamDestroy(B));
    }
  }
  return 0;
}
