// Which XCD does a workgroup run on?  s_getreg_b32 HW_REG_XCC_ID (id 20, bits 3:0) per one-wave workgroup of a 1024-workgroup grid: the histogram over XCDs and the
// first 32 workgroups' values (the persistent sweep's one-XCD form reads its placement this way).
//   build: hipcc --offload-arch=gfx950 -O3 -o profiles/probes/bin/xcc_id_probe profiles/probes/xcc_id_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
}
int main() {
  const int G = 1024;
  unsigned* d; CK(hipMalloc(&d, G * 4));
  hipLaunchKernelGGL(k, dim3(G), dim3(64), 0, 0, d);
  CK(hipDeviceSynchronize());
  unsigned h[G]; CK(hipMemcpy(h, d, G * 4, hipMemcpyDeviceToHost));
  int hist[16] = {0};
  for (int i = 0; i < G; ++i) hist[h[i] & 15]++;
  printf("raw register of workgroups 0..15:"); for (int i = 0; i < 16; ++i) printf(" %08x", h[i]); printf("\n");
  printf("XCC_ID[3:0] histogram over %d one-wave workgroups:", G); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\n");
  printf("XCC_ID of workgroups 0..31:"); for (int i = 0; i < 32; ++i) printf(" %u", h[i] & 15); printf("\n");
  return 0;
}
