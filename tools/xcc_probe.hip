// Which XCD does a workgroup land on?  (development aid: tools/xcc_probe.hip, run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
  const int n = 256;
  unsigned *d;
  hipMalloc(&d, n * sizeof(unsigned));
  std::vector<unsigned> h(n);
  for (int coop = 0; coop < 2; ++coop) {
    hipMemset(d, 0xFF, n * sizeof(unsigned));
    if (coop) {
      void *args[] = {&d};
      hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k), dim3(n), dim3(1024), args, 100 * 1024, nullptr);
    } else {
      hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
      hipLaunchKernelGGL(k, dim3(n), dim3(1024), 100 * 1024, nullptr, d);
    }
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost);
    printf("%s launch, XCC id of workgroups 0..31:", coop ? "cooperative" : "ordinary");
    for (int i = 0; i < 32; ++i) printf(" %u", h[i]);
    int same = 0;
    for (int i = 0; i < n; ++i) same += (h[i] == h[i % 8]);
    printf("\n  workgroups with XCC(i) == XCC(i mod 8): %d of %d\n", same, n);
  }
  return 0;
}
