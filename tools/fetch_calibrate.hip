// FETCH_SIZE / WRITE_SIZE calibration for the access widths the sweep kernel uses
// (4 B and 8 B per lane, coalesced), as MI355X_MICROARCH.md §HBM prescribes:
// stream a buffer of known size that is far larger than L2 + Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calibrate.hip -o /tmp/fetch_calibrate
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- /tmp/fetch_calibrate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <typename T>
__global__ void k_stream(const T *__restrict__ in, size_t n, unsigned long long *sink) {
  unsigned long long acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    T v = in[i];
    const unsigned char *p = reinterpret_cast<const unsigned char *>(&v);
    acc += p[0];
  }
  if (acc == 0x123456789abcdefull) *sink = acc;
}

int main() {
  const size_t bytes = 2ull << 30;  // 2 GiB >> 256 MiB Infinity Cache
  void *buf = nullptr;
  unsigned long long *sink = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **)&sink, 8) != hipSuccess) return 1;
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_stream<uint32_t>, dim3(2048), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, sink);
    hipLaunchKernelGGL(k_stream<uint2>, dim3(2048), dim3(256), 0, 0, (const uint2 *)buf, bytes / 8, sink);
    hipLaunchKernelGGL(k_stream<uint4>, dim3(2048), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, sink);
  }
  hipDeviceSynchronize();
  printf("streamed %zu bytes per kernel\n", bytes);
  return 0;
}
