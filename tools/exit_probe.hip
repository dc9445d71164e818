// Which HIP feature makes a process die in the runtime's exit handler under rocprofv3
// (SIGSEGV after "tool finalization" in tools/profile_team.py)?  Each mode uses ONE
// feature, releases everything it created, and returns from main normally.
//   hipcc --offload-arch=gfx950 -O2 tools/exit_probe.hip -o /tmp/exit_probe
//   rocprofv3 --kernel-trace --stats -d /tmp/x -- /tmp/exit_probe coop|fine|plain
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

__global__ void k_touch(unsigned *p) {
  if (threadIdx.x == 0) atomicAdd(p, 1u);
}

int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "plain";
  unsigned *p = nullptr;
  if (!strcmp(mode, "fine")) {
    if (hipExtMallocWithFlags(reinterpret_cast<void **>(&p), 4096, hipDeviceMallocFinegrained) != hipSuccess) return 2;
  } else {
    if (hipMalloc(reinterpret_cast<void **>(&p), 4096) != hipSuccess) return 2;
  }
  if (hipMemset(p, 0, 4096) != hipSuccess) return 3;
  hipStream_t s;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 4;
  if (!strcmp(mode, "coop")) {
    void *args[] = {&p};
    if (hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_touch), dim3(64), dim3(64), args, 0, s) != hipSuccess) return 5;
  } else {
    hipLaunchKernelGGL(k_touch, dim3(64), dim3(64), 0, s, p);
  }
  if (hipStreamSynchronize(s) != hipSuccess) return 6;
  unsigned v = 0;
  if (hipMemcpy(&v, p, 4, hipMemcpyDeviceToHost) != hipSuccess) return 7;
  (void)hipStreamDestroy(s);
  (void)hipFree(p);
  printf("%s: counter %u, leaving main\n", mode, v);
  fflush(stdout);
  return 0;
}
