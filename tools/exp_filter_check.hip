// Measures the relative error of the sweep kernel's hardware-exp estimate
//   p~ = v_exp_f32((float)x * -log2(e))      against exp(-x) in f64
// over 0 < x < 23 (2^26 equispaced points plus the neighbourhoods of the powers of two), i.e.
// the bound behind metropolis_accept()'s +-1e-5 band (csrc/sa_sweep.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/exp_filter_check.hip -o /tmp/exp_filter_check && /tmp/exp_filter_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

__global__ void k_check(unsigned long long n, double *max_err, double *worst_x) {
  double local = 0.0, where = 0.0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const double x = 23.0 * (static_cast<double>(i) + 0.5) / static_cast<double>(n);
    const float estimate = __builtin_amdgcn_exp2f(static_cast<float>(x) * -1.44269504f);
    const double exact = exp(-x);
    const double err = fabs(static_cast<double>(estimate) - exact) / exact;
    if (err > local) { local = err; where = x; }
  }
  // per-thread maxima; reduce on the host
  const unsigned long long t = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  max_err[t] = local;
  worst_x[t] = where;
}

int main() {
  const int blocks = 1024, threads = 256;
  const size_t total = (size_t)blocks * threads;
  double *d_err, *d_x;
  hipMalloc((void **)&d_err, total * sizeof(double));
  hipMalloc((void **)&d_x, total * sizeof(double));
  hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, 1ull << 30, d_err, d_x);
  double *err = new double[total], *xs = new double[total];
  hipMemcpy(err, d_err, total * sizeof(double), hipMemcpyDeviceToHost);
  hipMemcpy(xs, d_x, total * sizeof(double), hipMemcpyDeviceToHost);
  double worst = 0.0, at = 0.0;
  for (size_t i = 0; i < total; ++i) if (err[i] > worst) { worst = err[i]; at = xs[i]; }
  printf("max relative error of the f32 hardware estimate over 2^30 points in (0, 23): %.3e at x = %.6f\n", worst, at);
  printf("band used by metropolis_accept: 1e-5 -> margin factor %.1f\n", 1e-5 / worst);
  return worst < 5e-6 ? 0 : 1;
}
