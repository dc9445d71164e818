// Error state, device selection and the device-wide exclusive scan used by the
// coupling-build kernels.  gfx950 only.
#include "asp_common.hpp"

#include <atomic>

namespace asp {

ErrorState &error_state() {
  static thread_local ErrorState state;
  return state;
}

int set_error(int code, const char *fmt, ...) {
  ErrorState &s = error_state();
  s.code = code;
  va_list args;
  va_start(args, fmt);
  vsnprintf(s.message, sizeof s.message, fmt, args);
  va_end(args);
  return code;
}

namespace {
std::atomic<int> g_device{-1};  // process-wide choice of asp_set_device; -1 = HIP's default
}

void remember_device(int device) { g_device.store(device); }

int bind_device() {
  const int device = g_device.load();
  if (device >= 0) ASP_HIP_TRY(hipSetDevice(device));
  return ASP_OK;
}

int require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    return set_error(ASP_ERR_NO_DEVICE,
                     "no HIP device available (%s); libasp_hip has no CPU fallback",
                     e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  }
  return bind_device();
}

// ---------------------------------------------------------------------------
// Exclusive scan: tile = 256 threads x 8 items.  Three launches: per-tile
// totals, in-place exclusive scan of the totals by one workgroup, per-tile scan
// with the tile's base added.
// ---------------------------------------------------------------------------

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

__device__ __forceinline__ int64_t wave_inclusive_scan(int64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int64_t up = __shfl_up(v, d, 64);
    if (lane >= d) v += up;
  }
  return v;
}

// Exclusive scan of one value per thread across the 256-thread workgroup;
// returns the exclusive prefix and writes the workgroup total to *total.
__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t *total) {
  __shared__ int64_t wave_sums[kScanThreads / 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t incl = wave_inclusive_scan(v, lane);
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  int64_t base = 0;
  int64_t all = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    const int64_t s = wave_sums[w];
    if (w < wave) base += s;
    all += s;
  }
  __syncthreads();
  *total = all;
  return base + incl - v;
}

template <typename Tin>
__global__ __launch_bounds__(kScanThreads) void k_scan_tile_totals(const Tin *__restrict__ in,
                                                                  size_t n,
                                                                  int64_t *__restrict__ totals) {
  const size_t base = static_cast<size_t>(blockIdx.x) * kScanTile;
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const size_t i = base + static_cast<size_t>(k) * kScanThreads + threadIdx.x;
    if (i < n) s += static_cast<int64_t>(in[i]);
  }
  int64_t total;
  (void)block_exclusive_scan(s, &total);
  if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// One workgroup; totals[0..m) -> exclusive prefix in place, totals[m] = sum.
__global__ __launch_bounds__(kScanThreads) void k_scan_totals(int64_t *__restrict__ totals,
                                                             size_t m) {
  int64_t carry = 0;
  for (size_t start = 0; start < m; start += kScanThreads) {
    const size_t i = start + threadIdx.x;
    const int64_t v = i < m ? totals[i] : 0;
    int64_t chunk_total;
    const int64_t excl = block_exclusive_scan(v, &chunk_total);
    if (i < m) totals[i] = carry + excl;
    carry += chunk_total;
  }
  if (threadIdx.x == 0) totals[m] = carry;
}

template <typename Tin>
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const Tin *__restrict__ in, size_t n,
                                                            const int64_t *__restrict__ totals,
                                                            int64_t *__restrict__ out,
                                                            size_t num_tiles) {
  // Thread t owns kScanItems CONSECUTIVE items so the per-thread running sum is
  // in index order.
  const size_t base = static_cast<size_t>(blockIdx.x) * kScanTile +
                      static_cast<size_t>(threadIdx.x) * kScanItems;
  int64_t v[kScanItems];
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const size_t i = base + k;
    v[k] = i < n ? static_cast<int64_t>(in[i]) : 0;
    s += v[k];
  }
  int64_t tile_total;
  int64_t run = totals[blockIdx.x] + block_exclusive_scan(s, &tile_total);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const size_t i = base + k;
    if (i < n) out[i] = run;
    run += v[k];
  }
  if (blockIdx.x == num_tiles - 1 && threadIdx.x == 0) out[n] = totals[num_tiles];
}

__global__ void k_scan_empty(int64_t *out) { out[0] = 0; }

template <typename Tin>
int exclusive_scan_impl(const Tin *in, size_t n, int64_t *out, int64_t *scratch,
                        hipStream_t stream) {
  if (n == 0) {
    hipLaunchKernelGGL(k_scan_empty, dim3(1), dim3(1), 0, stream, out);
    ASP_HIP_TRY(hipGetLastError());
    return ASP_OK;
  }
  const size_t tiles = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(k_scan_tile_totals<Tin>, dim3(static_cast<unsigned>(tiles)),
                     dim3(kScanThreads), 0, stream, in, n, scratch);
  hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(kScanThreads), 0, stream, scratch, tiles);
  hipLaunchKernelGGL(k_scan_apply<Tin>, dim3(static_cast<unsigned>(tiles)), dim3(kScanThreads), 0,
                     stream, in, n, scratch, out, tiles);
  ASP_HIP_TRY(hipGetLastError());
  return ASP_OK;
}

}  // namespace

size_t scan_scratch_elems(size_t n) { return (n + kScanTile - 1) / kScanTile + 2; }

int exclusive_scan_i64(const int64_t *in, size_t n, int64_t *out, int64_t *scratch,
                       hipStream_t stream) {
  return exclusive_scan_impl<int64_t>(in, n, out, scratch, stream);
}
int exclusive_scan_u32(const uint32_t *in, size_t n, int64_t *out, int64_t *scratch,
                       hipStream_t stream) {
  return exclusive_scan_impl<uint32_t>(in, n, out, scratch, stream);
}

}  // namespace asp

// ---------------------------------------------------------------------------
// C ABI: status
// ---------------------------------------------------------------------------

extern "C" {

const char *asp_last_error(void) { return asp::error_state().message; }
int asp_last_error_code(void) { return asp::error_state().code; }
void asp_clear_error(void) {
  asp::error_state().code = ASP_OK;
  asp::error_state().message[0] = 0;
}

int asp_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e == hipErrorNoDevice) return 0;
  if (e != hipSuccess) {
    return asp::set_error(ASP_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  return n;
}

int asp_set_device(int device) {
  ASP_TRY(asp::require_device());
  ASP_HIP_TRY(hipSetDevice(device));
  asp::remember_device(device);
  return ASP_OK;
}

const char *asp_version(void) { return "0.1.0"; }

}  // extern "C"
