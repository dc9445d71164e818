// Error state, device selection and the device-wide exclusive scan used by the
// coupling-build kernels.  gfx950 only.
#include "asp_common.hpp"

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

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
std::atomic<int> g_touched{0};  // 1 once this process has made a HIP call through the library
}

void remember_device(int device) { g_device.store(device); }
int chosen_device() { return g_device.load(); }

int bind_device() {
  g_touched.store(1);
  const int device = g_device.load();
  if (device >= 0) ASP_HIP_TRY(hipSetDevice(device));
  return ASP_OK;
}

int device_touched() { return g_touched.load(); }

int require_device() {
  g_touched.store(1);
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
// Device memory pool
// ---------------------------------------------------------------------------

namespace {

struct PoolBlock {
  size_t bytes;
  int device;
};

struct Pool {
  std::mutex mutex;
  std::unordered_map<void *, PoolBlock> live;                 // handed out
  std::map<std::pair<int, size_t>, std::vector<void *>> idle;  // (device, class) -> blocks
  size_t idle_bytes = 0;
  // of 288 GB: sixteen pipeline threads each recycle ~0.3 GB of buffers, and a batched shuffled
  // anneal three sets of up to 16 GB of visiting orders per round of the pipeline — with a cap of
  // 16 GB those were given back to and taken from the driver every round (0.2 s per batch call)
  size_t cap = 64ull << 30;
  Pool() {
    if (const char *env = std::getenv("ASP_POOL_BYTES")) cap = std::strtoull(env, nullptr, 10);
  }
};

Pool &pool() {
  static Pool *p = new Pool;  // never destroyed: no HIP calls from static destructors
  return *p;
}

// Smallest of {4, 5, 6, 7} * 2^k (>= 512) that holds `bytes`.
size_t size_class(size_t bytes) {
  size_t base = 128;
  while (base * 7 < bytes) base <<= 1;
  for (size_t q = 4; q <= 7; ++q) {
    if (base * q >= bytes) return base * q;
  }
  return base * 8;
}

}  // namespace

int pool_alloc(size_t bytes, void **out) {
  Pool &p = pool();
  int device = 0;
  ASP_HIP_TRY(hipGetDevice(&device));
  const size_t cls = size_class(bytes);
  {
    std::lock_guard<std::mutex> lock(p.mutex);
    auto it = p.idle.find({device, cls});
    if (it != p.idle.end() && !it->second.empty()) {
      void *ptr = it->second.back();
      it->second.pop_back();
      p.idle_bytes -= cls;
      p.live[ptr] = PoolBlock{cls, device};
      *out = ptr;
      return ASP_OK;
    }
  }
  void *ptr = nullptr;
  hipError_t e = hipMalloc(&ptr, cls);
  if (e != hipSuccess) {
    // give the idle blocks of this device back and try once more
    std::vector<void *> drop;
    {
      std::lock_guard<std::mutex> lock(p.mutex);
      for (auto &kv : p.idle) {
        if (kv.first.first != device) continue;
        p.idle_bytes -= kv.first.second * kv.second.size();
        drop.insert(drop.end(), kv.second.begin(), kv.second.end());
        kv.second.clear();
      }
    }
    for (void *d : drop) (void)hipFree(d);
    e = hipMalloc(&ptr, cls);
  }
  if (e != hipSuccess) {
    return set_error(ASP_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", cls, hipGetErrorString(e));
  }
  std::lock_guard<std::mutex> lock(p.mutex);
  p.live[ptr] = PoolBlock{cls, device};
  *out = ptr;
  return ASP_OK;
}

void pool_free(void *ptr) {
  if (!ptr) return;
  Pool &p = pool();
  PoolBlock block{0, 0};
  {
    std::lock_guard<std::mutex> lock(p.mutex);
    auto it = p.live.find(ptr);
    if (it == p.live.end()) {  // not ours (cannot happen): release directly
      (void)hipFree(ptr);
      return;
    }
    block = it->second;
    p.live.erase(it);
    if (p.idle_bytes + block.bytes <= p.cap) {
      p.idle[{block.device, block.bytes}].push_back(ptr);
      p.idle_bytes += block.bytes;
      return;
    }
  }
  (void)hipFree(ptr);  // hipFree accepts a pointer of any device
}

namespace {

struct StreamPool {
  std::mutex mutex;
  std::map<int, std::vector<hipStream_t>> idle;  // device -> streams
  std::unordered_map<hipStream_t, int> device_of;
  struct Limits {
    int num_cus;
    size_t max_lds;
  };
  std::map<int, Limits> limits;
  bool closed = false;  // after asp_shutdown: released streams are destroyed, not kept
};

StreamPool &stream_pool() {
  static StreamPool *p = new StreamPool;
  return *p;
}

}  // namespace

int stream_acquire(hipStream_t *out) {
  StreamPool &p = stream_pool();
  int device = 0;
  ASP_HIP_TRY(hipGetDevice(&device));
  {
    std::lock_guard<std::mutex> lock(p.mutex);
    auto &idle = p.idle[device];
    if (!idle.empty()) {
      *out = idle.back();
      idle.pop_back();
      return ASP_OK;
    }
  }
  hipStream_t stream = nullptr;
  ASP_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  std::lock_guard<std::mutex> lock(p.mutex);
  p.device_of[stream] = device;
  *out = stream;
  return ASP_OK;
}

void stream_release(hipStream_t stream) {
  if (!stream) return;
  StreamPool &p = stream_pool();
  std::lock_guard<std::mutex> lock(p.mutex);
  auto it = p.device_of.find(stream);
  if (it == p.device_of.end()) {
    (void)hipStreamDestroy(stream);
    return;
  }
  auto &idle = p.idle[it->second];
  if (!p.closed && idle.size() < 64) {
    idle.push_back(stream);
  } else {
    p.device_of.erase(it);
    (void)hipStreamDestroy(stream);
  }
}

int shutdown_pools() {
  // wait for everything this process queued on the devices it used, then give back what the
  // library keeps between calls; blocks and streams still handed out stay with their owners
  std::vector<void *> blocks;
  std::vector<std::pair<int, hipStream_t>> streams;
  {
    Pool &p = pool();
    std::lock_guard<std::mutex> lock(p.mutex);
    for (auto &kv : p.idle) {
      blocks.insert(blocks.end(), kv.second.begin(), kv.second.end());
      kv.second.clear();
    }
    p.idle_bytes = 0;
    p.cap = 0;  // blocks released after the shutdown are freed directly
  }
  {
    StreamPool &sp = stream_pool();
    std::lock_guard<std::mutex> lock(sp.mutex);
    sp.closed = true;
    for (auto &kv : sp.idle) {
      for (hipStream_t s : kv.second) {
        streams.emplace_back(kv.first, s);
        sp.device_of.erase(s);
      }
      kv.second.clear();
    }
  }
  int current = 0;
  const bool have_device = hipGetDevice(&current) == hipSuccess;
  for (auto &ds : streams) {
    if (hipSetDevice(ds.first) != hipSuccess) continue;
    (void)hipStreamSynchronize(ds.second);
    (void)hipStreamDestroy(ds.second);
  }
  if (have_device) (void)hipSetDevice(current);
  for (void *b : blocks) (void)hipFree(b);
  return ASP_OK;
}

int device_limits(int *num_cus, size_t *max_lds) {
  StreamPool &p = stream_pool();
  int device = 0;
  ASP_HIP_TRY(hipGetDevice(&device));
  {
    std::lock_guard<std::mutex> lock(p.mutex);
    auto it = p.limits.find(device);
    if (it != p.limits.end()) {
      *num_cus = it->second.num_cus;
      *max_lds = it->second.max_lds;
      return ASP_OK;
    }
  }
  hipDeviceProp_t prop;
  ASP_HIP_TRY(hipGetDeviceProperties(&prop, device));
  StreamPool::Limits l{256, 160 * 1024};
  if (prop.multiProcessorCount > 0) l.num_cus = prop.multiProcessorCount;
  if (prop.sharedMemPerBlockOptin > 0) {
    l.max_lds = prop.sharedMemPerBlockOptin;
  } else if (prop.maxSharedMemoryPerMultiProcessor > 0) {
    l.max_lds = prop.maxSharedMemoryPerMultiProcessor;
  }
  std::lock_guard<std::mutex> lock(p.mutex);
  p.limits[device] = l;
  *num_cus = l.num_cus;
  *max_lds = l.max_lds;
  return ASP_OK;
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

int asp_device_touched(void) { return asp::device_touched(); }

int asp_device_count(void) {
  asp::require_device();  // (marks the process as having used HIP)
  asp_clear_error();
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

int asp_get_device(void) {
  const int chosen = asp::chosen_device();
  if (chosen >= 0) return chosen;
  int device = 0;
  ASP_HIP_TRY(hipGetDevice(&device));
  return device;
}

int asp_shutdown(void) {
  if (!asp::device_touched()) return ASP_OK;  // nothing was ever used: no HIP call at exit either
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ASP_OK;
  return asp::shutdown_pools();
}

const char *asp_version(void) { return "0.4.0"; }

}  // extern "C"
