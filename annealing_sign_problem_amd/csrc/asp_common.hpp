// Shared host-side plumbing for libasp_hip.so: error recording, HIP call
// checking, device buffers, device-wide exclusive scan.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <new>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "asp.h"

namespace asp {

// ---- error state (thread-local, read through asp_last_error) -------------
struct ErrorState {
  int code = ASP_OK;
  char message[512] = {0};
};
ErrorState &error_state();
int set_error(int code, const char *fmt, ...);

#define ASP_HIP_TRY(expr)                                                              \
  do {                                                                                 \
    hipError_t asp_hip_err_ = (expr);                                                  \
    if (asp_hip_err_ != hipSuccess) {                                                  \
      return ::asp::set_error(asp_hip_err_ == hipErrorNoDevice ? ASP_ERR_NO_DEVICE     \
                                                               : ASP_ERR_HIP,          \
                              "%s failed: %s (%s:%d)", #expr,                          \
                              hipGetErrorString(asp_hip_err_), __FILE__, __LINE__);    \
    }                                                                                  \
  } while (0)

#define ASP_TRY(expr)                    \
  do {                                   \
    int asp_rc_ = (expr);                \
    if (asp_rc_ != ASP_OK) return asp_rc_; \
  } while (0)

// Fails (recording ASP_ERR_NO_DEVICE) unless a device is usable; binds the calling thread
// to the device chosen with asp_set_device (HIP's current device is per thread).
int require_device();
// Only the binding (for entry points that receive an existing handle).
int bind_device();
int device_touched();  // 1 once require_device / bind_device ran in this process
void remember_device(int device);
int chosen_device();  // -1: asp_set_device was never called

// ---- pooled device memory ---------------------------------------------------
// hipMalloc/hipFree cost ~0.1-0.5 ms each and hipFree synchronises the device; a sampled-
// cluster run builds and drops three plans per cluster, tens of thousands of times.  Freed
// blocks are therefore kept per device in size classes (four per octave) up to a cap
// (ASP_POOL_BYTES, default 16 GiB; 0 disables) and handed out again.  Every entry point
// synchronises its stream before returning, so a block is idle when it comes back.
int pool_alloc(size_t bytes, void **out);
void pool_free(void *ptr);

// Non-blocking streams are recycled the same way (creating and destroying one costs more
// than a small plan's kernels); a released stream must be idle.
int stream_acquire(hipStream_t *out);
void stream_release(hipStream_t stream);

// A pooled stream for the duration of one synchronous entry point, so that calls from
// different host threads (sampled_components --jobs) overlap on the GPU instead of queueing
// on the null stream.  The entry point synchronises the stream before it returns.
struct ScopedStream {
  hipStream_t stream = nullptr;
  ScopedStream() = default;
  ScopedStream(const ScopedStream &) = delete;
  ScopedStream &operator=(const ScopedStream &) = delete;
  int acquire() { return stream_acquire(&stream); }
  ~ScopedStream() {
    if (stream) {
      (void)hipStreamSynchronize(stream);
      stream_release(stream);
    }
  }
};

// Declared AFTER the DeviceBuffers of an entry point (objects die in reverse order of
// declaration): waits for the stream first, so that on an early error return no buffer goes
// back to the shared pool while a kernel or an asynchronous copy still uses it.
struct StreamFence {
  hipStream_t stream = nullptr;
  explicit StreamFence(hipStream_t s) : stream(s) {}
  StreamFence(const StreamFence &) = delete;
  StreamFence &operator=(const StreamFence &) = delete;
  ~StreamFence() {
    if (stream) (void)hipStreamSynchronize(stream);
  }
};

// Releases what the library keeps alive between calls (idle pool blocks, pooled streams) after
// waiting for the device; see asp_shutdown in asp.h.
int shutdown_pools();

// Compute units and opt-in LDS bytes per workgroup of the current device (queried once).
int device_limits(int *num_cus, size_t *max_lds);

// ---- owning device buffer --------------------------------------------------
template <typename T>
struct DeviceBuffer {
  T *ptr = nullptr;
  size_t count = 0;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  ~DeviceBuffer() { release(); }
  void release() {
    if (ptr) pool_free(ptr);
    ptr = nullptr;
    count = 0;
  }
  // (Re)allocate for n elements (at least one, so pointers are never null).
  int alloc(size_t n) {
    release();
    size_t bytes = (n ? n : 1) * sizeof(T);
    void *raw = nullptr;
    const int rc = pool_alloc(bytes, &raw);
    if (rc != ASP_OK) {
      ptr = nullptr;
      return rc;
    }
    ptr = static_cast<T *>(raw);
    count = n;
    return ASP_OK;
  }
  // Grow-only: keep the allocation when it already holds n elements (no hipFree, which
  // synchronises the whole device, on the hot path of repeated calls).
  int ensure(size_t n) {
    if (ptr != nullptr && count >= n) return ASP_OK;
    return alloc(n);
  }
  int upload(const T *host, size_t n, hipStream_t stream) {
    if (n == 0) return ASP_OK;
    ASP_HIP_TRY(hipMemcpyAsync(ptr, host, n * sizeof(T), hipMemcpyHostToDevice, stream));
    return ASP_OK;
  }
  int download(T *host, size_t n, hipStream_t stream) const {
    if (n == 0) return ASP_OK;
    ASP_HIP_TRY(hipMemcpyAsync(host, ptr, n * sizeof(T), hipMemcpyDeviceToHost, stream));
    return ASP_OK;
  }
};

// ---- device-wide exclusive scan -------------------------------------------
// out[i] = sum_{j<i} in[j] for i in [0, n]; out has n + 1 elements (out[n] is
// the total).  `scratch` must hold scan_scratch_elems(n) int64.  Integer adds
// only, so the result is independent of the launch geometry.
size_t scan_scratch_elems(size_t n);
int exclusive_scan_i64(const int64_t *in, size_t n, int64_t *out, int64_t *scratch,
                       hipStream_t stream);
int exclusive_scan_u32(const uint32_t *in, size_t n, int64_t *out, int64_t *scratch,
                       hipStream_t stream);

}  // namespace asp
