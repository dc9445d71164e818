// Position of basis states in a large sorted list that stays on the device.
//
// The reference looks amplitudes up with `basis.batched_index(spins)` (common.py:813-818:
// lattice_symmetries' index of a representative) for every model it builds.  With the basis of
// the 36-site kagome sector — 31.5 million representatives — a numpy searchsorted of a 2e5-state
// cluster spends its time in cache misses (6 ms per call, a third of `make kagome_36`'s host
// time); the list lives in HBM once (252 MB) and the bisection of a whole cluster is one short
// kernel.  gfx950 only.

#include <hip/hip_runtime.h>

#include <cstdint>
#include <new>

#include "asp.h"
#include "asp_common.hpp"

struct asp_table {
  uint64_t n = 0;
  asp::DeviceBuffer<uint64_t> d_keys;
};

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void k_table_index(const uint64_t *__restrict__ keys, uint64_t n,
                                                         const uint64_t *__restrict__ queries,
                                                         uint64_t m, int64_t *__restrict__ index) {
  const uint64_t q = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (q >= m) return;
  const uint64_t needle = queries[q];
  uint64_t lo = 0, hi = n;  // keys[lo] <= needle < keys[hi] once the loop ends (if present)
  while (hi - lo > 1) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (keys[mid] <= needle) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  index[q] = (n > 0 && keys[lo] == needle) ? static_cast<int64_t>(lo) : -1;
}

}  // namespace

extern "C" {

int asp_table_create(uint64_t n, uint64_t const *sorted_keys, asp_table **out) {
  asp_clear_error();
  if (!out) return asp::set_error(ASP_ERR_INVALID, "null output pointer");
  *out = nullptr;
  if (n && !sorted_keys) return asp::set_error(ASP_ERR_INVALID, "null keys");
  ASP_TRY(asp::require_device());
  for (uint64_t i = 1; i < n; ++i) {
    if (sorted_keys[i - 1] >= sorted_keys[i]) {
      return asp::set_error(ASP_ERR_INVALID, "keys must be strictly ascending (position %llu)",
                            (unsigned long long)i);
    }
  }
  asp_table *t = new (std::nothrow) asp_table;
  if (!t) return asp::set_error(ASP_ERR_ALLOC, "out of host memory");
  t->n = n;
  int rc = t->d_keys.alloc(n);
  if (rc == ASP_OK) rc = t->d_keys.upload(sorted_keys, n, nullptr);
  if (rc == ASP_OK && hipStreamSynchronize(nullptr) != hipSuccess) {
    rc = asp::set_error(ASP_ERR_HIP, "upload of the key table failed");
  }
  if (rc != ASP_OK) {
    delete t;
    return rc;
  }
  *out = t;
  return ASP_OK;
}

void asp_table_destroy(asp_table *t) {
  if (!t) return;
  (void)asp::bind_device();
  delete t;
}

int asp_table_index(asp_table const *t, uint64_t m, uint64_t const *queries, int64_t *index) {
  asp_clear_error();
  if (!t) return asp::set_error(ASP_ERR_INVALID, "null table");
  ASP_TRY(asp::bind_device());
  if (m == 0) return ASP_OK;
  if (!queries || !index) return asp::set_error(ASP_ERR_INVALID, "null argument");
  asp::ScopedStream scoped;
  ASP_TRY(scoped.acquire());
  hipStream_t s = scoped.stream;
  asp::DeviceBuffer<uint64_t> d_queries;
  asp::DeviceBuffer<int64_t> d_index;
  asp::StreamFence fence(s);
  ASP_TRY(d_queries.alloc(m));
  ASP_TRY(d_index.alloc(m));
  ASP_TRY(d_queries.upload(queries, m, s));
  hipLaunchKernelGGL(k_table_index, dim3(static_cast<unsigned>((m + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, s, t->d_keys.ptr, t->n, d_queries.ptr, m, d_index.ptr);
  ASP_HIP_TRY(hipGetLastError());
  ASP_TRY(d_index.download(index, m, s));
  ASP_HIP_TRY(hipStreamSynchronize(s));
  return ASP_OK;
}

}  // extern "C"
