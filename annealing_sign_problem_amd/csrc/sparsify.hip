// Global-cutoff sparsification and component extraction on gfx950: the reference's
//   _sparsify_using_global_cutoff                 common.py:634-643   (numba)
//   0.5 * (M + M^T); eliminate_zeros              common.py:660-662   (scipy)
//   connected_components(directed=False)          common.py:664       (scipy.sparse.csgraph)
//   exchange[mask][:, mask]                       common.py:674       (scipy fancy indexing)
// in one call, CSR in and CSR out.
//
// HBM layout: indptr i64[K+1], indices i32[nnz], data f64[nnz] (rows sorted by column), frozen
// u8[K]; parent u32[K] (union-find forest), root u32[K]; keep u32[K]; new_index i64[K+1] (scan of keep);
// row_kept u32[K] -> out_indptr (scan); out_indices i32[], out_data f64[].
// Kernels (one stream): k_abs_max (exact: max is order-independent) -> k_hook_edges (one
// wavefront per row; an edge (i, j) exists iff 0.5 * (M'_ij + M'_ji) != 0 with M' the pruned
// matrix, M'_ji found by binary search in row j; lock-free union by atomicCAS, larger root under
// smaller, so the final root of a component is its smallest spin — deterministic) -> k_flatten ->
// k_mark_component (+ check that every frozen spin is in it) -> scan -> k_slice_rows<count> ->
// scan -> k_slice_rows<emit>.  Integer/latency-bound streaming over the non-zeros: 12 B/non-zero in, 12 B per
// kept non-zero out; no MFMA.
#include <algorithm>
#include <cstring>
#include <vector>

#include "asp_common.hpp"

namespace {

using asp::DeviceBuffer;
constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

__device__ __forceinline__ unsigned long long double_bits_abs(double x) {
  return static_cast<unsigned long long>(__double_as_longlong(x)) & 0x7FFFFFFFFFFFFFFFull;
}

// max |data| : the bit pattern of |x| orders like |x| for finite values.
__global__ __launch_bounds__(kThreads) void k_abs_max(const double *__restrict__ data, uint64_t nnz,
                                                     unsigned long long *__restrict__ out) {
  unsigned long long best = 0;
  for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x; k < nnz;
       k += static_cast<uint64_t>(gridDim.x) * kThreads) {
    const unsigned long long b = double_bits_abs(data[k]);
    best = b > best ? b : best;
  }
#pragma unroll
  for (int step = 1; step < 64; step <<= 1) {
    const unsigned long long other = __shfl_xor(best, step, 64);
    best = other > best ? other : best;
  }
  if ((threadIdx.x & 63) == 0 && best != 0) atomicMax(out, best);
}

struct GraphArgs {
  const int64_t *indptr;
  const int32_t *indices;
  const double *data;
  const uint8_t *frozen;
  uint32_t *parent;
  uint64_t num_spins;
  double threshold;  // reltol * max|J|
};

// The pruned element M'_ij (common.py:634-643): zero when weak, unless both ends are frozen.
__device__ __forceinline__ double pruned(const GraphArgs &a, uint32_t i, uint32_t j, double value) {
  if (a.frozen[i] && a.frozen[j]) return value;
  return fabs(value) < a.threshold ? 0.0 : value;
}

// M'_ji by binary search in row j (0 when the matrix has no such element).
__device__ __forceinline__ double pruned_transposed(const GraphArgs &a, uint32_t i, uint32_t j) {
  int64_t lo = a.indptr[j], hi = a.indptr[j + 1];
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (static_cast<uint32_t>(a.indices[mid]) < i) {
      lo = mid + 1;
    } else {
      hi = mid;
    }
  }
  if (lo < a.indptr[j + 1] && static_cast<uint32_t>(a.indices[lo]) == i) {
    return pruned(a, j, i, a.data[lo]);
  }
  return 0.0;
}

__device__ __forceinline__ uint32_t find_root(uint32_t *parent, uint32_t x) {
  uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != x) {
    const uint32_t g = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (g != p) {  // path halving: only ever replaces a parent by one of its ancestors
      __hip_atomic_store(&parent[x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    x = p;
    p = g;
  }
  return x;
}

__device__ __forceinline__ void unite(uint32_t *parent, uint32_t a, uint32_t b) {
  a = find_root(parent, a);
  b = find_root(parent, b);
  while (a != b) {
    if (a < b) {
      const uint32_t t = a;
      a = b;
      b = t;
    }  // a > b: hang the larger root under the smaller
    const uint32_t seen = atomicCAS(&parent[a], a, b);
    if (seen == a) return;
    a = find_root(parent, seen);  // somebody re-rooted a meanwhile
    b = find_root(parent, b);
  }
}

__global__ __launch_bounds__(kThreads) void k_init_parent(uint32_t *parent, uint64_t n) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) parent[i] = static_cast<uint32_t>(i);
}

__global__ __launch_bounds__(kThreads) void k_hook_edges(GraphArgs a) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t row = static_cast<uint64_t>(blockIdx.x) * kWaves + (threadIdx.x >> 6);
  if (row >= a.num_spins) return;
  const uint32_t i = static_cast<uint32_t>(row);
  const int64_t begin = a.indptr[i], end = a.indptr[i + 1];
  for (int64_t k = begin + lane; k < end; k += 64) {
    const uint32_t j = static_cast<uint32_t>(a.indices[k]);
    if (j == i) continue;
    const double mine = pruned(a, i, j, a.data[k]);
    const double theirs = pruned_transposed(a, i, j);
    // scipy: entry of M' + M'^T kept iff the sum is non-zero, scaled by 0.5, zeros eliminated
    if (__dmul_rn(0.5, __dadd_rn(mine, theirs)) != 0.0) unite(a.parent, i, j);
  }
}

// Roots into a SEPARATE array with a read-only walk: compressing in place here would let one
// thread's path halving overwrite the root another thread has just stored.
__global__ __launch_bounds__(kThreads) void k_flatten(const uint32_t *__restrict__ parent,
                                                     uint32_t *__restrict__ root, uint64_t n) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  uint32_t x = static_cast<uint32_t>(i);
  for (uint32_t p = parent[x]; p != x; p = parent[x]) x = p;
  root[i] = x;
}

// keep[i] = same component as the anchor; stray[0] counts frozen spins outside it.
__global__ __launch_bounds__(kThreads) void k_mark_component(const uint32_t *__restrict__ parent,
                                                            const uint8_t *__restrict__ frozen,
                                                            uint64_t n, uint64_t anchor,
                                                            uint32_t *__restrict__ keep,
                                                            uint32_t *__restrict__ stray) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  const bool in = parent[i] == parent[anchor];  // `parent` is k_flatten's root array here
  keep[i] = in ? 1u : 0u;
  if (frozen[i] && !in) atomicAdd(stray, 1u);
}

struct SliceArgs {
  const int64_t *indptr;
  const int32_t *indices;
  const double *data;
  const uint32_t *keep;
  const int64_t *new_index;  // exclusive scan of keep
  uint32_t *row_kept;        // per NEW row
  const int64_t *out_indptr;
  int32_t *out_indices;
  double *out_data;
  uint64_t num_spins;
};

template <bool EMIT>
__global__ __launch_bounds__(kThreads) void k_slice_rows(SliceArgs a) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t row = static_cast<uint64_t>(blockIdx.x) * kWaves + (threadIdx.x >> 6);
  if (row >= a.num_spins || !a.keep[row]) return;  // whole wavefront
  const int64_t new_row = a.new_index[row];
  const int64_t begin = a.indptr[row], end = a.indptr[row + 1];
  int64_t out = EMIT ? a.out_indptr[new_row] : 0;
  uint32_t total = 0;
  for (int64_t base = begin; base < end; base += 64) {
    const int64_t k = base + lane;
    const bool live = k < end && a.keep[a.indices[k]];
    const uint64_t votes = __ballot(live);
    if (EMIT && live) {
      const int64_t at = out + __popcll(votes & ((1ull << lane) - 1ull));
      a.out_indices[at] = static_cast<int32_t>(a.new_index[a.indices[k]]);
      a.out_data[at] = a.data[k];
    }
    const uint32_t here = static_cast<uint32_t>(__popcll(votes));
    out += here;
    total += here;
  }
  if (!EMIT && lane == 0) a.row_kept[new_row] = total;
}

unsigned grid_for(uint64_t items, uint64_t per_block) {
  return static_cast<unsigned>((items + per_block - 1) / per_block);
}

thread_local float g_last_ms = 0.0f;

}  // namespace

extern "C" float asp_sparsify_last_ms(void) { return g_last_ms; }

extern "C" int asp_sparsify_component(uint64_t num_spins, int64_t const *indptr,
                                      int32_t const *indices, double const *data,
                                      uint8_t const *is_frozen, double reltol, uint64_t anchor,
                                      uint8_t *keep, uint64_t *kept_spins, uint64_t capacity,
                                      int64_t *out_indptr, int32_t *out_indices, double *out_data,
                                      uint64_t *out_nnz) {
  asp_clear_error();
  ASP_TRY(asp::require_device());
  const uint64_t K = num_spins;
  if (!kept_spins || !out_nnz) return asp::set_error(ASP_ERR_INVALID, "null count pointer");
  *kept_spins = 0;
  *out_nnz = 0;
  if (K == 0) return ASP_OK;
  if (!indptr || !is_frozen || !keep) return asp::set_error(ASP_ERR_INVALID, "null input array");
  if (K >= 0xFFFFFFFFull) return asp::set_error(ASP_ERR_TOO_LARGE, "more than 2^32-2 spins");
  if (anchor >= K) return asp::set_error(ASP_ERR_INVALID, "anchor spin out of range");
  if (indptr[0] != 0) return asp::set_error(ASP_ERR_INVALID, "indptr[0] must be 0");
  const int64_t nnz_signed = indptr[K];
  if (nnz_signed < 0) return asp::set_error(ASP_ERR_INVALID, "negative number of non-zeros");
  const uint64_t nnz = static_cast<uint64_t>(nnz_signed);
  if (nnz && (!indices || !data)) return asp::set_error(ASP_ERR_INVALID, "null indices/data");
  for (uint64_t i = 0; i < K; ++i) {
    if (indptr[i + 1] < indptr[i]) return asp::set_error(ASP_ERR_INVALID, "indptr is not monotone");
    for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
      if (indices[k] < 0 || static_cast<uint64_t>(indices[k]) >= K) {
        return asp::set_error(ASP_ERR_INVALID, "column index out of range in row %llu",
                              (unsigned long long)i);
      }
      if (k > indptr[i] && indices[k - 1] >= indices[k]) {
        return asp::set_error(ASP_ERR_INVALID, "row %llu is not sorted and duplicate-free",
                              (unsigned long long)i);
      }
    }
  }
  asp::ScopedStream scoped;
  ASP_TRY(scoped.acquire());
  hipStream_t stream = scoped.stream;
  DeviceBuffer<int64_t> d_indptr, d_new_index, d_out_indptr, d_scratch;
  DeviceBuffer<int32_t> d_indices, d_out_indices;
  DeviceBuffer<double> d_data, d_out_data;
  DeviceBuffer<uint8_t> d_frozen;
  DeviceBuffer<uint32_t> d_parent, d_root, d_keep, d_row_kept, d_stray;
  DeviceBuffer<unsigned long long> d_max;
  asp::StreamFence fence(stream);  // error exits wait for the stream before the buffers go
  ASP_TRY(d_indptr.alloc(K + 1));
  ASP_TRY(d_indices.alloc(nnz));
  ASP_TRY(d_data.alloc(nnz));
  ASP_TRY(d_frozen.alloc(K));
  ASP_TRY(d_parent.alloc(K));
  ASP_TRY(d_root.alloc(K));
  ASP_TRY(d_keep.alloc(K));
  ASP_TRY(d_new_index.alloc(K + 1));
  ASP_TRY(d_scratch.alloc(asp::scan_scratch_elems(K)));
  ASP_TRY(d_stray.alloc(1));
  ASP_TRY(d_max.alloc(1));
  ASP_TRY(d_indptr.upload(indptr, K + 1, stream));
  ASP_TRY(d_indices.upload(indices, nnz, stream));
  ASP_TRY(d_data.upload(data, nnz, stream));
  ASP_TRY(d_frozen.upload(is_frozen, K, stream));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  ASP_HIP_TRY(hipEventCreate(&ev0));
  ASP_HIP_TRY(hipEventCreate(&ev1));
  struct Events {
    hipEvent_t a, b;
    ~Events() {
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
    }
  } events{ev0, ev1};
  ASP_HIP_TRY(hipEventRecord(ev0, stream));
  ASP_HIP_TRY(hipMemsetAsync(d_max.ptr, 0, sizeof(unsigned long long), stream));
  ASP_HIP_TRY(hipMemsetAsync(d_stray.ptr, 0, sizeof(uint32_t), stream));
  if (nnz > 0) {
    const unsigned blocks = std::min<unsigned>(grid_for(nnz, kThreads), 4096u);
    hipLaunchKernelGGL(k_abs_max, dim3(blocks), dim3(kThreads), 0, stream, d_data.ptr, nnz,
                       d_max.ptr);
  }
  unsigned long long max_bits = 0;
  ASP_HIP_TRY(hipMemcpyAsync(&max_bits, d_max.ptr, sizeof max_bits, hipMemcpyDeviceToHost, stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  double max_coupling;
  std::memcpy(&max_coupling, &max_bits, sizeof max_coupling);
  GraphArgs g{d_indptr.ptr, d_indices.ptr, d_data.ptr, d_frozen.ptr, d_parent.ptr, K,
              reltol * max_coupling};  // one rounded product, as numpy's `reltol * max_coupling`
  hipLaunchKernelGGL(k_init_parent, dim3(grid_for(K, kThreads)), dim3(kThreads), 0, stream,
                     d_parent.ptr, K);
  hipLaunchKernelGGL(k_hook_edges, dim3(grid_for(K, kWaves)), dim3(kThreads), 0, stream, g);
  hipLaunchKernelGGL(k_flatten, dim3(grid_for(K, kThreads)), dim3(kThreads), 0, stream,
                     d_parent.ptr, d_root.ptr, K);
  hipLaunchKernelGGL(k_mark_component, dim3(grid_for(K, kThreads)), dim3(kThreads), 0, stream,
                     d_root.ptr, d_frozen.ptr, K, anchor, d_keep.ptr, d_stray.ptr);
  ASP_HIP_TRY(hipGetLastError());
  ASP_TRY(asp::exclusive_scan_u32(d_keep.ptr, K, d_new_index.ptr, d_scratch.ptr, stream));
  int64_t kept = 0;
  uint32_t stray = 0;
  ASP_HIP_TRY(hipMemcpyAsync(&kept, d_new_index.ptr + K, sizeof kept, hipMemcpyDeviceToHost, stream));
  ASP_HIP_TRY(hipMemcpyAsync(&stray, d_stray.ptr, sizeof stray, hipMemcpyDeviceToHost, stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  if (stray != 0) {
    // the reference asserts this (common.py:666)
    return asp::set_error(ASP_ERR_INVALID,
                          "%u frozen spins are not connected to the anchor after the cutoff", stray);
  }
  *kept_spins = static_cast<uint64_t>(kept);
  {
    std::vector<uint32_t> keep_words(K);
    ASP_HIP_TRY(hipMemcpyAsync(keep_words.data(), d_keep.ptr, K * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, stream));
    ASP_HIP_TRY(hipStreamSynchronize(stream));
    for (uint64_t i = 0; i < K; ++i) keep[i] = static_cast<uint8_t>(keep_words[i]);
  }
  // the kept block of the UN-pruned matrix (common.py:674)
  ASP_TRY(d_row_kept.alloc(static_cast<uint64_t>(kept)));
  ASP_TRY(d_out_indptr.alloc(static_cast<uint64_t>(kept) + 1));
  SliceArgs sargs{d_indptr.ptr, d_indices.ptr, d_data.ptr,  d_keep.ptr, d_new_index.ptr,
                  d_row_kept.ptr, nullptr,       nullptr,     nullptr,    K};
  hipLaunchKernelGGL(k_slice_rows<false>, dim3(grid_for(K, kWaves)), dim3(kThreads), 0, stream,
                     sargs);
  ASP_HIP_TRY(hipGetLastError());
  ASP_TRY(asp::exclusive_scan_u32(d_row_kept.ptr, static_cast<uint64_t>(kept), d_out_indptr.ptr,
                                  d_scratch.ptr, stream));
  int64_t kept_nnz = 0;
  ASP_HIP_TRY(hipMemcpyAsync(&kept_nnz, d_out_indptr.ptr + kept, sizeof kept_nnz,
                             hipMemcpyDeviceToHost, stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  *out_nnz = static_cast<uint64_t>(kept_nnz);
  if (!out_indptr && !out_indices && !out_data && capacity == 0) {  // mask only
    ASP_HIP_TRY(hipEventRecord(ev1, stream));
    ASP_HIP_TRY(hipStreamSynchronize(stream));
    (void)hipEventElapsedTime(&g_last_ms, ev0, ev1);
    return ASP_OK;
  }
  if (!out_indptr || !out_indices || !out_data || *out_nnz > capacity) {
    return asp::set_error(ASP_ERR_INVALID, "%llu kept couplings do not fit capacity %llu",
                          (unsigned long long)*out_nnz, (unsigned long long)capacity);
  }
  ASP_TRY(d_out_indices.alloc(*out_nnz));
  ASP_TRY(d_out_data.alloc(*out_nnz));
  sargs.out_indptr = d_out_indptr.ptr;
  sargs.out_indices = d_out_indices.ptr;
  sargs.out_data = d_out_data.ptr;
  hipLaunchKernelGGL(k_slice_rows<true>, dim3(grid_for(K, kWaves)), dim3(kThreads), 0, stream,
                     sargs);
  ASP_HIP_TRY(hipGetLastError());
  ASP_HIP_TRY(hipEventRecord(ev1, stream));
  ASP_TRY(d_out_indptr.download(out_indptr, static_cast<uint64_t>(kept) + 1, stream));
  ASP_TRY(d_out_indices.download(out_indices, *out_nnz, stream));
  ASP_TRY(d_out_data.download(out_data, *out_nnz, stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  (void)hipEventElapsedTime(&g_last_ms, ev0, ev1);
  return ASP_OK;
}
