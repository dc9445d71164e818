// Coupling build on gfx950: HIP replacement for the reference's
// cbits/build_matrix.c (build_matrix :22-65, ls_bits512_cmp :7-20,
// extract_signs :67-76) behind the same C symbols.
//
// Data in HBM (all SoA except the 512-bit keys, which keep the reference's AoS
// layout because that is what the ABI hands over):
//   table   ls_bits512[K]   sorted, unique           table0  u64[K] = words[0]
//   needles ls_bits512[N]   flat, row-major by row
//   counts i64[K], psi f64[K], coeffs f64[N], other_psi f64[N], other_counts i64[K]
//   offsets i64[K+1] (scan of other_counts), found i32[N] (table index or -1),
//   row_hits u32[K], row_start i64[K+1] (scan of row_hits)
//   out: row u32[N], col u32[N], elements f64[N], field f64[K]
//
// Launch sequence of one build (all on one stream, no host round trip except
// reading back nnz):
//   k_split_word0      table -> table0 (8 B/key probe array instead of 64 B)
//   scan               other_counts -> offsets
//   k_search           one lane per needle: binary search, word 0 first
//   k_row_hits         one lane per row: number of hits in its needle range
//   scan               row_hits -> row_start
//   k_emit             one lane per row: COO triples in input order + the row's
//                      field as a left-to-right sum (the reference's rounding)
//
// Arithmetic: __dmul_rn / __dadd_rn keep every product and the field add
// separately rounded (no FMA contraction), matching the reference binary.
// Memory-bound integer/f64 streaming: no MFMA, no LDS tiling of the payload.
#include "asp_common.hpp"

namespace {

using asp::DeviceBuffer;

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void k_split_word0(const ls_bits512 *__restrict__ table,
                                                         uint64_t n,
                                                         uint64_t *__restrict__ table0) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) table0[i] = table[i].words[0];
}

// Three-way compare of words[1..7] (word 0 already known equal); the order of
// cbits/build_matrix.c:11-18.
__device__ __forceinline__ int compare_tail(const ls_bits512 *__restrict__ a,
                                            const ls_bits512 *__restrict__ b) {
#pragma unroll
  for (int w = 1; w < 8; ++w) {
    const uint64_t x = a->words[w];
    const uint64_t y = b->words[w];
    if (x != y) return x < y ? -1 : 1;
  }
  return 0;
}

// Index of *needle in the sorted unique table, or -1.
__device__ __forceinline__ int32_t find_key(const uint64_t *__restrict__ table0,
                                            const ls_bits512 *__restrict__ table, uint64_t n,
                                            const ls_bits512 *__restrict__ needle) {
  const uint64_t n0 = needle->words[0];
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    const uint64_t t0 = table0[mid];
    int c;
    if (t0 != n0) {
      c = t0 < n0 ? -1 : 1;
    } else {
      c = compare_tail(&table[mid], needle);
      if (c == 0) return static_cast<int32_t>(mid);
    }
    if (c < 0) {
      lo = mid + 1;
    } else {
      hi = mid;
    }
  }
  return -1;
}

__global__ __launch_bounds__(kThreads) void k_search(const uint64_t *__restrict__ table0,
                                                    const ls_bits512 *__restrict__ table,
                                                    uint64_t num_spins,
                                                    const ls_bits512 *__restrict__ needles,
                                                    uint64_t num_other,
                                                    int32_t *__restrict__ found) {
  const uint64_t e = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e < num_other) found[e] = find_key(table0, table, num_spins, &needles[e]);
}

__global__ __launch_bounds__(kThreads) void k_row_hits(const int64_t *__restrict__ offsets,
                                                      const int32_t *__restrict__ found,
                                                      uint64_t num_spins,
                                                      uint32_t *__restrict__ row_hits) {
  const uint64_t r = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (r >= num_spins) return;
  uint32_t hits = 0;
  for (int64_t e = offsets[r]; e < offsets[r + 1]; ++e) hits += found[e] >= 0 ? 1u : 0u;
  row_hits[r] = hits;
}

__global__ __launch_bounds__(kThreads) void k_emit(
    const int64_t *__restrict__ offsets, const int64_t *__restrict__ row_start,
    const int32_t *__restrict__ found, const int64_t *__restrict__ counts,
    const double *__restrict__ psi, const double *__restrict__ coeffs,
    const double *__restrict__ other_psi, uint64_t num_spins, uint32_t *__restrict__ out_row,
    uint32_t *__restrict__ out_col, double *__restrict__ out_elements,
    double *__restrict__ out_field) {
  const uint64_t r = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (r >= num_spins) return;
  const double c = static_cast<double>(counts[r]);  // exact int64 -> f64 conversion as in C
  const double a = fabs(psi[r]);
  int64_t w = row_start[r];
  double f = 0.0;
  for (int64_t e = offsets[r]; e < offsets[r + 1]; ++e) {
    const int32_t pos = found[e];
    // ((counts * coeff) * |psi|) * x, each product rounded: build_matrix.c:41-42,49
    const double head = __dmul_rn(__dmul_rn(c, coeffs[e]), a);
    if (pos >= 0) {
      out_row[w] = static_cast<uint32_t>(r);
      out_col[w] = static_cast<uint32_t>(pos);
      out_elements[w] = __dmul_rn(head, fabs(other_psi[e]));
      ++w;
    } else {
      f = __dadd_rn(f, __dmul_rn(head, other_psi[e]));
    }
  }
  out_field[r] = f;
}

// extract_signs: one wavefront builds one 64-bit word with a ballot.
__global__ __launch_bounds__(kThreads) void k_extract_signs(const double *__restrict__ psi,
                                                           uint64_t num_spins,
                                                           uint64_t *__restrict__ signs) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const bool positive = i < num_spins && psi[i] > 0.0;  // NaN, 0 -> false
  const uint64_t word = __ballot(positive);
  if ((threadIdx.x & 63) == 0 && i < num_spins) signs[i >> 6] = word;
}

inline unsigned blocks_for(uint64_t n) { return static_cast<unsigned>((n + kThreads - 1) / kThreads); }

}  // namespace

struct asp_build {
  uint64_t num_spins = 0;
  uint64_t num_other = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  float last_ms = 0.0f;
  uint64_t last_nnz = 0;
  bool uploaded = false;
  DeviceBuffer<ls_bits512> table, needles;
  DeviceBuffer<uint64_t> table0;
  DeviceBuffer<int64_t> counts, other_counts, offsets, row_start, scratch;
  DeviceBuffer<double> psi, coeffs, other_psi, elements, field;
  DeviceBuffer<int32_t> found;
  DeviceBuffer<uint32_t> row_hits, out_row, out_col;
};

extern "C" {

asp_build *asp_build_create(uint64_t num_spins, uint64_t num_other) {
  if (asp::require_device() != ASP_OK) return nullptr;
  if (num_spins >= (1ull << 31) || num_other >= (1ull << 40)) {
    asp::set_error(ASP_ERR_TOO_LARGE, "build of %llu rows / %llu connections is out of range",
                   (unsigned long long)num_spins, (unsigned long long)num_other);
    return nullptr;
  }
  asp_build *b = new (std::nothrow) asp_build();
  if (!b) {
    asp::set_error(ASP_ERR_ALLOC, "out of host memory");
    return nullptr;
  }
  b->num_spins = num_spins;
  b->num_other = num_other;
  const uint64_t K = num_spins, N = num_other;
  bool ok = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreate(&b->ev_start) == hipSuccess && hipEventCreate(&b->ev_stop) == hipSuccess;
  if (!ok) asp::set_error(ASP_ERR_HIP, "could not create HIP stream/events");
  const size_t scratch = asp::scan_scratch_elems(K > N ? K : N);
  ok = ok && b->table.alloc(K) == ASP_OK && b->needles.alloc(N) == ASP_OK &&
       b->table0.alloc(K) == ASP_OK && b->counts.alloc(K) == ASP_OK &&
       b->other_counts.alloc(K) == ASP_OK && b->offsets.alloc(K + 1) == ASP_OK &&
       b->row_start.alloc(K + 1) == ASP_OK && b->scratch.alloc(scratch) == ASP_OK &&
       b->psi.alloc(K) == ASP_OK && b->coeffs.alloc(N) == ASP_OK &&
       b->other_psi.alloc(N) == ASP_OK && b->elements.alloc(N) == ASP_OK &&
       b->field.alloc(K) == ASP_OK && b->found.alloc(N) == ASP_OK &&
       b->row_hits.alloc(K) == ASP_OK && b->out_row.alloc(N) == ASP_OK &&
       b->out_col.alloc(N) == ASP_OK;
  if (!ok) {
    asp_build_destroy(b);
    return nullptr;
  }
  return b;
}

void asp_build_destroy(asp_build *b) {
  if (!b) return;
  if (b->ev_start) (void)hipEventDestroy(b->ev_start);
  if (b->ev_stop) (void)hipEventDestroy(b->ev_stop);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

int asp_build_upload(asp_build *b, ls_bits512 const *spins, int64_t const *counts,
                     double const *psi, ls_bits512 const *other_spins,
                     double const *other_coeffs, int64_t const *other_counts,
                     double const *other_psi) {
  if (!b) return asp::set_error(ASP_ERR_INVALID, "null build handle");
  const uint64_t K = b->num_spins, N = b->num_other;
  if ((K && (!spins || !counts || !psi || !other_counts)) ||
      (N && (!other_spins || !other_coeffs || !other_psi))) {
    return asp::set_error(ASP_ERR_INVALID, "null input array");
  }
  // The flat arrays must hold exactly sum(other_counts) entries
  // (cbits/build_matrix.c:32-36 walks them with that many increments).
  uint64_t total = 0;
  for (uint64_t r = 0; r < K; ++r) {
    if (other_counts[r] < 0) {
      return asp::set_error(ASP_ERR_INVALID, "other_counts[%llu] is negative",
                            (unsigned long long)r);
    }
    total += static_cast<uint64_t>(other_counts[r]);
  }
  if (total != N) {
    return asp::set_error(ASP_ERR_INVALID, "sum(other_counts) = %llu but num_other = %llu",
                          (unsigned long long)total, (unsigned long long)N);
  }
  ASP_TRY(b->table.upload(spins, K, b->stream));
  ASP_TRY(b->counts.upload(counts, K, b->stream));
  ASP_TRY(b->psi.upload(psi, K, b->stream));
  ASP_TRY(b->needles.upload(other_spins, N, b->stream));
  ASP_TRY(b->coeffs.upload(other_coeffs, N, b->stream));
  ASP_TRY(b->other_counts.upload(other_counts, K, b->stream));
  ASP_TRY(b->other_psi.upload(other_psi, N, b->stream));
  ASP_HIP_TRY(hipStreamSynchronize(b->stream));
  b->uploaded = true;
  return ASP_OK;
}

int asp_build_run(asp_build *b, uint64_t *nnz) {
  if (!b) return asp::set_error(ASP_ERR_INVALID, "null build handle");
  if (!b->uploaded) return asp::set_error(ASP_ERR_INVALID, "asp_build_run before asp_build_upload");
  const uint64_t K = b->num_spins, N = b->num_other;
  hipStream_t s = b->stream;
  ASP_HIP_TRY(hipEventRecord(b->ev_start, s));
  if (K > 0) {
    hipLaunchKernelGGL(k_split_word0, dim3(blocks_for(K)), dim3(kThreads), 0, s, b->table.ptr, K,
                       b->table0.ptr);
  }
  ASP_TRY(asp::exclusive_scan_i64(b->other_counts.ptr, K, b->offsets.ptr, b->scratch.ptr, s));
  if (N > 0) {
    hipLaunchKernelGGL(k_search, dim3(blocks_for(N)), dim3(kThreads), 0, s, b->table0.ptr,
                       b->table.ptr, K, b->needles.ptr, N, b->found.ptr);
  }
  if (K > 0) {
    hipLaunchKernelGGL(k_row_hits, dim3(blocks_for(K)), dim3(kThreads), 0, s, b->offsets.ptr,
                       b->found.ptr, K, b->row_hits.ptr);
  }
  ASP_TRY(asp::exclusive_scan_u32(b->row_hits.ptr, K, b->row_start.ptr, b->scratch.ptr, s));
  if (K > 0) {
    hipLaunchKernelGGL(k_emit, dim3(blocks_for(K)), dim3(kThreads), 0, s, b->offsets.ptr,
                       b->row_start.ptr, b->found.ptr, b->counts.ptr, b->psi.ptr, b->coeffs.ptr,
                       b->other_psi.ptr, K, b->out_row.ptr, b->out_col.ptr, b->elements.ptr,
                       b->field.ptr);
  }
  ASP_HIP_TRY(hipGetLastError());
  ASP_HIP_TRY(hipEventRecord(b->ev_stop, s));
  int64_t total = 0;
  ASP_HIP_TRY(hipMemcpyAsync(&total, b->row_start.ptr + K, sizeof total, hipMemcpyDeviceToHost, s));
  ASP_HIP_TRY(hipStreamSynchronize(s));
  ASP_HIP_TRY(hipEventElapsedTime(&b->last_ms, b->ev_start, b->ev_stop));
  b->last_nnz = static_cast<uint64_t>(total);
  if (nnz) *nnz = b->last_nnz;
  return ASP_OK;
}

float asp_build_last_ms(asp_build const *b) { return b ? b->last_ms : 0.0f; }

int asp_build_download(asp_build *b, uint32_t *row_indices, uint32_t *col_indices,
                       double *elements, double *field) {
  if (!b) return asp::set_error(ASP_ERR_INVALID, "null build handle");
  const uint64_t n = b->last_nnz;
  if (row_indices) ASP_TRY(b->out_row.download(row_indices, n, b->stream));
  if (col_indices) ASP_TRY(b->out_col.download(col_indices, n, b->stream));
  if (elements) ASP_TRY(b->elements.download(elements, n, b->stream));
  if (field) ASP_TRY(b->field.download(field, b->num_spins, b->stream));
  ASP_HIP_TRY(hipStreamSynchronize(b->stream));
  return ASP_OK;
}

// ---- the reference's own symbols (cbits/build_matrix.h:7-14) ---------------

uint64_t build_matrix(uint64_t num_spins, ls_bits512 const spins[], int64_t const *counts,
                      double const *psi, ls_bits512 const *other_spins,
                      double const *other_coeffs, int64_t const *other_counts,
                      double const *other_psi, uint32_t *row_indices, uint32_t *col_indices,
                      double *elements, double *field) {
  asp_clear_error();
  if (num_spins && !other_counts) {
    asp::set_error(ASP_ERR_INVALID, "null other_counts");
    return 0;
  }
  uint64_t num_other = 0;
  for (uint64_t r = 0; r < num_spins; ++r) {
    if (other_counts[r] < 0) {
      asp::set_error(ASP_ERR_INVALID, "other_counts[%llu] is negative", (unsigned long long)r);
      return 0;
    }
    num_other += static_cast<uint64_t>(other_counts[r]);
  }
  asp_build *b = asp_build_create(num_spins, num_other);
  if (!b) return 0;
  uint64_t nnz = 0;
  int rc = asp_build_upload(b, spins, counts, psi, other_spins, other_coeffs, other_counts,
                            other_psi);
  if (rc == ASP_OK) rc = asp_build_run(b, &nnz);
  if (rc == ASP_OK) rc = asp_build_download(b, row_indices, col_indices, elements, field);
  asp_build_destroy(b);
  return rc == ASP_OK ? nnz : 0;
}

void extract_signs(uint64_t num_spins, double const *psi, uint64_t *signs) {
  asp_clear_error();
  if (num_spins == 0) return;
  if (asp::require_device() != ASP_OK) return;
  if (!psi || !signs) {
    asp::set_error(ASP_ERR_INVALID, "null argument");
    return;
  }
  const uint64_t words = (num_spins + 63) / 64;
  DeviceBuffer<double> d_psi;
  DeviceBuffer<uint64_t> d_signs;
  if (d_psi.alloc(num_spins) != ASP_OK || d_signs.alloc(words) != ASP_OK) return;
  auto fail = [](hipError_t e) {
    if (e != hipSuccess) asp::set_error(ASP_ERR_HIP, "extract_signs: %s", hipGetErrorString(e));
    return e != hipSuccess;
  };
  if (fail(hipMemcpy(d_psi.ptr, psi, num_spins * sizeof(double), hipMemcpyHostToDevice))) return;
  hipLaunchKernelGGL(k_extract_signs, dim3(blocks_for(num_spins)), dim3(kThreads), 0, nullptr,
                     d_psi.ptr, num_spins, d_signs.ptr);
  if (fail(hipGetLastError())) return;
  if (fail(hipMemcpy(signs, d_signs.ptr, words * sizeof(uint64_t), hipMemcpyDeviceToHost))) return;
}

}  // extern "C"
