// Symmetry sectors at full size: the list of representatives of a sector and the sector's
// Hamiltonian as a device-resident ELL matrix, for exact diagonalisation ON the device.
//
// The reference takes its ground states from SpinED output files (common.py:783-803: HDF5
// datasets /basis/representatives and /hamiltonian/eigenvectors); those files are not part of the
// reference tree, and for the 36-site kagome model (heisenberg_kagome_36.yaml:7-29: 144 lattice
// permutations x spin inversion, Sz = 0) the sector holds 31.5 million representatives out of
// 9.08e9 states — out of reach of the host route (operators.SpinBasis.build + scipy).  On this
// device it is small: the representatives take 0.25 GB, the whole Hamiltonian with f64 values
// 27 GB of the 288 GB, and a Lanczos step is one pass over that matrix
// (annealing_sign_problem_amd/sector_ed.py drives it).
//
//   asp_sector_enumerate   candidates of the given magnetisation -> those that are the smallest
//                          member of their orbit, in three filter passes with compaction between
//                          them (a state survives k images with probability ~1/k, so almost all
//                          of the 144 x 2 images are only ever computed for the survivors), then
//                          a radix sort and the norms
//   asp_sector_rows        one row per representative: every off-diagonal transition's target ->
//                          representative, character, bisection in the sorted list; value
//                          c * chi * norm(r') / norm(r) as in `k_symmetrise`
//   asp_sector_matvec      y = H x over the ELL arrays (column-major slots: coalesced)
//
// Conventions (bit i = site i, permutation tables, characters, norms) are those of
// operator_apply.hip / symmetry.py.  gfx950 only.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "asp.h"
#include "asp_common.hpp"
#include "operator_internal.hpp"

namespace {

using asp::Bond;
using asp::DeviceBuffer;
using asp::SymmetryArgs;

constexpr int kThreads = 256;

unsigned grid_for(uint64_t items, uint64_t per_block) {
  return static_cast<unsigned>((items + per_block - 1) / per_block);
}

// Image of x under one group element (site i -> dst[i]).  `dst` is uniform over the wavefront:
// the table is read four entries at a time with scalar loads; the state is moved as two 32-bit
// halves (one v_bfe_u32 + one v_lshl_or_b32 per site).
__device__ __forceinline__ uint64_t permuted(const uint8_t *dst, uint32_t n, uint64_t x) {
  const uint32_t *row = reinterpret_cast<const uint32_t *>(dst);
  const uint32_t xl = static_cast<uint32_t>(x), xh = static_cast<uint32_t>(x >> 32);
  uint32_t yl = 0, yh = 0;
  for (uint32_t w = 0; 4u * w < n; ++w) {
    const uint32_t four = row[w];
    const uint32_t half = (w < 8u) ? xl : xh;  // sites 4w .. 4w+3 lie in one half
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      const uint32_t bit = (half >> ((4u * w + j) & 31u)) & 1u;  // sites >= n are zero bits
      const uint32_t d = (four >> (8u * j)) & 0xFFu;
      if (d < 32u) {
        yl |= bit << d;
      } else {
        yh |= bit << (d - 32u);
      }
    }
  }
  return (static_cast<uint64_t>(yh) << 32) | yl;
}

// Is x smaller than or equal to its images under elements [e_begin, e_end)?
__device__ __forceinline__ bool is_smallest(const SymmetryArgs &g, uint32_t e_begin, uint32_t e_end,
                                            uint64_t x) {
  for (uint32_t e = e_begin; e < e_end; ++e) {
    const uint64_t y = e == 0 ? x : permuted(g.table + static_cast<size_t>(e) * 64u, g.number_spins, x);
    if (y < x) return false;
    if (g.inversion != 0 && (~y & g.mask) < x) return false;
  }
  return true;
}

// Wavefront-aggregated append (every lane of the wavefront calls it).
__device__ __forceinline__ void append(bool keep, uint64_t x, uint64_t *__restrict__ out,
                                       unsigned long long *__restrict__ count, uint64_t capacity) {
  const uint64_t ballot = __ballot(keep);
  if (ballot == 0) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t leader = static_cast<uint32_t>(__ffsll(static_cast<long long>(ballot))) - 1u;
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(count, static_cast<unsigned long long>(__popcll(ballot)));
  base = __shfl(base, static_cast<int>(leader), 64);
  if (keep) {
    const uint64_t at = base + static_cast<uint64_t>(__popcll(ballot & ((1ull << lane) - 1ull)));
    if (at < capacity) out[at] = x;
  }
}

// First pass: workgroup = one value of the high bits; its candidates are that value joined with
// every low word of the matching population count (all low words when weight < 0).
__global__ __launch_bounds__(kThreads) void k_sector_generate(
    SymmetryArgs g, uint32_t e_end, const uint32_t *__restrict__ low_words,
    const uint32_t *__restrict__ low_offsets, uint32_t lo_bits, int32_t weight,
    uint64_t *__restrict__ out, unsigned long long *__restrict__ count, uint64_t capacity) {
  const uint64_t high = blockIdx.x;
  uint32_t begin = 0, end = 1u << lo_bits;
  if (weight >= 0) {
    const int32_t need = weight - __popcll(high);
    if (need < 0 || need > static_cast<int32_t>(lo_bits)) return;
    begin = low_offsets[need];
    end = low_offsets[need + 1];
  }
  for (uint32_t first = begin; first < end; first += kThreads) {  // uniform trip count
    const uint32_t at = first + threadIdx.x;
    const bool live = at < end;
    const uint64_t x = (high << lo_bits) | (live ? (weight >= 0 ? low_words[at] : at) : 0u);
    const bool keep = live && is_smallest(g, 0, e_end, x);
    append(keep, x, out, count, capacity);
  }
}

// Later passes: the survivors of the pass before against the next range of elements.
__global__ __launch_bounds__(kThreads) void k_sector_filter(SymmetryArgs g, uint32_t e_begin,
                                                           uint32_t e_end,
                                                           const uint64_t *__restrict__ in, uint64_t n,
                                                           uint64_t *__restrict__ out,
                                                           unsigned long long *__restrict__ count,
                                                           uint64_t capacity) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const bool live = i < n;
  const uint64_t x = live ? in[i] : 0;
  const bool keep = live && is_smallest(g, e_begin, e_end, x);
  append(keep, x, out, count, capacity);
}

// Sum of the characters of the stabiliser of x (>= 1 unless spin inversion has character -1).
__device__ __forceinline__ int32_t stabiliser_sum(const SymmetryArgs &g, uint64_t x) {
  if (g.num_permutations == 0) return 1;
  int32_t sum = 0;
  for (uint32_t e = 0; e < g.num_permutations; ++e) {
    const uint64_t y = e == 0 ? x : permuted(g.table + static_cast<size_t>(e) * 64u, g.number_spins, x);
    sum += y == x ? 1 : 0;
    if (g.inversion != 0) sum += (~y & g.mask) == x ? g.inversion : 0;
  }
  return sum;
}

__device__ __forceinline__ double norm_of(const SymmetryArgs &g, int32_t stabiliser) {
  if (g.num_permutations == 0) return 1.0;
  const double order = static_cast<double>(g.num_permutations) * (g.inversion != 0 ? 2.0 : 1.0);
  // the expression of operator_apply.hip / symmetry.py: sqrt(max(stabiliser, 0) / |G|)
  return sqrt(static_cast<double>(stabiliser > 0 ? stabiliser : 0) / order);
}

// Drops the orbit minima that are not part of the sector (norm 0).
__global__ __launch_bounds__(kThreads) void k_sector_in_sector(SymmetryArgs g,
                                                              const uint64_t *__restrict__ in,
                                                              uint64_t n, uint64_t *__restrict__ out,
                                                              unsigned long long *__restrict__ count,
                                                              uint64_t capacity) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const bool live = i < n;
  const uint64_t x = live ? in[i] : 0;
  const bool keep = live && stabiliser_sum(g, x) > 0;
  append(keep, x, out, count, capacity);
}

__global__ __launch_bounds__(kThreads) void k_sector_norms(SymmetryArgs g,
                                                          const uint64_t *__restrict__ reps,
                                                          uint64_t n, double *__restrict__ norms) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  norms[i] = norm_of(g, stabiliser_sum(g, reps[i]));
}

struct RowsArgs {
  SymmetryArgs g;
  const Bond *bonds;
  const uint16_t *transitions;  // (bond << 2) | (src ^ dst), only those some source can take
  uint32_t num_bonds;
  uint32_t num_transitions;  // <= 128
  const uint64_t *reps;      // sorted
  const double *norms;
  uint64_t n;
  uint32_t width;
  uint32_t *idx;   // [width][n]
  double *val;     // [width][n]
  double *diag;    // [n]
  unsigned long long *overflow;  // rows with more than `width` entries (cannot happen)
};

__device__ __forceinline__ uint32_t bond_state(const Bond &bond, uint64_t key) {
  return static_cast<uint32_t>(((key >> bond.a) & 1ull) * 2ull + ((key >> bond.b) & 1ull));
}

// One row per thread.  The transitions a row takes are collected in a bit mask first and then
// popped one per iteration, so that every lane of the wavefront walks the group for a target of
// its own instead of idling while its neighbours' bonds are flippable and its own are not.
__global__ __launch_bounds__(kThreads) void k_sector_rows(RowsArgs a) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const bool live = i < a.n;
  const uint64_t key = live ? a.reps[i] : 0;
  const double my_norm = live ? a.norms[i] : 1.0;
  double diagonal = 0.0;
  for (uint32_t b = 0; b < a.num_bonds; ++b) {  // left to right, as k_apply sums it
    const Bond &bond = a.bonds[b];
    const uint32_t src = bond_state(bond, key);
    diagonal = __dadd_rn(diagonal, bond.m[src * 4u + src]);
  }
  uint64_t todo[2] = {0, 0};
  if (live) {
    for (uint32_t t = 0; t < a.num_transitions; ++t) {
      const uint32_t code = a.transitions[t];
      const Bond &bond = a.bonds[code >> 2];
      const uint32_t src = bond_state(bond, key);
      const uint32_t dst = src ^ (code & 3u);
      if (bond.m[dst * 4u + src] != 0.0) todo[t >> 6] |= 1ull << (t & 63u);
    }
  }
  uint32_t slot = 0;
  while (__ballot((todo[0] | todo[1]) != 0) != 0) {
    const bool mine = (todo[0] | todo[1]) != 0;
    uint32_t t = 0;
    if (todo[0]) {
      t = static_cast<uint32_t>(__ffsll(static_cast<long long>(todo[0]))) - 1u;
      todo[0] &= todo[0] - 1;
    } else if (todo[1]) {
      t = 64u + static_cast<uint32_t>(__ffsll(static_cast<long long>(todo[1]))) - 1u;
      todo[1] &= todo[1] - 1;
    }
    const uint32_t code = a.transitions[t];
    const Bond &bond = a.bonds[code >> 2];
    const uint32_t src = bond_state(bond, key);
    const uint32_t dst = src ^ (code & 3u);
    const double c = bond.m[dst * 4u + src];
    const uint64_t target = key ^ bond.flip[code & 3u];
    // representative of the target and the character of an element that maps onto it
    uint64_t best = target;
    bool through_flip = false;
    for (uint32_t e = 0; e < a.g.num_permutations; ++e) {
      const uint64_t y =
          e == 0 ? target : permuted(a.g.table + static_cast<size_t>(e) * 64u, a.g.number_spins, target);
      if (y < best) {
        best = y;
        through_flip = false;
      }
      if (a.g.inversion != 0) {
        const uint64_t z = ~y & a.g.mask;
        if (z < best) {
          best = z;
          through_flip = true;
        }
      }
    }
    if (!mine) continue;
    // bisection: reps[lo] <= best < reps[hi]
    uint64_t lo = 0, hi = a.n;
    while (hi - lo > 1) {
      const uint64_t mid = (lo + hi) / 2;
      if (a.reps[mid] <= best) {
        lo = mid;
      } else {
        hi = mid;
      }
    }
    if (a.reps[lo] != best) continue;  // the target's orbit is not part of the sector
    if (slot >= a.width) {
      atomicAdd(a.overflow, 1ull);
      continue;
    }
    const double character = (through_flip && a.g.inversion < 0) ? -1.0 : 1.0;
    a.idx[static_cast<uint64_t>(slot) * a.n + i] = static_cast<uint32_t>(lo);
    a.val[static_cast<uint64_t>(slot) * a.n + i] =
        __ddiv_rn(__dmul_rn(__dmul_rn(c, character), a.norms[lo]), my_norm);
    ++slot;
  }
  if (!live) return;
  a.diag[i] = diagonal;
  for (; slot < a.width; ++slot) {  // padding: 0 * x[i]
    a.idx[static_cast<uint64_t>(slot) * a.n + i] = static_cast<uint32_t>(i);
    a.val[static_cast<uint64_t>(slot) * a.n + i] = 0.0;
  }
}

// y = H x.  The matrix streams through once (non-temporal loads: x is what should stay in the
// caches), x is gathered.
__global__ __launch_bounds__(kThreads) void k_sector_matvec(uint64_t n, uint32_t width,
                                                           const uint32_t *__restrict__ idx,
                                                           const double *__restrict__ val,
                                                           const double *__restrict__ diag,
                                                           const double *__restrict__ x,
                                                           double *__restrict__ y) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  double acc = __dmul_rn(diag[i], x[i]);
  for (uint32_t k = 0; k < width; ++k) {
    const uint64_t at = static_cast<uint64_t>(k) * n + i;
    const uint32_t j = __builtin_nontemporal_load(idx + at);
    const double v = __builtin_nontemporal_load(val + at);
    acc = __fma_rn(v, x[j], acc);
  }
  y[i] = acc;
}

double binomial(uint32_t n, uint32_t k) {
  if (k > n) return 0.0;
  double r = 1.0;
  for (uint32_t j = 1; j <= k; ++j) r = r * static_cast<double>(n - k + j) / static_cast<double>(j);
  return r;
}

int check(const asp_operator *op) {
  if (!op) return asp::set_error(ASP_ERR_INVALID, "null operator");
  return asp::bind_device();
}

}  // namespace

extern "C" {

int asp_sector_enumerate(asp_operator const *op, int32_t hamming_weight, uint64_t capacity,
                         uint64_t *reps_dev, double *norms_dev, uint64_t *count) {
  asp_clear_error();
  ASP_TRY(check(op));
  if (!count) return asp::set_error(ASP_ERR_INVALID, "null count");
  *count = 0;
  const uint32_t n = op->number_spins;
  if (n > 48) return asp::set_error(ASP_ERR_TOO_LARGE, "sector enumeration handles up to 48 spins");
  if (hamming_weight > static_cast<int32_t>(n)) {
    return asp::set_error(ASP_ERR_INVALID, "hamming_weight exceeds the number of spins");
  }
  if (capacity && !reps_dev) return asp::set_error(ASP_ERR_INVALID, "null output");
  const uint32_t lo_bits = std::min(n / 2u + (n & 1u), 20u);
  const uint32_t hi_bits = n - lo_bits;
  const double candidates = hamming_weight >= 0 ? binomial(n, static_cast<uint32_t>(hamming_weight))
                                                : std::ldexp(1.0, static_cast<int>(n));
  // low words grouped by population count, ascending inside a group
  std::vector<uint32_t> offsets(lo_bits + 2, 0), words(size_t{1} << lo_bits);
  for (uint32_t w = 0; w < (1u << lo_bits); ++w) ++offsets[static_cast<uint32_t>(__builtin_popcount(w)) + 1];
  for (uint32_t p = 0; p <= lo_bits; ++p) offsets[p + 1] += offsets[p];
  {
    std::vector<uint32_t> fill(offsets.begin(), offsets.end() - 1);
    for (uint32_t w = 0; w < (1u << lo_bits); ++w) words[fill[static_cast<uint32_t>(__builtin_popcount(w))]++] = w;
  }
  const SymmetryArgs g = op->symmetry();
  const uint32_t P = op->num_permutations;
  const uint32_t images_per_element = op->inversion != 0 ? 2u : 1u;
  // pass boundaries: elements [0, e1), [e1, e2), [e2, P)
  const uint32_t e1 = std::min(P, 9u), e2 = std::min(P, 41u);
  auto room = [&](uint32_t elements) {  // survivors of `elements` elements, generously
    const double images = std::max(1.0, static_cast<double>(elements * images_per_element));
    return static_cast<uint64_t>(candidates / images * 2.0) + (1ull << 20);
  };
  asp::ScopedStream scoped;
  ASP_TRY(scoped.acquire());
  hipStream_t s = scoped.stream;
  DeviceBuffer<uint32_t> d_words, d_offsets;
  DeviceBuffer<uint64_t> d_a, d_b;
  DeviceBuffer<unsigned long long> d_count;
  DeviceBuffer<uint8_t> d_temp;
  asp::StreamFence fence(s);
  ASP_TRY(d_words.alloc(words.size()));
  ASP_TRY(d_offsets.alloc(offsets.size()));
  ASP_TRY(d_count.alloc(4));
  ASP_TRY(d_words.upload(words.data(), words.size(), s));
  ASP_TRY(d_offsets.upload(offsets.data(), offsets.size(), s));
  ASP_HIP_TRY(hipMemsetAsync(d_count.ptr, 0, 4 * sizeof(unsigned long long), s));
  uint64_t cap_a = P ? room(e1) : static_cast<uint64_t>(candidates) + 1;
  ASP_TRY(d_a.alloc(cap_a));
  hipLaunchKernelGGL(k_sector_generate, dim3(1u << hi_bits), dim3(kThreads), 0, s, g, e1, d_words.ptr,
                     d_offsets.ptr, lo_bits, hamming_weight, d_a.ptr, d_count.ptr + 0, cap_a);
  ASP_HIP_TRY(hipGetLastError());
  unsigned long long have = 0;
  auto fetch = [&](int which, uint64_t cap) -> int {
    ASP_HIP_TRY(hipMemcpyAsync(&have, d_count.ptr + which, sizeof have, hipMemcpyDeviceToHost, s));
    ASP_HIP_TRY(hipStreamSynchronize(s));
    if (have > cap) {
      return asp::set_error(ASP_ERR_TOO_LARGE, "sector enumeration: %llu survivors of a pass exceed "
                                               "the room of %llu", have, (unsigned long long)cap);
    }
    return ASP_OK;
  };
  ASP_TRY(fetch(0, cap_a));
  DeviceBuffer<uint64_t> *cur = &d_a, *other = &d_b;
  int pass = 1;
  const uint32_t bounds[3] = {e1, e2, P};
  for (int stage = 0; stage < 2; ++stage) {
    if (bounds[stage] >= bounds[stage + 1]) continue;
    const uint64_t n_in = have;
    const uint64_t cap = std::min<uint64_t>(n_in, room(bounds[stage + 1])) + 1;
    ASP_TRY(other->alloc(cap));
    if (n_in) {
      hipLaunchKernelGGL(k_sector_filter, dim3(grid_for(n_in, kThreads)), dim3(kThreads), 0, s, g,
                         bounds[stage], bounds[stage + 1], cur->ptr, n_in, other->ptr,
                         d_count.ptr + pass, cap);
      ASP_HIP_TRY(hipGetLastError());
    }
    ASP_TRY(fetch(pass, cap));
    ++pass;
    std::swap(cur, other);
    other->release();
  }
  if (op->inversion < 0 && have) {  // orbits whose stabiliser holds an element of character -1
    const uint64_t n_in = have;
    ASP_TRY(other->alloc(n_in));
    hipLaunchKernelGGL(k_sector_in_sector, dim3(grid_for(n_in, kThreads)), dim3(kThreads), 0, s, g,
                       cur->ptr, n_in, other->ptr, d_count.ptr + pass, n_in);
    ASP_HIP_TRY(hipGetLastError());
    ASP_TRY(fetch(pass, n_in));
    std::swap(cur, other);
    other->release();
  }
  *count = have;
  if (have > capacity) {
    if (capacity == 0) return ASP_OK;  // sizing call
    return asp::set_error(ASP_ERR_TOO_LARGE, "%llu representatives do not fit the capacity of %llu",
                          have, (unsigned long long)capacity);
  }
  if (have == 0) return ASP_OK;
  size_t temp_bytes = 0;
  ASP_HIP_TRY(rocprim::radix_sort_keys(nullptr, temp_bytes, cur->ptr, reps_dev, have, 0, n, s));
  ASP_TRY(d_temp.alloc(temp_bytes ? temp_bytes : 1));
  ASP_HIP_TRY(rocprim::radix_sort_keys(d_temp.ptr, temp_bytes, cur->ptr, reps_dev, have, 0, n, s));
  if (norms_dev) {
    hipLaunchKernelGGL(k_sector_norms, dim3(grid_for(have, kThreads)), dim3(kThreads), 0, s, g,
                       reps_dev, have, norms_dev);
    ASP_HIP_TRY(hipGetLastError());
  }
  ASP_HIP_TRY(hipStreamSynchronize(s));
  return ASP_OK;
}

uint32_t asp_sector_width(asp_operator const *op) { return op ? op->max_connections - 1u : 0u; }

int asp_sector_rows(asp_operator const *op, uint64_t n, uint64_t const *reps_dev,
                    double const *norms_dev, uint32_t width, uint32_t *idx_dev, double *val_dev,
                    double *diag_dev) {
  asp_clear_error();
  ASP_TRY(check(op));
  if (n == 0) return ASP_OK;
  if (!reps_dev || !norms_dev || !idx_dev || !val_dev || !diag_dev) {
    return asp::set_error(ASP_ERR_INVALID, "null argument");
  }
  if (n >= 0xFFFFFFFFull) return asp::set_error(ASP_ERR_TOO_LARGE, "row indices are 32 bits wide");
  if (width < op->max_connections - 1u) {
    return asp::set_error(ASP_ERR_INVALID, "width %u is below asp_sector_width() = %u", width,
                          op->max_connections - 1u);
  }
  // the transitions some source state can take
  std::vector<uint16_t> transitions;
  for (uint32_t b = 0; b < op->num_bonds; ++b) {
    for (uint32_t x = 1; x < 4; ++x) {
      bool used = false;
      for (uint32_t src = 0; src < 4; ++src) used = used || op->bonds[b].m[(src ^ x) * 4 + src] != 0.0;
      if (used) transitions.push_back(static_cast<uint16_t>((b << 2) | x));
    }
  }
  if (transitions.size() > 128) {
    return asp::set_error(ASP_ERR_TOO_LARGE, "more than 128 off-diagonal transitions per state");
  }
  asp::ScopedStream scoped;
  ASP_TRY(scoped.acquire());
  hipStream_t s = scoped.stream;
  DeviceBuffer<uint16_t> d_transitions;
  DeviceBuffer<unsigned long long> d_overflow;
  asp::StreamFence fence(s);
  ASP_TRY(d_transitions.alloc(transitions.size()));
  ASP_TRY(d_overflow.alloc(1));
  ASP_TRY(d_transitions.upload(transitions.data(), transitions.size(), s));
  ASP_HIP_TRY(hipMemsetAsync(d_overflow.ptr, 0, sizeof(unsigned long long), s));
  RowsArgs a{};
  a.g = op->symmetry();
  a.bonds = op->d_bonds.ptr;
  a.transitions = d_transitions.ptr;
  a.num_bonds = op->num_bonds;
  a.num_transitions = static_cast<uint32_t>(transitions.size());
  a.reps = reps_dev;
  a.norms = norms_dev;
  a.n = n;
  a.width = width;
  a.idx = idx_dev;
  a.val = val_dev;
  a.diag = diag_dev;
  a.overflow = d_overflow.ptr;
  hipLaunchKernelGGL(k_sector_rows, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, s, a);
  ASP_HIP_TRY(hipGetLastError());
  unsigned long long overflow = 0;
  ASP_HIP_TRY(hipMemcpyAsync(&overflow, d_overflow.ptr, sizeof overflow, hipMemcpyDeviceToHost, s));
  ASP_HIP_TRY(hipStreamSynchronize(s));
  if (overflow) return asp::set_error(ASP_ERR_INVALID, "%llu entries did not fit the row width", overflow);
  return ASP_OK;
}

int asp_sector_matvec(uint64_t n, uint32_t width, uint32_t const *idx_dev, double const *val_dev,
                      double const *diag_dev, double const *x_dev, double *y_dev) {
  asp_clear_error();
  ASP_TRY(asp::bind_device());
  if (n == 0) return ASP_OK;
  if (!diag_dev || !x_dev || !y_dev || (width && (!idx_dev || !val_dev))) {
    return asp::set_error(ASP_ERR_INVALID, "null argument");
  }
  if (x_dev == y_dev) return asp::set_error(ASP_ERR_INVALID, "x and y must not alias");
  asp::ScopedStream scoped;
  ASP_TRY(scoped.acquire());
  hipLaunchKernelGGL(k_sector_matvec, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, scoped.stream, n,
                     width, idx_dev, val_dev, diag_dev, x_dev, y_dev);
  ASP_HIP_TRY(hipGetLastError());
  ASP_HIP_TRY(hipStreamSynchronize(scoped.stream));
  return ASP_OK;
}

}  // extern "C"
