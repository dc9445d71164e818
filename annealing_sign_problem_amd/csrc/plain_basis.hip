// y = H x in a fixed-magnetisation basis WITHOUT lattice symmetries, matrix-free.
//
// sk_32_1.yaml (the reference's third large target, Makefile:129-141) is such a basis:
// C(32,16) = 6.0e8 states, 496 bonds of c * sigma.sigma — 1.5e11 matrix elements, too many to keep
// (csrc/sector_basis.hip keeps the 36-site kagome sector's 1.1e9), but cheap to regenerate,
// because a fixed-magnetisation basis in ascending order has a closed-form index (Lin tables):
//
//   state = (high word, low word);  index = offset[high] + rank[low]
//
// with rank[low] = position of the low word among the low words of its population count — itself
// two-level: rank = before[low >> 8][popcount] + rank8[low & 255], 9 KB of LDS instead of a
// 128 KB table, so that two workgroups of 1024 threads share a CU (the product is bound by the
// latency of its gathers).  One workgroup owns one high word: its states are consecutive, and a
// bond falls in one of three classes handled without per-lane address arithmetic:
//
//   both sites in the high word   the flip is uniform: x is read at the SAME ranks of another
//                                 high word's block — a coalesced stream
//   one site in each              the high part of the target is uniform (scalar offset), the
//                                 low part is an LDS rank lookup
//   both sites in the low word    same block, LDS rank lookup
//
// Only magnetisation-conserving bonds are accepted (off-diagonal weight on 01 <-> 10 only), which
// is what every model of the reference has.  gfx950 only.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <new>
#include <vector>

#include "asp.h"
#include "asp_common.hpp"
#include "operator_internal.hpp"

namespace {

using asp::DeviceBuffer;

constexpr int kThreads = 1024;

struct PlainBond {
  double diag[4];   // <src| M |src>, src = 2 * bit(first site) + bit(second site)
  double row1;      // <01| M |10>: row state 01
  double row2;      // <10| M |01>: row state 10
  uint32_t p, q;    // bit positions inside their words (first and second site of the bond)
  uint32_t kind;    // 0: both high, 1: first high / second low, 2: first low / second high, 3: both low
  uint32_t pad;
};

struct PlainArgs {
  const PlainBond *bonds;
  uint32_t num_bonds;
  uint32_t lo_bits;
  int32_t weight;
  const uint64_t *offset;       // [2^hi_bits + 1]
  const uint16_t *before;       // [2^(lo_bits - 8)][lo_bits + 1]: low words of that class with a smaller top part
  const uint16_t *rank8;        // [256]: rank of the bottom byte among the bytes of its population count
  uint32_t before_entries;
  const uint32_t *words;        // low words grouped by population count
  const uint32_t *class_start;  // [lo_bits + 2]
  const double *x;
  double *y;
  uint64_t *states;  // k_plain_states only
};

__global__ __launch_bounds__(kThreads) void k_plain_matvec(PlainArgs a) {
  extern __shared__ uint16_t tables[];  // before[...], then rank8[256]
  const uint32_t hi = blockIdx.x;
  const int32_t k = a.weight - __popc(hi);
  if (k < 0 || k > static_cast<int32_t>(a.lo_bits)) return;  // whole workgroup
  uint16_t *rank8 = tables + a.before_entries;
  for (uint32_t i = threadIdx.x; i < a.before_entries; i += kThreads) tables[i] = a.before[i];
  for (uint32_t i = threadIdx.x; i < 256u; i += kThreads) rank8[i] = a.rank8[i];
  __syncthreads();
  const uint32_t classes = a.lo_bits + 1u;
  // position of a low word among the low words of its population count
  auto rank_of = [&](uint32_t low) -> uint32_t {
    return static_cast<uint32_t>(tables[(low >> 8) * classes + static_cast<uint32_t>(__popc(low))]) +
           rank8[low & 255u];
  };
  const uint32_t begin = a.class_start[k];
  const uint32_t size = a.class_start[k + 1] - begin;
  const uint64_t base = a.offset[hi];
  for (uint32_t r0 = 0; r0 < size; r0 += kThreads) {
    const uint32_t r = r0 + threadIdx.x;
    const bool live = r < size;
    const uint32_t lo = live ? a.words[begin + r] : 0u;
    const double mine = live ? a.x[base + r] : 0.0;
    double acc = 0.0, diagonal = 0.0;
    for (uint32_t b = 0; b < a.num_bonds; ++b) {  // uniform loop; bond fields are scalar
      const PlainBond &bond = a.bonds[b];
      if (bond.kind == 0) {
        const uint32_t bp = (hi >> bond.p) & 1u, bq = (hi >> bond.q) & 1u;
        const uint32_t src = 2u * bp + bq;
        diagonal += bond.diag[src];
        if (bp != bq) {
          const uint64_t other = a.offset[hi ^ ((1u << bond.p) | (1u << bond.q))];
          const double c = bp ? bond.row2 : bond.row1;
          if (live) acc = __fma_rn(c, a.x[other + r], acc);
        }
      } else if (bond.kind == 3) {
        const uint32_t bp = (lo >> bond.p) & 1u, bq = (lo >> bond.q) & 1u;
        const uint32_t src = 2u * bp + bq;
        const double d = src & 2u ? (src & 1u ? bond.diag[3] : bond.diag[2])
                                  : (src & 1u ? bond.diag[1] : bond.diag[0]);
        diagonal += d;
        if (live) {
          if (bp != bq) {
            const uint32_t target = lo ^ ((1u << bond.p) | (1u << bond.q));
            acc = __fma_rn(bp ? bond.row2 : bond.row1, a.x[base + rank_of(target)], acc);
          }
        }
      } else {
        // one site in each word: `up` = the bit in the high word (uniform)
        const bool first_high = bond.kind == 1;
        const uint32_t hbit = first_high ? bond.p : bond.q, lbit = first_high ? bond.q : bond.p;
        const uint32_t bh = (hi >> hbit) & 1u, bl = (lo >> lbit) & 1u;
        const uint32_t src = first_high ? 2u * bh + bl : 2u * bl + bh;
        const double d = src & 2u ? (src & 1u ? bond.diag[3] : bond.diag[2])
                                  : (src & 1u ? bond.diag[1] : bond.diag[0]);
        diagonal += d;
        if (live) {
          if (bh != bl) {
            const uint64_t other = a.offset[hi ^ (1u << hbit)];
            // row state: first site's bit * 2 + second site's bit; 10 -> row2, 01 -> row1
            const uint32_t first_bit = first_high ? bh : bl;
            acc = __fma_rn(first_bit ? bond.row2 : bond.row1,
                           a.x[other + rank_of(lo ^ (1u << lbit))], acc);
          }
        }
      }
    }
    if (live) a.y[base + r] = __fma_rn(diagonal, mine, acc);
  }
}

__global__ __launch_bounds__(kThreads) void k_plain_states(PlainArgs a) {
  const uint32_t hi = blockIdx.x;
  const int32_t k = a.weight - __popc(hi);
  if (k < 0 || k > static_cast<int32_t>(a.lo_bits)) return;
  const uint32_t begin = a.class_start[k];
  const uint32_t size = a.class_start[k + 1] - begin;
  const uint64_t base = a.offset[hi];
  for (uint32_t r = threadIdx.x; r < size; r += kThreads) {
    a.states[base + r] = (static_cast<uint64_t>(hi) << a.lo_bits) | a.words[begin + r];
  }
}

}  // namespace

struct asp_plain_basis {
  uint32_t n = 0, lo_bits = 0, hi_bits = 0;
  int32_t weight = 0;
  uint64_t dimension = 0;
  uint32_t num_bonds = 0;
  DeviceBuffer<PlainBond> d_bonds;
  DeviceBuffer<uint64_t> d_offset;
  DeviceBuffer<uint16_t> d_before, d_rank8;
  uint32_t before_entries = 0;
  DeviceBuffer<uint32_t> d_words, d_class_start;
  PlainArgs args() const {
    PlainArgs a{};
    a.bonds = d_bonds.ptr;
    a.num_bonds = num_bonds;
    a.lo_bits = lo_bits;
    a.weight = weight;
    a.offset = d_offset.ptr;
    a.before = d_before.ptr;
    a.rank8 = d_rank8.ptr;
    a.before_entries = before_entries;
    a.words = d_words.ptr;
    a.class_start = d_class_start.ptr;
    return a;
  }
};

extern "C" {

int asp_plain_basis_create(asp_operator const *op, int32_t hamming_weight, asp_plain_basis **out) {
  asp_clear_error();
  if (!out) return asp::set_error(ASP_ERR_INVALID, "null output pointer");
  *out = nullptr;
  if (!op) return asp::set_error(ASP_ERR_INVALID, "null operator");
  ASP_TRY(asp::bind_device());
  if (op->num_permutations != 0) {
    return asp::set_error(ASP_ERR_INVALID, "the matrix-free product handles bases without symmetries "
                                           "(symmetry sectors: asp_sector_rows)");
  }
  const uint32_t n = op->number_spins;
  if (n < 2 || n > 36) return asp::set_error(ASP_ERR_TOO_LARGE, "2..36 spins");
  if (hamming_weight < 0 || hamming_weight > static_cast<int32_t>(n)) {
    return asp::set_error(ASP_ERR_INVALID, "a magnetisation sector is required (0 <= hamming_weight <= n)");
  }
  const uint32_t lo_bits = std::min(16u, (n + 1u) / 2u), hi_bits = n - lo_bits;
  std::vector<PlainBond> bonds(op->num_bonds);
  for (uint32_t b = 0; b < op->num_bonds; ++b) {
    const asp::Bond &in = op->bonds[b];
    for (uint32_t dst = 0; dst < 4; ++dst) {
      for (uint32_t src = 0; src < 4; ++src) {
        const bool exchange = (dst == 1 && src == 2) || (dst == 2 && src == 1);
        if (dst != src && !exchange && in.m[dst * 4 + src] != 0.0) {
          return asp::set_error(ASP_ERR_INVALID, "bond %u does not conserve the magnetisation", b);
        }
      }
    }
    PlainBond &o = bonds[b];
    for (uint32_t s = 0; s < 4; ++s) o.diag[s] = in.m[s * 4 + s];
    o.row1 = in.m[1 * 4 + 2];
    o.row2 = in.m[2 * 4 + 1];
    const bool a_high = in.a >= lo_bits, b_high = in.b >= lo_bits;
    o.p = a_high ? in.a - lo_bits : in.a;
    o.q = b_high ? in.b - lo_bits : in.b;
    o.kind = a_high ? (b_high ? 0u : 1u) : (b_high ? 2u : 3u);
    o.pad = 0;
  }
  // bonds inside the high word first (their reads are streams), then mixed, then low
  std::stable_sort(bonds.begin(), bonds.end(), [](const PlainBond &l, const PlainBond &r) {
    auto key = [](uint32_t kind) { return kind == 0 ? 0 : (kind == 3 ? 2 : 1); };
    return key(l.kind) < key(r.kind);
  });
  std::vector<uint32_t> class_start(lo_bits + 2, 0), words(size_t{1} << lo_bits);
  for (uint32_t w = 0; w < (1u << lo_bits); ++w) ++class_start[static_cast<uint32_t>(__builtin_popcount(w)) + 1];
  for (uint32_t p = 0; p <= lo_bits; ++p) class_start[p + 1] += class_start[p];
  {
    std::vector<uint32_t> fill(class_start.begin(), class_start.end() - 1);
    for (uint32_t w = 0; w < (1u << lo_bits); ++w) words[fill[static_cast<uint32_t>(__builtin_popcount(w))]++] = w;
  }
  // rank of a low word inside its class = (words of the class with a smaller top part) + (rank
  // of its bottom byte among the bytes of the same population count): ascending order is
  // top-part major
  const uint32_t bottom_bits = std::min(lo_bits, 8u), top_values = 1u << (lo_bits - bottom_bits);
  std::vector<uint16_t> rank8(256, 0), before(static_cast<size_t>(top_values) * (lo_bits + 1), 0);
  {
    uint32_t seen[9] = {0};
    for (uint32_t v = 0; v < (1u << bottom_bits); ++v) rank8[v] = static_cast<uint16_t>(seen[__builtin_popcount(v)]++);
    for (uint32_t c = 0; c <= lo_bits; ++c) {
      uint32_t running = 0;
      for (uint32_t t = 0; t < top_values; ++t) {
        before[static_cast<size_t>(t) * (lo_bits + 1) + c] = static_cast<uint16_t>(running);
        const int32_t rest = static_cast<int32_t>(c) - __builtin_popcount(t);
        if (rest >= 0 && rest <= static_cast<int32_t>(bottom_bits)) running += seen[rest];  // C(bottom_bits, rest)
      }
    }
  }
  std::vector<uint64_t> offset((size_t{1} << hi_bits) + 1, 0);
  for (uint64_t h = 0; h < (1ull << hi_bits); ++h) {
    const int32_t k = hamming_weight - __builtin_popcountll(h);
    const uint64_t size = (k < 0 || k > static_cast<int32_t>(lo_bits)) ? 0 : class_start[k + 1] - class_start[k];
    offset[h + 1] = offset[h] + size;
  }
  asp_plain_basis *pb = new (std::nothrow) asp_plain_basis;
  if (!pb) return asp::set_error(ASP_ERR_ALLOC, "out of host memory");
  pb->n = n;
  pb->lo_bits = lo_bits;
  pb->hi_bits = hi_bits;
  pb->weight = hamming_weight;
  pb->dimension = offset.back();
  pb->num_bonds = op->num_bonds;
  int rc = pb->d_bonds.alloc(bonds.size());
  if (rc == ASP_OK) rc = pb->d_offset.alloc(offset.size());
  if (rc == ASP_OK) rc = pb->d_before.alloc(before.size());
  if (rc == ASP_OK) rc = pb->d_rank8.alloc(rank8.size());
  if (rc == ASP_OK) rc = pb->d_words.alloc(words.size());
  if (rc == ASP_OK) rc = pb->d_class_start.alloc(class_start.size());
  if (rc == ASP_OK) rc = pb->d_bonds.upload(bonds.data(), bonds.size(), nullptr);
  if (rc == ASP_OK) rc = pb->d_offset.upload(offset.data(), offset.size(), nullptr);
  if (rc == ASP_OK) rc = pb->d_before.upload(before.data(), before.size(), nullptr);
  if (rc == ASP_OK) rc = pb->d_rank8.upload(rank8.data(), rank8.size(), nullptr);
  if (rc == ASP_OK) rc = pb->d_words.upload(words.data(), words.size(), nullptr);
  if (rc == ASP_OK) rc = pb->d_class_start.upload(class_start.data(), class_start.size(), nullptr);
  if (rc == ASP_OK && hipStreamSynchronize(nullptr) != hipSuccess) {
    rc = asp::set_error(ASP_ERR_HIP, "upload of the basis tables failed");
  }
  pb->before_entries = static_cast<uint32_t>(before.size());
  if (rc != ASP_OK) {
    delete pb;
    return rc;
  }
  *out = pb;
  return ASP_OK;
}

void asp_plain_basis_destroy(asp_plain_basis *pb) {
  if (!pb) return;
  (void)asp::bind_device();
  delete pb;
}

uint64_t asp_plain_basis_dimension(asp_plain_basis const *pb) { return pb ? pb->dimension : 0; }

int asp_plain_basis_states(asp_plain_basis const *pb, uint64_t *states_dev) {
  asp_clear_error();
  if (!pb || !states_dev) return asp::set_error(ASP_ERR_INVALID, "null argument");
  ASP_TRY(asp::bind_device());
  asp::ScopedStream scoped;
  ASP_TRY(scoped.acquire());
  PlainArgs a = pb->args();
  a.states = states_dev;
  hipLaunchKernelGGL(k_plain_states, dim3(1u << pb->hi_bits), dim3(kThreads), 0, scoped.stream, a);
  ASP_HIP_TRY(hipGetLastError());
  ASP_HIP_TRY(hipStreamSynchronize(scoped.stream));
  return ASP_OK;
}

int asp_plain_matvec(asp_plain_basis const *pb, double const *x_dev, double *y_dev) {
  asp_clear_error();
  if (!pb || !x_dev || !y_dev) return asp::set_error(ASP_ERR_INVALID, "null argument");
  if (x_dev == y_dev) return asp::set_error(ASP_ERR_INVALID, "x and y must not alias");
  ASP_TRY(asp::bind_device());
  asp::ScopedStream scoped;
  ASP_TRY(scoped.acquire());
  PlainArgs a = pb->args();
  a.x = x_dev;
  a.y = y_dev;
  hipLaunchKernelGGL(k_plain_matvec, dim3(1u << pb->hi_bits), dim3(kThreads),
                     sizeof(uint16_t) * (pb->before_entries + 256u), scoped.stream, a);
  ASP_HIP_TRY(hipGetLastError());
  ASP_HIP_TRY(hipStreamSynchronize(scoped.stream));
  return ASP_OK;
}

}  // extern "C"
