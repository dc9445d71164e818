// Hamiltonian action on bit-packed basis states and the coupling build fused with
// it, on gfx950.  Replaces, for symmetry-free two-site operators, the chain
//   ls.Operator.batched_apply -> _batched_apply          common.py:85-106
//   _clipped_search_sorted + membership                   common.py:116-128,173
//   _make_ising_model_compute_elements                    common.py:71-82
//   csr_matrix; 0.5 * (M + M.T); sort_indices; tocoo      common.py:190-196
//   make_hamiltonian_extension's np.unique                common.py:516-522
// of the reference's make_ising_model, without ever materialising the ~37 (kagome_36)
// to ~257 (sk_32) connections per state in host memory.
//
// HBM layout: keys u64[K] sorted; psi f64[K]; bonds (struct Bond, 144 B) [B];
//   slots u64[2^s >= 2K]  open-addressing hash of keys {fingerprint:32 | index+1:32};
//   row_nnz u32[K] -> row_start i64[K+1] (device scan); out row/col i32[nnz], val f64[nnz].
// Mapping: one wavefront per basis state, one lane per bond (bonds in chunks of 64).  Every
// connection of a row is decided by the lane that owns its bond: the target key is one XOR,
// its membership one hash probe (table and keys are L2-resident: 24 B/state), and BOTH matrix
// elements of the pair — M_ij seen from row i and M_ji seen from row j — follow from the
// bond's 4x4 matrix, because state j carries the bond's bits in the swapped configuration.  So
// J = (M + M^T)/2 needs no transpose pass.  Rows are sorted by column in LDS (rank by counting).
// Compulsory HBM traffic: 16 B/state in, 16 B/non-zero out; integer/latency-bound, no MFMA.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "asp_common.hpp"
#include "operator_internal.hpp"

namespace {

using asp::Bond;
using asp::DeviceBuffer;
using asp::SymmetryArgs;

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

__global__ __launch_bounds__(kThreads) void k_key_insert(const uint64_t *__restrict__ keys,
                                                        uint64_t n,
                                                        unsigned long long *__restrict__ slots,
                                                        uint64_t mask) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = mix64(keys[i]);
  const unsigned long long entry = (h & 0xFFFFFFFF00000000ull) | (i + 1);
  uint64_t at = h & mask;
  while (atomicCAS(&slots[at], 0ull, entry) != 0ull) at = (at + 1) & mask;
}

// Index of `needle` in keys, or -1.
__device__ __forceinline__ int64_t find_key(const unsigned long long *__restrict__ slots,
                                            uint64_t mask, const uint64_t *__restrict__ keys,
                                            uint64_t needle) {
  const uint64_t h = mix64(needle);
  const uint32_t fingerprint = static_cast<uint32_t>(h >> 32);
  for (uint64_t at = h & mask;; at = (at + 1) & mask) {
    const unsigned long long slot = slots[at];
    if (slot == 0) return -1;
    if (static_cast<uint32_t>(slot >> 32) != fingerprint) continue;
    const uint32_t idx = static_cast<uint32_t>(slot) - 1u;
    if (keys[idx] == needle) return idx;
  }
}

__device__ __forceinline__ uint32_t wave_exclusive_scan_u32(uint32_t v, uint32_t lane,
                                                            uint32_t *total) {
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= static_cast<uint32_t>(d)) incl += up;
  }
  *total = __shfl(incl, 63, 64);
  return incl - v;
}

// Bits of the bond in `key`: src = 2 * b_a + b_b.
__device__ __forceinline__ uint32_t bond_state(const Bond &bond, uint64_t key) {
  return static_cast<uint32_t>(((key >> bond.a) & 1ull) * 2ull + ((key >> bond.b) & 1ull));
}

// ---------------------------------------------------------------------------
// batched_apply
// ---------------------------------------------------------------------------

// EMIT = false: other_counts only.  EMIT = true: entries at offsets[i]...
template <bool EMIT>
__global__ __launch_bounds__(kThreads) void k_apply(const Bond *__restrict__ bonds,
                                                   uint32_t num_bonds,
                                                   const uint64_t *__restrict__ keys, uint64_t n,
                                                   const int64_t *__restrict__ offsets,
                                                   int64_t *__restrict__ counts,
                                                   uint64_t *__restrict__ other_keys,
                                                   double *__restrict__ other_coeffs) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kWaves + (threadIdx.x >> 6);
  if (i >= n) return;  // whole wavefront
  const uint64_t key = keys[i];
  const int64_t base = EMIT ? offsets[i] : 0;
  uint32_t written = 1;  // slot 0 is the diagonal entry
  double diagonal = 0.0;
  for (uint32_t first = 0; first < num_bonds; first += 64u) {
    const uint32_t bi = first + lane;
    const bool live = bi < num_bonds;
    uint32_t src = 0, mine = 0;
    double dval = 0.0;
    if (live) {
      const Bond &bond = bonds[bi];
      src = bond_state(bond, key);
      dval = bond.m[src * 4u + src];
#pragma unroll
      for (uint32_t dst = 0; dst < 4; ++dst) {
        if (dst != src && bond.m[dst * 4u + src] != 0.0) ++mine;
      }
    }
    // the diagonal is the left-to-right sum over bonds (numpy's `diagonal += m[k, k]`)
    const uint32_t here = min(64u, num_bonds - first);
    for (uint32_t j = 0; j < here; ++j) diagonal = __dadd_rn(diagonal, __shfl(dval, j, 64));
    uint32_t total;
    const uint32_t before = wave_exclusive_scan_u32(mine, lane, &total);
    if (EMIT && live && mine) {
      const Bond &bond = bonds[bi];
      int64_t at = base + written + before;
#pragma unroll
      for (uint32_t dst = 0; dst < 4; ++dst) {
        const double c = bond.m[dst * 4u + src];
        if (dst != src && c != 0.0) {
          other_keys[at] = key ^ bond.flip[src ^ dst];
          other_coeffs[at] = c;
          ++at;
        }
      }
    }
    written += total;
  }
  if (lane == 0) {
    if (EMIT) {
      other_keys[base] = key;
      other_coeffs[base] = diagonal;
    } else {
      counts[i] = written;
    }
  }
}

// ---------------------------------------------------------------------------
// fused coupling build
// ---------------------------------------------------------------------------

struct IsingArgs {
  const Bond *bonds;
  const uint64_t *keys;
  const double *psi;
  const unsigned long long *slots;
  uint64_t mask;
  const int64_t *row_start;  // EMIT only
  uint32_t *row_nnz;         // count pass
  int32_t *row;
  int32_t *col;
  double *val;
  uint64_t num_spins;
  uint32_t num_bonds;
  uint32_t row_capacity;  // LDS entries per wavefront (EMIT only)
};

// M_rj + M_jr from the lane's bond for transition src -> dst, 0 when j is outside the cluster.
__device__ __forceinline__ double pair_coupling(const IsingArgs &a, const Bond &bond, uint64_t key,
                                                double psi_r, uint32_t src, uint32_t dst,
                                                int64_t *col) {
  const double fwd = bond.m[dst * 4u + src];  // H_jr: row r's connection to j
  const double rev = bond.m[src * 4u + dst];  // H_rj: row j's connection back to r
  *col = -1;
  if (fwd == 0.0 && rev == 0.0) return 0.0;
  const int64_t j = find_key(a.slots, a.mask, a.keys, key ^ bond.flip[src ^ dst]);
  if (j < 0) return 0.0;
  *col = j;
  const double psi_j = fabs(a.psi[j]);
  // M_rj = (coeff * |psi_j|) * |psi_r| (common.py:79,81), likewise M_jr from row j; an
  // element the operator does not have is an absent entry, i.e. contributes nothing
  double sum = 0.0;
  if (fwd != 0.0) sum = __dadd_rn(sum, __dmul_rn(__dmul_rn(fwd, psi_j), psi_r));
  if (rev != 0.0) sum = __dadd_rn(sum, __dmul_rn(__dmul_rn(rev, psi_r), psi_j));
  return sum;
}

template <bool EMIT>
__global__ __launch_bounds__(kThreads) void k_ising_rows(IsingArgs a) {
  extern __shared__ __align__(16) uint8_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint64_t r = static_cast<uint64_t>(blockIdx.x) * kWaves + wave;
  if (r >= a.num_spins) return;  // whole wavefront; no workgroup barrier below
  // per-wavefront staging area: cols i32[capacity] | vals f64[capacity]
  double *vals = reinterpret_cast<double *>(lds) + static_cast<size_t>(wave) * a.row_capacity;
  int32_t *cols = reinterpret_cast<int32_t *>(reinterpret_cast<double *>(lds) +
                                              static_cast<size_t>(kWaves) * a.row_capacity) +
                  static_cast<size_t>(wave) * a.row_capacity;
  const uint64_t key = a.keys[r];
  const double psi_r = fabs(a.psi[r]);
  uint32_t kept = 0;
  double diagonal = 0.0;
  for (uint32_t first = 0; first < a.num_bonds; first += 64u) {
    const uint32_t bi = first + lane;
    const bool live = bi < a.num_bonds;
    double dval = 0.0;
    double v[3] = {0.0, 0.0, 0.0};
    int64_t c[3] = {-1, -1, -1};
    uint32_t mine = 0;
    if (live) {
      const Bond &bond = a.bonds[bi];
      const uint32_t src = bond_state(bond, key);
      dval = bond.m[src * 4u + src];
      uint32_t slot = 0;
#pragma unroll
      for (uint32_t dst = 0; dst < 4; ++dst) {
        if (dst == src) continue;
        int64_t col;
        // scipy keeps an entry of M + M^T iff the sum is non-zero, then scales it by 0.5
        const double x = pair_coupling(a, bond, key, psi_r, src, dst, &col);
        if (x != 0.0) {
          v[slot] = __dmul_rn(0.5, x);
          c[slot] = col;
          ++slot;
        }
      }
      mine = slot;
    }
    const uint32_t here = min(64u, a.num_bonds - first);
    for (uint32_t j = 0; j < here; ++j) diagonal = __dadd_rn(diagonal, __shfl(dval, j, 64));
    uint32_t total;
    const uint32_t before = wave_exclusive_scan_u32(mine, lane, &total);
    if (EMIT) {
#pragma unroll
      for (uint32_t s = 0; s < 3; ++s) {
        if (s < mine) {
          cols[kept + before + s] = static_cast<int32_t>(c[s]);
          vals[kept + before + s] = v[s];
        }
      }
    }
    kept += total;
  }
  // J_rr = 0.5 * (M_rr + M_rr), M_rr = (diag * |psi_r|) * |psi_r|
  const double m_rr = __dmul_rn(__dmul_rn(diagonal, psi_r), psi_r);
  const double twice = __dadd_rn(m_rr, m_rr);
  const double j_rr = __dmul_rn(0.5, twice);
  const bool has_diag = twice != 0.0;
  if (EMIT && has_diag && lane == 0) {
    cols[kept] = static_cast<int32_t>(r);
    vals[kept] = j_rr;
  }
  kept += has_diag ? 1u : 0u;
  if (!EMIT) {
    if (lane == 0) a.row_nnz[r] = kept;
    return;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  // columns of a row are distinct: rank = number of smaller columns
  const int64_t out = a.row_start[r];
  for (uint32_t e = lane; e < kept; e += 64u) {
    const int32_t mine_col = cols[e];
    uint32_t rank = 0;
    for (uint32_t o = 0; o < kept; ++o) rank += cols[o] < mine_col ? 1u : 0u;
    a.row[out + rank] = static_cast<int32_t>(r);
    a.col[out + rank] = mine_col;
    a.val[out + rank] = vals[e];
  }
}

// ---------------------------------------------------------------------------
// coupling build when a row may reach a state more than once (symmetry-adapted bases,
// single-site flips): the reference's arithmetic with its duplicates
// ---------------------------------------------------------------------------
// make_ising_model builds csr_matrix((elements, clipped indices, offsets)) — unsorted, with
// duplicate columns and explicit zeros for targets outside the cluster — and lets scipy form
// 0.5 * (M + M.T) (common.py:190-196).  scipy's csr + csr on such input sums, per row, the
// duplicates of each operand in storage order starting from 0 (A_row[col] += Ax), adds the two
// sums, and keeps the entry iff the result is non-zero; M.T's duplicates come in the storage
// order of the row they sit in.  So with  Mhat_rj = ((0 + e_1) + e_2) + ...  over row r's entries
// with column j in connection order:  J_rj = 0.5 * (Mhat_rj + Mhat_jr), kept iff the sum != 0.
// Entries outside the cluster carry the value 0 in the reference and never change a sum.
//   k_merge_rows   one wavefront per row: columns through the hash, elements
//                  (coeff * |psi_j|) * |psi_r|, duplicates summed in connection order, the row's
//                  distinct columns sorted -> Mhat as CSR
//   k_sym_rows     one wavefront per row: Mhat_jr by bisection in row j, prune, COO out;
//                  counts entries whose mirror is missing from Mhat (one-directional matrix
//                  elements): the caller then leaves the job to the host route

struct MergeArgs {
  const uint64_t *keys;
  const double *psi;
  const unsigned long long *slots;
  uint64_t mask;
  const int64_t *offsets;       // connections of row r: [offsets[r], offsets[r+1])
  const uint64_t *other_keys;   // targets (representatives for a symmetric basis)
  const double *other_coeffs;
  uint64_t num_spins;
  uint32_t row_capacity;        // LDS entries per wavefront
  const int64_t *mrow_start;    // EMIT
  uint32_t *mrow_nnz;           // count pass
  int32_t *mcol;
  double *mval;
};

template <bool EMIT>
__global__ __launch_bounds__(kThreads) void k_merge_rows(MergeArgs a) {
  extern __shared__ __align__(16) uint8_t lds[];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint64_t r = static_cast<uint64_t>(blockIdx.x) * kWaves + wave;
  if (r >= a.num_spins) return;  // whole wavefront; only wave-level barriers below
  // per wavefront: vals f64[cap] | cols i32[cap] | leader u8[cap]
  double *vals = reinterpret_cast<double *>(lds) + static_cast<size_t>(wave) * a.row_capacity;
  int32_t *cols = reinterpret_cast<int32_t *>(reinterpret_cast<double *>(lds) +
                                              static_cast<size_t>(kWaves) * a.row_capacity) +
                  static_cast<size_t>(wave) * a.row_capacity;
  uint8_t *leader = reinterpret_cast<uint8_t *>(reinterpret_cast<int32_t *>(
                        reinterpret_cast<double *>(lds) + static_cast<size_t>(kWaves) * a.row_capacity) +
                    static_cast<size_t>(kWaves) * a.row_capacity) +
                    static_cast<size_t>(wave) * a.row_capacity;
  const int64_t begin = a.offsets[r];
  const uint32_t n = static_cast<uint32_t>(a.offsets[r + 1] - begin);
  const double psi_r = fabs(a.psi[r]);
  for (uint32_t i = lane; i < n; i += 64u) {
    const int64_t j = find_key(a.slots, a.mask, a.keys, a.other_keys[begin + i]);
    cols[i] = static_cast<int32_t>(j);
    // (coeff * |psi_j|) * |psi_r|, each product rounded (common.py:79,81)
    vals[i] = j < 0 ? 0.0 : __dmul_rn(__dmul_rn(a.other_coeffs[begin + i], fabs(a.psi[j])), psi_r);
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  uint32_t mine = 0;
  for (uint32_t i = lane; i < n; i += 64u) {
    bool first = cols[i] >= 0;
    for (uint32_t k = 0; k < i && first; ++k) first = cols[k] != cols[i];
    leader[i] = first ? 1 : 0;
    mine += first ? 1u : 0u;
  }
  uint32_t distinct = mine;
#pragma unroll
  for (int step = 1; step < 64; step <<= 1) distinct += __shfl_xor(distinct, step, 64);
  if (!EMIT) {
    if (lane == 0) a.mrow_nnz[r] = distinct;
    return;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  const int64_t out = a.mrow_start[r];
  for (uint32_t i = lane; i < n; i += 64u) {
    if (!leader[i]) continue;
    const int32_t c = cols[i];
    double sum = 0.0;  // scipy: A_row[col] starts at 0 and takes the duplicates in storage order
    uint32_t rank = 0;
    for (uint32_t k = 0; k < n; ++k) {
      if (k >= i && cols[k] == c) sum = __dadd_rn(sum, vals[k]);
      rank += (leader[k] && cols[k] < c) ? 1u : 0u;
    }
    a.mcol[out + rank] = c;
    a.mval[out + rank] = sum;
  }
}

struct SymArgs {
  const int64_t *mrow_start;  // num_spins + 1
  const int32_t *mcol;
  const double *mval;
  uint64_t num_spins;
  const int64_t *row_start;  // EMIT
  uint32_t *row_nnz;         // count pass
  unsigned long long *missing_mirrors;
  int32_t *row;
  int32_t *col;
  double *val;
};

template <bool EMIT>
__global__ __launch_bounds__(kThreads) void k_sym_rows(SymArgs a) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t r = static_cast<uint64_t>(blockIdx.x) * kWaves + (threadIdx.x >> 6);
  if (r >= a.num_spins) return;
  const int64_t begin = a.mrow_start[r], end = a.mrow_start[r + 1];
  int64_t written = EMIT ? a.row_start[r] : 0;
  uint32_t kept = 0;
  for (int64_t base = begin; base < end; base += 64) {
    const int64_t e = base + lane;
    bool keep = false;
    int32_t j = -1;
    double x = 0.0;
    if (e < end) {
      j = a.mcol[e];
      const double forward = a.mval[e];
      // Mhat_jr: bisection among row j's sorted columns
      int64_t lo = a.mrow_start[j], hi = a.mrow_start[j + 1];
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (static_cast<uint64_t>(a.mcol[mid]) < r) {
          lo = mid + 1;
        } else {
          hi = mid;
        }
      }
      const bool mirrored = lo < a.mrow_start[j + 1] && static_cast<uint64_t>(a.mcol[lo]) == r;
      if (!mirrored && !EMIT) atomicAdd(a.missing_mirrors, 1ull);
      x = __dadd_rn(forward, mirrored ? a.mval[lo] : 0.0);  // (M + M.T)_rj
      keep = x != 0.0;
    }
    const uint64_t ballot = __ballot(keep);
    if (EMIT && keep) {
      const int64_t at = written + __popcll(ballot & ((1ull << lane) - 1ull));
      a.row[at] = static_cast<int32_t>(r);
      a.col[at] = j;
      a.val[at] = __dmul_rn(0.5, x);
    }
    written += __popcll(ballot);
    kept += static_cast<uint32_t>(__popcll(ballot));
  }
  if (!EMIT && lane == 0) a.row_nnz[r] = kept;
}

// ---------------------------------------------------------------------------
// extension: sorted unique targets
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(kThreads) void k_flag_first(const uint64_t *__restrict__ sorted,
                                                        uint64_t n, uint32_t *__restrict__ flag) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(kThreads) void k_scatter_first(const uint64_t *__restrict__ sorted,
                                                           uint64_t n,
                                                           const uint32_t *__restrict__ flag,
                                                           const int64_t *__restrict__ position,
                                                           uint64_t *__restrict__ out) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  if (flag[i]) out[position[i]] = sorted[i];
}

// ---------------------------------------------------------------------------
// sector-0 symmetry-adapted bases (annealing_sign_problem_amd/symmetry.py)
// ---------------------------------------------------------------------------
// Representative = smallest image under the group; norm^2 = (sum of the characters of the
// stabiliser) / |G|.  One lane per state, all lanes walk the permutations together, so the
// destination table (u8[P][64], site i -> table[g][i]) is read with scalar loads.  Spin inversion
// needs no second walk: the flipped image is the complement of the plain one.

struct StateInfo {
  uint64_t representative;
  double character;  // of an element mapping the state onto its representative
  double norm;       // 0: the state is outside the sector
};

__device__ __forceinline__ StateInfo state_info(const SymmetryArgs &g, uint64_t x) {
  uint64_t best = x;  // the identity is element 0
  int32_t stabiliser = 0;
  bool through_flip = false;
  for (uint32_t e = 0; e < g.num_permutations; ++e) {
    const uint8_t *dst = g.table + static_cast<size_t>(e) * 64u;
    uint64_t y = 0;
    for (uint32_t i = 0; i < g.number_spins; ++i) {
      y |= ((x >> i) & 1ull) << dst[i];
    }
    stabiliser += y == x ? 1 : 0;
    if (y < best) {
      best = y;
      through_flip = false;
    }
    if (g.inversion != 0) {
      const uint64_t z = ~y & g.mask;
      stabiliser += z == x ? g.inversion : 0;
      if (z < best) {
        best = z;
        through_flip = true;
      }
    }
  }
  const double order = static_cast<double>(g.num_permutations) * (g.inversion != 0 ? 2.0 : 1.0);
  StateInfo out;
  out.representative = best;
  out.character = (through_flip && g.inversion < 0) ? -1.0 : 1.0;
  // the same expression as the numpy reference: sqrt(max(stabiliser, 0) / |G|)
  out.norm = sqrt(static_cast<double>(stabiliser > 0 ? stabiliser : 0) / order);
  return out;
}

// norm of every source state (row)
__global__ __launch_bounds__(kThreads) void k_source_norms(SymmetryArgs g,
                                                          const uint64_t *__restrict__ keys,
                                                          uint64_t n, double *__restrict__ norms) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  norms[i] = state_info(g, keys[i]).norm;
}

// Every emitted connection: target -> representative, coefficient -> c * chi * norm(target) /
// norm(source).  The row of an entry is found by bisection in the offsets.
__global__ __launch_bounds__(kThreads) void k_symmetrise(SymmetryArgs g,
                                                        const int64_t *__restrict__ offsets,
                                                        uint64_t num_rows,
                                                        const double *__restrict__ source_norms,
                                                        uint64_t total,
                                                        uint64_t *__restrict__ other_keys,
                                                        double *__restrict__ other_coeffs) {
  const uint64_t e = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= total) return;
  uint64_t lo = 0, hi = num_rows;  // offsets[lo] <= e < offsets[hi]
  while (hi - lo > 1) {
    const uint64_t mid = (lo + hi) / 2;
    if (static_cast<uint64_t>(offsets[mid]) <= e) {
      lo = mid;
    } else {
      hi = mid;
    }
  }
  const StateInfo info = state_info(g, other_keys[e]);
  other_keys[e] = info.representative;
  // numpy: values * character * norm / source_norm, left to right
  other_coeffs[e] = __ddiv_rn(__dmul_rn(__dmul_rn(other_coeffs[e], info.character), info.norm),
                              source_norms[lo]);
}

// k_source_norms + k_symmetrise for whole rows, through the LINEARITY of a bit permutation over
// XOR: a target of row r is t = s ^ m with m the few bits its bond flips, so its image under group
// element e is image_e(s) ^ image_e(m).  One wavefront per row: the images of the source state
// under all elements once (lanes over the elements; LDS), which also gives the source's
// stabiliser, i.e. its norm; then lanes over the row's targets, every element costing one LDS
// broadcast read, two table bytes and an XOR instead of a number_spins-bit permutation.  The
// elements are walked in the same order with the same comparisons as state_info, so the
// representatives, characters and norms are the same bits.  (Measured on the 36-site kagome model,
// 144 permutations x inversion: the two kernels it replaces were half of the device time of the
// sampled-cluster pipeline, profiles/r03_pipeline_greedy_kernel_stats.csv.)
// LDS: images u64[kWaves][P] | destination table, transposed, u8[number_spins][P].
__global__ __launch_bounds__(kThreads) void k_symmetrise_rows(SymmetryArgs g,
                                                             const uint64_t *__restrict__ keys,
                                                             uint64_t num_rows,
                                                             const int64_t *__restrict__ offsets,
                                                             double *__restrict__ source_norms,
                                                             uint64_t *__restrict__ other_keys,
                                                             double *__restrict__ other_coeffs) {
  extern __shared__ __align__(16) uint8_t symmetry_lds[];
  const uint32_t P = g.num_permutations, ns = g.number_spins;
  uint64_t *images = reinterpret_cast<uint64_t *>(symmetry_lds);
  uint8_t *where = symmetry_lds + sizeof(uint64_t) * kWaves * P;  // where[i * P + e]: destination of site i
  for (uint32_t idx = threadIdx.x; idx < ns * P; idx += kThreads) {
    const uint32_t i = idx / P, e = idx - i * P;
    where[idx] = g.table[static_cast<size_t>(e) * 64u + i];
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t r = static_cast<uint64_t>(blockIdx.x) * kWaves + wave;
  const bool row_ok = r < num_rows;
  uint64_t *mine = images + static_cast<size_t>(wave) * P;
  const uint64_t s = row_ok ? keys[r] : 0ull;  // (wavefront-uniform)
  __syncthreads();
  // ---- images of the source under every element; its stabiliser ----
  int32_t stabiliser = 0;
  if (row_ok) {
    for (uint32_t e = lane; e < P; e += 64u) {
      uint64_t y = 0;
      for (uint64_t x = s; x != 0; x &= x - 1) {
        y |= 1ull << where[static_cast<uint32_t>(__builtin_ctzll(x)) * P + e];
      }
      mine[e] = y;
      stabiliser += y == s ? 1 : 0;
      if (g.inversion != 0) stabiliser += (~y & g.mask) == s ? g.inversion : 0;
    }
  }
  for (int step = 1; step < 64; step <<= 1) stabiliser += __shfl_xor(stabiliser, step, 64);
  __syncthreads();  // (the images were written by other lanes of this wavefront)
  if (!row_ok) return;
  const double order = static_cast<double>(P) * (g.inversion != 0 ? 2.0 : 1.0);
  // the same expression as state_info / the numpy reference: sqrt(max(stabiliser, 0) / |G|)
  const double source_norm = sqrt(static_cast<double>(stabiliser > 0 ? stabiliser : 0) / order);
  if (lane == 0) source_norms[r] = source_norm;
  // ---- the row's targets, 64 at a time ----
  const int64_t begin = offsets[r], end = offsets[r + 1];
  for (int64_t base = begin; base < end; base += 64) {
    const int64_t at = base + lane;
    const bool on = at < end;
    const uint64_t t = on ? other_keys[at] : s;
    const uint64_t m = t ^ s;
    const int flipped = __popcll(m);
    // the usual case, at most two flipped bits: their sites once, outside the walk over the group
    const uint32_t i0 = flipped ? static_cast<uint32_t>(__builtin_ctzll(m)) * P : 0u;
    const uint32_t i1 = flipped ? static_cast<uint32_t>(63 - __builtin_clzll(m)) * P : 0u;
    uint64_t best = t;  // the identity is element 0
    int32_t fixed = 0;
    bool through_flip = false;
    for (uint32_t e = 0; e < P; ++e) {
      uint64_t image_m = 0;
      if (flipped > 2) {
        for (uint64_t x = m; x != 0; x &= x - 1) {
          image_m |= 1ull << where[static_cast<uint32_t>(__builtin_ctzll(x)) * P + e];
        }
      } else if (flipped != 0) {
        image_m = (1ull << where[i0 + e]) | (1ull << where[i1 + e]);
      }
      const uint64_t y = mine[e] ^ image_m;
      fixed += y == t ? 1 : 0;
      if (y < best) {
        best = y;
        through_flip = false;
      }
      if (g.inversion != 0) {
        const uint64_t z = ~y & g.mask;
        fixed += z == t ? g.inversion : 0;
        if (z < best) {
          best = z;
          through_flip = true;
        }
      }
    }
    if (on) {
      const double character = (through_flip && g.inversion < 0) ? -1.0 : 1.0;
      const double norm = sqrt(static_cast<double>(fixed > 0 ? fixed : 0) / order);
      other_keys[at] = best;
      // numpy: values * character * norm / source_norm, left to right (as k_symmetrise)
      other_coeffs[at] = __ddiv_rn(__dmul_rn(__dmul_rn(other_coeffs[at], character), norm), source_norm);
    }
  }
}

thread_local float g_last_ms = 0.0f;

struct Timer {
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t stream = nullptr;
  int start(hipStream_t s) {
    stream = s;
    ASP_HIP_TRY(hipEventCreate(&ev0));
    ASP_HIP_TRY(hipEventCreate(&ev1));
    ASP_HIP_TRY(hipEventRecord(ev0, stream));
    return ASP_OK;
  }
  int stop() {
    ASP_HIP_TRY(hipEventRecord(ev1, stream));
    return ASP_OK;
  }
  void finish() {
    if (ev0 && ev1) (void)hipEventElapsedTime(&g_last_ms, ev0, ev1);
  }
  ~Timer() {
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
  }
};

unsigned grid_for(uint64_t items, uint64_t per_block) {
  return static_cast<unsigned>((items + per_block - 1) / per_block);
}

}  // namespace

namespace {

// other_counts / offsets / total of a batch; on return the inputs are resident.
struct ApplyBatch {
  DeviceBuffer<uint64_t> d_keys;
  DeviceBuffer<int64_t> d_counts, d_offsets, d_scratch;
  uint64_t total = 0;
};

int count_connections(const asp_operator *op, uint64_t n, const uint64_t *keys, ApplyBatch *w,
                      hipStream_t stream) {
  ASP_TRY(w->d_keys.alloc(n));
  ASP_TRY(w->d_counts.alloc(n));
  ASP_TRY(w->d_offsets.alloc(n + 1));
  ASP_TRY(w->d_scratch.alloc(asp::scan_scratch_elems(n)));
  ASP_TRY(w->d_keys.upload(keys, n, stream));
  if (n > 0) {
    hipLaunchKernelGGL(k_apply<false>, dim3(grid_for(n, kWaves)), dim3(kThreads), 0, stream,
                       op->d_bonds.ptr, op->num_bonds, w->d_keys.ptr, n, nullptr, w->d_counts.ptr,
                       nullptr, nullptr);
    ASP_HIP_TRY(hipGetLastError());
  }
  ASP_TRY(asp::exclusive_scan_i64(w->d_counts.ptr, n, w->d_offsets.ptr, w->d_scratch.ptr, stream));
  int64_t total = 0;
  ASP_HIP_TRY(hipMemcpyAsync(&total, w->d_offsets.ptr + n, sizeof total, hipMemcpyDeviceToHost,
                             stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  w->total = static_cast<uint64_t>(total);
  return ASP_OK;
}

// After k_apply<true>: representatives and rescaled coefficients for a symmetry-adapted basis.
// Fails with ASP_ERR_INVALID when a source state lies outside the sector (norm 0).
int symmetrise_batch(const asp_operator *op, uint64_t n, const ApplyBatch &w, uint64_t *d_other,
                     double *d_coeffs, DeviceBuffer<double> *d_norms, hipStream_t stream) {
  if (op->num_permutations == 0 || n == 0) return ASP_OK;
  ASP_TRY(d_norms->alloc(n));
  const SymmetryArgs g = op->symmetry();
  // a wavefront per row (k_symmetrise_rows) when the group's tables fit the LDS; the entry-wise
  // kernels otherwise (ASP_SYMMETRISE_ROWS=0: always; tests compare the two)
  const size_t lds = sizeof(uint64_t) * kWaves * static_cast<size_t>(g.num_permutations) +
                     static_cast<size_t>(g.number_spins) * g.num_permutations;
  bool by_rows = lds <= 160u * 1024u;
  if (const char *env = std::getenv("ASP_SYMMETRISE_ROWS")) by_rows = by_rows && std::atoi(env) != 0;
  if (by_rows) {
    if (lds > 64u * 1024u) {
      ASP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_symmetrise_rows),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    }
    hipLaunchKernelGGL(k_symmetrise_rows, dim3(grid_for(n, kWaves)), dim3(kThreads), lds, stream, g,
                       w.d_keys.ptr, n, w.d_offsets.ptr, d_norms->ptr, d_other, d_coeffs);
  } else {
    hipLaunchKernelGGL(k_source_norms, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, stream, g,
                       w.d_keys.ptr, n, d_norms->ptr);
    hipLaunchKernelGGL(k_symmetrise, dim3(grid_for(w.total, kThreads)), dim3(kThreads), 0, stream, g,
                       w.d_offsets.ptr, n, d_norms->ptr, w.total, d_other, d_coeffs);
  }
  ASP_HIP_TRY(hipGetLastError());
  return ASP_OK;
}

// asp_operator_ising for operators whose rows may reach a state twice; `missing` receives the
// number of entries without a mirror (non-zero: the caller must not use the result).
int ising_with_duplicates(const asp_operator *op, uint64_t K, const uint64_t *keys,
                          const double *psi, uint64_t capacity, int32_t *row, int64_t *indptr,
                          int32_t *col, double *val, uint64_t *nnz, hipStream_t stream) {
  Timer timer;
  ApplyBatch w;
  DeviceBuffer<uint64_t> d_other;
  DeviceBuffer<double> d_coeffs, d_norms, d_psi, d_mval, d_val;
  DeviceBuffer<unsigned long long> d_slots, d_missing;
  DeviceBuffer<uint32_t> d_mrow_nnz, d_row_nnz;
  DeviceBuffer<int64_t> d_mrow_start, d_row_start, d_scratch;
  DeviceBuffer<int32_t> d_mcol, d_row, d_col;
  asp::StreamFence fence(stream);
  ASP_TRY(timer.start(stream));
  ASP_TRY(count_connections(op, K, keys, &w, stream));
  ASP_TRY(d_other.alloc(w.total));
  ASP_TRY(d_coeffs.alloc(w.total));
  hipLaunchKernelGGL(k_apply<true>, dim3(grid_for(K, kWaves)), dim3(kThreads), 0, stream,
                     op->d_bonds.ptr, op->num_bonds, w.d_keys.ptr, K, w.d_offsets.ptr, nullptr,
                     d_other.ptr, d_coeffs.ptr);
  ASP_HIP_TRY(hipGetLastError());
  ASP_TRY(symmetrise_batch(op, K, w, d_other.ptr, d_coeffs.ptr, &d_norms, stream));
  uint64_t slots_n = 1024;
  while (slots_n < 2 * K) slots_n <<= 1;
  ASP_TRY(d_psi.alloc(K));
  ASP_TRY(d_slots.alloc(slots_n));
  ASP_TRY(d_missing.alloc(1));
  ASP_TRY(d_mrow_nnz.alloc(K));
  ASP_TRY(d_row_nnz.alloc(K));
  ASP_TRY(d_mrow_start.alloc(K + 1));
  ASP_TRY(d_row_start.alloc(K + 1));
  ASP_TRY(d_scratch.alloc(asp::scan_scratch_elems(K)));
  ASP_TRY(d_psi.upload(psi, K, stream));
  ASP_HIP_TRY(hipMemsetAsync(d_slots.ptr, 0, slots_n * sizeof(unsigned long long), stream));
  ASP_HIP_TRY(hipMemsetAsync(d_missing.ptr, 0, sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(k_key_insert, dim3(grid_for(K, kThreads)), dim3(kThreads), 0, stream,
                     w.d_keys.ptr, K, d_slots.ptr, slots_n - 1);
  MergeArgs m{};
  m.keys = w.d_keys.ptr;
  m.psi = d_psi.ptr;
  m.slots = d_slots.ptr;
  m.mask = slots_n - 1;
  m.offsets = w.d_offsets.ptr;
  m.other_keys = d_other.ptr;
  m.other_coeffs = d_coeffs.ptr;
  m.num_spins = K;
  m.row_capacity = (op->max_connections + 7u) & ~7u;
  m.mrow_nnz = d_mrow_nnz.ptr;
  const size_t lds = static_cast<size_t>(kWaves) * m.row_capacity * (sizeof(double) + sizeof(int32_t) + 1);
  if (lds > 160u * 1024u) {
    return asp::set_error(ASP_ERR_TOO_LARGE, "%u connections per row need %zu bytes of LDS",
                          op->max_connections, lds);
  }
  for (const void *kernel : {reinterpret_cast<const void *>(k_merge_rows<false>),
                             reinterpret_cast<const void *>(k_merge_rows<true>)}) {
    if (lds > 64u * 1024u) {
      ASP_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds)));
    }
  }
  hipLaunchKernelGGL(k_merge_rows<false>, dim3(grid_for(K, kWaves)), dim3(kThreads), lds, stream, m);
  ASP_HIP_TRY(hipGetLastError());
  ASP_TRY(asp::exclusive_scan_u32(d_mrow_nnz.ptr, K, d_mrow_start.ptr, d_scratch.ptr, stream));
  int64_t merged = 0;
  ASP_HIP_TRY(hipMemcpyAsync(&merged, d_mrow_start.ptr + K, sizeof merged, hipMemcpyDeviceToHost,
                             stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  ASP_TRY(d_mcol.alloc(static_cast<uint64_t>(merged)));
  ASP_TRY(d_mval.alloc(static_cast<uint64_t>(merged)));
  m.mrow_start = d_mrow_start.ptr;
  m.mcol = d_mcol.ptr;
  m.mval = d_mval.ptr;
  hipLaunchKernelGGL(k_merge_rows<true>, dim3(grid_for(K, kWaves)), dim3(kThreads), lds, stream, m);
  SymArgs y{};
  y.mrow_start = d_mrow_start.ptr;
  y.mcol = d_mcol.ptr;
  y.mval = d_mval.ptr;
  y.num_spins = K;
  y.row_nnz = d_row_nnz.ptr;
  y.missing_mirrors = d_missing.ptr;
  hipLaunchKernelGGL(k_sym_rows<false>, dim3(grid_for(K, kWaves)), dim3(kThreads), 0, stream, y);
  ASP_HIP_TRY(hipGetLastError());
  ASP_TRY(asp::exclusive_scan_u32(d_row_nnz.ptr, K, d_row_start.ptr, d_scratch.ptr, stream));
  int64_t total = 0;
  unsigned long long missing = 0;
  ASP_HIP_TRY(hipMemcpyAsync(&total, d_row_start.ptr + K, sizeof total, hipMemcpyDeviceToHost, stream));
  ASP_HIP_TRY(hipMemcpyAsync(&missing, d_missing.ptr, sizeof missing, hipMemcpyDeviceToHost, stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  if (missing != 0) {
    return asp::set_error(ASP_ERR_INVALID,
                          "%llu couplings have no mirror element (one-directional matrix elements): "
                          "use the host route", missing);
  }
  *nnz = static_cast<uint64_t>(total);
  if (capacity == 0 && !row && !indptr && !col && !val) {  // sizing call
    ASP_TRY(timer.stop());
    ASP_HIP_TRY(hipStreamSynchronize(stream));
    timer.finish();
    return ASP_OK;
  }
  if (*nnz > capacity || (!row && !indptr) || !col || !val) {
    return asp::set_error(ASP_ERR_INVALID, "%llu couplings do not fit capacity %llu",
                          (unsigned long long)*nnz, (unsigned long long)capacity);
  }
  ASP_TRY(d_row.alloc(*nnz));
  ASP_TRY(d_col.alloc(*nnz));
  ASP_TRY(d_val.alloc(*nnz));
  y.row_start = d_row_start.ptr;
  y.row = d_row.ptr;
  y.col = d_col.ptr;
  y.val = d_val.ptr;
  hipLaunchKernelGGL(k_sym_rows<true>, dim3(grid_for(K, kWaves)), dim3(kThreads), 0, stream, y);
  ASP_HIP_TRY(hipGetLastError());
  ASP_TRY(timer.stop());
  if (row) ASP_TRY(d_row.download(row, *nnz, stream));
  if (indptr) ASP_TRY(d_row_start.download(indptr, K + 1, stream));
  ASP_TRY(d_col.download(col, *nnz, stream));
  ASP_TRY(d_val.download(val, *nnz, stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  timer.finish();
  return ASP_OK;
}

int check_operator(const asp_operator *op) {
  if (!op) return asp::set_error(ASP_ERR_INVALID, "null operator");
  return asp::bind_device();
}

}  // namespace

extern "C" {

float asp_operator_last_ms(void) { return g_last_ms; }

int asp_operator_create(uint32_t number_spins, uint32_t num_bonds, uint8_t const *site_a,
                        uint8_t const *site_b, double const *matrices, asp_operator **out) {
  asp_clear_error();
  if (!out) return asp::set_error(ASP_ERR_INVALID, "null output pointer");
  *out = nullptr;
  if (number_spins == 0 || number_spins > 64) {
    return asp::set_error(ASP_ERR_INVALID, "number_spins must be in 1..64 (got %u)", number_spins);
  }
  if (num_bonds && (!site_a || !site_b || !matrices)) {
    return asp::set_error(ASP_ERR_INVALID, "null bond arrays");
  }
  ASP_TRY(asp::require_device());
  asp_operator *op = new (std::nothrow) asp_operator;
  if (!op) return asp::set_error(ASP_ERR_ALLOC, "out of host memory");
  op->number_spins = number_spins;
  op->num_bonds = num_bonds;
  op->bonds.resize(num_bonds);
  std::vector<uint64_t> masks;
  uint64_t max_conn = 1;
  for (uint32_t k = 0; k < num_bonds; ++k) {
    Bond &bond = op->bonds[k];
    const uint32_t a = site_a[k], b = site_b[k];
    if (a >= number_spins || b >= number_spins || a == b) {
      delete op;
      return asp::set_error(ASP_ERR_INVALID, "invalid bond %u: (%u, %u)", k, a, b);
    }
    for (int e = 0; e < 16; ++e) {
      const double x = matrices[static_cast<size_t>(k) * 16 + e];
      if (!(x == x) || x - x != 0.0) {
        delete op;
        return asp::set_error(ASP_ERR_INVALID, "bond %u has a non-finite matrix element", k);
      }
      bond.m[e] = x;
    }
    bond.a = a;
    bond.b = b;
    bond.flip[0] = 0;
    bond.flip[1] = 1ull << b;
    bond.flip[2] = 1ull << a;
    bond.flip[3] = (1ull << a) | (1ull << b);
    uint32_t worst = 0;
    bool used[4] = {false, false, false, false};
    for (uint32_t src = 0; src < 4; ++src) {
      uint32_t here = 0;
      for (uint32_t dst = 0; dst < 4; ++dst) {
        if (dst == src) continue;
        if (bond.m[dst * 4 + src] != 0.0) ++here;
        if (bond.m[dst * 4 + src] != 0.0 || bond.m[src * 4 + dst] != 0.0) used[src ^ dst] = true;
      }
      worst = worst > here ? worst : here;
    }
    max_conn += worst;
    for (uint32_t x = 1; x < 4; ++x) {
      if (used[x]) masks.push_back(bond.flip[x]);
    }
  }
  std::sort(masks.begin(), masks.end());
  op->unique_targets = std::adjacent_find(masks.begin(), masks.end()) == masks.end();
  op->max_connections = static_cast<uint32_t>(max_conn);
  int rc = op->d_bonds.alloc(num_bonds);
  if (rc == ASP_OK) rc = op->d_bonds.upload(op->bonds.data(), num_bonds, nullptr);
  if (rc == ASP_OK && hipStreamSynchronize(nullptr) != hipSuccess) {
    rc = asp::set_error(ASP_ERR_HIP, "upload of the bond table failed");
  }
  if (rc != ASP_OK) {
    delete op;
    return rc;
  }
  *out = op;
  return ASP_OK;
}

void asp_operator_destroy(asp_operator *op) {
  if (!op) return;
  (void)asp::bind_device();
  delete op;
}

int asp_operator_set_symmetry(asp_operator *op, uint32_t num_permutations, uint8_t const *table,
                              int32_t spin_inversion) {
  asp_clear_error();
  ASP_TRY(check_operator(op));
  if (spin_inversion < -1 || spin_inversion > 1) {
    return asp::set_error(ASP_ERR_INVALID, "spin_inversion must be -1, 0 or 1");
  }
  if (num_permutations == 0 || !table) {
    return asp::set_error(ASP_ERR_INVALID, "a group holds at least the identity");
  }
  for (uint32_t e = 0; e < num_permutations; ++e) {
    uint64_t seen = 0;
    for (uint32_t i = 0; i < op->number_spins; ++i) {
      const uint32_t d = table[static_cast<size_t>(e) * 64 + i];
      if (d >= op->number_spins || ((seen >> d) & 1ull)) {
        return asp::set_error(ASP_ERR_INVALID, "element %u is not a permutation of the sites", e);
      }
      seen |= 1ull << d;
    }
    if (e == 0) {
      for (uint32_t i = 0; i < op->number_spins; ++i) {
        if (table[i] != i) return asp::set_error(ASP_ERR_INVALID, "element 0 must be the identity");
      }
    }
  }
  ASP_TRY(op->d_table.alloc(static_cast<size_t>(num_permutations) * 64));
  ASP_TRY(op->d_table.upload(table, static_cast<size_t>(num_permutations) * 64, nullptr));
  ASP_HIP_TRY(hipStreamSynchronize(nullptr));
  op->num_permutations = num_permutations;
  op->inversion = spin_inversion;
  // equal targets are no longer excluded: two connections may share a representative
  op->unique_targets = false;
  return ASP_OK;
}

int asp_operator_state_info(asp_operator const *op, uint64_t n, uint64_t const *keys,
                            uint64_t *representatives, double *characters, double *norms) {
  asp_clear_error();
  ASP_TRY(check_operator(op));
  if (n == 0) return ASP_OK;
  if (!keys) return asp::set_error(ASP_ERR_INVALID, "null keys");
  if (op->num_permutations == 0) {
    return asp::set_error(ASP_ERR_INVALID, "the operator's basis has no symmetries");
  }
  asp::ScopedStream scoped;
  ASP_TRY(scoped.acquire());
  hipStream_t stream = scoped.stream;
  // one "row" per key with a single entry (the key itself, coefficient 1, source norm 1):
  // k_symmetrise then leaves (representative, character * norm) behind
  DeviceBuffer<uint64_t> d_keys;
  DeviceBuffer<double> d_coeffs, d_ones;
  DeviceBuffer<int64_t> d_offsets;
  asp::StreamFence fence(stream);
  ASP_TRY(d_keys.alloc(n));
  ASP_TRY(d_coeffs.alloc(n));
  ASP_TRY(d_ones.alloc(n));
  ASP_TRY(d_offsets.alloc(n + 1));
  std::vector<double> ones(n, 1.0);
  std::vector<int64_t> iota(n + 1);
  for (uint64_t i = 0; i <= n; ++i) iota[i] = static_cast<int64_t>(i);
  ASP_TRY(d_keys.upload(keys, n, stream));
  ASP_TRY(d_coeffs.upload(ones.data(), n, stream));
  ASP_TRY(d_ones.upload(ones.data(), n, stream));
  ASP_TRY(d_offsets.upload(iota.data(), n + 1, stream));
  const SymmetryArgs g = op->symmetry();
  if (norms) {
    // norms alone come from k_source_norms; characters need the second pass
    DeviceBuffer<double> d_norms;
    ASP_TRY(d_norms.alloc(n));
    hipLaunchKernelGGL(k_source_norms, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, stream, g,
                       d_keys.ptr, n, d_norms.ptr);
    ASP_HIP_TRY(hipGetLastError());
    ASP_TRY(d_norms.download(norms, n, stream));
    ASP_HIP_TRY(hipStreamSynchronize(stream));
  }
  hipLaunchKernelGGL(k_symmetrise, dim3(grid_for(n, kThreads)), dim3(kThreads), 0, stream, g,
                     d_offsets.ptr, n, d_ones.ptr, n, d_keys.ptr, d_coeffs.ptr);
  ASP_HIP_TRY(hipGetLastError());
  if (representatives) ASP_TRY(d_keys.download(representatives, n, stream));
  std::vector<double> signed_norm(characters ? n : 0);
  if (characters) ASP_TRY(d_coeffs.download(signed_norm.data(), n, stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  if (characters) {
    for (uint64_t i = 0; i < n; ++i) characters[i] = signed_norm[i] < 0.0 ? -1.0 : 1.0;
  }
  return ASP_OK;
}

int asp_operator_unique_targets(asp_operator const *op) { return op && op->unique_targets ? 1 : 0; }

uint32_t asp_operator_max_connections(asp_operator const *op) {
  return op ? op->max_connections : 0;
}

int asp_operator_apply(asp_operator const *op, uint64_t n, uint64_t const *keys, uint64_t capacity,
                       uint64_t *other_keys, double *other_coeffs, int64_t *other_counts,
                       uint64_t *total) {
  asp_clear_error();
  ASP_TRY(check_operator(op));
  if (n && !keys) return asp::set_error(ASP_ERR_INVALID, "null keys");
  asp::ScopedStream scoped;
  ASP_TRY(scoped.acquire());
  hipStream_t stream = scoped.stream;
  Timer timer;
  ApplyBatch w;
  asp::StreamFence fence_w(stream);  // (and the later buffers have a fence of their own)
  ASP_TRY(timer.start(stream));
  ASP_TRY(count_connections(op, n, keys, &w, stream));
  if (total) *total = w.total;
  if (other_counts) ASP_TRY(w.d_counts.download(other_counts, n, stream));
  if (w.total > capacity || !other_keys || !other_coeffs) {
    ASP_HIP_TRY(hipStreamSynchronize(stream));
    if (capacity == 0 && !other_keys && !other_coeffs) return ASP_OK;  // sizing call
    return asp::set_error(ASP_ERR_INVALID, "%llu connections do not fit capacity %llu",
                          (unsigned long long)w.total, (unsigned long long)capacity);
  }
  DeviceBuffer<uint64_t> d_other;
  DeviceBuffer<double> d_coeffs, d_norms;
  asp::StreamFence fence(stream);  // error exits wait for the stream before the buffers go
  ASP_TRY(d_other.alloc(w.total));
  ASP_TRY(d_coeffs.alloc(w.total));
  if (n > 0) {
    hipLaunchKernelGGL(k_apply<true>, dim3(grid_for(n, kWaves)), dim3(kThreads), 0, stream,
                       op->d_bonds.ptr, op->num_bonds, w.d_keys.ptr, n, w.d_offsets.ptr, nullptr,
                       d_other.ptr, d_coeffs.ptr);
    ASP_HIP_TRY(hipGetLastError());
    ASP_TRY(symmetrise_batch(op, n, w, d_other.ptr, d_coeffs.ptr, &d_norms, stream));
  }
  ASP_TRY(timer.stop());
  ASP_TRY(d_other.download(other_keys, w.total, stream));
  ASP_TRY(d_coeffs.download(other_coeffs, w.total, stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  timer.finish();
  return ASP_OK;
}

namespace {

int operator_ising(asp_operator const *op, uint64_t num_spins, uint64_t const *keys,
                   double const *psi, uint64_t capacity, int32_t *row, int64_t *indptr, int32_t *col,
                   double *val, uint64_t *nnz) {
  asp_clear_error();
  ASP_TRY(check_operator(op));
  if (!nnz) return asp::set_error(ASP_ERR_INVALID, "null nnz pointer");
  *nnz = 0;
  const uint64_t K = num_spins;
  if (K && (!keys || !psi)) return asp::set_error(ASP_ERR_INVALID, "null input array");
  if (K >= 0x7FFFFFFFull) return asp::set_error(ASP_ERR_TOO_LARGE, "more than 2^31-1 spins");
  for (uint64_t i = 1; i < K; ++i) {
    if (keys[i - 1] >= keys[i]) {
      return asp::set_error(ASP_ERR_INVALID, "keys are not sorted and unique");
    }
  }
  if (K == 0) return ASP_OK;
  asp::ScopedStream scoped;
  ASP_TRY(scoped.acquire());
  hipStream_t stream = scoped.stream;
  if (!op->unique_targets) {
    // rows may reach a state twice (symmetry-adapted bases, single-site flips): the variant that
    // keeps the reference's duplicate arithmetic
    return ising_with_duplicates(op, K, keys, psi, capacity, row, indptr, col, val, nnz, stream);
  }
  uint64_t slots_n = 1024;
  while (slots_n < 2 * K) slots_n <<= 1;
  DeviceBuffer<uint64_t> d_keys;
  DeviceBuffer<double> d_psi, d_val;
  DeviceBuffer<unsigned long long> d_slots;
  DeviceBuffer<uint32_t> d_row_nnz;
  DeviceBuffer<int64_t> d_row_start, d_scratch;
  DeviceBuffer<int32_t> d_row, d_col;
  asp::StreamFence fence(stream);  // error exits wait for the stream before the buffers go
  ASP_TRY(d_keys.alloc(K));
  ASP_TRY(d_psi.alloc(K));
  ASP_TRY(d_slots.alloc(slots_n));
  ASP_TRY(d_row_nnz.alloc(K));
  ASP_TRY(d_row_start.alloc(K + 1));
  ASP_TRY(d_scratch.alloc(asp::scan_scratch_elems(K)));
  ASP_TRY(d_keys.upload(keys, K, stream));
  ASP_TRY(d_psi.upload(psi, K, stream));
  Timer timer;
  ASP_TRY(timer.start(stream));
  ASP_HIP_TRY(hipMemsetAsync(d_slots.ptr, 0, slots_n * sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(k_key_insert, dim3(grid_for(K, kThreads)), dim3(kThreads), 0, stream,
                     d_keys.ptr, K, d_slots.ptr, slots_n - 1);
  IsingArgs a{};
  a.bonds = op->d_bonds.ptr;
  a.keys = d_keys.ptr;
  a.psi = d_psi.ptr;
  a.slots = d_slots.ptr;
  a.mask = slots_n - 1;
  a.row_nnz = d_row_nnz.ptr;
  a.num_spins = K;
  a.num_bonds = op->num_bonds;
  a.row_capacity = 0;
  hipLaunchKernelGGL(k_ising_rows<false>, dim3(grid_for(K, kWaves)), dim3(kThreads), 0, stream, a);
  ASP_HIP_TRY(hipGetLastError());
  ASP_TRY(asp::exclusive_scan_u32(d_row_nnz.ptr, K, d_row_start.ptr, d_scratch.ptr, stream));
  int64_t total = 0;
  ASP_HIP_TRY(hipMemcpyAsync(&total, d_row_start.ptr + K, sizeof total, hipMemcpyDeviceToHost,
                             stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  *nnz = static_cast<uint64_t>(total);
  if (capacity == 0 && !row && !indptr && !col && !val) {  // sizing call
    ASP_TRY(timer.stop());
    ASP_HIP_TRY(hipStreamSynchronize(stream));
    timer.finish();
    return ASP_OK;
  }
  if (*nnz > capacity || (!row && !indptr) || !col || !val) {
    return asp::set_error(ASP_ERR_INVALID, "%llu couplings do not fit capacity %llu",
                          (unsigned long long)*nnz, (unsigned long long)capacity);
  }
  ASP_TRY(d_row.alloc(*nnz));
  ASP_TRY(d_col.alloc(*nnz));
  ASP_TRY(d_val.alloc(*nnz));
  a.row_start = d_row_start.ptr;
  a.row = d_row.ptr;
  a.col = d_col.ptr;
  a.val = d_val.ptr;
  a.row_capacity = (op->max_connections + 1u) & ~1u;  // even: keeps the i32 area 8-byte aligned
  // the pair (rev != 0, fwd == 0) can add entries beyond max_connections: 3 per bond is the cap
  const uint32_t cap_all = 3u * op->num_bonds + 2u;
  if (a.row_capacity < cap_all) a.row_capacity = (cap_all + 1u) & ~1u;
  const size_t lds = static_cast<size_t>(kWaves) * a.row_capacity * (sizeof(double) + sizeof(int32_t));
  if (lds > 160u * 1024u) {
    return asp::set_error(ASP_ERR_TOO_LARGE, "%u bonds need %zu bytes of LDS per workgroup",
                          op->num_bonds, lds);
  }
  if (lds > 64u * 1024u) {
    ASP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ising_rows<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
  }
  hipLaunchKernelGGL(k_ising_rows<true>, dim3(grid_for(K, kWaves)), dim3(kThreads), lds, stream, a);
  ASP_HIP_TRY(hipGetLastError());
  ASP_TRY(timer.stop());
  if (row) ASP_TRY(d_row.download(row, *nnz, stream));
  if (indptr) ASP_TRY(d_row_start.download(indptr, K + 1, stream));
  ASP_TRY(d_col.download(col, *nnz, stream));
  ASP_TRY(d_val.download(val, *nnz, stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  timer.finish();
  return ASP_OK;
}

}  // namespace

int asp_operator_ising(asp_operator const *op, uint64_t num_spins, uint64_t const *keys,
                       double const *psi, uint64_t capacity, int32_t *row, int32_t *col,
                       double *val, uint64_t *nnz) {
  return operator_ising(op, num_spins, keys, psi, capacity, row, nullptr, col, val, nnz);
}

int asp_operator_ising_csr(asp_operator const *op, uint64_t num_spins, uint64_t const *keys,
                           double const *psi, uint64_t capacity, int64_t *indptr, int32_t *col,
                           double *val, uint64_t *nnz) {
  return operator_ising(op, num_spins, keys, psi, capacity, nullptr, indptr, col, val, nnz);
}

int asp_operator_extend(asp_operator const *op, uint64_t n, uint64_t const *keys,
                        uint64_t capacity, uint64_t *out, uint64_t *count) {
  asp_clear_error();
  ASP_TRY(check_operator(op));
  if (!count) return asp::set_error(ASP_ERR_INVALID, "null count pointer");
  *count = 0;
  if (n && !keys) return asp::set_error(ASP_ERR_INVALID, "null keys");
  if (n == 0) return ASP_OK;
  asp::ScopedStream scoped;
  ASP_TRY(scoped.acquire());
  hipStream_t stream = scoped.stream;
  Timer timer;
  ApplyBatch w;
  asp::StreamFence fence_w(stream);  // (and the later buffers have a fence of their own)
  ASP_TRY(timer.start(stream));
  ASP_TRY(count_connections(op, n, keys, &w, stream));
  const uint64_t N = w.total;
  if (N >= (1ull << 32)) return asp::set_error(ASP_ERR_TOO_LARGE, "more than 2^32 connections");
  DeviceBuffer<uint64_t> d_targets, d_sorted, d_unique;
  DeviceBuffer<double> d_coeffs, d_norms;  // written by k_apply<true>, not used here
  DeviceBuffer<uint32_t> d_flag;
  DeviceBuffer<int64_t> d_pos, d_scratch;
  DeviceBuffer<uint8_t> d_temp;
  asp::StreamFence fence(stream);  // error exits wait for the stream before the buffers go
  ASP_TRY(d_targets.alloc(N));
  ASP_TRY(d_sorted.alloc(N));
  ASP_TRY(d_coeffs.alloc(N));
  ASP_TRY(d_flag.alloc(N));
  ASP_TRY(d_pos.alloc(N + 1));
  ASP_TRY(d_scratch.alloc(asp::scan_scratch_elems(N)));
  hipLaunchKernelGGL(k_apply<true>, dim3(grid_for(n, kWaves)), dim3(kThreads), 0, stream,
                     op->d_bonds.ptr, op->num_bonds, w.d_keys.ptr, n, w.d_offsets.ptr, nullptr,
                     d_targets.ptr, d_coeffs.ptr);
  ASP_HIP_TRY(hipGetLastError());
  // symmetry-adapted basis: the extension is the set of the targets' REPRESENTATIVES
  ASP_TRY(symmetrise_batch(op, n, w, d_targets.ptr, d_coeffs.ptr, &d_norms, stream));
  size_t temp_bytes = 0;
  ASP_HIP_TRY(rocprim::radix_sort_keys(nullptr, temp_bytes, d_targets.ptr, d_sorted.ptr, N, 0,
                                       op->number_spins, stream));
  ASP_TRY(d_temp.alloc(temp_bytes ? temp_bytes : 1));
  ASP_HIP_TRY(rocprim::radix_sort_keys(d_temp.ptr, temp_bytes, d_targets.ptr, d_sorted.ptr, N, 0,
                                       op->number_spins, stream));
  hipLaunchKernelGGL(k_flag_first, dim3(grid_for(N, kThreads)), dim3(kThreads), 0, stream,
                     d_sorted.ptr, N, d_flag.ptr);
  ASP_TRY(asp::exclusive_scan_u32(d_flag.ptr, N, d_pos.ptr, d_scratch.ptr, stream));
  int64_t unique = 0;
  ASP_HIP_TRY(hipMemcpyAsync(&unique, d_pos.ptr + N, sizeof unique, hipMemcpyDeviceToHost, stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  *count = static_cast<uint64_t>(unique);
  if (capacity == 0 && !out) {  // sizing call
    ASP_TRY(timer.stop());
    ASP_HIP_TRY(hipStreamSynchronize(stream));
    timer.finish();
    return ASP_OK;
  }
  if (*count > capacity || !out) {
    return asp::set_error(ASP_ERR_INVALID, "%llu states do not fit capacity %llu",
                          (unsigned long long)*count, (unsigned long long)capacity);
  }
  ASP_TRY(d_unique.alloc(*count));
  hipLaunchKernelGGL(k_scatter_first, dim3(grid_for(N, kThreads)), dim3(kThreads), 0, stream,
                     d_sorted.ptr, N, d_flag.ptr, d_pos.ptr, d_unique.ptr);
  ASP_HIP_TRY(hipGetLastError());
  ASP_TRY(timer.stop());
  ASP_TRY(d_unique.download(out, *count, stream));
  ASP_HIP_TRY(hipStreamSynchronize(stream));
  timer.finish();
  return ASP_OK;
}

}  // extern "C"
