// Coupling build on gfx950: HIP replacement for the reference's
// cbits/build_matrix.c (build_matrix :22-65, ls_bits512_cmp :7-20,
// extract_signs :67-76) behind the same C symbols.
//
// Data in HBM (all SoA except the 512-bit keys, which keep the reference's AoS
// layout because that is what the ABI hands over):
//   table   ls_bits512[K]   sorted, unique           table0  u64[K] = words[0]
//   slots   u64[2^s]        open-addressing hash of table0: {fingerprint:32 | index+1:32}
//   needles ls_bits512[N]   flat, row-major by row
//   counts i64[K], psi f64[K], coeffs f64[N], other_psi f64[N], other_counts i64[K]
//   offsets i64[K+1] (prefix sums of other_counts, formed on the host while the upload checks
//   them), found i32[N] (table index or -1), row_hits u32[K],
//   block_total u32[row blocks], tile_total u64[2][row blocks / 64] (hits of 64 row blocks)
//   out: row u32[N], col u32[N], elements f64[N], field f64[K]
//
// One build = THREE launches on one stream, no host round trip except nnz (round 2: twelve —
// two memsets, two three-kernel scans, four kernels):
//   k_insert_keys   table -> table0 (words[0]), a flag "some table key has a non-zero word
//                   1..7", and table0 -> slots (load factor <= 1/2, linear probing, atomicCAS)
//   k_search_rows   32 lanes per row: each lane resolves one connection through the hash
//                   (the reference bsearches, cbits/build_matrix.c:37-38; on a unique table
//                   any exact lookup returns the same element), row hit count by ballot; the
//                   block's total goes to block_total and, atomically, to its tile's total
//   k_emit_rows     first the block's output position WITHOUT a scan kernel: the totals of the
//                   tiles before its own (<= K / 512 numbers) plus the totals of the blocks
//                   before it inside its tile (<= 63) — integer sums, so the order does not
//                   matter; then 32 lanes per row: COO triples in input order at ballot-prefix
//                   positions (coalesced), and the row's field as a LEFT-TO-RIGHT sum over its
//                   misses (the reference's rounding, cbits/build_matrix.c:49); last, every block
//                   zeroes its share of the hash slots and block 0 the flag and the OTHER parity's
//                   tile totals: the next run finds them clean (the slots are zeroed at creation)
//
// Arithmetic: __dmul_rn / __dadd_rn keep every product and the field add
// separately rounded (no FMA contraction), matching the reference binary.
// Memory-bound integer/f64 streaming: no MFMA, no LDS tiling of the payload.
#include <vector>

#include "asp_common.hpp"

namespace {

using asp::DeviceBuffer;

constexpr int kThreads = 256;
constexpr int kRowLanes = 32;                         // lanes cooperating on one row
constexpr int kRowsPerBlock = kThreads / kRowLanes;   // 8

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

__global__ __launch_bounds__(kThreads) void k_insert_keys(const ls_bits512 *__restrict__ table,
                                                         uint64_t n,
                                                         uint64_t *__restrict__ table0,
                                                         uint32_t *__restrict__ tail_flag,
                                                         unsigned long long *__restrict__ slots,
                                                         uint64_t mask) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  const ls_bits512 key = table[i];
  table0[i] = key.words[0];
  uint64_t tail = 0;
#pragma unroll
  for (int w = 1; w < 8; ++w) tail |= key.words[w];
  if (tail != 0) atomicOr(tail_flag, 1u);
  const uint64_t h = mix64(key.words[0]);
  const unsigned long long entry = (h & 0xFFFFFFFF00000000ull) | (i + 1);
  uint64_t at = h & mask;
  while (atomicCAS(&slots[at], 0ull, entry) != 0ull) at = (at + 1) & mask;
}

// Index of the needle (its eight words in registers) in the table, or -1.
__device__ __forceinline__ int32_t find_key(const unsigned long long *__restrict__ slots,
                                            uint64_t mask, const uint64_t *__restrict__ table0,
                                            const ls_bits512 *__restrict__ table,
                                            bool table_has_tails, const uint64_t (&needle)[8]) {
  const uint64_t n0 = needle[0];
  const uint64_t h = mix64(n0);
  const uint32_t fingerprint = static_cast<uint32_t>(h >> 32);
  uint64_t needle_tail = 0;
#pragma unroll
  for (int w = 1; w < 8; ++w) needle_tail |= needle[w];
  for (uint64_t at = h & mask;; at = (at + 1) & mask) {
    const unsigned long long slot = slots[at];
    if (slot == 0) return -1;
    if (static_cast<uint32_t>(slot >> 32) != fingerprint) continue;
    const uint32_t idx = static_cast<uint32_t>(slot) - 1u;
    if (table0[idx] != n0) continue;
    // word 0 matches: the full 512-bit keys must agree (cbits/build_matrix.c:11-18)
    bool same;
    if (table_has_tails) {
      uint64_t diff = 0;
#pragma unroll
      for (int w = 1; w < 8; ++w) diff |= table[idx].words[w] ^ needle[w];
      same = diff == 0;
    } else {
      same = needle_tail == 0;
    }
    if (same) return static_cast<int32_t>(idx);
  }
}

// Ballot restricted to this lane's 32-lane half.
__device__ __forceinline__ uint32_t half_ballot(bool predicate, uint32_t lane) {
  const uint64_t all = __ballot(predicate);
  return static_cast<uint32_t>(all >> (lane & 32u));
}

__global__ __launch_bounds__(kThreads) void k_search_rows(
    const unsigned long long *__restrict__ slots, uint64_t mask,
    const uint64_t *__restrict__ table0, const ls_bits512 *__restrict__ table,
    const uint32_t *__restrict__ tail_flag, uint64_t num_spins,
    const ls_bits512 *__restrict__ needles, const int64_t *__restrict__ offsets,
    int32_t *__restrict__ found, uint32_t *__restrict__ row_hits,
    uint32_t *__restrict__ block_total, unsigned long long *__restrict__ tile_total) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t sub = threadIdx.x & (kRowLanes - 1);
  const uint64_t r = static_cast<uint64_t>(blockIdx.x) * kRowsPerBlock + threadIdx.x / kRowLanes;
  const bool row_ok = r < num_spins;
  const int64_t begin = row_ok ? offsets[r] : 0;
  const int64_t end = row_ok ? offsets[r + 1] : 0;
  const bool has_tails = *tail_flag != 0;
  uint32_t hits = 0;
  // The 32 needles a half-wavefront resolves per trip are 2 KiB of CONTIGUOUS memory (AoS
  // keys of one row).  Each lane fetches four 16-byte chunks of that run (fully coalesced,
  // 512 contiguous bytes per half and instruction) into LDS and then reads its own key
  // back transposed, instead of eight loads per lane at a 64-byte stride.
  __shared__ uint4 stage[kThreads * 4];
  uint4 *mine = stage + (threadIdx.x / kRowLanes) * (kRowLanes * 4);
  // both halves of the wavefront iterate until the longer row is done, so the ballot is
  // always executed by all 64 lanes
  const int64_t other_len = __shfl_xor(end - begin, 32, 64);
  const int64_t trips = ((end - begin > other_len ? end - begin : other_len) + kRowLanes - 1) / kRowLanes;
  for (int64_t it = 0; it < trips; ++it) {
    const int64_t run = begin + it * kRowLanes;
    const int64_t valid = end - run;  // needles of this trip that exist (may be <= 0)
    const uint4 *src = reinterpret_cast<const uint4 *>(needles + run);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t chunk = i * kRowLanes + sub;  // 4 chunks per key
      if (static_cast<int64_t>(chunk >> 2) < valid) mine[chunk] = src[chunk];
    }
    __builtin_amdgcn_wave_barrier();  // same wavefront: LDS ops are in order, keep them so
    const int64_t e = run + sub;
    const bool active = e < end;
    int32_t idx = -1;
    if (active) {
      uint64_t key[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint4 q = mine[sub * 4 + j];
        key[2 * j] = (static_cast<uint64_t>(q.y) << 32) | q.x;
        key[2 * j + 1] = (static_cast<uint64_t>(q.w) << 32) | q.z;
      }
      idx = find_key(slots, mask, table0, table, has_tails, key);
      found[e] = idx;
    }
    hits += __popc(half_ballot(idx >= 0, lane));
    __builtin_amdgcn_wave_barrier();  // the next trip overwrites the staging area
  }
  if (row_ok && sub == 0) row_hits[r] = hits;
  // hits of the block's rows: to the block's total and to the total of its tile of 64 blocks
  __shared__ uint32_t hits_of_row[kRowsPerBlock];
  if (sub == 0) hits_of_row[threadIdx.x / kRowLanes] = row_ok ? hits : 0u;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < kRowsPerBlock; ++k) total += hits_of_row[k];
    block_total[blockIdx.x] = total;
    if (total) atomicAdd(&tile_total[blockIdx.x >> 6], static_cast<unsigned long long>(total));
  }
}

struct EmitTotals {
  const uint32_t *row_hits;                // [K]
  const uint32_t *block_total;             // [row blocks]
  const unsigned long long *tile_total;    // [tiles] of this run
  unsigned long long *nnz;                 // the build's number of couplings
  // left clean for the next run:
  unsigned long long *slots;
  uint64_t slot_count;
  uint32_t *tail_flag;
  unsigned long long *next_tile_total;     // the other parity's tiles
  uint32_t num_tiles;
};

__global__ __launch_bounds__(kThreads) void k_emit_rows(
    const int64_t *__restrict__ offsets, EmitTotals totals,
    const int32_t *__restrict__ found, const int64_t *__restrict__ counts,
    const double *__restrict__ psi, const double *__restrict__ coeffs,
    const double *__restrict__ other_psi, uint64_t num_spins, uint32_t *__restrict__ out_row,
    uint32_t *__restrict__ out_col, double *__restrict__ out_elements,
    double *__restrict__ out_field) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t sub = threadIdx.x & (kRowLanes - 1);
  const uint32_t half_base = lane & 32u;
  const uint32_t row_in_block = threadIdx.x / kRowLanes;
  const uint64_t r = static_cast<uint64_t>(blockIdx.x) * kRowsPerBlock + row_in_block;
  const bool row_ok = r < num_spins;
  // ---- where this block's couplings start: tiles before its tile + blocks before it in the tile
  __shared__ unsigned long long block_base;
  __shared__ uint32_t hits_of_row[kRowsPerBlock];
  if (threadIdx.x == 0) block_base = 0ull;
  if (sub == 0) hits_of_row[row_in_block] = row_ok ? totals.row_hits[r] : 0u;
  __syncthreads();
  {
    const uint32_t tile = blockIdx.x >> 6, first_of_tile = blockIdx.x & ~63u;
    unsigned long long mine = 0;
    for (uint32_t t = threadIdx.x; t < tile; t += kThreads) mine += totals.tile_total[t];
    if (threadIdx.x < blockIdx.x - first_of_tile) mine += totals.block_total[first_of_tile + threadIdx.x];
    for (int step = 1; step < 64; step <<= 1) mine += __shfl_xor(mine, step, 64);
    if (lane == 0 && mine) atomicAdd(&block_base, mine);
  }
  __syncthreads();
  int64_t w = static_cast<int64_t>(block_base);
  for (uint32_t k = 0; k < row_in_block; ++k) w += hits_of_row[k];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    unsigned long long all = block_base;
    for (int k = 0; k < kRowsPerBlock; ++k) all += hits_of_row[k];
    *totals.nnz = all;
  }
  const int64_t begin = row_ok ? offsets[r] : 0;
  const int64_t end = row_ok ? offsets[r + 1] : 0;
  const double c = row_ok ? static_cast<double>(counts[r]) : 0.0;  // exact i64 -> f64 as in C
  const double a = row_ok ? fabs(psi[r]) : 0.0;
  double f = 0.0;
  const int64_t other_len = __shfl_xor(end - begin, 32, 64);
  const int64_t trips = ((end - begin > other_len ? end - begin : other_len) + kRowLanes - 1) / kRowLanes;
  for (int64_t it = 0; it < trips; ++it) {
    const int64_t e = begin + it * kRowLanes + sub;
    const bool active = e < end;
    const int32_t pos = active ? found[e] : -1;
    // ((counts * coeff) * |psi|) * x, each product rounded: build_matrix.c:41-42,49
    const double head = active ? __dmul_rn(__dmul_rn(c, coeffs[e]), a) : 0.0;
    const double x = active ? other_psi[e] : 0.0;
    const bool hit = active && pos >= 0;
    const uint32_t hit_mask = half_ballot(hit, lane);
    if (hit) {
      const int64_t at = w + __popc(hit_mask & ((1u << sub) - 1u));
      out_row[at] = static_cast<uint32_t>(r);
      out_col[at] = static_cast<uint32_t>(pos);
      out_elements[at] = __dmul_rn(head, fabs(x));
    }
    w += __popc(hit_mask);
    // misses: every lane of the half replays the row's sequential sum
    const double t = __dmul_rn(head, x);
    uint32_t miss_mask = half_ballot(active && pos < 0, lane);
    const uint32_t other_mask = __shfl_xor(miss_mask, 32, 64);
    uint32_t rounds = __popc(miss_mask) > __popc(other_mask) ? __popc(miss_mask) : __popc(other_mask);
    for (; rounds > 0; --rounds) {  // uniform trip count: the shuffle runs with all lanes on
      const bool take = miss_mask != 0;
      const uint32_t j = take ? static_cast<uint32_t>(__ffs(miss_mask) - 1) : 0u;
      const double tj = __shfl(t, static_cast<int>(half_base + j), 64);
      if (take) {
        f = __dadd_rn(f, tj);
        miss_mask &= miss_mask - 1u;
      }
    }
  }
  if (row_ok && sub == 0) out_field[r] = f;
  // ---- leave the tables clean for the next run (nobody reads the slots or the flag any more) ----
  {
    const uint64_t share = (totals.slot_count + gridDim.x - 1) / gridDim.x;
    const uint64_t first = static_cast<uint64_t>(blockIdx.x) * share;
    const uint64_t last = first + share < totals.slot_count ? first + share : totals.slot_count;
    for (uint64_t k = first + threadIdx.x; k < last; k += kThreads) totals.slots[k] = 0ull;
    if (blockIdx.x == 0) {
      if (threadIdx.x == 0) *totals.tail_flag = 0u;
      for (uint32_t t = threadIdx.x; t < totals.num_tiles; t += kThreads) totals.next_tile_total[t] = 0ull;
    }
  }
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
inline unsigned row_blocks_for(uint64_t rows) {
  return static_cast<unsigned>((rows + kRowsPerBlock - 1) / kRowsPerBlock);
}

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
  DeviceBuffer<unsigned long long> slots;
  DeviceBuffer<uint32_t> tail_flag;
  uint64_t slot_mask = 0;
  DeviceBuffer<int64_t> counts, offsets;
  DeviceBuffer<double> psi, coeffs, other_psi, elements, field;
  DeviceBuffer<int32_t> found;
  DeviceBuffer<uint32_t> row_hits, block_total, out_row, out_col;
  DeviceBuffer<unsigned long long> tile_total, nnz;  // tile_total: [2][num_tiles], one parity per run
  uint32_t num_tiles = 0;
  uint32_t parity = 0;
  bool clean = false;  // slots, flag and this parity's tile totals are zero (k_emit_rows leaves them so)
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
  bool ok = asp::stream_acquire(&b->stream) == ASP_OK &&
            hipEventCreate(&b->ev_start) == hipSuccess && hipEventCreate(&b->ev_stop) == hipSuccess;
  if (!ok) asp::set_error(ASP_ERR_HIP, "could not create HIP stream/events");
  uint64_t slot_count = 64;
  while (slot_count < 2 * K) slot_count <<= 1;  // load factor <= 1/2
  b->slot_mask = slot_count - 1;
  const uint64_t row_blocks = row_blocks_for(K);
  b->num_tiles = static_cast<uint32_t>((row_blocks + 63) / 64);
  ok = ok && b->table.alloc(K) == ASP_OK && b->needles.alloc(N) == ASP_OK &&
       b->table0.alloc(K) == ASP_OK && b->slots.alloc(b->slot_mask + 1) == ASP_OK &&
       b->tail_flag.alloc(1) == ASP_OK && b->counts.alloc(K) == ASP_OK &&
       b->offsets.alloc(K + 1) == ASP_OK &&
       b->psi.alloc(K) == ASP_OK && b->coeffs.alloc(N) == ASP_OK &&
       b->other_psi.alloc(N) == ASP_OK && b->elements.alloc(N) == ASP_OK &&
       b->field.alloc(K) == ASP_OK && b->found.alloc(N) == ASP_OK &&
       b->row_hits.alloc(K) == ASP_OK && b->block_total.alloc(row_blocks) == ASP_OK &&
       b->tile_total.alloc(2ull * b->num_tiles) == ASP_OK && b->nnz.alloc(1) == ASP_OK &&
       b->out_row.alloc(N) == ASP_OK && b->out_col.alloc(N) == ASP_OK;
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
  if (b->stream) {
    (void)hipStreamSynchronize(b->stream);
    asp::stream_release(b->stream);
  }
  delete b;
}

int asp_build_upload(asp_build *b, ls_bits512 const *spins, int64_t const *counts,
                     double const *psi, ls_bits512 const *other_spins,
                     double const *other_coeffs, int64_t const *other_counts,
                     double const *other_psi) {
  if (!b) return asp::set_error(ASP_ERR_INVALID, "null build handle");
  ASP_TRY(asp::bind_device());
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
  // offsets of the rows in the flat arrays: the prefix sums of the counts just walked
  std::vector<int64_t> offsets(K + 1);
  offsets[0] = 0;
  for (uint64_t r = 0; r < K; ++r) offsets[r + 1] = offsets[r] + other_counts[r];
  ASP_TRY(b->table.upload(spins, K, b->stream));
  ASP_TRY(b->counts.upload(counts, K, b->stream));
  ASP_TRY(b->psi.upload(psi, K, b->stream));
  ASP_TRY(b->needles.upload(other_spins, N, b->stream));
  ASP_TRY(b->coeffs.upload(other_coeffs, N, b->stream));
  ASP_TRY(b->offsets.upload(offsets.data(), K + 1, b->stream));
  ASP_TRY(b->other_psi.upload(other_psi, N, b->stream));
  ASP_HIP_TRY(hipStreamSynchronize(b->stream));
  b->uploaded = true;
  return ASP_OK;
}

int asp_build_run(asp_build *b, uint64_t *nnz) {
  if (!b) return asp::set_error(ASP_ERR_INVALID, "null build handle");
  if (!b->uploaded) return asp::set_error(ASP_ERR_INVALID, "asp_build_run before asp_build_upload");
  ASP_TRY(asp::bind_device());
  const uint64_t K = b->num_spins;
  hipStream_t s = b->stream;
  if (!b->clean) {
    // first run of the handle, or the one before it failed half-way: k_emit_rows leaves these zero
    ASP_HIP_TRY(hipMemsetAsync(b->slots.ptr, 0, (b->slot_mask + 1) * sizeof(unsigned long long), s));
    ASP_HIP_TRY(hipMemsetAsync(b->tail_flag.ptr, 0, sizeof(uint32_t), s));
    ASP_HIP_TRY(hipMemsetAsync(b->tile_total.ptr, 0, 2ull * b->num_tiles * sizeof(unsigned long long), s));
  }
  b->clean = false;
  ASP_HIP_TRY(hipEventRecord(b->ev_start, s));
  if (K > 0) {
    unsigned long long *tiles = b->tile_total.ptr + static_cast<size_t>(b->parity) * b->num_tiles;
    unsigned long long *other_tiles = b->tile_total.ptr + static_cast<size_t>(b->parity ^ 1u) * b->num_tiles;
    hipLaunchKernelGGL(k_insert_keys, dim3(blocks_for(K)), dim3(kThreads), 0, s, b->table.ptr, K,
                       b->table0.ptr, b->tail_flag.ptr, b->slots.ptr, b->slot_mask);
    hipLaunchKernelGGL(k_search_rows, dim3(row_blocks_for(K)), dim3(kThreads), 0, s, b->slots.ptr,
                       b->slot_mask, b->table0.ptr, b->table.ptr, b->tail_flag.ptr, K,
                       b->needles.ptr, b->offsets.ptr, b->found.ptr, b->row_hits.ptr,
                       b->block_total.ptr, tiles);
    EmitTotals totals{b->row_hits.ptr, b->block_total.ptr, tiles, b->nnz.ptr, b->slots.ptr,
                      b->slot_mask + 1, b->tail_flag.ptr, other_tiles, b->num_tiles};
    hipLaunchKernelGGL(k_emit_rows, dim3(row_blocks_for(K)), dim3(kThreads), 0, s, b->offsets.ptr,
                       totals, b->found.ptr, b->counts.ptr, b->psi.ptr, b->coeffs.ptr,
                       b->other_psi.ptr, K, b->out_row.ptr, b->out_col.ptr, b->elements.ptr,
                       b->field.ptr);
  }
  ASP_HIP_TRY(hipGetLastError());
  ASP_HIP_TRY(hipEventRecord(b->ev_stop, s));
  unsigned long long total = 0;  // (K = 0: no kernel ran)
  if (K > 0) ASP_HIP_TRY(hipMemcpyAsync(&total, b->nnz.ptr, sizeof total, hipMemcpyDeviceToHost, s));
  ASP_HIP_TRY(hipStreamSynchronize(s));
  ASP_HIP_TRY(hipEventElapsedTime(&b->last_ms, b->ev_start, b->ev_stop));
  // this run's tile totals were read by its last kernel; the other parity's are zero again
  b->parity ^= 1u;
  b->clean = K > 0;
  b->last_nnz = static_cast<uint64_t>(total);
  if (nnz) *nnz = b->last_nnz;
  return ASP_OK;
}

float asp_build_last_ms(asp_build const *b) { return b ? b->last_ms : 0.0f; }

int asp_build_download(asp_build *b, uint32_t *row_indices, uint32_t *col_indices,
                       double *elements, double *field) {
  if (!b) return asp::set_error(ASP_ERR_INVALID, "null build handle");
  ASP_TRY(asp::bind_device());
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
