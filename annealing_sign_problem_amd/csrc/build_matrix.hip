// Coupling build on gfx950: HIP replacement for the reference's
// cbits/build_matrix.c (build_matrix :22-65, ls_bits512_cmp :7-20,
// extract_signs :67-76) behind the same C symbols.
//
// Data in HBM (all SoA except the 512-bit keys, which keep the reference's AoS
// layout because that is what the ABI hands over):
//   table   ls_bits512[K]   sorted, unique           table0  u64[K] = words[0]
//   slots   u64[2^s]        blocked hash of table0: {fingerprint:32 | index+1:32}, load factor
//                           <= 1/2; a key is in its home BUCKET of eight slots (64 bytes, one
//                           probe) unless that is full, then in the next bucket that is not
//   needles ls_bits512[N]   flat, row-major by row
//   counts i64[K], psi f64[K], coeffs f64[N], other_psi f64[N], other_counts i64[K]
//   offsets i64[K+1] (prefix sums of other_counts, formed on the host while the upload checks
//   them), found i32[N] (table index or -1),
//   group_hits u32[N / 64] (hits among 64 consecutive needles), chunk_total u32[N / 2048],
//   super_total u64[2][N / 131072] (one parity per run)
//   out: row u32[N], col u32[N], elements f64[N], field f64[K]
//
// One build = THREE launches on one stream, no host round trip except nnz:
//   k_insert_keys   table -> table0 (words[0]) and table0 -> slots (atomicCAS on the slots of the
//                   bucket, from a slot the hash picks).  Whether some table key has a non-zero
//                   word 1..7 the host notes while it uploads the keys.
//   k_search_flat   the needles as ONE FLAT ARRAY, whatever row they belong to: a wavefront
//                   resolves 64 consecutive needles per trip (4 KiB of contiguous memory, loaded
//                   coalesced and requested BEFORE the previous trip's probe — the loads depend on
//                   nothing, one round trip carries both), each lane one connection through
//                   the hash: its bucket in four loads issued together, fingerprint candidates
//                   verified against table0 and, for keys beyond 64 spins, against the other
//                   seven words (the reference bsearches,
//                   cbits/build_matrix.c:37-38; on a unique table any exact lookup returns the
//                   same element).  Hits are counted per group of 64 needles (ballot), per chunk
//                   (a workgroup) and, atomically, per super-chunk.  Round 3 searched row by row
//                   (32 lanes per row: offsets -> needles -> slot -> verify, four dependent round
//                   trips per wavefront and nothing else in flight): 110 us of a 170 us build.
//   k_emit_rows     first the block's output position WITHOUT a scan kernel: hits before the
//                   first needle of its first row = super-chunks before (<= N / 131072 numbers) +
//                   chunks before inside the super-chunk (<= 63) + groups before inside the chunk
//                   (<= 31) + hits among the <= 63 needles before it inside its group — integer
//                   sums, so the order does not matter; 32 lanes per row: the row's hits, its
//                   first 32 connections and those totals are loaded in front of ONE barrier;
//                   then COO triples in input order at ballot-prefix
//                   positions (coalesced), and the row's field as a LEFT-TO-RIGHT sum over its
//                   misses (the reference's rounding, cbits/build_matrix.c:49); last, every block
//                   zeroes its share of the hash slots and block 0 the OTHER parity's super-chunk
//                   totals: the next run finds them clean
//
// Arithmetic: __dmul_rn / __dadd_rn keep every product and the field add
// separately rounded (no FMA contraction), matching the reference binary.
// Memory-bound integer/f64 streaming: no MFMA, no LDS tiling of the payload.
#include <vector>

#include "asp_common.hpp"

namespace {

using asp::DeviceBuffer;

// Timing-only ablations of k_search_flat (results are WRONG when set; never set in the product
// build): 1 no probe, 2 probe without the verifying load, 3 every trip re-reads the chunk's first
// 4 KiB (no HBM stream), 4 no pass through LDS (a lane takes the first piece it loaded for its
// key), 5 one 16-byte load per probe instead of the bucket's four, 6 no result stores
#ifndef ASP_BUILD_ABL
#define ASP_BUILD_ABL 0
#endif

constexpr int kThreads = 256;
constexpr int kRowLanes = 32;                         // lanes cooperating on one row (k_emit_rows)
constexpr int kRowsPerBlock = kThreads / kRowLanes;   // 8
constexpr uint32_t kGroup = 64;                       // needles a wavefront resolves per trip
constexpr uint32_t kSearchWaves = kThreads / 64;      // 4
#ifndef ASP_BUILD_SLOTS_PER_KEY
#define ASP_BUILD_SLOTS_PER_KEY 2
#endif
constexpr uint64_t kSlotsPerKey = ASP_BUILD_SLOTS_PER_KEY;  // the hash's load factor is at most its inverse
#ifndef ASP_BUILD_TRIPS
#define ASP_BUILD_TRIPS 8
#endif
constexpr uint32_t kTrips = ASP_BUILD_TRIPS;          // trips of a wavefront of k_search_flat (4 / 8 / 16: the same)
constexpr uint32_t kGroupsPerChunk = kSearchWaves * kTrips;  // 32: a chunk = a workgroup = 2048 needles
constexpr uint32_t kChunk = kGroupsPerChunk * kGroup;
constexpr uint32_t kChunksPerSuper = 64;              // 131072 needles
static_assert(kGroupsPerChunk <= 64 && kChunksPerSuper <= 64, "one wavefront sums a level");
constexpr uint32_t kStagePlaces = kGroup * 5;         // 16-byte places of a wavefront's LDS staging area
// (slots of a bucket: 8, 4 or 2.  Measured at K = 1e5, search / insert in us: eight 70.4 / 12.5,
// four — half the bytes per probe, ~8 % of the look-ups in a second bucket — 66.8 / 15.1, two
// 67.8 / 13.6, two at a load factor <= 1/4 66.5 / 9.4: 126-128 us per build whichever.  The
// timing-only ablations above say where the search's 70 us are: 22 without any probe (the keys
// alone, largely from the 256 MB last-level cache on repeated runs), 56 with one 16-byte load per
// probe and the chain cut there, 59 without the verifying load, 68 without the stores, 42 with
// the keys re-read from L2 — the dependent look-up, not the bytes.)
#ifndef ASP_BUILD_BUCKET
#define ASP_BUILD_BUCKET 8
#endif
constexpr uint32_t kBucket = ASP_BUILD_BUCKET;        // hash slots a probe reads

// One hash slot: {fingerprint: the hash's high word | index + 1}, 0 = empty.  Eight bytes, so that
// the whole hash (load factor <= 1/2: 2 MiB at K = 1e5) and the first words it is verified against
// (table0, 0.8 MiB) stay in an XCD's 4 MiB of L2 beside the needles streaming through it: with
// the key itself in a 16-byte slot (8 MiB at load factor 1/4) half of the probes missed the L2 and
// the look-ups fetched as many bytes from HBM as the needles (FETCH_SIZE, round 4).
using Slot = unsigned long long;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// Blocked hashing: a key lives in its home bucket (eight slots, 64 bytes) unless that is full,
// then in the next bucket that is not.  A look-up therefore reads whole buckets: a match is
// wherever it is in the bucket, and a bucket with an empty slot ends the chain.
__global__ __launch_bounds__(kThreads) void k_insert_keys(const ls_bits512 *__restrict__ table,
                                                         uint64_t n, uint64_t *__restrict__ table0,
                                                         Slot *__restrict__ slots, uint64_t bucket_mask) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  const uint64_t word0 = table[i].words[0];
  table0[i] = word0;
  const uint64_t h = mix64(word0);
  const Slot entry = (h & 0xFFFFFFFF00000000ull) | (i + 1);
  // (the keys of a bucket start at different slots of it: fewer of them fight over one slot)
  const uint32_t first = static_cast<uint32_t>(h >> 29) & (kBucket - 1);
  for (uint64_t b = h & bucket_mask;; b = (b + 1) & bucket_mask) {
    for (uint32_t t = 0; t < kBucket; ++t) {
      if (atomicCAS(&slots[b * kBucket + ((first + t) & (kBucket - 1))], 0ull, entry) == 0ull) return;
    }
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Loads AS WRITTEN (left to itself hipcc loads slot after slot, each behind the test of the one
// before: a dependent round trip per slot; a volatile access becomes a system-scope load past the
// L2).  vmcnt counts in order, so waiting for these loads is waiting for every older one too — the
// asm's wait costs what the compiler's would; nothing of the asm is outstanding when the
// compiler's own waits run.
// The slots of one bucket, two to a 16-byte load:
__device__ __forceinline__ void load_bucket(const Slot *bucket, u32x4 (&s)[kBucket / 2]) {
  static_assert(kBucket == 8 || kBucket == 4 || kBucket == 2, "the loads are spelled out");
#if ASP_BUILD_ABL == 5
  {
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(s[0]) : "v"(bucket) : "memory");
    for (uint32_t j = 1; j < kBucket / 2; ++j) s[j] = u32x4{0u, 0u, 0u, 0u};
  }
#elif ASP_BUILD_BUCKET == 8
  {
    asm volatile(
        "global_load_dwordx4 %0, %4, off\n\t"
        "global_load_dwordx4 %1, %4, off offset:16\n\t"
        "global_load_dwordx4 %2, %4, off offset:32\n\t"
        "global_load_dwordx4 %3, %4, off offset:48\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3])
        : "v"(bucket)
        : "memory");
  }
#elif ASP_BUILD_BUCKET == 2
  {
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(s[0]) : "v"(bucket) : "memory");
  }
#else
  {
    asm volatile(
        "global_load_dwordx4 %0, %2, off\n\t"
        "global_load_dwordx4 %1, %2, off offset:16\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(s[0]), "=&v"(s[1])
        : "v"(bucket)
        : "memory");
  }
#endif
}

// Index in the table of the needle with hash h, or -1 (`verify(index)`: the needle IS
// table[index]): a lane reads the needle's home bucket whole — the chain ends there in all but
// ~1 % of the look-ups — and moves on only from a full bucket.
template <typename Verify>
__device__ __forceinline__ int32_t find_key(const Slot *__restrict__ slots, uint64_t bucket_mask, uint64_t h,
                                            Verify verify) {
  const uint32_t fingerprint = static_cast<uint32_t>(h >> 32);
  for (uint64_t b = h & bucket_mask;; b = (b + 1) & bucket_mask) {
    u32x4 s[kBucket / 2];
    load_bucket(slots + b * kBucket, s);
    uint32_t empty = 0, match = 0;
#pragma unroll
    for (uint32_t j = 0; j < kBucket / 2; ++j) {  // slot 2 j = {x: index + 1, y: fingerprint}, 2 j + 1 = {z, w}
      empty |= (s[j].x == 0u ? 1u : 0u) << (2 * j) | (s[j].z == 0u ? 1u : 0u) << (2 * j + 1);
      match |= (s[j].y == fingerprint ? 1u : 0u) << (2 * j) | (s[j].w == fingerprint ? 1u : 0u) << (2 * j + 1);
    }
    match &= ~empty;  // (an empty slot is all zeros, and zero is a fingerprint like any other)
    while (match) {  // one candidate in all but 2^-32 of the cases (keys that share their first word aside)
      const uint32_t j = static_cast<uint32_t>(__ffs(match) - 1);
      uint32_t index1 = 0;
#pragma unroll
      for (uint32_t k = 0; k < kBucket / 2; ++k) {
        index1 = 2 * k == j ? s[k].x : index1;
        index1 = 2 * k + 1 == j ? s[k].z : index1;
      }
      if (verify(index1 - 1u)) return static_cast<int32_t>(index1 - 1u);
      match &= match - 1u;
    }
    if (empty) return -1;
  }
}

// Ballot restricted to this lane's 32-lane half.
__device__ __forceinline__ uint32_t half_ballot(bool predicate, uint32_t lane) {
  const uint64_t all = __ballot(predicate);
  return static_cast<uint32_t>(all >> (lane & 32u));
}

// Workgroup c resolves the needles [c kChunk, (c + 1) kChunk): wavefront w the groups k 4 + w.
template <bool HAS_TAILS>
__device__ __forceinline__ void search_flat_body(
    const Slot *__restrict__ slots, uint64_t bucket_mask, const uint64_t *__restrict__ table0,
    const ls_bits512 *__restrict__ table, const ls_bits512 *__restrict__ needles, uint64_t num_needles,
    int32_t *__restrict__ found, uint32_t *__restrict__ group_hits, uint32_t *__restrict__ chunk_total,
    unsigned long long *__restrict__ super_total, uint4 *stage, uint32_t *hits_of_wave) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint64_t chunk_first = static_cast<uint64_t>(blockIdx.x) * kChunk;
  // The 64 needles of a trip are 4 KiB of CONTIGUOUS memory (AoS keys).  Each lane fetches four
  // 16-byte pieces of that run (fully coalesced, 1 KiB per instruction), the wavefront passes them
  // through LDS and every lane reads its own key back — instead of eight loads per lane at a
  // 64-byte stride.  (In LDS a key takes FIVE 16-byte places, 80 bytes: the eight lanes of a pass of
  // the read-back then cover the banks once; at 64 every second lane shares four banks.)
  uint4 *mine = stage + wave * kStagePlaces;
  const u32x4 *src = reinterpret_cast<const u32x4 *>(needles);
  const uint64_t last_piece = num_needles * 4u - 1u;  // (num_needles > 0: the launch is skipped otherwise)
  // loads of trip k: unconditional, on clamped addresses (a group past the end re-reads the last
  // piece: under a branch hipcc would wait for every load where it is issued)
  uint4 buf[4];
  auto request = [&](uint32_t k) {
    const uint32_t slice = ASP_BUILD_ABL == 3 ? 0u : k;
    const uint64_t piece0 = (chunk_first + static_cast<uint64_t>(slice * kSearchWaves + wave) * kGroup) * 4u + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint64_t piece = piece0 + static_cast<uint64_t>(i) * kGroup;
      const u32x4 v = __builtin_nontemporal_load(src + (piece < last_piece ? piece : last_piece));
      buf[i] = uint4{v.x, v.y, v.z, v.w};
    }
  };
  uint32_t hits = 0;  // of this wavefront's groups (wave-uniform)
  request(0);
#pragma unroll 1
  for (uint32_t k = 0; k < kTrips; ++k) {
    const uint64_t group = static_cast<uint64_t>(blockIdx.x) * kGroupsPerChunk + k * kSearchWaves + wave;
    const uint64_t first = group * kGroup;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t piece = i * kGroup + lane;  // piece & 3 of key piece >> 2
      if (ASP_BUILD_ABL != 4) mine[piece + (piece >> 2)] = buf[i];
    }
    __builtin_amdgcn_wave_barrier();  // same wavefront: LDS ops are in order, keep them so
    uint64_t key[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 q = ASP_BUILD_ABL == 4 ? buf[0] : mine[lane * 5 + j];
      key[2 * j] = (static_cast<uint64_t>(q.y) << 32) | q.x;
      key[2 * j + 1] = (static_cast<uint64_t>(q.w) << 32) | q.z;
    }
    __builtin_amdgcn_wave_barrier();  // the next trip overwrites the staging area
    // The registers are free: the next trip's needles are requested BEFORE this trip's probe and
    // arrive with it — one round trip to HBM per trip carries both; the verifying load that
    // follows hits the L2.
    if (k + 1u < kTrips) request(k + 1u);
    if (first >= num_needles) continue;  // (wave-uniform) nothing of this group exists
    const uint64_t e = first + lane;
    uint64_t needle_tail = 0;
#pragma unroll
    for (int w = 1; w < 8; ++w) needle_tail |= key[w];
    // HAS_TAILS = false: no table key has a non-zero word 1..7, so a needle with one is a miss
    // without a probe
    const bool wanted = e < num_needles && (HAS_TAILS || needle_tail == 0);
    const uint64_t h = mix64(key[0]);
    auto verify = [&](uint32_t at) {
      if (ASP_BUILD_ABL == 2) return true;
      if (table0[at] != key[0]) return false;  // (an L2 hit: 8 K bytes in all)
      if (!HAS_TAILS) return true;
      // word 0 matches: the full 512-bit keys must agree (cbits/build_matrix.c:11-18)
      uint64_t diff = 0;
#pragma unroll
      for (int w = 1; w < 8; ++w) diff |= table[at].words[w] ^ key[w];
      return diff == 0;
    };
    // (tried: four lanes per bucket, 16 bytes each, the hash by shuffle and the verdict through LDS —
    // a quarter of the lines per load instruction, and slower: 100 against 71 us)
    int32_t idx = -1;
    if (ASP_BUILD_ABL != 1 && wanted) idx = find_key(slots, bucket_mask, h, verify);
    if (e < num_needles && (ASP_BUILD_ABL != 6 || idx == 12345)) found[e] = idx;
    const uint32_t hcount = static_cast<uint32_t>(__popcll(__ballot(idx >= 0)));
    if (lane == 0 && (ASP_BUILD_ABL != 6 || hcount == 99)) group_hits[group] = hcount;
    hits += hcount;
  }
  // hits of the chunk: to its own total and, atomically, to the total of its super-chunk
  if (lane == 0) hits_of_wave[wave] = hits;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
#pragma unroll
    for (uint32_t k = 0; k < kSearchWaves; ++k) total += hits_of_wave[k];
    chunk_total[blockIdx.x] = total;
    if (total) atomicAdd(&super_total[blockIdx.x / kChunksPerSuper], static_cast<unsigned long long>(total));
  }
}

// HAS_TAILS (whether some table key has a non-zero word 1..7: the host sees the keys when it
// uploads them) picks the instantiation, so that the common case — keys of at most 64 spins — does
// not carry the registers of the eight-word comparison.
template <bool HAS_TAILS>
__global__ __launch_bounds__(kThreads) void k_search_flat(
    const Slot *__restrict__ slots, uint64_t bucket_mask, const uint64_t *__restrict__ table0,
    const ls_bits512 *__restrict__ table, const ls_bits512 *__restrict__ needles, uint64_t num_needles,
    int32_t *__restrict__ found, uint32_t *__restrict__ group_hits, uint32_t *__restrict__ chunk_total,
    unsigned long long *__restrict__ super_total) {
  __shared__ uint4 stage[kSearchWaves * kStagePlaces];
  __shared__ uint32_t hits_of_wave[kSearchWaves];
  search_flat_body<HAS_TAILS>(slots, bucket_mask, table0, table, needles, num_needles, found, group_hits, chunk_total,
                              super_total, stage, hits_of_wave);
}

struct EmitTotals {
  const uint32_t *group_hits;              // [groups]
  const uint32_t *chunk_total;             // [chunks]
  const unsigned long long *super_total;   // [supers] of this run
  unsigned long long *nnz;                 // the build's number of couplings
  // left clean for the next run:
  Slot *slots;
  uint64_t slot_count;
  unsigned long long *next_super_total;    // the other parity's totals
  uint32_t num_supers;
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
  // Everything a row needs is requested in TWO rounds of loads, then one barrier: (1) the row's
  // bounds, its count and amplitude, and the first needle of the block; (2) what the bounds point
  // at — the first 32 connections of the row (look-up result, coefficient, amplitude: kept in
  // registers for the emission below) and this wavefront's share of the hit totals before the
  // block.  (Until round 4's last build the block position, the rows' hits and the emission were
  // three phases with a barrier and a round trip each: 45 us at K = 1e5.)
  const int64_t begin = row_ok ? offsets[r] : 0;
  const int64_t end = row_ok ? offsets[r + 1] : 0;
  const uint64_t o = static_cast<uint64_t>(offsets[static_cast<uint64_t>(blockIdx.x) * kRowsPerBlock]);
  const double c = row_ok ? static_cast<double>(counts[r]) : 0.0;  // exact i64 -> f64 as in C
  const double a = row_ok ? fabs(psi[r]) : 0.0;
  const int64_t other_len = __shfl_xor(end - begin, 32, 64);
  // both halves of the wavefront iterate until the longer row is done, so a ballot is
  // always executed by all 64 lanes
  const int64_t trips = ((end - begin > other_len ? end - begin : other_len) + kRowLanes - 1) / kRowLanes;
  __shared__ unsigned long long base_part[kThreads / 64];
  __shared__ uint32_t hits_of_row[kRowsPerBlock];
  const int64_t e0 = begin + sub;
  const bool active0 = e0 < end;
  const int32_t pos0 = active0 ? found[e0] : -1;
  const double coeff0 = active0 ? coeffs[e0] : 0.0;
  const double x0 = active0 ? other_psi[e0] : 0.0;
  {
    // ---- where this block's couplings start: the hits before the first needle of its first row
    const uint64_t g0 = o / kGroup, c0 = g0 / kGroupsPerChunk, s0 = c0 / kChunksPerSuper;
    const uint32_t wave = threadIdx.x >> 6;
    unsigned long long mine = 0;
    if (wave == 0) {
      for (uint64_t t = lane; t < s0; t += 64) mine += totals.super_total[t];
    } else if (wave == 1) {
      const uint64_t ch = s0 * kChunksPerSuper + lane;
      if (ch < c0) mine = totals.chunk_total[ch];
    } else if (wave == 2) {
      const uint64_t g = c0 * kGroupsPerChunk + lane;
      if (g < g0) mine = totals.group_hits[g];
    } else {
      const uint64_t e = g0 * kGroup + lane;
      if (e < o) mine = found[e] >= 0 ? 1u : 0u;
    }
    for (int step = 1; step < 64; step <<= 1) mine += __shfl_xor(mine, step, 64);
    if (lane == 0) base_part[wave] = mine;
  }
  {
    // ---- the hits of the block's rows
    uint32_t hits = 0;
    for (int64_t it = 0; it < trips; ++it) {
      const int64_t e = begin + it * kRowLanes + sub;
      const int32_t pos = it == 0 ? pos0 : (e < end ? found[e] : -1);
      hits += __popc(half_ballot(pos >= 0, lane));
    }
    if (sub == 0) hits_of_row[row_in_block] = hits;
  }
  __syncthreads();
  unsigned long long block_base = 0;
#pragma unroll
  for (uint32_t k = 0; k < kThreads / 64; ++k) block_base += base_part[k];
  int64_t w = static_cast<int64_t>(block_base);
  for (uint32_t k = 0; k < row_in_block; ++k) w += hits_of_row[k];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    unsigned long long all = block_base;
    for (int k = 0; k < kRowsPerBlock; ++k) all += hits_of_row[k];
    *totals.nnz = all;
  }
  double f = 0.0;
  for (int64_t it = 0; it < trips; ++it) {
    const int64_t e = begin + it * kRowLanes + sub;
    const bool active = e < end;
    const int32_t pos = it == 0 ? pos0 : (active ? found[e] : -1);
    // ((counts * coeff) * |psi|) * x, each product rounded: build_matrix.c:41-42,49
    const double coeff = it == 0 ? coeff0 : (active ? coeffs[e] : 0.0);
    const double head = active ? __dmul_rn(__dmul_rn(c, coeff), a) : 0.0;
    const double x = it == 0 ? x0 : (active ? other_psi[e] : 0.0);
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
  // ---- leave the tables clean for the next run (nobody reads the slots any more) ----
  {
    const uint64_t share = (totals.slot_count + gridDim.x - 1) / gridDim.x;
    const uint64_t first = static_cast<uint64_t>(blockIdx.x) * share;
    const uint64_t last = first + share < totals.slot_count ? first + share : totals.slot_count;
    for (uint64_t k = first + threadIdx.x; k < last; k += kThreads) totals.slots[k] = 0ull;
    if (blockIdx.x == 0) {
      for (uint32_t t = threadIdx.x; t < totals.num_supers; t += kThreads) totals.next_super_total[t] = 0ull;
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
  DeviceBuffer<Slot> slots;
  bool has_tails = false;  // some table key has a non-zero word 1..7 (seen by asp_build_upload)
  uint64_t slot_mask = 0;
  DeviceBuffer<int64_t> counts, offsets;
  DeviceBuffer<double> psi, coeffs, other_psi, elements, field;
  DeviceBuffer<int32_t> found;
  DeviceBuffer<uint32_t> group_hits, chunk_total, out_row, out_col;
  DeviceBuffer<unsigned long long> super_total, nnz;  // super_total: [2][num_supers], one parity per run
  uint32_t num_chunks = 0, num_supers = 0;
  uint32_t parity = 0;
  bool clean = false;  // slots and this parity's super totals are zero (k_emit_rows leaves them so)
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
  while (slot_count < kSlotsPerKey * K) slot_count <<= 1;  // load factor <= 1 / kSlotsPerKey
  b->slot_mask = slot_count - 1;
  const uint64_t num_groups = (N + kGroup - 1) / kGroup;
  b->num_chunks = static_cast<uint32_t>((N + kChunk - 1) / kChunk);  // (N < 2^40)
  b->num_supers = (b->num_chunks + kChunksPerSuper - 1) / kChunksPerSuper;
  ok = ok && b->table.alloc(K) == ASP_OK && b->needles.alloc(N) == ASP_OK &&
       b->table0.alloc(K) == ASP_OK && b->slots.alloc(b->slot_mask + 1) == ASP_OK &&
       b->counts.alloc(K) == ASP_OK &&
       b->offsets.alloc(K + 1) == ASP_OK &&
       b->psi.alloc(K) == ASP_OK && b->coeffs.alloc(N) == ASP_OK &&
       b->other_psi.alloc(N) == ASP_OK && b->elements.alloc(N) == ASP_OK &&
       b->field.alloc(K) == ASP_OK && b->found.alloc(N) == ASP_OK &&
       b->group_hits.alloc(num_groups) == ASP_OK && b->chunk_total.alloc(b->num_chunks) == ASP_OK &&
       b->super_total.alloc(2ull * b->num_supers) == ASP_OK && b->nnz.alloc(1) == ASP_OK &&
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
  // does any table key reach beyond its first word?  (the search kernel's instantiation)
  bool tails = false;
  for (uint64_t r = 0; r < K && !tails; ++r) {
    uint64_t rest = 0;
    for (int w = 1; w < 8; ++w) rest |= spins[r].words[w];
    tails = rest != 0;
  }
  b->has_tails = tails;
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
    ASP_HIP_TRY(hipMemsetAsync(b->slots.ptr, 0, (b->slot_mask + 1) * sizeof(Slot), s));
    if (b->num_supers) {
      ASP_HIP_TRY(hipMemsetAsync(b->super_total.ptr, 0, 2ull * b->num_supers * sizeof(unsigned long long), s));
    }
  }
  b->clean = false;
  ASP_HIP_TRY(hipEventRecord(b->ev_start, s));
  if (K > 0) {
    unsigned long long *supers = b->super_total.ptr + static_cast<size_t>(b->parity) * b->num_supers;
    unsigned long long *other_supers = b->super_total.ptr + static_cast<size_t>(b->parity ^ 1u) * b->num_supers;
    hipLaunchKernelGGL(k_insert_keys, dim3(blocks_for(K)), dim3(kThreads), 0, s, b->table.ptr, K,
                       b->table0.ptr, b->slots.ptr, b->slot_mask / kBucket);
    if (b->num_chunks) {  // (a build without a single connection searches nothing)
      hipLaunchKernelGGL(b->has_tails ? k_search_flat<true> : k_search_flat<false>, dim3(b->num_chunks),
                         dim3(kThreads), 0, s, b->slots.ptr, b->slot_mask / kBucket, b->table0.ptr, b->table.ptr,
                         b->needles.ptr, b->num_other, b->found.ptr, b->group_hits.ptr, b->chunk_total.ptr,
                         supers);
    }
    EmitTotals totals{b->group_hits.ptr, b->chunk_total.ptr, supers, b->nnz.ptr, b->slots.ptr,
                      b->slot_mask + 1, other_supers, b->num_supers};
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
  // this run's super-chunk totals were read by its last kernel; the other parity's are zero again
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
