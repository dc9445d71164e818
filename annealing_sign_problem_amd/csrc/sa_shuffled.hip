// Shuffled sweep "ASP-SA-1S" on gfx950: Metropolis annealing with a FRESH
// RANDOM VISITING ORDER EVERY SWEEP — the order the reference's annealer uses
// (ising_glass_annealer behind annealing_sign_problem/common.py:242-248 and
// experiments/full_hilbert_space.py:212-218; DESIGN.md §6.1: its published success
// probabilities are reproduced by this order and by no fixed one).
//
// Sweep t visits the spins in ascending (priority, index), priority_t(i) = word 0 of
// Philox4x32-10(counter (i, t, 0xFFFFFFFE, 0), key seed); the same order for every chain.  A
// sequential sweep in that order equals visiting the LEVELS of the priority graph one after
// another: level(i) = 1 + max level(j) over the neighbours j that come before i.  Spins of a
// level are pairwise non-adjacent, so a level is updated in parallel, and any order inside a
// level gives the same bits.
//
// Mapping to the machine — two kernels in a pipeline, a chunk of sweeps at a time:
//   * k_shuffled_orders (one workgroup per SWEEP): priorities, then the levels by a topological
//     peel (every spin counts its earlier neighbours; spins at zero form level 0; a finished
//     level decrements the counters of its later neighbours, those reaching zero form the next
//     level — 2 nnz edge visits per sweep, not levels x nnz), a counting sort of every level by
//     row length, and the sweep's couplings written as a LEVEL-MAJOR SLICED ELL in exactly the
//     quad-interleaved layout the colour-ordered kernel streams (csrc/sa_plan.cpp): a block of
//     64 same-level spins, width = its longest row, columns = LDS addresses of the neighbours.
//     The order depends on (seed, t) only, so this is done ONCE per sweep for all chains and
//     all workgroups of the call;
//   * k_sa_sweep_shuffled (one workgroup per group of M chains): spins in LDS in ORIGINAL
//     order (a word per spin, byte m = chain m, for M <= 4 — the one-instruction SDWA sign — or a
//     byte per spin, bit m = chain m, for M = 8); a wavefront per block of the current level,
//     lane = spin, three coalesced 16-byte loads per four couplings shared by the M chains, the
//     proposal arithmetic, random words and exact integer energy bookkeeping of ASP-SA-1
//     (csrc/sa_sweep.hip), one workgroup barrier per level.
//   The order kernel of chunk c + 1 runs on a second stream beside the sweep kernel of chunk c;
//   chain state waits in HBM between chunks.  Integer + f64 work bound by per-level latency
//   and VALU issue: no MFMA.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "asp_common.hpp"
#include "sa_device.hpp"
#include "sa_internal.hpp"
#include "sa_plan.hpp"

namespace {

using asp::DeviceBuffer;
using asp::kDummySpin;
using namespace asp::dev;

constexpr uint32_t kPriorityCounter = 0xFFFFFFFEu;  // counter word 2 of the priority draw
constexpr uint32_t kOrderThreads = 1024;            // threads of an order workgroup (large K)
constexpr uint32_t kClassCap = 63;                  // rows of >= 63 quads share the last class
constexpr uint64_t kQuadLimit = 0x100000;           // quads of one sweep: 3 KiB each within 32-bit byte offsets

// status words shared by the kernels of a call (device memory, zeroed per attempt)
enum : uint32_t { kStatBad = 0, kStatLevels = 1, kStatBlocks = 2, kStatQuads = 3, kStatWords = 4 };
constexpr uint32_t kTimingSlots = 6, kTimingWaves = 8;  // u64[waves][slots] behind the status words (ASP_SHUF_TIMING)
constexpr uint32_t kOrderTimingSlots = 8;               // u64[slots] behind those: wavefront 0 of the order kernel
constexpr uint32_t kStatusWords = 4 + 2 * kTimingSlots * kTimingWaves + 2 * kOrderTimingSlots;

// ---------------------------------------------------------------------------
// Order kernel
// ---------------------------------------------------------------------------

struct OrderArgs {
  // static: rows of A padded to quads (sa_plan.hpp RowQuads)
  const uint32_t *rq_ptr;  // [K + 1]
  const uint4 *rq_col;     // [quads]
  const double2 *rq_val;   // [quads][2]
  const double *field;     // [K]
  uint64_t seed;
  uint32_t num_spins, first_sweep, count;
  uint32_t level_cap, block_cap, quad_cap;  // capacities per sweep of the outputs below
  uint32_t stream_kib;                      // KiB between the coupling streams of consecutive sweeps
  uint32_t log_s;                           // a block is S = 1 << log_s spins of one level (S = 4 .. 64)
  uint32_t lanes_per_row;                   // power of two <= 64: lanes sharing a row in the graph passes
  uint32_t threads;                         // threads of this problem's order workgroups (a multiple of 64)
  uint32_t num_quads, max_quads;            // quads of all rows of A / of its longest row
  uint32_t lds_arrays;                      // 1: priorities, counters and the order in LDS (PeelArrays)
  uint32_t lds_counters;                    // 1 / 2: (with finish_only) the whole peel in the per-sweep workgroup, its
                                            //    counters as bytes / nibbles in LDS, order and later-masks in HBM
  // The WIDE path (large clusters: the per-spin arrays in HBM): priorities, counts, every level of
  // the peel and the stream are kernels of their own over ALL sweeps of the chunk (k_order_prio,
  // k_order_counts, k_order_level, k_order_stream) and only what needs a sweep's tables in LDS
  // stays in a workgroup per sweep (k_shuffled_orders with finish_only = 1).
  uint32_t finish_only;                     // 1: the first wide_levels levels are peeled by k_order_level launches
  uint32_t wide_levels;                     // number of those launches per chunk
  uint32_t *peel_ctl;                       // [count][8]: members of level l % 3 | first position of level l % 3 | -
  uint32_t *level_start_g;                  // [count][level_cap + 2] first position of every level
  uint8_t *later;                           // [count][quads of A] bit j: entry j of the quad is a LATER neighbour
  uint32_t col_shift;                       // columns are written as (neighbour << col_shift): LDS addresses
  // scratch, [count][K] each
  uint32_t *prio, *indeg, *order;
  // outputs, per sweep of the chunk
  uint32_t *level_block;  // [count][level_cap + 1] first block of level l; entry [levels] = blocks
  uint32_t *num_levels;   // [count]
  uint2 *block_meta;      // [count][block_cap] {offset of the block in the sweep's stream in slabs, quads}
  uint32_t *spin_of_pos;  // [count][block_cap * S] scratch: kDummySpin = padding lane
  // The sweep's couplings as ONE stream per sweep, block after block in level-major order, in
  // SLABS of 16 S bytes (1 KiB for S = 64).  A block is one header slab — per spin of the block 16
  // bytes: its index u32 (kDummySpin = padding lane), 4 bytes unused, its field f64 — followed by
  // three slabs per quad of couplings: columns uint4[S], values double2[S] (entries 0, 1), values
  // double2[S] (entries 2, 3); the quad layout of csrc/sa_plan.cpp.  One buffer resource and one
  // scalar offset address all of a block; a lane's part of a slab is ONE 16-byte load.
  uint8_t *stream;        // [count][stream_kib KiB]
  uint32_t *status;
};

// Exclusive prefix sum of a[0..n) in LDS, in place; returns the total.  All threads call it;
// `carry` is one LDS word of scratch, `wave_tot` 16.
__device__ uint32_t block_exclusive_scan(uint32_t *a, uint32_t n, uint32_t *wave_tot, uint32_t *carry,
                                         uint32_t nthreads) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, waves = nthreads >> 6;
  if (tid == 0) *carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += nthreads) {
    const uint32_t i = base + tid;
    const uint32_t v = i < n ? a[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int step = 1; step < 64; step <<= 1) {
      const uint32_t o = __shfl_up(inc, step, 64);
      if (lane >= static_cast<uint32_t>(step)) inc += o;
    }
    if (lane == 63u) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t before = *carry;
    for (uint32_t w = 0; w < wave; ++w) before += wave_tot[w];
    if (i < n) a[i] = before + inc - v;
    __syncthreads();
    if (tid == 0) {
      uint32_t total = *carry;
      for (uint32_t w = 0; w < waves; ++w) total += wave_tot[w];
      *carry = total;
    }
    __syncthreads();
  }
  return *carry;
}

__device__ __forceinline__ bool comes_before(uint32_t pa, uint32_t a, uint32_t pb, uint32_t b) {
  return pa < pb || (pa == pb && a < b);
}

// One pass of a wavefront over the coupling stream of a sweep: the 64 / S blocks b0 .. of the
// sweep (lane = (block of the pass, slot)): header slab and quad slabs of each.  `first` / `quads`:
// slab offset and width of every block (LDS or HBM).
template <typename Args>
__device__ __forceinline__ void write_stream_pass(const Args &a, uint8_t *stream, const uint32_t *sop, uint32_t b,
                                                  bool live, uint32_t first_slab, uint32_t quads_of_block,
                                                  uint32_t lane) {
  const uint32_t S = 1u << a.log_s;
  const uint32_t slot = lane & (S - 1u);
  const uint32_t slab_bytes = 16u << a.log_s;
  const uint32_t i = live ? sop[(b << a.log_s) + slot] : kDummySpin;
  const bool real = i != kDummySpin;
  const uint32_t row = real ? a.rq_ptr[i] : 0u;
  const uint32_t mine = real ? a.rq_ptr[i + 1] - row : 0u;
  const uint32_t own = real ? i << a.col_shift : 0u;  // padding reads the lane's own spin (x +0.0)
  const uint32_t quads = live ? quads_of_block : 0u;
  uint32_t most = quads;  // the widest block of the pass
#pragma unroll
  for (int step = 1; step < 64; step <<= 1) most = max(most, static_cast<uint32_t>(__shfl_xor(most, step, 64)));
  uint8_t *block = stream + static_cast<uint64_t>(live ? first_slab : 0u) * slab_bytes;
  if (live) {
    const double h = real ? a.field[i] : 0.0;
    const unsigned long long hb = static_cast<unsigned long long>(__double_as_longlong(h));
    reinterpret_cast<uint4 *>(block)[slot] =
        make_uint4(i, 0u, static_cast<uint32_t>(hb), static_cast<uint32_t>(hb >> 32));
  }
  for (uint32_t q = 0; q < most; ++q) {
    if (!live || q >= quads) continue;
    uint4 c = make_uint4(own, own, own, own);
    double2 v01 = make_double2(0.0, 0.0), v23 = make_double2(0.0, 0.0);
    if (q < mine) {
      // (padding entries of a row carry the row's own index: shifted like every column)
      c = a.rq_col[row + q];
      c.x <<= a.col_shift;
      c.y <<= a.col_shift;
      c.z <<= a.col_shift;
      c.w <<= a.col_shift;
      v01 = a.rq_val[static_cast<uint64_t>(row + q) * 2u];
      v23 = a.rq_val[static_cast<uint64_t>(row + q) * 2u + 1u];
    }
    uint8_t *quad = block + static_cast<uint64_t>(1u + 3u * q) * slab_bytes;
    reinterpret_cast<uint4 *>(quad)[slot] = c;
    reinterpret_cast<double2 *>(quad + slab_bytes)[slot] = v01;
    reinterpret_cast<double2 *>(quad + 2u * slab_bytes)[slot] = v23;
  }
}

// The three per-spin arrays of the peel — priority, number of earlier neighbours still unvisited,
// visiting order — in HBM scratch (clusters of any size), or IN LDS (LDSA; a.lds_arrays): 4 + 1 + 2
// bytes per spin, counters packed four to a word (rows of at most 255 couplings), the order as
// 16-bit indices (K < 65 536).  With the arrays in HBM every level of the peel is a chain of L2
// round trips — read the level's spins, gather their neighbours' priorities, a device-scope
// atomic per later neighbour —, about 4 us per level and 0.97 s of device time for the 128-problem
// production batch (orders alone; the sweeps alone took 1.7 s): LDS atomics and gathers cut a
// level to a fraction of a microsecond.
// Where the peel's per-spin arrays live.  kPeelLdsCounters / kPeelLdsNibbles (large clusters): only the
// counters in LDS — a byte per spin, or FOUR BITS with an escape: a spin with 15 or more earlier
// neighbours keeps the nibble 15 for good and counts down in its exact HBM counter instead (whether
// a spin is such a one never changes, so a plain read of the nibble decides).
enum : int { kPeelHbm = 0, kPeelLds = 1, kPeelLdsCounters = 2, kPeelLdsNibbles = 3 };
template <int MODE>
struct PeelArrays {
  static constexpr bool kOrderInLds = MODE == kPeelLds, kCountersInLds = MODE != kPeelHbm;
  uint32_t *prio;
  uint32_t *indeg;  // kCountersInLds: bytes (kPeelLdsNibbles: nibbles) packed in words
  uint32_t *exact;  // kPeelLdsNibbles: the counters as words in HBM
  void *order;      // kOrderInLds: uint16_t
  __device__ __forceinline__ uint32_t order_at(uint32_t m) const {
    if constexpr (kOrderInLds) return static_cast<const uint16_t *>(order)[m];
    return static_cast<const uint32_t *>(order)[m];
  }
  __device__ __forceinline__ void append(uint32_t *tail, uint32_t i) const {
    const uint32_t at = atomicAdd(tail, 1u);
    if constexpr (kOrderInLds) {
      static_cast<uint16_t *>(order)[at] = static_cast<uint16_t>(i);
    } else {
      static_cast<uint32_t *>(order)[at] = i;
    }
  }
  // (called once per spin, on zeroed words in the LDS form)
  __device__ __forceinline__ void set_count(uint32_t i, uint32_t count) const {
    if constexpr (MODE == kPeelLds) {
      if (count) atomicAdd(indeg + (i >> 2), count << (8u * (i & 3u)));
    } else if constexpr (MODE == kPeelHbm) {
      // (relaxed device-scope store: the decrements are device-scope atomics)
      __hip_atomic_store(indeg + i, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (kPeelLdsCounters / kPeelLdsNibbles: loaded from k_order_counts' words, see the prologue)
  }
  // The same in two steps, so that the atomics of many neighbours are in flight together: the word
  // returned by the atomic (`peeked`: the neighbour's LDS word read beforehand, nibbles only), and
  // what it says.
  __device__ __forceinline__ uint32_t peek(uint32_t n) const {
    if constexpr (MODE == kPeelLdsNibbles) return indeg[n >> 3];
    return 0u;
  }
  __device__ __forceinline__ bool counts_in_hbm(uint32_t n, uint32_t peeked) const {
    return MODE == kPeelLdsNibbles && ((peeked >> (4u * (n & 7u))) & 15u) == 15u;
  }
  __device__ __forceinline__ uint32_t decrement(uint32_t n, uint32_t peeked) const {
    if constexpr (MODE == kPeelLdsNibbles) {
      if (counts_in_hbm(n, peeked)) return atomicSub(exact + n, 1u);
      return atomicSub(indeg + (n >> 3), 1u << (4u * (n & 7u)));
    } else if constexpr (kCountersInLds) {
      return atomicSub(indeg + (n >> 2), 1u << (8u * (n & 3u)));
    } else {
      return atomicSub(indeg + n, 1u);
    }
  }
  __device__ __forceinline__ bool was_last(uint32_t n, uint32_t peeked, uint32_t returned) const {
    if constexpr (MODE == kPeelLdsNibbles) {
      return counts_in_hbm(n, peeked) ? returned == 1u : ((returned >> (4u * (n & 7u))) & 15u) == 1u;
    } else if constexpr (kCountersInLds) {
      return ((returned >> (8u * (n & 3u))) & 0xFFu) == 1u;
    } else {
      return returned == 1u;
    }
  }
  // one earlier neighbour of spin n has been visited: true when it was the last one
  __device__ __forceinline__ bool visited_one(uint32_t n) const {
    if constexpr (MODE == kPeelLdsNibbles) {
      const uint32_t shift = 4u * (n & 7u);
      if (((indeg[n >> 3] >> shift) & 15u) == 15u) return atomicSub(exact + n, 1u) == 1u;
      return ((atomicSub(indeg + (n >> 3), 1u << shift) >> shift) & 15u) == 1u;
    } else if constexpr (kCountersInLds) {
      const uint32_t shift = 8u * (n & 3u);
      return ((atomicSub(indeg + (n >> 2), 1u << shift) >> shift) & 0xFFu) == 1u;
    } else {
      return atomicSub(indeg + n, 1u) == 1u;
    }
  }
};

// (`Args` is OrderArgs, or OrderArgs in the constant address space: the batched kernel reads its
// problem's descriptor from a table, like k_sa_sweep_batch does; `s` = sweep of the chunk)
#ifndef ASP_SHUF_PEEL_ROWS
#define ASP_SHUF_PEEL_ROWS 4
#endif
constexpr int kPeelRows = ASP_SHUF_PEEL_ROWS;  // rows in flight per group of lanes (kPeelLdsCounters)

template <int MODE, typename Args>
__device__ __forceinline__ void shuffled_orders_impl(const Args &a, const uint32_t s) {
  extern __shared__ __align__(16) uint8_t lds[];
  const uint32_t K = a.num_spins;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  // The problem's own number of threads: in a shared launch (k_shuffled_orders_batch) the workgroups
  // have the threads of the LARGEST problem; the wavefronts a small problem has no use for end here
  // (a barrier waits for the surviving wavefronts of its workgroup only).
  const uint32_t nthreads = a.threads;
  if (tid >= nthreads) return;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), waves = nthreads >> 6;
  const uint32_t t = a.first_sweep + s;
  // blocks of S spins: a wavefront handles 64 / S of them per pass, lane = (block of the pass, slot)
  const uint32_t S = 1u << a.log_s, per_pass = 64u >> a.log_s;
  const uint32_t slot = lane & (S - 1u), pass_block = lane >> a.log_s;
  // ctl: [0] tail of the order, [1..2] level ends (ping-pong), [3] carry of the scans
  uint32_t *ctl = reinterpret_cast<uint32_t *>(lds);
  uint32_t *wave_tot = ctl + 8;                         // 16
  uint32_t *level_start = wave_tot + 16;                // level_cap + 2
  uint32_t *level_block = level_start + a.level_cap + 2;  // level_cap + 2
  uint32_t *block_quads = level_block + a.level_cap + 2;  // block_cap + 1
  uint32_t *block_first = block_quads + a.block_cap + 1;  // block_cap + 1
  uint32_t *hist = block_first + a.block_cap + 1;         // waves * 64
  uint32_t *cursor = hist + waves * 64u;                  // waves * 64
  PeelArrays<MODE> peel;
  if constexpr (MODE == kPeelLds) {
    peel.prio = cursor + waves * 64u;            // K
    peel.indeg = peel.prio + K;                  // ceil(K / 4) words of four counters
    peel.order = peel.indeg + ((K + 3u) >> 2);   // K uint16_t
    for (uint32_t w = tid; w < ((K + 3u) >> 2); w += nthreads) peel.indeg[w] = 0u;
  } else if constexpr (MODE == kPeelLdsCounters || MODE == kPeelLdsNibbles) {
    // the counters take the place of the block tables, which nothing reads before the peel is over
    peel.prio = nullptr;
    peel.indeg = block_quads;                    // ceil(K / 4) words of four counters (K / 8: of eight)
    peel.exact = a.indeg + static_cast<uint64_t>(s) * K;
    peel.order = a.order + static_cast<uint64_t>(s) * K;
  } else {
    peel.prio = a.prio + static_cast<uint64_t>(s) * K;
    peel.indeg = a.indeg + static_cast<uint64_t>(s) * K;
    peel.order = a.order + static_cast<uint64_t>(s) * K;
  }
  uint32_t *prio = peel.prio;
  const uint32_t key0 = static_cast<uint32_t>(a.seed), key1 = static_cast<uint32_t>(a.seed >> 32);
#if ASP_SHUF_TIMING
  unsigned long long oticks[kOrderTimingSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long otick_last = __builtin_readcyclecounter();
#define ASP_OTICK(slot)                                             \
  do {                                                              \
    const unsigned long long now_ = __builtin_readcyclecounter();   \
    oticks[slot] += now_ - otick_last;                              \
    otick_last = now_;                                              \
  } while (0)
#else
#define ASP_OTICK(slot) do {} while (0)
#endif

  // G lanes share a row: one quad of four neighbours per lane and trip
  const uint32_t G = a.lanes_per_row;
  const uint32_t sub = tid & (G - 1u), gid = tid / G, groups = nthreads / G;
  uint32_t levels = 0, begin = 0, end = 0;
  if (a.finish_only) {
    // The wide path: k_order_level has peeled the first a.wide_levels levels as grids over the
    // chunk (the bulk of the spins: a sweep's levels shrink from K / 25 spins to a handful); the
    // first positions of those levels are in HBM.  Either one of them is K — the peel is complete
    // — or this workgroup peels the short rest itself, from where the last launch stopped: the
    // members of level wide_levels are in the pair's control words.
    // (lds_counters: NO level launches — the whole peel is this workgroup's, from level 0 as
    // k_order_counts left it)
    constexpr bool kOwnPeel = MODE == kPeelLdsCounters || MODE == kPeelLdsNibbles;
    const uint32_t W = kOwnPeel ? 0u : min(a.wide_levels, a.level_cap);
    if constexpr (kOwnPeel) {
      constexpr uint32_t kBits = MODE == kPeelLdsNibbles ? 4u : 8u, kPer = 32u / kBits, kMax = (1u << kBits) - 1u;
      const uint32_t *counts = a.indeg + static_cast<uint64_t>(s) * K;
      for (uint32_t w = tid; w < (K + kPer - 1u) / kPer; w += nthreads) {
        uint32_t packed = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPer; ++j) {
          const uint32_t i = kPer * w + j;
          if (i < K) packed |= min(counts[i], kMax) << (kBits * j);
        }
        peel.indeg[w] = packed;
      }
    }
    const uint32_t *from = a.level_start_g + static_cast<uint64_t>(s) * (a.level_cap + 2u);
    const uint32_t *pair = a.peel_ctl + static_cast<uint64_t>(s) * 8u;
    for (uint32_t l = tid; l <= W; l += nthreads) level_start[l] = from[l];
    if (tid == 0) ctl[4] = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t l = 1u + tid; l <= W; l += nthreads) {
      if (level_start[l] == K) atomicMin(&ctl[4], l);
    }
    __syncthreads();
    if (ctl[4] != 0xFFFFFFFFu) {
      levels = ctl[4];  // (begin == end: nothing left to peel)
    } else if (!kOwnPeel && a.wide_levels > a.level_cap) {
      // (a shared launch sequence longer than this problem's level table, and the peel is not
      // complete inside the table: more levels than its capacity — the host repeats the call)
      levels = a.level_cap + 1u;
    } else {
      levels = W;
      begin = level_start[W];
      end = begin + pair[W % 3u];
      if (tid == 0) {
        ctl[0] = end;
        level_start[W + 1u] = end;
      }
      __syncthreads();
    }
  } else {
    // ---- 1. priorities ----
    for (uint32_t i = tid; i < K; i += nthreads) {
      prio[i] = philox4x32_10(i, t, kPriorityCounter, 0u, key0, key1).w[0];
    }
    if (tid == 0) ctl[0] = 0;
    __syncthreads();
    ASP_OTICK(0);

    // ---- 2. number of earlier neighbours; spins without any open level 0 ----
    for (uint32_t i = gid; i < K; i += groups) {
      const uint32_t pi = prio[i];
      const uint32_t q1 = a.rq_ptr[i + 1];
      uint32_t count = 0;
      for (uint32_t q = a.rq_ptr[i] + sub; q < q1; q += G) {
        const uint4 c = a.rq_col[q];
        const uint32_t cs[4] = {c.x, c.y, c.z, c.w};
  #pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (cs[j] != i && comes_before(prio[cs[j]], cs[j], pi, i)) ++count;
        }
      }
      for (uint32_t step = G >> 1; step > 0; step >>= 1) count += __shfl_xor(count, step, 64);
      if (sub == 0) {
        peel.set_count(i, count);
        if (count == 0) peel.append(&ctl[0], i);
      }
    }
    __syncthreads();
    if (tid == 0) {
      ctl[1] = ctl[0];
      level_start[0] = 0;
      level_start[1] = ctl[0];
    }
    __syncthreads();

    end = ctl[1];
  }
  ASP_OTICK(1);
  // ---- 3. peel the levels ----
  while (begin < end) {
    ++levels;
    if constexpr (MODE == kPeelLdsCounters || MODE == kPeelLdsNibbles) {
      // A level is a chain of dependent HBM reads per row — its spin, its row pointers, its quads
      // and their later-masks — in front of LDS atomics: kPeelRows rows per group of lanes are in
      // flight at once (one workgroup has only its own loads to hide that latency with).
      constexpr int U = kPeelRows;
      const uint8_t *later = a.later + static_cast<uint64_t>(s) * a.num_quads;
      const uint32_t *order = static_cast<const uint32_t *>(peel.order);
      auto visit = [&](const uint4 c, const uint32_t mask) {
        const uint32_t cs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if ((mask >> j) & 1u) {
            if (peel.visited_one(cs[j])) peel.append(&ctl[0], cs[j]);
          }
        }
      };
      uint32_t *order_w = static_cast<uint32_t *>(peel.order);
      uint32_t *row_quads = a.prio + static_cast<uint64_t>(s) * K;  // (the priorities: of no use after the counts)
      // (the trip count is the SAME for every lane of the workgroup: the appends below are wave-wide)
      for (uint32_t base = begin; base < end; base += groups * U) {
        const uint32_t m0 = base + gid;
        // (every load below is UNCONDITIONAL — a lane without a row or a quad reads a valid address and
        // discards the value —: loads under a branch are waited for one by one)
        uint32_t row[U], q0[U], q1[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t m = m0 + static_cast<uint32_t>(u) * groups;
          live[u] = m < end;
          row[u] = order[live[u] ? m : begin];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          q0[u] = a.rq_ptr[row[u]];
          q1[u] = a.rq_ptr[row[u] + 1u];
        }
        // (the rows' lengths beside the order, for the sort of the levels: no gathers there)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (sub == 0 && live[u]) row_quads[m0 + static_cast<uint32_t>(u) * groups] = q1[u] - q0[u];
        }
        uint32_t cs[U][4], mask[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t q = q0[u] + sub;
          const bool have = live[u] && q < q1[u];
          const uint32_t at = have ? q : 0u;
          const uint4 c = a.rq_col[at];
          mask[u] = later[at];
          mask[u] = have ? mask[u] : 0u;
          cs[u][0] = c.x, cs[u][1] = c.y, cs[u][2] = c.z, cs[u][3] = c.w;
          if (!live[u]) q1[u] = 0u;  // (no remainder either)
        }
        // the decrements of the batch's 4 U entries per lane: issued together, read afterwards
        uint32_t peeked[U][4], returned[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int j = 0; j < 4; ++j) peeked[u][j] = peel.peek(cs[u][j]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            returned[u][j] = 0u;
            if ((mask[u] >> j) & 1u) returned[u][j] = peel.decrement(cs[u][j], peeked[u][j]);
          }
        }
        // the spins whose last earlier neighbour this was: ONE atomic on the order's tail per
        // wavefront and batch (ballots give every lane its place), not one per spin
        uint32_t total = 0;
        uint64_t pushing[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool hit = ((mask[u] >> j) & 1u) && peel.was_last(cs[u][j], peeked[u][j], returned[u][j]);
            pushing[u][j] = __ballot(hit);
            total += static_cast<uint32_t>(__builtin_popcountll(pushing[u][j]));
          }
        }
        if (total) {  // (uniform over the wavefront)
          uint32_t first = 0;
          if (lane == 0) first = atomicAdd(&ctl[0], total);
          first = __builtin_amdgcn_readfirstlane(first);
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint64_t p = pushing[u][j];
              if ((p >> lane) & 1ull) {
                const uint32_t before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(p >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(p), 0u));
                order_w[first + before] = cs[u][j];
              }
              first += static_cast<uint32_t>(__builtin_popcountll(p));
            }
          }
        }
        // (rows of more than G quads: a lane at a time)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          for (uint32_t q = q0[u] + sub + G; q < q1[u]; q += G) visit(a.rq_col[q], later[q]);
        }
      }
    } else
    for (uint32_t m = begin + gid; m < end; m += groups) {
      const uint32_t i = peel.order_at(m);
      const uint32_t pi = prio[i];
      const uint32_t q1 = a.rq_ptr[i + 1];
      for (uint32_t q = a.rq_ptr[i] + sub; q < q1; q += G) {
        const uint4 c = a.rq_col[q];
        const uint32_t cs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t n = cs[j];
          if (n != i && comes_before(pi, i, prio[n], n)) {
            if (peel.visited_one(n)) peel.append(&ctl[0], n);
          }
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      ctl[1 + (levels & 1u)] = ctl[0];
      if (levels + 1u <= a.level_cap + 1u) level_start[levels + 1u] = ctl[0];
    }
    __syncthreads();
    begin = end;
    end = ctl[1 + (levels & 1u)];
  }
  // (begin == K here: the priority order is a total order, so the peel reaches every spin)
  const uint32_t L = levels;
  bool bad = L > a.level_cap;

  ASP_OTICK(2);
  // ---- 4. blocks of the levels ----
  if (!bad) {
    for (uint32_t l = tid; l < L; l += nthreads) {
      level_block[l] = (level_start[l + 1] - level_start[l] + S - 1u) >> a.log_s;
    }
    __syncthreads();
  }
  uint32_t B = 0;
  if (!bad) {
    B = block_exclusive_scan(level_block, L, wave_tot, ctl + 3, nthreads);
    if (tid == 0) level_block[L] = B;
    __syncthreads();
    bad = B > a.block_cap;
  }
  if (bad) {
    if (tid == 0) {
      atomicMax(a.status + kStatLevels, L);
      atomicMax(a.status + kStatBlocks, B);
      atomicOr(a.status + kStatBad, 1u);
      a.num_levels[s] = 0;
    }
    return;
  }

  ASP_OTICK(3);
  // ---- 5. every level sorted by descending row length (counting sort, a wavefront per level) ----
  uint32_t *sop = a.spin_of_pos + (static_cast<uint64_t>(s) * a.block_cap << a.log_s);
  for (uint32_t base = 0; base < L; base += waves) {
    const uint32_t l = base + wave;
    uint32_t lo = 0, n = 0, pos0 = 0;
    if (l < L) {
      lo = level_start[l];
      n = level_start[l + 1] - lo;
      pos0 = level_block[l] << a.log_s;
    }
    hist[wave * 64u + lane] = 0;
    __syncthreads();
    for (uint32_t m = lane; m < n; m += 64u) {
      uint32_t quads;
      if constexpr (MODE == kPeelLdsCounters || MODE == kPeelLdsNibbles) {
        quads = a.prio[static_cast<uint64_t>(s) * K + lo + m];  // (written by the peel, position by position)
      } else {
        const uint32_t i = peel.order_at(lo + m);
        quads = a.rq_ptr[i + 1] - a.rq_ptr[i];
      }
      atomicAdd(&hist[wave * 64u + min(quads, kClassCap)], 1u);
    }
    __syncthreads();
    {
      const uint32_t own = hist[wave * 64u + lane];
      uint32_t suffix = own;  // members of this class and of the longer ones
#pragma unroll
      for (int step = 1; step < 64; step <<= 1) {
        const uint32_t o = __shfl_down(suffix, step, 64);
        if (lane + static_cast<uint32_t>(step) < 64u) suffix += o;
      }
      cursor[wave * 64u + lane] = suffix - own;
    }
    __syncthreads();
    for (uint32_t m = lane; m < n; m += 64u) {
      const uint32_t i = peel.order_at(lo + m);
      uint32_t quads;
      if constexpr (MODE == kPeelLdsCounters || MODE == kPeelLdsNibbles) {
        quads = a.prio[static_cast<uint64_t>(s) * K + lo + m];
      } else {
        quads = a.rq_ptr[i + 1] - a.rq_ptr[i];
      }
      sop[pos0 + atomicAdd(&cursor[wave * 64u + min(quads, kClassCap)], 1u)] = i;
    }
    for (uint32_t m = n + lane; m < ((n + S - 1u) & ~(S - 1u)); m += 64u) sop[pos0 + m] = kDummySpin;
    __syncthreads();
  }

  ASP_OTICK(4);
  // ---- 6. block widths and their prefix sum ----
  for (uint32_t b0 = wave * per_pass; b0 < B; b0 += waves * per_pass) {
    const uint32_t b = b0 + pass_block;
    const uint32_t i = b < B ? sop[(b << a.log_s) + slot] : kDummySpin;
    uint32_t w = i == kDummySpin ? 0u : a.rq_ptr[i + 1] - a.rq_ptr[i];
#pragma unroll
    for (int step = 1; step < 64; step <<= 1) {  // maximum over the S lanes of the block
      const uint32_t o = static_cast<uint32_t>(__shfl_xor(w, step, 64));
      if (static_cast<uint32_t>(step) < S) w = max(w, o);
    }
    if (slot == 0 && b < B) {
      block_quads[b] = w;
      block_first[b] = 1u + 3u * w;  // slabs of the block: header + quads
    }
  }
  __syncthreads();
  const uint32_t slabs = block_exclusive_scan(block_first, B, wave_tot, ctl + 3, nthreads);
  const uint32_t Q = (slabs - B) / 3u;
  if (Q > a.quad_cap) {
    if (tid == 0) {
      atomicMax(a.status + kStatQuads, Q);
      atomicOr(a.status + kStatBad, 1u);
      a.num_levels[s] = 0;
    }
    return;
  }

  ASP_OTICK(5);
  // ---- 7. the sweep's coupling stream (the wide path writes it from a grid of its own) ----
  if (!a.finish_only) {
    uint8_t *stream = a.stream + static_cast<uint64_t>(s) * a.stream_kib * 1024u;
    for (uint32_t b0 = wave * per_pass; b0 < B; b0 += waves * per_pass) {
      const uint32_t b = b0 + pass_block;
      const bool live = b < B;
      write_stream_pass(a, stream, sop, b, live, live ? block_first[b] : 0u, live ? block_quads[b] : 0u, lane);
    }
  }

  ASP_OTICK(6);
  // ---- 8. the sweep's tables ----
  uint32_t *out_lb = a.level_block + static_cast<uint64_t>(s) * (a.level_cap + 1u);
  for (uint32_t l = tid; l <= L; l += nthreads) out_lb[l] = level_block[l];
  uint2 *out_meta = a.block_meta + static_cast<uint64_t>(s) * a.block_cap;
  for (uint32_t b = tid; b < B; b += nthreads) out_meta[b] = make_uint2(block_first[b], block_quads[b]);
  if (tid == 0) {
    a.num_levels[s] = L;
    atomicMax(a.status + kStatLevels, L);
    atomicMax(a.status + kStatBlocks, B);
    atomicMax(a.status + kStatQuads, Q);
  }
  ASP_OTICK(7);
#if ASP_SHUF_TIMING
  if (tid == 0) {
    unsigned long long *out = reinterpret_cast<unsigned long long *>(a.status + kStatWords) + kTimingSlots * kTimingWaves;
    for (uint32_t k = 0; k < kOrderTimingSlots; ++k) atomicAdd(out + k, oticks[k]);
  }
#endif
}

// (WIDE: the per-sweep workgroup of a cluster of the wide path — a kernel of its own: its batched
// peel holds 81 registers per lane where the others need 54, and at 1024 threads per workgroup that
// is the difference between one and two workgroups per compute unit for the small clusters)
template <bool WIDE, typename Args>
__device__ __forceinline__ void shuffled_orders_body(const Args &a, const uint32_t s) {
  if constexpr (WIDE) {
    if (a.lds_counters == 1u) {
      shuffled_orders_impl<kPeelLdsCounters>(a, s);
    } else if (a.lds_counters == 2u) {
      shuffled_orders_impl<kPeelLdsNibbles>(a, s);
    } else {
      shuffled_orders_impl<kPeelHbm>(a, s);
    }
  } else {
    if (a.lds_arrays) {
      shuffled_orders_impl<kPeelLds>(a, s);
    } else {
      shuffled_orders_impl<kPeelHbm>(a, s);
    }
  }
}

// ---------------------------------------------------------------------------
// The wide path of the order build (clusters whose per-spin arrays do not fit the LDS).
// One workgroup per sweep is a chain of dependent HBM accesses per row — read the level's spin,
// its row pointers, its columns, the neighbours' priorities, a device-scope atomic — with a
// handful of rows in flight: 13.5 ms per chunk for the order-2 models of the kagome_36 pipeline
// (1.5e5 .. 3e5 spins), 70 % of its annealing time, beside sweep kernels of 6-8 ms
// (profiles/r04_pipeline_trace.txt).  Here every phase without a dependence between rows is a
// grid over ALL (problem, sweep) pairs of the chunk — priorities, counts, the stream —, and the
// peel is LEVEL-SYNCHRONOUS ACROSS THE GRID: launch l visits level l of every pair (the launch
// boundary is the barrier), rows by the thousand in flight.  Per pair: cnt[l % 3] members of
// level l, start[l % 3] its first position in the order (launch l reads slot l % 3, appends
// through slot (l + 1) % 3 and clears slot (l + 2) % 3 for launch l + 1).
// ---------------------------------------------------------------------------
constexpr uint32_t kWideThreads = 256;

template <typename Args>
__device__ __forceinline__ void order_prio_body(const Args &a, uint32_t s, uint32_t part, uint32_t parts) {
  if (!a.finish_only) return;  // (a problem of the fused path in a shared launch)
  const uint32_t K = a.num_spins, t = a.first_sweep + s;
  const uint32_t key0 = static_cast<uint32_t>(a.seed), key1 = static_cast<uint32_t>(a.seed >> 32);
  uint32_t *prio = a.prio + static_cast<uint64_t>(s) * K;
  for (uint32_t i = part * blockDim.x + threadIdx.x; i < K; i += parts * blockDim.x) {
    prio[i] = philox4x32_10(i, t, kPriorityCounter, 0u, key0, key1).w[0];
  }
  if (part == 0) {  // the pair's control words and level table start from zero
    uint32_t *ctl = a.peel_ctl + static_cast<uint64_t>(s) * 8u;
    if (threadIdx.x < 8u) ctl[threadIdx.x] = 0u;
    uint32_t *starts = a.level_start_g + static_cast<uint64_t>(s) * (a.level_cap + 2u);
    for (uint32_t l = threadIdx.x; l < a.level_cap + 2u; l += blockDim.x) starts[l] = 0u;
  }
}

template <typename Args>
__device__ __forceinline__ void order_counts_body(const Args &a, uint32_t s, uint32_t part, uint32_t parts) {
  if (!a.finish_only) return;
  const uint32_t K = a.num_spins;
  const uint32_t G = a.lanes_per_row, tid = threadIdx.x;
  const uint32_t sub = tid & (G - 1u), gid = tid / G, groups = blockDim.x / G;
  const uint32_t *prio = a.prio + static_cast<uint64_t>(s) * K;
  uint32_t *indeg = a.indeg + static_cast<uint64_t>(s) * K;
  uint32_t *order = a.order + static_cast<uint64_t>(s) * K;
  uint32_t *ctl = a.peel_ctl + static_cast<uint64_t>(s) * 8u;
  // (trip counts are uniform over the G lanes of a row; the shuffles below stay inside them)
  // later[q]: which entries of quad q are neighbours that come AFTER the row's spin — the peel then
  // needs no priorities (four random gathers per quad and a link of its dependence chain less)
  uint8_t *later = a.later + static_cast<uint64_t>(s) * a.num_quads;
  for (uint32_t i = part * groups + gid; i < K; i += parts * groups) {
    const uint32_t pi = prio[i];
    const uint32_t q1 = a.rq_ptr[i + 1];
    uint32_t count = 0;
    for (uint32_t q = a.rq_ptr[i] + sub; q < q1; q += G) {
      const uint4 c = a.rq_col[q];
      const uint32_t cs[4] = {c.x, c.y, c.z, c.w};
      uint32_t mask = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (cs[j] != i) {
          if (comes_before(prio[cs[j]], cs[j], pi, i)) {
            ++count;
          } else {
            mask |= 1u << j;
          }
        }
      }
      later[q] = static_cast<uint8_t>(mask);
    }
    for (uint32_t step = G >> 1; step > 0; step >>= 1) count += __shfl_xor(count, step, 64);
    if (sub == 0) {
      indeg[i] = count;
      if (count == 0) order[atomicAdd(&ctl[0], 1u)] = i;  // level 0
    }
  }
}

// `push`: this lane appends `value` to the list at list[base + ...] whose length is *tail — ONE atomic
// per wavefront instruction (the lanes that push are counted by a ballot), not one per lane: a level
// of thousands of spins otherwise serialises on its tail counter.
__device__ __forceinline__ void wave_append(bool push, uint32_t value, uint32_t *list, uint32_t base,
                                            uint32_t *tail) {
  const uint64_t pushing = __ballot(push);
  if (pushing == 0ull) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t leader = static_cast<uint32_t>(__builtin_ctzll(pushing));
  uint32_t first = 0;
  if (lane == leader) first = atomicAdd(tail, static_cast<uint32_t>(__builtin_popcountll(pushing)));
  first = __shfl(first, static_cast<int>(leader), 64);
  if (push) list[base + first + static_cast<uint32_t>(__builtin_popcountll(pushing & ((1ull << lane) - 1ull)))] = value;
}

template <typename Args>
__device__ __forceinline__ void order_level_body(const Args &a, uint32_t s, uint32_t part, uint32_t parts,
                                                 uint32_t l) {
  if (!a.finish_only || a.lds_counters) return;  // (fused path / the peel is the per-sweep workgroup's)
  uint32_t *ctl = a.peel_ctl + static_cast<uint64_t>(s) * 8u;
  const uint32_t lo = ctl[3u + l % 3u], n = ctl[l % 3u];  // (written by earlier launches: stable here)
  if (n == 0) return;                                     // the pair has no level l: peeled already
  const uint32_t K = a.num_spins;
  const uint32_t G = a.lanes_per_row, tid = threadIdx.x;
  const uint32_t sub = tid & (G - 1u), gid = tid / G, groups = blockDim.x / G;
  uint32_t *indeg = a.indeg + static_cast<uint64_t>(s) * K;
  uint32_t *order = a.order + static_cast<uint64_t>(s) * K;
  uint32_t *next = ctl + (l + 1u) % 3u;
  const uint32_t base = lo + n;  // first position of level l + 1
  if (part == 0 && tid == 0) {
    ctl[3u + (l + 1u) % 3u] = base;
    ctl[(l + 2u) % 3u] = 0u;  // (level l - 1's count: read by launch l - 1 only)
    if (l + 1u <= a.level_cap + 1u) a.level_start_g[static_cast<uint64_t>(s) * (a.level_cap + 2u) + l + 1u] = base;
  }
  const uint8_t *later = a.later + static_cast<uint64_t>(s) * a.num_quads;
  // (every lane of the wavefront runs the same number of trips — lanes without a row or a quad
  // take part in the ballots of wave_append with nothing to push)
  const uint32_t trips = (n + parts * groups - 1u) / (parts * groups);
  const uint32_t quad_trips = (a.max_quads + G - 1u) / G;
  for (uint32_t trip = 0; trip < trips; ++trip) {
    const uint32_t m = (trip * parts + part) * groups + gid;
    const bool have_row = m < n;
    const uint32_t i = have_row ? order[lo + m] : 0u;
    const uint32_t q0 = have_row ? a.rq_ptr[i] : 0u, q1 = have_row ? a.rq_ptr[i + 1] : 0u;
    // (rows are at most max_quads long; the loop bound is uniform, most rows end earlier)
    uint32_t longest = q1 - q0;
#pragma unroll
    for (int step = 1; step < 64; step <<= 1) longest = max(longest, static_cast<uint32_t>(__shfl_xor(longest, step, 64)));
    const uint32_t my_trips = min(quad_trips, (longest + G - 1u) / G);
    for (uint32_t qt = 0; qt < my_trips; ++qt) {
      const uint32_t q = q0 + qt * G + sub;
      const bool have_quad = q < q1;
      uint4 c = make_uint4(0u, 0u, 0u, 0u);
      uint32_t mask = 0;
      if (have_quad) {
        c = a.rq_col[q];
        mask = later[q];
      }
      const uint32_t cs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool is_later = (mask >> j) & 1u;
        const bool last = is_later && atomicSub(indeg + cs[j], 1u) == 1u;
        wave_append(last, cs[j], order, base, next);
      }
    }
  }
}

template <typename Args>
__device__ __forceinline__ void order_stream_body(const Args &a, uint32_t s, uint32_t part, uint32_t parts) {
  if (!a.finish_only) return;
  const uint32_t L = a.num_levels[s];
  if (L == 0) return;  // (the sweep ran out of room: flagged, the host repeats the call)
  const uint32_t B = a.level_block[static_cast<uint64_t>(s) * (a.level_cap + 1u) + L];
  const uint2 *meta = a.block_meta + static_cast<uint64_t>(s) * a.block_cap;
  const uint32_t *sop = a.spin_of_pos + (static_cast<uint64_t>(s) * a.block_cap << a.log_s);
  uint8_t *stream = a.stream + static_cast<uint64_t>(s) * a.stream_kib * 1024u;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
  const uint32_t per_pass = 64u >> a.log_s, pass_block = lane >> a.log_s;
  for (uint32_t b0 = (part * waves + wave) * per_pass; b0 < B; b0 += parts * waves * per_pass) {
    const uint32_t b = b0 + pass_block;
    const bool live = b < B;
    const uint2 info = live ? meta[b] : make_uint2(0u, 0u);
    write_stream_pass(a, stream, sop, b, live, info.x, info.y, lane);
  }
}

// Launch shapes: blockIdx -> (problem, sweep of the chunk, part of the pair's work); `problems` of a
// single call is one descriptor (the same kernels serve asp_sa_anneal_shuffled and the batch).
// XCD-LOCAL PAIRS: workgroups are dealt round-robin over the chip's eight XCDs (blocks b and b + 8
// share one), each with an L2 of its own, and the peel of a pair is atomics on the pair's counters
// and list tails: with the parts of a pair in consecutive blocks, every counter line travelled
// between eight L2s.  The parts of one pair are therefore blocks of EQUAL index mod 8 — the same in
// every launch of the chain, so the priorities, counters, masks and lists of a pair stay in one L2
// from launch to launch.  (A placement for speed only: the atomics are device-scope.)
struct WideGrid {
  uint32_t count, parts;  // sweeps of the chunk, workgroups per (problem, sweep)
  uint32_t pairs;         // problems x count
};
__host__ __device__ inline uint32_t wide_grid_blocks(uint32_t pairs, uint32_t parts) {
  return ((pairs + 7u) & ~7u) * parts;
}
template <typename F>
__device__ __forceinline__ void wide_dispatch(const OrderArgs *problems, WideGrid g, F body) {
  using ConstArgs = const OrderArgs __attribute__((address_space(4)));
  const uint32_t xcd = blockIdx.x & 7u, y = blockIdx.x >> 3;
  const uint32_t pair = (y / g.parts) * 8u + xcd, part = y % g.parts;
  if (pair >= g.pairs) return;
  const uint32_t problem = pair / g.count;
  ConstArgs *a = reinterpret_cast<ConstArgs *>(reinterpret_cast<uintptr_t>(problems + problem));
  body(*a, pair - problem * g.count, part);
}
__global__ __launch_bounds__(kWideThreads) void k_order_prio(const OrderArgs *problems, WideGrid g) {
  wide_dispatch(problems, g, [&](const auto &a, uint32_t s, uint32_t part) { order_prio_body(a, s, part, g.parts); });
}
__global__ __launch_bounds__(kWideThreads) void k_order_counts(const OrderArgs *problems, WideGrid g) {
  wide_dispatch(problems, g, [&](const auto &a, uint32_t s, uint32_t part) { order_counts_body(a, s, part, g.parts); });
}
__global__ __launch_bounds__(kWideThreads) void k_order_level(const OrderArgs *problems, WideGrid g, uint32_t level) {
  wide_dispatch(problems, g,
                [&](const auto &a, uint32_t s, uint32_t part) { order_level_body(a, s, part, g.parts, level); });
}
__global__ __launch_bounds__(kWideThreads) void k_order_stream(const OrderArgs *problems, WideGrid g) {
  wide_dispatch(problems, g, [&](const auto &a, uint32_t s, uint32_t part) { order_stream_body(a, s, part, g.parts); });
}

template <bool WIDE>
__global__ __launch_bounds__(kOrderThreads) void k_shuffled_orders(OrderArgs a) {
  shuffled_orders_body<WIDE>(a, blockIdx.x);
}

// Many problems, `count` sweeps of each: workgroup -> (problem, sweep).  WIDE: the launch is for the
// problems of the wide path only, else for the others only — two launches, each with the LDS its own
// problems need (the finish workgroup of a large cluster holds its counters).
template <bool WIDE>
__global__ __launch_bounds__(kOrderThreads) void k_shuffled_orders_batch(const OrderArgs *problems, uint32_t count) {
  using ConstArgs = const OrderArgs __attribute__((address_space(4)));
  const uint32_t problem = blockIdx.x / count;
  ConstArgs *a = reinterpret_cast<ConstArgs *>(reinterpret_cast<uintptr_t>(problems + problem));
  if ((a->finish_only != 0u) != WIDE) return;
  shuffled_orders_body<WIDE>(*a, blockIdx.x - problem * count);
}

// ---------------------------------------------------------------------------
// Sweep kernel
// ---------------------------------------------------------------------------

struct ShuffledArgs {
  // the chunk's visiting orders (k_shuffled_orders)
  const uint32_t *level_block;
  const uint32_t *num_levels;
  const uint2 *block_meta;
  const uint8_t *stream;  // the coupling streams of the chunk's sweeps (k_shuffled_orders)
  const uint32_t *status;
  const double *betas;  // all sweeps of the call
  const uint64_t *x0;   // packed original-order start configuration or nullptr
  uint8_t *state;       // [groups][K] bit m = chain m of the group is -1 (between chunks)
  uint64_t *best;       // [groups * M][words] best configurations, packed original order (bit = +1)
  long long *e_cur, *e_best;     // [groups * M] tracked energies (fixed point)
  unsigned long long *accepted;  // [groups * M]
  uint64_t seed;
  double scale;
  uint32_t num_spins, words, level_cap, block_cap, stream_kib;
  uint32_t first_sweep, chunk_sweeps, replica_first, initialise;
  // lane packing (kernels with PK = true): a block is S = 1 << log_s spins and a wavefront visits
  // it for G = 64 / S groups of M chains at once, lane = (group, spin); a workgroup holds the spins
  // of its G groups.  groups_total: groups of the whole call (the last workgroup may have fewer).
  uint32_t log_s, groups_total;
  uint32_t waves;  // wavefronts (per team) of this problem's workgroups
};

// Spins stay in LDS in ORIGINAL order, in one of four layouts: a 32-bit word per spin (kWide: byte
// m = 0x80 * chain m is -1; up to four chains), a byte per spin (kBytes: bit m; up to eight), and —
// for clusters beyond the capacity of bytes — four bits per spin (kNibbles: up to four chains,
// ~2.5e5 spins) or one bit per spin (kBits: one chain, ~6e5 spins).  In the packed layouts several
// spins share a word, so a flip is an LDS atomic XOR and a neighbour read is a byte load plus a
// variable shift; the bits above chain m's are ignored by the sign instructions.
template <int LAYOUT>
inline constexpr bool kPackedLayout = LAYOUT == kNibbles || LAYOUT == kBits;
template <int LAYOUT>
inline constexpr uint32_t kSpinBits = LAYOUT == kNibbles ? 4u : (LAYOUT == kBits ? 1u : 8u);

// Negative-spin mask of the M chains (bit m) <-> the LDS representation of one spin.
template <int LAYOUT>
__device__ __forceinline__ uint32_t to_lds(uint32_t mask) {
  return LAYOUT == kWide ? spread_mask(mask) : mask;
}
// The LDS representation of spin i.
template <int LAYOUT>
__device__ __forceinline__ uint32_t read_spin(const uint8_t *spins, uint32_t i) {
  if constexpr (LAYOUT == kGlobal) {
    // a 32-bit word per spin in HBM (one chain): written by other wavefronts of the workgroup in
    // earlier levels — read at device scope, past the compute unit's vector L1
    return __hip_atomic_load(reinterpret_cast<const uint32_t *>(spins) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if constexpr (LAYOUT == kWide) {
    return reinterpret_cast<const uint32_t *>(spins)[i];
  } else if constexpr (kPackedLayout<LAYOUT>) {
    constexpr uint32_t B = kSpinBits<LAYOUT>;
    return (reinterpret_cast<const uint32_t *>(spins)[(i * B) >> 5] >> ((i * B) & 31u)) & ((1u << B) - 1u);
  } else {
    return spins[i];
  }
}
template <int LAYOUT>
__device__ __forceinline__ uint32_t from_lds(uint32_t v) {
  if constexpr (LAYOUT == kWide) {
    return ((v >> 7) & 1u) | ((v >> 14) & 2u) | ((v >> 21) & 4u) | ((v >> 28) & 8u);
  } else {
    return v;
  }
}

// The chains of `mask` (workgroup-uniform) improved on their best energies: their configurations, packed
// in original order, to a.best.  A wavefront packs a ROW of up to 64 consecutive words — word k of the
// row is a ballot over 64 spins, kept by lane k — and stores the row at once: a store per ballot from
// one lane was 13 % of the sweep kernel on a 150 000-spin cluster (2 344 words x 4 chains, every sweep
// of the first half of a ladder improves).
template <int M, int LAYOUT, typename Args>
__device__ __forceinline__ void snapshot_original(const uint8_t *spins, const Args &a,
                                                  uint32_t group, uint32_t mask, uint32_t nthreads) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), waves = nthreads >> 6;
  const uint32_t row = min(64u, max(1u, (a.words + waves - 1u) / waves));
  for (uint32_t w0 = wave * row; w0 < a.words; w0 += waves * row) {
    const uint32_t count = min(row, a.words - w0);  // (uniform over the wavefront)
    uint32_t lo[M], hi[M];
#pragma unroll
    for (int m = 0; m < M; ++m) lo[m] = hi[m] = 0u;
    // (four words at a time, their LDS reads in flight together; words past the row's end read
    // nothing and are kept by no lane that stores)
    for (uint32_t k0 = 0; k0 < count; k0 += 4u) {
      uint32_t neg[4];
#pragma unroll
      for (uint32_t j = 0; j < 4u; ++j) {
        const uint32_t i = (w0 + k0 + j) * 64u + lane;
        neg[j] = (1u << M) - 1u;
        if (i < a.num_spins) neg[j] = from_lds<LAYOUT>(read_spin<LAYOUT>(spins, i));
      }
#pragma unroll
      for (uint32_t j = 0; j < 4u; ++j) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          if (!((mask >> m) & 1u)) continue;  // workgroup-uniform
          const uint64_t word = __ballot(!((neg[j] >> m) & 1u));
          if (lane == k0 + j) {
            lo[m] = static_cast<uint32_t>(word);
            hi[m] = static_cast<uint32_t>(word >> 32);
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (!((mask >> m) & 1u)) continue;
      if (lane < count) {
        a.best[(static_cast<uint64_t>(group) * M + m) * a.words + w0 + lane] =
            static_cast<uint64_t>(lo[m]) | (static_cast<uint64_t>(hi[m]) << 32);
      }
    }
  }
}

// ---- the row sums of one quad, in two pinned stages ----
// This kernel runs ONE or two wavefronts per SIMD (the levels of a sweep are short), so what
// the instruction stream of a single wavefront looks like matters: with the register budget of
// <= 2 wavefronts per SIMD hipcc hoists the sign instructions of later terms over the FMAs of
// earlier ones and pays a register copy per term and chain (the multiplier's high word is
// rewritten in place).  Signs and FMAs are therefore volatile asm in source order, and the LDS
// gather of a quad is a stage of its own, issued one quad ahead of its use.
// One quad of the sweep's ELL through BUFFER loads: the address is (resource: the sweep's
// array, uniform) + (scalar offset: the quad) + (lane * 16), so a load costs no vector
// address arithmetic — with global loads hipcc spends a 64-bit VALU add per load, and this kernel
// issues up to 36 of them per block visit on its critical path.
using BufferRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufferRsrc make_rsrc(const void *base) {
  // raw buffer, no range checking to speak of (4 GiB), gfx9 DATA_FORMAT word
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xFFFFFFFF, 0x00020000);
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
// Four consecutive couplings of one lane as the three loads deliver them (native vectors: a
// bit cast, no component shuffling — a shuffle would have to wait for the load)
struct HeldQuad {
  u32x4 c;
  f64x2 v01, v23;
};
// `at` = byte offset of the quad in the sweep's stream (scalar); lane16 = lane * 16
__device__ __forceinline__ void load_quad_buffer(HeldQuad &q, BufferRsrc stream, uint32_t at, uint32_t lane16) {
  q.c = __builtin_amdgcn_raw_buffer_load_b128(stream, lane16, at, 0);
  q.v01 = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(stream, lane16 + 1024u, at, 0));
  q.v23 = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(stream, lane16 + 2048u, at, 0));
}
// Blocks of S < 64 spins (lane packing): the three slabs of a quad are `slab` = 16 S bytes apart —
// scalar offsets, one vector offset (slot * 16); the groups of a wavefront read the same bytes
__device__ __forceinline__ void load_quad_slabs(HeldQuad &q, BufferRsrc stream, uint32_t at, uint32_t lane16,
                                                uint32_t slab) {
  q.c = __builtin_amdgcn_raw_buffer_load_b128(stream, lane16, at, 0);
  q.v01 = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(stream, lane16, at + slab, 0));
  q.v23 = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(stream, lane16, at + 2u * slab, 0));
}

// (`gbase`, lane packing only: LDS byte address of the spins of the lane's group; `hbm`, kGlobal only:
// the chain's spin words in HBM)
template <int LAYOUT, bool PK = false>
__device__ __forceinline__ void gather_quad(const HeldQuad &q, uint32_t (&s)[4], uint32_t gbase = 0,
                                            const uint8_t *hbm = nullptr) {
  const uint32_t cs[4] = {q.c.x, q.c.y, q.c.z, q.c.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // columns are LDS byte addresses (sa_device.hpp: the spins start at LDS address 0)
    if constexpr (LAYOUT == kGlobal) {  // (columns: byte offsets of the spins' words)
      s[j] = __hip_atomic_load(reinterpret_cast<const uint32_t *>(hbm + cs[j]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (LAYOUT == kWide && PK) {
      s[j] = *reinterpret_cast<LdsWord *>(static_cast<uintptr_t>(cs[j] + gbase));
    } else if constexpr (LAYOUT == kWide) {
      s[j] = *reinterpret_cast<LdsWord *>(static_cast<uintptr_t>(cs[j]));
    } else if constexpr (LAYOUT == kNibbles) {  // (packed layouts: the column is the spin's index)
      s[j] = static_cast<uint32_t>(*reinterpret_cast<LdsByte *>(static_cast<uintptr_t>(cs[j] >> 1))) >>
             ((cs[j] & 1u) << 2);
    } else if constexpr (LAYOUT == kBits) {
      s[j] = static_cast<uint32_t>(*reinterpret_cast<LdsByte *>(static_cast<uintptr_t>(cs[j] >> 3))) >>
             (cs[j] & 7u);
    } else {
      s[j] = *reinterpret_cast<LdsByte *>(static_cast<uintptr_t>(cs[j]));
    }
  }
}

// high word of +-1.0 from byte m of a wide spin word, in place (see wide_factor)
#define ASP_SDWA_SIGN(sel)                                                                         \
  asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD " \
               "src1_sel:" sel : "+v"(hi) : "v"(0x3Fu), "v"(word))
__device__ __forceinline__ double pinned_wide_factor(uint32_t word, int m, uint32_t &hi) {
  switch (m) {
    case 0: ASP_SDWA_SIGN("BYTE_0"); break;
    case 1: ASP_SDWA_SIGN("BYTE_1"); break;
    case 2: ASP_SDWA_SIGN("BYTE_2"); break;
    default: ASP_SDWA_SIGN("BYTE_3"); break;
  }
  return __hiloint2double(static_cast<int>(hi), 0);
}
#undef ASP_SDWA_SIGN

template <int M, int LAYOUT>
__device__ __forceinline__ void apply_quad(const HeldQuad &q, const uint32_t (&s)[4], double (&acc)[M],
                                           uint32_t (&one_hi)[4]) {
  const double vs[4] = {q.v01.x, q.v01.y, q.v23.x, q.v23.y};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // four chains at a time: their signs, then their FMAs (no FMA directly behind the
    // instruction that builds its multiplier); each acc[m] receives its terms in ascending k
#pragma unroll
    for (int m0 = 0; m0 < M; m0 += 4) {
      double f[4];
#pragma unroll
      for (int m = m0; m < M && m < m0 + 4; ++m) {
        if constexpr (LAYOUT == kWide) {
          f[m - m0] = pinned_wide_factor(s[j], m, one_hi[m & 3]);
        } else {
          uint32_t hi;
          if (m == 0) {
            asm volatile("v_lshl_or_b32 %0, %1, 31, %2" : "=v"(hi) : "v"(s[j]), "s"(0x3FF00000u));
          } else {
            // bit m to bit 0 with a RIGHT shift (a plain VOP2), then the m = 0 instruction
            uint32_t down;
            asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(down) : "v"(static_cast<uint32_t>(m)), "v"(s[j]));
            asm volatile("v_lshl_or_b32 %0, %1, 31, %2" : "=v"(hi) : "v"(down), "s"(0x3FF00000u));
          }
          f[m - m0] = __hiloint2double(static_cast<int>(hi), 0);
        }
      }
#pragma unroll
      for (int m = m0; m < M && m < m0 + 4; ++m) {
        asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[m]) : "v"(vs[j]), "v"(f[m - m0]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Timing-only ablations (results are WRONG when set; never set in the product build):
// 1 no row sums, 2 no accept phase, 3 no coupling loads, 4 no level barriers, 5 a cheap hash for
// Philox, 6 a cheap compare for the exp filter
#ifndef ASP_SHUF_ABL
#define ASP_SHUF_ABL 0
#endif

// Development aid (-DASP_SHUF_TIMING=1): every wavefront of workgroup 0 adds up the shader-clock
// cycles it spends in the row sums, the request, the accept phase, the level barriers and the
// per-sweep bookkeeping; the totals land behind the status words (tools/time_shuffled.py --timing).
#ifndef ASP_SHUF_TIMING
#define ASP_SHUF_TIMING 0
#endif
#ifndef ASP_SHUF_ABLATE_ENV
#define ASP_SHUF_ABLATE_ENV 0  // 1: honour $ASP_SHUFFLED_ABLATE (timing-only launches that skip work: WRONG results)
#endif
#if ASP_SHUF_TIMING
#define ASP_TICK(slot)                                  \
  do {                                                  \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    ticks[slot] += now_ - tick_last;                    \
    tick_last = now_;                                   \
  } while (0)
#else
#define ASP_TICK(slot) do {} while (0)
#endif

constexpr uint32_t kNoBlock = 0xFFFFFFFFu;
#ifndef ASP_SHUF_HELD_QUADS
#define ASP_SHUF_HELD_QUADS 12
#endif
// quads of a block kept in registers (wider blocks stream the rest).  Scanned in round 4 on the
// production batch and on K = 1e4 x 1024 chains (-DASP_SHUF_HELD_QUADS=6/8/10/12): fewer held quads
// free registers (10: the order kernel fits beside two sweep wavefronts per SIMD; 6: three sweep
// wavefronts per SIMD) but every step down lengthens a visit, 22.3 / 25.1 / 27.8 / 31.6 ms for
// 12 / 10 / 8 / 6 on the single cluster.
constexpr int kHeldQuads = ASP_SHUF_HELD_QUADS;
#ifndef ASP_SHUF_FIRST_QUADS
#define ASP_SHUF_FIRST_QUADS 6
#endif
// quads of the next block requested BEFORE the accept phase (the rest after it: what is requested
// late must land during the level barrier, so as much as the miss queue takes goes out early)
constexpr int kFirstQuads = ASP_SHUF_FIRST_QUADS < kHeldQuads ? ASP_SHUF_FIRST_QUADS : kHeldQuads;

// The loads of a block do not depend on the spins, so a wavefront fetches its NEXT block — of this
// level or of the next one — while it finishes the current one: the quads of the next block are
// requested as soon as the k-loop has consumed the registers that held the current block's, and
// arrive during the accept phase and the level barrier.  Per level the critical path is then
// LDS gathers + FMAs + accept, not a chain of global-memory round trips.
//
// TEAMS = 2 (asp_sa_set_shuffled_teams; never chosen automatically): the wavefronts of a workgroup
// form two teams; both visit the same blocks at the same time, team t for chains t M .. t M + M - 1
// of the group (its half of every LDS spin word or byte) — two busy wavefronts per SIMD and
// level on half the instructions per visit.  Built to test whether a visit is bound by the issue
// rate of its lone wavefront; it is not (the fill rate from L2 binds: DESIGN.md §4.9), and this
// form is slower.  Kept because it is cheap and its parity is tested.
// PK = true (lane packing, wide layout only): a block of the order kernel is S = 1 << log_s < 64
// spins and a wavefront visits it for G = 64 / S groups of M chains at once — lane = (group,
// spin) —, so that a level of a handful of spins (clusters of 1e2 .. 3e3 spins: 3 .. 40 spins
// per level) fills its wavefront with CHAINS instead of padding lanes.  The workgroup holds the
// spins of its G groups (group g at LDS byte address g K 4); the G groups of a wavefront read
// the same bytes of the coupling stream (one request per address), a neighbour's LDS address is
// column + base of the lane's group (one VALU add per gather), the energy bookkeeping is
// reduced over the S lanes of a group.  Everything else — arithmetic, random words, the order
// of the terms — is the lane = spin kernel's; every chain is bit for bit the same.
template <int M, int LAYOUT, int TEAMS, bool PK, typename Args>
__device__ __forceinline__ void shuffled_sweep_body(const Args &a, const uint32_t wg) {
  constexpr bool WIDE = LAYOUT == kWide;
  constexpr int MT = M * TEAMS;  // chains of a group
  constexpr bool PACKED = kPackedLayout<LAYOUT>;
  // kGlobal: a 32-bit word per spin in HBM (one chain per workgroup) — clusters beyond the LDS even
  // at a bit per spin (~6e5 spins): every neighbour gather is an L2 access, the slow path that makes
  // the DEFAULT visiting order accept a cluster of any size, as the colour order does
  constexpr bool GLOBAL = LAYOUT == kGlobal;
  static_assert(LAYOUT == kWide || LAYOUT == kBytes || PACKED || GLOBAL,
                "spins are LDS words, bytes, nibbles or bits, or words in HBM");
  static_assert(!GLOBAL || (M == 1 && TEAMS == 1 && !PK), "spins in HBM: one chain per workgroup");
  static_assert(!WIDE || MT <= 4, "the wide layout holds up to four chains");
  static_assert(!PACKED || (TEAMS == 1 && M <= static_cast<int>(kSpinBits<LAYOUT>)),
                "a packed layout holds as many chains as it has bits per spin, in one team");
  static_assert(MT <= 8 && (TEAMS == 1 || TEAMS == 2), "a byte holds eight chains");
  static_assert(!PK || (WIDE && TEAMS == 1), "lane packing: wide layout, one team");
  constexpr uint32_t CH = PK ? 64u : 8u;  // chains of a workgroup the bookkeeping arrays hold
  extern __shared__ __align__(16) uint8_t lds[];
  if (a.status[kStatBad] != 0u) return;  // an order kernel ran out of room: the host repeats the call
#if ASP_ABS_LDS
  if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t *)lds) != 0) {
    __builtin_trap();
  }
#endif
  const uint32_t K = a.num_spins;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  // The problem's own wavefronts: in a shared launch (k_sa_sweep_shuffled_batch) the workgroups have
  // the wavefronts of the problem that wants the most; the ones this problem has no use for end
  // here (a barrier waits for the surviving wavefronts of its workgroup only).
  const uint32_t nthreads = a.waves * 64u * TEAMS;
  if (tid >= nthreads) return;
  const uint32_t log_s = PK ? a.log_s : 6u;
  const uint32_t G = 64u >> log_s;                         // groups of the workgroup
  const uint32_t slot = lane & ((1u << log_s) - 1u);       // the lane's spin slot inside a block
  const uint32_t lg = PK ? lane >> log_s : 0u;             // the lane's group inside the workgroup
  const uint32_t group = wg * G + lg;                      // ... and inside the call
  const bool group_live = !PK || group < a.groups_total;   // (the last workgroup may have idle groups)
  const uint32_t gbase = PK ? lg * K * 4u : 0u;            // LDS byte address of the group's spins
  // original order: K words (byte m = 0x80 * chain m is -1) or K bytes, per group; kGlobal: the
  // chain's words in a.state (there a word per spin), which also carries them between chunks
  uint8_t *spins = GLOBAL ? a.state + static_cast<uint64_t>(wg) * K * 4u : lds;
  uint32_t *wide = reinterpret_cast<uint32_t *>(lds);
  const uint32_t P = GLOBAL ? 0u
                            : ((WIDE ? G * K * 4u : (PACKED ? (K * kSpinBits<LAYOUT> + 7u) / 8u : K)) + 15u) & ~15u;
  long long *delta = reinterpret_cast<long long *>(lds + P);  // [CH] energy change of the running sweep
  long long *book = delta + CH;  // [c] current tracked energy, [CH + c] best, [2 CH + c] accepted flips
  uint32_t *improved_flag = reinterpret_cast<uint32_t *>(book + 3 * CH);  // [2] bit c: chain c improved
  // the running sweep's tables, copied from HBM once per sweep (they were written by another
  // kernel, possibly on another XCD: a scalar load of one entry is a full memory round trip,
  // and the level loop would pay two of them per level)
  uint2 *meta = reinterpret_cast<uint2 *>(improved_flag + 4);             // [block_cap]
  uint32_t *level_block = reinterpret_cast<uint32_t *>(meta + a.block_cap);  // [level_cap + 2]
  // (`wave` / `waves`: this wavefront's index inside its team and the team's size)
  const uint32_t waves = a.waves;
  const uint32_t team = TEAMS == 1 ? 0u : __builtin_amdgcn_readfirstlane(tid >> 6) / waves;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6) - team * waves;
#ifdef ASP_SHUF_PRIO
  // experiment: the wavefront with the widest blocks of every level (wave 0: the levels are sorted
  // by row length) issues first where it shares a SIMD with another workgroup's wavefront
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
  } else if (wave == 1) {
    __builtin_amdgcn_s_setprio(2);
  } else if (wave == 2) {
    __builtin_amdgcn_s_setprio(1);
  }
#endif
  const uint32_t r0 = a.replica_first + group * MT;  // first chain of the (lane's) group
  const uint32_t c0 = team * M;                      // first chain of this wavefront's team
  // this team's chains inside a spin's LDS word (byte per chain) or byte (bit per chain)
  const uint32_t team_shift = team * (WIDE ? 8u * M : static_cast<uint32_t>(M));
  const uint32_t key0 = static_cast<uint32_t>(a.seed), key1 = static_cast<uint32_t>(a.seed >> 32);
  const uint64_t first_chain = static_cast<uint64_t>(wg) * G * MT;  // of the workgroup, in the call's arrays
  const uint64_t chains_total = PK ? static_cast<uint64_t>(a.groups_total) * MT : ~0ull;

  // ---- chain state: fresh, or where the previous chunk left it ----
  if constexpr (PACKED) {  // spins of different threads share a word: OR them into zeroed words
    for (uint32_t w = tid; w < P / 4u; w += nthreads) wide[w] = 0u;
    __syncthreads();
  }
  for (uint32_t g = 0; g < G; ++g) {
    const uint32_t gg = wg * G + g;
    const bool live = !PK || gg < a.groups_total;  // (workgroup-uniform)
    const uint8_t *state_g = a.state + static_cast<uint64_t>(gg) * K;
    const uint32_t r0g = a.replica_first + gg * MT;
    if (GLOBAL && !a.initialise) break;  // (the words are where the previous chunk left them)
    for (uint32_t i = tid; i < K; i += nthreads) {
      uint32_t mask;
      if (!live) {
        mask = 0;
      } else if (!a.initialise) {
        mask = state_g[i];
      } else if (a.x0 != nullptr) {
        mask = ((a.x0[i >> 6] >> (i & 63u)) & 1ull) ? 0u : ((1u << MT) - 1u);
      } else {
        mask = 0;
        Philox4 rnd{};
        uint32_t have = 0xFFFFFFFFu;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const uint32_t r = r0g + m;
          if (m == 0 || (r >> 2) != have) {
            have = r >> 2;
            rnd = philox4x32_10(i, 0xFFFFFFFFu, have, 0u, key0, key1);
          }
          mask |= ((pick_word(rnd, r & 3u) & 1u) ^ 1u) << m;  // bit 0 of the word: 1 -> s = +1
        }
      }
      if constexpr (GLOBAL) {
        __hip_atomic_store(reinterpret_cast<uint32_t *>(spins) + i, mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if constexpr (WIDE) {
        wide[g * K + i] = spread_mask(mask);
      } else if constexpr (PACKED) {
        constexpr uint32_t B = kSpinBits<LAYOUT>;
        if (mask) atomicOr(wide + ((i * B) >> 5), mask << ((i * B) & 31u));
      } else {
        spins[i] = static_cast<uint8_t>(mask);
      }
    }
  }
  for (uint32_t x = tid; x < 4u * CH; x += nthreads) {  // delta[CH] | book[3 CH]
    long long v = 0;
    const uint32_t c = x & (CH - 1u), which = x / CH;
    const uint64_t at = first_chain + c;
    if (!a.initialise && which >= 1u && c < G * MT && at < chains_total) {
      v = which == 1u ? a.e_cur[at] : (which == 2u ? a.e_best[at] : static_cast<long long>(a.accepted[at]));
    }
    delta[x] = v;
  }
  if (tid < 2) improved_flag[tid] = 0;
  if constexpr (GLOBAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stores above, before the barrier)
  __syncthreads();
  if (a.initialise) {
    for (uint32_t g = 0; g < G; ++g) {
      if (PK && wg * G + g >= a.groups_total) break;
      snapshot_original<MT, LAYOUT>(spins + (PK ? g * K * 4u : 0u), a, wg * G + g, (1u << MT) - 1u, nthreads);
    }
    __syncthreads();
  }

  uint32_t one_hi[4] = {0x3FF00000u, 0x3FF00000u, 0x3FF00000u, 0x3FF00000u};
#if ASP_SHUF_TIMING
  unsigned long long ticks[kTimingSlots] = {0, 0, 0, 0, 0, 0};
  unsigned long long tick_last = __builtin_readcyclecounter();
#endif
  // byte offsets of a lane inside the three slabs of a quad (lane packing: slabs of 16 S bytes)
  const uint32_t lane16 = slot * 16u;
  const uint32_t slab_bytes = 16u << log_s;
  for (uint32_t tt = 0; tt < a.chunk_sweeps; ++tt) {
    const uint32_t t = a.first_sweep + tt;
    const double beta = a.betas[t];
    const uint32_t levels = a.num_levels[tt];
    {
      const uint32_t *g_level_block = a.level_block + static_cast<uint64_t>(tt) * (a.level_cap + 1u);
      const uint2 *g_meta = a.block_meta + static_cast<uint64_t>(tt) * a.block_cap;
      for (uint32_t l = tid; l <= levels; l += nthreads) level_block[l] = g_level_block[l];
      const uint32_t blocks = g_level_block[levels];
      for (uint32_t b = tid; b < blocks; b += nthreads) meta[b] = g_meta[b];
      __syncthreads();  // (the previous sweep's last use of the tables is behind its final barriers)
    }
    const BufferRsrc stream = make_rsrc(a.stream + static_cast<uint64_t>(tt) * a.stream_kib * 1024u);
    long long q_acc[M];
    uint32_t n_acc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      q_acc[m] = 0;
      n_acc[m] = 0;
    }
    // The block this wavefront holds in registers (requested one visit ahead): kNoBlock = none,
    // then held_quads = 0 and every lane is a padding lane.
    uint32_t held_quads = 0, held_spin = kDummySpin, held_first = 0;
    double held_h = 0.0;
    HeldQuad hq[kHeldQuads];
    // Requests block `nb`: what does not depend on other loads, in two instalments — a compute
    // unit keeps only so many cache misses in flight (a burst of 36 KiB per wavefront stalls the
    // issuing wavefront until the queue drains), so the header and the first half of the quads go
    // out after the row sums, the second half after the accept phase.  Every register set is
    // (re)defined by every request — those beyond the block's width with "any value" — so that
    // nothing of the previous block stays live across the visit; ONE call site each inside the
    // loops, so that the registers of the quads are not duplicated.
    auto load_quad_at = [&](HeldQuad &q, uint32_t j) {  // quad j of the held block
      if constexpr (PK) {
        load_quad_slabs(q, stream, held_first + (1u + 3u * j) * slab_bytes, lane16, slab_bytes);
      } else {
        load_quad_buffer(q, stream, held_first + 1024u + j * 3072u, lane16);
      }
    };
    auto request_quads = [&](int lo, int hi) {
#pragma unroll
      for (int j = lo; j < hi; ++j) {
        if (ASP_SHUF_ABL != 3 && static_cast<uint32_t>(j) < held_quads) {
          load_quad_at(hq[j], static_cast<uint32_t>(j));
        } else {
          hq[j].c = __builtin_nondeterministic_value(hq[j].c);
          hq[j].v01 = __builtin_nondeterministic_value(hq[j].v01);
          hq[j].v23 = __builtin_nondeterministic_value(hq[j].v23);
        }
      }
    };
    auto request_first = [&](uint32_t nb) {
      if (nb == kNoBlock) {
        held_quads = 0;
        held_spin = kDummySpin;
      } else {
        const uint2 info = meta[nb];
        held_quads = __builtin_amdgcn_readfirstlane(info.y);
        held_first = __builtin_amdgcn_readfirstlane(info.x) << (4u + log_s);  // byte offset of the block
        // the header slab: {spin of the lane, -, field of that spin} in one 16-byte load
        const u32x4 head = __builtin_amdgcn_raw_buffer_load_b128(stream, lane16, held_first, 0);
        held_spin = head.x;
        held_h = __hiloint2double(static_cast<int>(head.w), static_cast<int>(head.z));
      }
      request_quads(0, kFirstQuads);
    };
    auto request_rest = [&]() { request_quads(kFirstQuads, kHeldQuads); };
    uint32_t lb_begin = level_block[0];
    uint32_t lb_end = level_block[levels ? 1u : 0u];
    {
      const uint32_t first = lb_begin + wave;
      request_first(first < lb_end ? first : kNoBlock);
      request_rest();
    }
    for (uint32_t l = 0; l < levels; ++l) {
      // the level after this one: its blocks are lb_end .. lb_after
      const uint32_t lb_after = level_block[l + 2u <= levels ? l + 2u : levels];
      uint32_t b = lb_begin + wave;
      ASP_TICK(5);
      for (;;) {
        const bool busy = b < lb_end;  // (wave-uniform) the held block is block b of this level
        // this wavefront's next block: the next round of this level, else its block of the next level
        uint32_t nb = b + waves;
        bool same_level = true;
        if (!busy || nb >= lb_end) {
          same_level = false;
          nb = lb_end + wave;
          if (nb >= lb_after) nb = kNoBlock;
        }
        const uint32_t spin = held_spin;
        const bool valid = spin != kDummySpin && group_live;
        const uint32_t me = valid ? spin : 0u;
        const double h = held_h;
        double acc[M];
#pragma unroll
        for (int m = 0; m < M; ++m) acc[m] = 0.0;
        if (busy && ASP_SHUF_ABL != 1) {
          const uint32_t quads = held_quads;
          // the row sums in the order k = 0, 1, 2, ... of the row (= ascending column: the
          // oracle's); the LDS gather of quad j + 1 is issued before the FMAs of quad j (past the
          // block's last quad it reads whatever the registers held: harmless, never applied)
          uint32_t sa[4], sb[4];
          gather_quad<LAYOUT, PK>(hq[0], sa, gbase, spins);
#pragma unroll
          for (int j = 0; j < kHeldQuads; ++j) {
            if (static_cast<uint32_t>(j) < quads) {
              if (j + 1 < kHeldQuads) gather_quad<LAYOUT, PK>(hq[j + 1], (j & 1) ? sa : sb, gbase, spins);
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (TEAMS > 1) {  // this team's chains to the low end (a plain VOP2 shift)
#pragma unroll
                for (int k = 0; k < 4; ++k) ((j & 1) ? sb : sa)[k] >>= team_shift;
              }
              apply_quad<M, LAYOUT>(hq[j], (j & 1) ? sb : sa, acc, one_hi);
            }
          }
          if (quads > static_cast<uint32_t>(kHeldQuads)) {
            // a block wider than the registers hold (rows of more than 4 kHeldQuads couplings):
            // the rest streams with one quad in flight
            HeldQuad qa, qb;
            load_quad_at(qa, kHeldQuads);
            for (uint32_t j = kHeldQuads; j < quads; ++j) {
              // (one quad past the block at the end: never used)
              load_quad_at(qb, j + 1u);
              __builtin_amdgcn_sched_barrier(0);
              gather_quad<LAYOUT, PK>(qa, sa, gbase, spins);
              if constexpr (TEAMS > 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sa[k] >>= team_shift;
              }
              apply_quad<M, LAYOUT>(qa, sa, acc, one_hi);
              qa = qb;
            }
          }
        }
        ASP_TICK(0);
        // the registers are free: the next block's loads fly during the accept phase and the barrier
        // (a wavefront without a block in this level holds nothing and requests its block of the
        // next level, if it has one there)
        __builtin_amdgcn_sched_barrier(0);
        request_first(nb);
        __builtin_amdgcn_sched_barrier(0);
        ASP_TICK(1);
        if (busy && ASP_SHUF_ABL != 2) {
          // (this team's chains only: the partner team owns the other half of the word / byte)
          const uint32_t own = PK ? wide[(gbase >> 2) + me] : read_spin<LAYOUT>(spins, me) >> team_shift;
          bool need = false;  // some proposal of this lane needs a random number
          double de[M];
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const double g = __dadd_rn(acc[m], h);
            const bool negative = (own >> (WIDE ? 8 * m + 7 : m)) & 1u;  // s = -1
            de[m] = __dmul_rn(negative ? 2.0 : -2.0, g);
            // dE <= 0 is accepted and beta * dE >= 23 rejected whatever the draw
            need = need || (valid && !(de[m] <= 0.0) && !(__dmul_rn(beta, de[m]) >= 23.0));
          }
          uint32_t flip = 0;
          if (__ballot(need) != 0ull) {
            Philox4 rnd{};
            uint32_t have = 0xFFFFFFFFu;
#pragma unroll
            for (int m = 0; m < M; ++m) {
              const uint32_t r = r0 + c0 + m;
              if (m == 0 || (r >> 2) != have) {
                have = r >> 2;
#if ASP_SHUF_ABL == 5
                rnd = Philox4{{spin * 2654435761u ^ t, spin ^ (t * 40503u), spin + have, t ^ key0}};
#else
                rnd = philox4x32_10(spin, t, have, 0u, key0, key1);
#endif
              }
#if ASP_SHUF_ABL == 6
              const bool accept = valid && (de[m] <= 0.0 || pick_word(rnd, r & 3u) <
                                                                static_cast<uint32_t>(__dmul_rn(beta, de[m])));
#else
              const bool accept = valid && (de[m] <= 0.0 || metropolis_accept_word(pick_word(rnd, r & 3u),
                                                                                   __dmul_rn(beta, de[m])));
#endif
              flip |= (accept ? 1u : 0u) << m;
            }
          } else {
#pragma unroll
            for (int m = 0; m < M; ++m) flip |= ((valid && de[m] <= 0.0) ? 1u : 0u) << m;
          }
#pragma unroll
          for (int m = 0; m < M; ++m) {
            if ((flip >> m) & 1u) {
              // rint(dE * 2^S): |dE * 2^S| < 2^51 by the plan's S
              q_acc[m] += __double_as_longlong(__dadd_rn(__dmul_rn(de[m], a.scale), 0x1.8p52)) -
                          0x4338000000000000ll;
              n_acc[m] += 1;
            }
          }
          if (flip) {  // no neighbour of this spin is in the level: nobody reads it before the barrier
            if constexpr (WIDE && TEAMS == 2) {
              // this team's half of the word: the partner wavefront may be writing the other half
              if constexpr (M == 2) {
                reinterpret_cast<uint16_t *>(wide + me)[team] = static_cast<uint16_t>(own ^ spread_mask(flip));
              } else {
                reinterpret_cast<uint8_t *>(wide + me)[team] = static_cast<uint8_t>(own ^ spread_mask(flip));
              }
            } else if constexpr (GLOBAL) {
              __hip_atomic_store(reinterpret_cast<uint32_t *>(spins) + me, own ^ flip, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
            } else if constexpr (WIDE) {
              wide[(gbase >> 2) + me] = own ^ spread_mask(flip);
            } else if constexpr (PACKED) {
              // other lanes own the other spins of the word and may flip in the same instruction
              constexpr uint32_t B = kSpinBits<LAYOUT>;
              atomicXor(wide + ((me * B) >> 5), flip << ((me * B) & 31u));
            } else if constexpr (TEAMS == 2) {
              // bits of two teams in one byte: an LDS atomic keeps the partner's flips
              atomicXor(reinterpret_cast<uint32_t *>(spins) + (me >> 2), (flip << team_shift) << (8u * (me & 3u)));
            } else {
              spins[me] = static_cast<uint8_t>(own ^ flip);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        request_rest();
        __builtin_amdgcn_sched_barrier(0);
        ASP_TICK(2);
        if (!same_level) break;
        b = nb;
      }
#if ASP_SHUF_ABL != 4
      // (spins in HBM: the flips must have arrived before another wavefront gathers them)
      if constexpr (GLOBAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#endif
      ASP_TICK(3);
      lb_begin = lb_end;
      lb_end = lb_after;
    }

    // ---- exact (integer) reduction of the sweep's energy change ----
    // (lane packing: over the S lanes of the lane's group; else over the wavefront)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      long long v = q_acc[m], n = static_cast<long long>(n_acc[m]);
#pragma unroll
      for (int step = 1; step < 64; step <<= 1) {
        const long long ov = __shfl_xor(v, step, 64), on = __shfl_xor(n, step, 64);
        if (!PK || static_cast<uint32_t>(step) < (1u << log_s)) {
          v += ov;
          n += on;
        }
      }
      if (slot == 0 && n != 0) {
        const uint32_t c = (PK ? lg * M : c0) + m;
        atomicAdd(reinterpret_cast<unsigned long long *>(&delta[c]), static_cast<unsigned long long>(v));
        atomicAdd(reinterpret_cast<unsigned long long *>(&book[2 * CH + c]), static_cast<unsigned long long>(n));
      }
    }
    __syncthreads();
    if (tid < G * MT) {
      const long long e = book[tid] + delta[tid];
      book[tid] = e;
      delta[tid] = 0;
      if (e < book[CH + tid]) {
        book[CH + tid] = e;
        atomicOr(improved_flag + (tid >> 5), 1u << (tid & 31u));
      }
    }
    __syncthreads();
    const uint64_t improved = static_cast<uint64_t>(improved_flag[0]) | (static_cast<uint64_t>(improved_flag[1]) << 32);
    if (improved) {
      for (uint32_t g = 0; g < G; ++g) {
        const uint32_t mask = static_cast<uint32_t>(improved >> (g * MT)) & ((1u << MT) - 1u);
        if (mask) snapshot_original<MT, LAYOUT>(spins + (PK ? g * K * 4u : 0u), a, wg * G + g, mask, nthreads);
      }
    }
    __syncthreads();
    if (tid < 2) improved_flag[tid] = 0;  // next write to it is two barriers away
    ASP_TICK(4);
  }
#if ASP_SHUF_TIMING
  if (wg == 0 && lane == 0 && wave < kTimingWaves) {
    unsigned long long *out = reinterpret_cast<unsigned long long *>(const_cast<uint32_t *>(a.status) + kStatWords) +
                              wave * kTimingSlots;
    for (uint32_t k = 0; k < kTimingSlots; ++k) atomicAdd(out + k, ticks[k]);
  }
#endif

  for (uint32_t g = 0; g < G; ++g) {
    if (GLOBAL) break;  // (the spin words ARE the state)
    const uint32_t gg = wg * G + g;
    if (PK && gg >= a.groups_total) break;
    uint8_t *state_g = a.state + static_cast<uint64_t>(gg) * K;
    const uint8_t *spins_g = spins + (PK ? g * K * 4u : 0u);
    for (uint32_t i = tid; i < K; i += nthreads) {
      state_g[i] = static_cast<uint8_t>(from_lds<LAYOUT>(read_spin<LAYOUT>(spins_g, i)));
    }
  }
  if (tid < G * MT && first_chain + tid < chains_total) {
    const uint64_t at = first_chain + tid;
    a.e_cur[at] = book[tid];
    a.e_best[at] = book[CH + tid];
    a.accepted[at] = static_cast<unsigned long long>(book[2 * CH + tid]);
  }
}

using ShuffledKernel = void (*)(ShuffledArgs);

template <int M, int LAYOUT, int TEAMS, bool PK = false>
__global__ __launch_bounds__(512) void k_sa_sweep_shuffled(ShuffledArgs a) {
  shuffled_sweep_body<M, LAYOUT, TEAMS, PK>(a, blockIdx.x);
}

// Many problems in one launch: workgroup -> (problem, workgroup of the problem) through a slot
// table, the problem's arguments through a descriptor table (csrc/sa_sweep.hip: k_sa_sweep_batch).
struct ShuffledSlot {
  uint32_t problem, group;
};
template <int M, int LAYOUT, bool PK = false>
__global__ __launch_bounds__(512) void k_sa_sweep_shuffled_batch(const ShuffledArgs *problems,
                                                                 const ShuffledSlot *slots) {
  using ConstArgs = const ShuffledArgs __attribute__((address_space(4)));
  const ShuffledSlot slot = slots[blockIdx.x];
  if (slot.problem == 0xFFFFFFFFu) return;  // (padding of the XCD-aware slot table)
  ConstArgs *a = reinterpret_cast<ConstArgs *>(
      reinterpret_cast<uintptr_t>(problems + __builtin_amdgcn_readfirstlane(slot.problem)));
  shuffled_sweep_body<M, LAYOUT, 1, PK>(*a, __builtin_amdgcn_readfirstlane(slot.group));
}

// m = chains per group; teams = 2: two teams of m / 2 chains (wide: m = 4 or 2; bytes: m = 8 or 4);
// packed_lanes: blocks of fewer than 64 spins, several groups per wavefront (wide layout, one team)
ShuffledKernel shuffled_kernel_for(int m, int layout, int teams = 1, bool packed_lanes = false) {
  if (packed_lanes) {
    if (layout != kWide || teams != 1) return nullptr;
    switch (m) {
      case 1: return k_sa_sweep_shuffled<1, kWide, 1, true>;
      case 2: return k_sa_sweep_shuffled<2, kWide, 1, true>;
      case 4: return k_sa_sweep_shuffled<4, kWide, 1, true>;
      default: return nullptr;
    }
  }
  if (teams == 2) {
    if (layout == kWide) {
      switch (m) {
        case 2: return k_sa_sweep_shuffled<1, kWide, 2>;
        case 4: return k_sa_sweep_shuffled<2, kWide, 2>;
        default: return nullptr;
      }
    }
    if (layout != kBytes) return nullptr;
    switch (m) {
      case 4: return k_sa_sweep_shuffled<2, kBytes, 2>;
      case 8: return k_sa_sweep_shuffled<4, kBytes, 2>;
      default: return nullptr;
    }
  }
  if (layout == kWide) {
    switch (m) {
      case 1: return k_sa_sweep_shuffled<1, kWide, 1>;
      case 2: return k_sa_sweep_shuffled<2, kWide, 1>;
      case 4: return k_sa_sweep_shuffled<4, kWide, 1>;
      default: return nullptr;
    }
  }
  if (layout == kNibbles) {
    switch (m) {
      case 1: return k_sa_sweep_shuffled<1, kNibbles, 1>;
      case 2: return k_sa_sweep_shuffled<2, kNibbles, 1>;
      case 4: return k_sa_sweep_shuffled<4, kNibbles, 1>;
      default: return nullptr;
    }
  }
  if (layout == kBits) return m == 1 ? k_sa_sweep_shuffled<1, kBits, 1> : nullptr;
  if (layout == kGlobal) return m == 1 ? k_sa_sweep_shuffled<1, kGlobal, 1> : nullptr;
  switch (m) {
    case 1: return k_sa_sweep_shuffled<1, kBytes, 1>;
    case 2: return k_sa_sweep_shuffled<2, kBytes, 1>;
    case 4: return k_sa_sweep_shuffled<4, kBytes, 1>;
    case 8: return k_sa_sweep_shuffled<8, kBytes, 1>;
    default: return nullptr;
  }
}

// spins (of `groups` groups: lane packing) | delta[CH] book[3 CH] | flags (16 B) | meta[block_cap] |
// level_block[level_cap + 2]; CH = 8 chains, 64 with lane packing (groups > 0)
size_t sweep_lds_bytes(uint64_t K, int layout, uint32_t level_cap, uint32_t block_cap, uint32_t groups = 0) {
  uint64_t spin_bytes = layout == kGlobal ? 0 : (layout == kWide ? K * 4 : (layout == kNibbles ? (K + 1) / 2 : (layout == kBits ? (K + 7) / 8 : K)));
  if (groups) spin_bytes *= groups;
  return ((spin_bytes + 15) & ~size_t{15}) + (groups ? 256 : 32) * sizeof(long long) + 16 +
         static_cast<size_t>(block_cap) * sizeof(uint2) + (static_cast<size_t>(level_cap) + 2) * sizeof(uint32_t);
}

// The layout of a run with m chains per workgroup: the fastest that fits (-1: none does).
int shuffled_layout_for(uint64_t K, int m, uint32_t level_cap, uint32_t block_cap, size_t max_lds) {
  if (m <= 4 && sweep_lds_bytes(K, kWide, level_cap, block_cap) <= max_lds) return kWide;
  if (sweep_lds_bytes(K, kBytes, level_cap, block_cap) <= max_lds) return kBytes;
  if (m <= 4 && sweep_lds_bytes(K, kNibbles, level_cap, block_cap) <= max_lds) return kNibbles;
  if (m == 1 && sweep_lds_bytes(K, kBits, level_cap, block_cap) <= max_lds) return kBits;
  // beyond a bit per spin in LDS: the spins of a chain as words in HBM (one chain per workgroup)
  if (m == 1 && sweep_lds_bytes(K, kGlobal, level_cap, block_cap) <= max_lds) return kGlobal;
  return -1;
}

// (`arrays_of`: K when the peel's per-spin arrays live in LDS — kPeelLds —, else 0; `counter_words`:
// words of the counters when only they do — kPeelLdsCounters / kPeelLdsNibbles: they lie over the
// block tables)
size_t order_lds_bytes(uint32_t level_cap, uint32_t block_cap, uint32_t waves, uint64_t arrays_of = 0,
                       uint64_t counter_words = 0) {
  const size_t tables = 2 * (static_cast<size_t>(block_cap) + 1) + 2 * static_cast<size_t>(waves) * 64;
  return sizeof(uint32_t) * (8 + 16 + 2 * (static_cast<size_t>(level_cap) + 2) +
                             std::max<size_t>(tables, counter_words) +
                             arrays_of + (arrays_of + 3) / 4 + (arrays_of + 1) / 2);
}

uint32_t next_pow2(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Static device data of the shuffled sweep: uploaded once per plan.
int ensure_static(asp_sa_plan *p) {
  if (p->rq_ptr.ptr) return ASP_OK;
  const asp::SaHostLayout &L = p->host;
  asp::RowQuads rows;
  ASP_TRY(asp::build_row_quads(L, &rows));
  std::vector<double> field(L.num_spins);
  for (uint64_t i = 0; i < L.num_spins; ++i) field[i] = L.field_pos[L.pos_of_spin[i]];
  hipStream_t s = p->stream;
  ASP_TRY(asp::upload_vector(p->rq_col, rows.col, s));
  ASP_TRY(asp::upload_vector(p->rq_val, rows.val, s));
  ASP_TRY(asp::upload_vector(p->field_dev, field, s));
  ASP_TRY(asp::upload_vector(p->rq_ptr, rows.quad_ptr, s));
  ASP_HIP_TRY(hipStreamSynchronize(s));  // the host vectors die with this scope
  p->rq_quads = rows.quad_ptr.back();
  p->rq_max_quads = rows.max_quads;
  return ASP_OK;
}

// Level launches of the wide path per chunk, given the levels a sweep is expected to have: the first
// three quarters of them.  A sweep's levels shrink roughly linearly from ~2.5 x the mean to a handful
// of spins, so those launches peel ~94 % of the spins with thousands of rows in flight each, and the
// per-sweep workgroup (k_shuffled_orders, finish_only) peels the short rest itself, however many levels
// that is: no launch count has to be guessed right.  (Scanned on the kagome_36 pipeline, 64 clusters:
// 20 / 30 / 45 launches and 0.6 of the expected levels 29.8 / 29.0 / 24.6 / 25.3 s; one launch per
// level of the capacity trimmed after the first chunk, with a retry when a sweep overran it, 24.1 s:
// a level launch costs ~0.2 ms beside the sweep kernels, the tail of a sweep in ONE workgroup about
// as much per level.)
uint32_t wide_level_launches(double expected_levels, uint32_t level_cap) {
  uint32_t launches = static_cast<uint32_t>(std::ceil(0.75 * expected_levels));
  if (const char *env = std::getenv("ASP_SHUFFLED_WIDE_LEVELS")) {  // tests, measurements
    launches = static_cast<uint32_t>(std::strtoul(env, nullptr, 10));
  }
  return std::max(1u, std::min(launches, level_cap));
}

// The wide launches of one chunk in front of the per-sweep workgroups (k_shuffled_orders with
// finish_only): priorities, counts, one launch per level.  `problems`: device descriptors of the
// chunk (P of them), `now` sweeps; parts: workgroups per (problem, sweep).
struct WidePartsMax {
  uint32_t prio = 1, counts = 1, level = 1, stream = 1, levels = 0;
};
int launch_wide_front(hipStream_t os, const OrderArgs *problems, uint32_t P, uint32_t now, const WidePartsMax &w) {
  const uint32_t pairs = P * now;
  hipLaunchKernelGGL(k_order_prio, dim3(wide_grid_blocks(pairs, w.prio)), dim3(kWideThreads), 0, os, problems,
                     WideGrid{now, w.prio, pairs});
  ASP_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_order_counts, dim3(wide_grid_blocks(pairs, w.counts)), dim3(kWideThreads), 0, os, problems,
                     WideGrid{now, w.counts, pairs});
  ASP_HIP_TRY(hipGetLastError());
  for (uint32_t l = 0; l < w.levels; ++l) {
    hipLaunchKernelGGL(k_order_level, dim3(wide_grid_blocks(pairs, w.level)), dim3(kWideThreads), 0, os, problems,
                       WideGrid{now, w.level, pairs}, l);
  }
  ASP_HIP_TRY(hipGetLastError());
  return ASP_OK;
}
int launch_wide_stream(hipStream_t os, const OrderArgs *problems, uint32_t P, uint32_t now, const WidePartsMax &w) {
  const uint32_t pairs = P * now;
  hipLaunchKernelGGL(k_order_stream, dim3(wide_grid_blocks(pairs, w.stream)), dim3(kWideThreads), 0, os, problems,
                     WideGrid{now, w.stream, pairs});
  ASP_HIP_TRY(hipGetLastError());
  return ASP_OK;
}

// ---------------------------------------------------------------------------
// One call = one ShuffledRun: set-up, then attempts (enqueue everything, collect the status
// words; an order kernel that ran out of room makes the run grow its capacities and try again
// from the chains' initial configurations), then the energies and the copies to the caller.
// asp_sa_anneal_shuffled drives one run; the batched entry point enqueues the attempts of many
// runs — each on its plan's own stream pair — before it waits for any of them, so a batch of
// small clusters overlaps on the device.
// ---------------------------------------------------------------------------
struct ShuffledRun {
  asp_sa_plan *p = nullptr;
  uint64_t seed = 0;
  const double *betas = nullptr;
  uint32_t num_sweeps = 0, repetitions = 0, replica_offset = 0;
  const uint64_t *x0 = nullptr;
  uint64_t *out_x = nullptr;
  double *out_e = nullptr;
  uint64_t budget = 3ull << 30;  // bytes of visiting orders per buffer set

  uint64_t K = 0;
  uint32_t words = 0, groups = 0, waves = 1, level_cap = 0, quad_cap = 0, order_threads = 64, lanes_per_row = 1;
  // lane packing: blocks of 1 << log_s spins, 64 >> log_s groups per workgroup, `wgs` workgroups
  uint32_t log_s = 6, wgs = 0;
  bool packed_lanes = false;
  uint32_t blocks_of_spins() const { return static_cast<uint32_t>((K + (1u << log_s) - 1) >> log_s); }
  uint64_t padded = 0;
  int m = 1, teams = 1, attempt = 0;
  int forced_m = 0;      // chains per workgroup chosen by the batched driver (0: by this run)
  uint32_t order_threads_cap = 0;  // set by the batched driver: threads of an order workgroup at most
  bool batch_saturates = false;    // set by the batched driver: the batch has more workgroups than the chip holds
  bool batched = false;  // launched by the batched driver: no timing events of its own
  bool trivial = false;  // nothing to launch (no spins or no chains)
  uint32_t status[kStatWords] = {0, 0, 0, 0};

  // Order kernels of two chunks may be in flight at once (a stream and a scratch area each), beside
  // the sweep kernel of a third: an order workgroup is a chain of dependent global accesses (23 ms
  // for a sweep of 1e5 spins), so their THROUGHPUT is workgroups in flight / that latency, and a
  // 128-sweep call has only 32 sweeps per chunk to offer.
  // (round 4, -DASP_SHUF_LANES=2/3/4/6: no difference on the kagome_36 pipeline — 23.6 / 23.7 / 23.1 s —
  // nor on the synthetic batches, and every lane is one more buffer set to allocate)
#ifndef ASP_SHUF_LANES
#define ASP_SHUF_LANES 2
#endif
  static constexpr int kLanes = ASP_SHUF_LANES, kSets = kLanes + 1;
  asp::ScopedStream order_stream[kLanes];
  hipEvent_t ordered[kSets] = {}, swept[kSets] = {};
  DeviceBuffer<double> d_betas, d_partial, d_e;
  DeviceBuffer<uint64_t> d_x0, d_best, d_perm;
  DeviceBuffer<uint8_t> d_state;
  DeviceBuffer<long long> d_ecur, d_ebest;
  DeviceBuffer<unsigned long long> d_accepted;
  DeviceBuffer<uint32_t> d_status, d_prio[kLanes], d_indeg[kLanes], d_order[kLanes];
  struct OrderSet {
    DeviceBuffer<uint32_t> level_block, num_levels, spin_of_pos;
    DeviceBuffer<uint2> block_meta;
    DeviceBuffer<uint8_t> stream;
  } sets[kSets];

  ShuffledRun() = default;
  ShuffledRun(const ShuffledRun &) = delete;
  ShuffledRun &operator=(const ShuffledRun &) = delete;
  ~ShuffledRun() {
    // (the buffers are released after this body: first wait for whatever still uses them)
    for (auto &o : order_stream) {
      if (o.stream) (void)hipStreamSynchronize(o.stream);
    }
    if (p && p->stream) (void)hipStreamSynchronize(p->stream);
    for (hipEvent_t e : ordered) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : swept) if (e) (void)hipEventDestroy(e);
  }

  int setup() {
    if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
    if (repetitions == 0) {
      trivial = true;
      return ASP_OK;
    }
    if (!out_x || !out_e || (num_sweeps && !betas)) return asp::set_error(ASP_ERR_INVALID, "null argument");
    if (num_sweeps >= 0xFFFFFFFEu) return asp::set_error(ASP_ERR_INVALID, "num_sweeps too large");
    if (static_cast<uint64_t>(replica_offset) + repetitions + 8 > 0xFFFFFFFFull) {
      return asp::set_error(ASP_ERR_INVALID, "replica ids exceed 32 bits");
    }
    for (uint32_t t = 0; t < num_sweeps; ++t) {
      if (!(betas[t] >= 0.0)) return asp::set_error(ASP_ERR_INVALID, "betas[%u] is not >= 0", t);
    }
    const asp::SaHostLayout &L = p->host;
    K = L.num_spins;
    words = static_cast<uint32_t>((K + 63) / 64);
    p->last_sweep_ms = p->last_total_ms = p->last_order_ms = 0.0f;
    if (K == 0) {
      for (uint32_t r = 0; r < repetitions; ++r) out_e[r] = 0.0;  // (host pointers: nothing ran)
      trivial = true;
      return ASP_OK;
    }
    ASP_TRY(ensure_static(p));
    // chains per workgroup: as many as still leave a workgroup per compute unit
    m = 1;
    if (forced_m) {
      m = forced_m;
    } else if (p->shuffled_m) {
      m = p->shuffled_m;
    } else {
      // (eight chains — the byte layout — only when that still leaves TWO workgroups per compute
      // unit: 2048 chains on K = 12 870 run at 99.7 G flips/s as 512 groups of four and at 70 G
      // as 256 groups of eight, profiles/r03_shuffled_scan.txt)
      for (int cand : {8, 4, 2}) {
        const uint32_t want_groups = static_cast<uint32_t>(p->num_cus) * (cand == 8 ? 2u : 1u);
        if ((repetitions + cand - 1) / cand >= want_groups) {
          m = cand;
          break;
        }
      }
    }
    const int wanted_m = m;
    const double mean_degree = std::max(1.0, static_cast<double>(L.a_col.size()) / static_cast<double>(K));
    // levels of a sweep: the longest descending-priority path, about 2.5 x the mean degree on
    // the clusters of this problem (measured: 29 at degree 8, 59-69 at degree 23); the last
    // call's count when there is one
    const double levels_guess = p->last_shuffled_levels > 0 ? p->last_shuffled_levels : 2.5 * mean_degree + 4.0;
    expected_levels = levels_guess;
    waves = static_cast<uint32_t>(p->shuffled_waves);
    // one wavefront more than the blocks of an average level: the first levels of a sweep are its
    // widest (K = 12 870 alone on the chip: +6 %) — but not in a batch that oversubscribes the chip,
    // where a wavefront mostly waiting at level barriers holds 256 registers of a SIMD that another
    // problem's workgroup could run in (128-problem production batch, sweeps alone: 1.69 -> 1.56 s)
    uint32_t waves_extra = batch_saturates ? 0u : 1u;
    if (const char *env = std::getenv("ASP_SHUFFLED_WAVES_EXTRA")) waves_extra = static_cast<uint32_t>(std::atoi(env));
    if (!waves) {
      // a wavefront per block of an average level, and one more: the first blocks of a level are
      // its widest (K = 12 870: 4 blocks per level, 4 -> 8 wavefronts +6 %)
      waves = static_cast<uint32_t>(std::ceil(static_cast<double>(K) / levels_guess / 64.0)) + waves_extra;
      waves = std::max(1u, std::min(8u, waves));
    }
    // Two teams of m / 2 chains (k_sa_sweep_shuffled, TEAMS): only on request.  Measured at
    // K = 12 870 (profiles/r03_shuffled_scan.txt): 59 against 63-66 G flips/s with 1024 chains, 66
    // against 100 with 2048 — a visit is bound by the compute unit's fill rate from L2 (a block is
    // 22 KiB, four or five blocks per level and CU), not by the issue rate of its wavefront, and
    // the partner team adds load instructions without adding bandwidth.
    teams = p->shuffled_teams == 2 && m >= 2 && waves <= 4 ? 2 : 1;
    groups = (repetitions + m - 1) / m;
    padded = static_cast<uint64_t>(groups) * m;
    level_cap = static_cast<uint32_t>(std::min<double>(static_cast<double>(K), 2.0 * levels_guess + 32.0));
    if (const char *env = std::getenv("ASP_SHUFFLED_LEVEL_CAP")) {  // test hook: provoke the retry
      level_cap = std::max(1u, static_cast<uint32_t>(std::strtoul(env, nullptr, 10)));
    }
    // Clusters beyond the capacity of a byte per spin: four bits per spin and at most four chains
    // per workgroup, or a bit per spin and one chain (the largest order-2 models of the
    // sampled-cluster pipeline: 1.5e5..3.2e5 spins)
    while (m > 1 && shuffled_layout_for(K, m, level_cap, words + level_cap, p->max_lds) < 0) m >>= 1;
    if (m != wanted_m) {
      groups = (repetitions + m - 1) / m;
      padded = static_cast<uint64_t>(groups) * m;
    }
    {
      const int fits = shuffled_layout_for(K, m, level_cap, words + level_cap, p->max_lds);
      if (fits == kNibbles || fits == kBits) teams = 1;  // (the packed layouts have one team)
    }
    // Lane packing: a level holds about K / levels spins.  When that is well below a wavefront, the
    // order kernel cuts the levels into blocks of S < 64 spins and a wavefront visits a block for
    // G = 64 / S groups of chains at once (k_sa_sweep_shuffled<.., PK = true>): a production-sized
    // cluster (1e2 .. 3e3 spins) has 3 .. 40 spins per level, so 16 workgroups of four chains with
    // a few live lanes each become one to four workgroups with full wavefronts.  S = three quarters
    // of the mean level rounded up to a power of two (levels run from ~2.5 x the mean down to one
    // spin; smaller blocks waste fewer slots on the rounding), at least 4, and large enough that
    // the call has G groups to fill the lanes with and that G x K spin words fit the LDS.
    log_s = 6;
    packed_lanes = false;
    if (teams == 1 && m <= 4 && !std::getenv("ASP_SHUFFLED_NO_PACKING")) {
      const double mean_level = static_cast<double>(K) / levels_guess;
      // ... and a QUARTER of the mean level in a batch that oversubscribes the chip: blocks of 16-32
      // spins up to ~8000 spins, i.e. half to a quarter of the workgroups (each with more wavefronts
      // and G = 2-4 groups).  What such a batch needs is to be RESIDENT at once: at two wavefronts
      // per SIMD the 128-problem production mix was 7 % over the chip's capacity with blocks of 64
      // above 2500 spins, and every chunk launch then takes two critical paths instead of one
      // (2560 sweeps, sweeps alone: 0.80 s; with factor <= 0.5: 0.62 s = the critical path of the
      // largest clusters, the same for 64 .. 112 problems; profiles/r04_shuffled_batch_trace.txt).
      // A call that does not fill the chip keeps blocks of 64 where a level has them: lane packing
      // costs an add per gather and per-lane chain ids (K = 3000 x 64 chains: 43.7 against 49.5 ms).
      double s_factor = batch_saturates ? 0.25 : 0.75;
      if (const char *env = std::getenv("ASP_SHUFFLED_S_FACTOR")) s_factor = std::atof(env);
      uint32_t want = std::max(4u, std::min(64u, next_pow2(static_cast<uint32_t>(std::ceil(s_factor * mean_level)))));
      if (const char *env = std::getenv("ASP_SHUFFLED_LOG_S")) {
        want = 1u << std::max(2l, std::min(6l, std::strtol(env, nullptr, 10)));
      }
      while (want < 64u && 64u / want > groups) want <<= 1;
      while (want < 64u && sweep_lds_bytes(K, kWide, level_cap, static_cast<uint32_t>((K + want - 1) / want) + level_cap,
                                           64u / want) > p->max_lds) {
        want <<= 1;
      }
      if (want < 64u) {
        packed_lanes = true;
        while ((1u << log_s) > want) --log_s;
        if (!p->shuffled_waves) {
          waves = std::max(1u, std::min(8u, static_cast<uint32_t>(std::ceil(mean_level / want)) + waves_extra));
        }
      }
    }
    wgs = (groups + (64u >> log_s) - 1) / (64u >> log_s);
    quad_cap = 0;  // 0: derive from level_cap
    order_threads = K >= 4096 ? kOrderThreads : (K >= 512 ? 256u : 64u);
    if (order_threads_cap) order_threads = std::min(order_threads, order_threads_cap);
    lanes_per_row = std::min(64u, std::min(order_threads, next_pow2(std::max(
        1u, static_cast<uint32_t>(std::ceil(mean_degree / 4.0))))));
    if (const char *env = std::getenv("ASP_SHUFFLED_LANES_PER_ROW")) {  // measurements
      lanes_per_row = std::min(64u, next_pow2(std::max(1u, static_cast<uint32_t>(std::atoi(env)))));
    }
    if (const char *env = std::getenv("ASP_SHUFFLED_BYTES")) budget = std::strtoull(env, nullptr, 10);

    // (the run's own order streams and events: made by enqueue(), which the batched driver — with
    // its shared streams — never calls; 128 problems x (2 streams, 6 events) were 0.1 s of a batch)
    hipStream_t s = p->stream;
    ASP_TRY(d_betas.alloc(num_sweeps));
    // (a byte per spin and group between chunks — or, beyond every LDS layout, the chains' spin words)
    const bool spins_in_hbm = shuffled_layout_for(K, m, level_cap, words + level_cap, p->max_lds) == kGlobal;
    ASP_TRY(d_state.alloc(static_cast<uint64_t>(groups) * K * (spins_in_hbm ? 4 : 1)));
    ASP_TRY(d_best.alloc(padded * words));
    ASP_TRY(d_ecur.alloc(padded));
    ASP_TRY(d_ebest.alloc(padded));
    ASP_TRY(d_accepted.alloc(padded));
    ASP_TRY(d_perm.alloc(static_cast<uint64_t>(repetitions) * L.num_blocks));
    ASP_TRY(d_partial.alloc(static_cast<uint64_t>(repetitions) * L.num_blocks));
    ASP_TRY(d_e.alloc(repetitions));
    ASP_TRY(d_status.alloc(kStatusWords));
    ASP_TRY(d_betas.upload(betas, num_sweeps, s));
    if (x0) {
      ASP_TRY(d_x0.alloc(words));
      ASP_TRY(d_x0.upload(x0, words, s));
    }
    return ASP_OK;
  }

  // ---- one attempt: plan (capacities -> kernels, buffers, argument templates), then launches ----
  uint32_t block_cap = 0, stream_kib = 0, chunk = 0;
  int layout = kBytes;
  size_t lds = 0, order_lds = 0;
  bool order_in_lds = false;
  bool wide_orders = false;  // the order build as grids over the chunk (k_order_*): large clusters
  int counters_in_lds = 0;  // ... of which the peel is the per-sweep workgroup's: 1 bytes, 2 nibbles in LDS
  uint32_t wide_levels = 0;  // level launches per chunk of the wide path (wide_level_launches)
  double expected_levels = 0.0;  // levels a sweep is expected to have (setup())
  DeviceBuffer<uint32_t> d_peel_ctl[kLanes], d_level_start[kLanes];
  DeviceBuffer<uint8_t> d_later[kLanes];
  DeviceBuffer<OrderArgs> d_oargs;  // descriptors of every chunk (the single call's wide launches)
  std::vector<OrderArgs> h_oargs;

  // Workgroups per (problem, sweep) of the wide kernels for a cluster of `spins` spins.
  struct WideParts {
    uint32_t prio, counts, level, stream;
  };
  static WideParts wide_parts(uint64_t spins, uint32_t lanes_per_row, uint32_t log_s) {
    auto clamp = [](uint64_t v, uint64_t hi) { return static_cast<uint32_t>(std::max<uint64_t>(1, std::min(v, hi))); };
    const uint64_t groups = kWideThreads / std::max(1u, lanes_per_row);  // rows per workgroup and trip
    WideParts w;
    w.prio = clamp((spins + kWideThreads * 8 - 1) / (kWideThreads * 8), 64);
    w.counts = clamp((spins + groups * 16 - 1) / (groups * 16), 128);
    // (a level: ~ 1/60 of the spins, the first ones twice that; a row is a chain of six dependent HBM
    // accesses, so a level wants ALL its rows in flight at once: one or two trips per group of lanes)
    w.level = clamp((spins / 48 + groups * 2 - 1) / (groups * 2), 64);
    const uint64_t blocks = (spins >> log_s) + 64, per_workgroup = (kWideThreads / 64) * (64u >> log_s) * 8;
    w.stream = clamp((blocks + per_workgroup - 1) / per_workgroup, 128);
    return w;
  }
  ShuffledKernel kernel = nullptr;
  int nsets = 1, nlanes = 1;
  OrderArgs oa{};
  ShuffledArgs sa{};

  // Sweeps per chunk this run would choose by itself for its present capacities.
  int plan_sizes() {
    const uint32_t max_quads = p->rq_max_quads;
    const uint32_t S = 1u << log_s;
    block_cap = blocks_of_spins() + level_cap;
    if (packed_lanes && sweep_lds_bytes(K, kWide, level_cap, block_cap, 64u >> log_s) > p->max_lds) {
      // (the capacities grew since setup(): back to blocks of 64 spins, one group per workgroup)
      packed_lanes = false;
      log_s = 6;
      wgs = groups;
      return plan_sizes();
    }
    // a word per spin (the one-instruction sign) when that fits the LDS beside the sweep's tables,
    // else a byte, else four bits (m <= 4) or one (m = 1)
    layout = packed_lanes ? kWide : shuffled_layout_for(K, m, level_cap, block_cap, p->max_lds);
    if (layout < 0) {
      return asp::set_error(ASP_ERR_TOO_LARGE, "the shuffled sweep keeps its spins in LDS: %llu spins with %d "
                                               "chains per workgroup do not fit", (unsigned long long)K, m);
    }
    lds = sweep_lds_bytes(K, layout, level_cap, block_cap, packed_lanes ? 64u >> log_s : 0u);
    if (!quad_cap) {
      // exact class sort (rows below 63 quads): a block is no wider than every row of the block
      // before it in its level, so the blocks hold at most (sum of the row quads) / 64 + one
      // widest block per level; otherwise every block may be as wide as the longest row
      const uint64_t tight = (static_cast<uint64_t>(p->rq_quads) + S - 1) / S +
                             static_cast<uint64_t>(level_cap) * max_quads;
      const uint64_t loose = static_cast<uint64_t>(block_cap) * max_quads;
      quad_cap = static_cast<uint32_t>(std::min<uint64_t>(max_quads < kClassCap ? tight : loose, kQuadLimit));
    }
    // KiB of one sweep's stream: slabs of 16 S bytes — a header slab per block, three per quad,
    // and slack for the sweep kernel's read of one quad past a wide block (< 4 GiB: 32-bit
    // scalar offsets)
    stream_kib = static_cast<uint32_t>(((static_cast<uint64_t>(block_cap) + 3ull * (quad_cap + 2ull)) * 16 * S + 1023) / 1024);
    const uint64_t per_sweep = static_cast<uint64_t>(stream_kib) * 1024 + static_cast<uint64_t>(block_cap) * (4 * S + 8) +
                               (level_cap + 1ull) * 4 + 12ull * K;
    chunk = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(256, budget / per_sweep)));
    // (at least two chunks per order lane: the pipeline needs them)
    chunk = std::max(1u, std::min(chunk, (num_sweeps + 2 * kLanes - 1) / (2 * kLanes)));
    // the peel's arrays in LDS when they fit half of it (the sweep workgroups of the previous chunk
    // are resident beside the order workgroups) and the counters fit a byte
    order_in_lds = K < 65536 && p->rq_max_quads * 4u <= 255u && !std::getenv("ASP_SHUFFLED_ORDER_IN_HBM") &&
                   order_lds_bytes(level_cap, block_cap, order_threads / 64, K) <= p->max_lds * 9 / 16;
    wide_orders = !order_in_lds && !std::getenv("ASP_SHUFFLED_ORDER_FUSED");
    // Beyond that: priorities, counts and the later-masks from grids over the chunk, and the PEEL in
    // the per-sweep workgroup with its counters as bytes in LDS when those fit (K <= ~150 000) —
    // the peel's decrements are random atomics, and in HBM the chip retires ~27 G of them a second
    // whatever their locality (tools/atomics_probe.hip); else level launches (k_order_level).
    counters_in_lds = 0;
    if (wide_orders && !std::getenv("ASP_SHUFFLED_COUNTERS_IN_HBM")) {
      const uint32_t ow = order_threads / 64;
      const bool bytes_fit = p->rq_max_quads * 4u <= 255u && !std::getenv("ASP_SHUFFLED_COUNTER_NIBBLES") &&
                             order_lds_bytes(level_cap, block_cap, ow, 0, (K + 3) / 4) <= p->max_lds;
      if (bytes_fit) {
        counters_in_lds = 1;
      } else if (order_lds_bytes(level_cap, block_cap, ow, 0, (K + 7) / 8) <= p->max_lds) {
        counters_in_lds = 2;  // nibbles; spins with 15 or more earlier neighbours count in HBM
      }
    }
    order_lds = order_lds_bytes(level_cap, block_cap, order_threads / 64, order_in_lds ? K : 0,
                                counters_in_lds == 1 ? (K + 3) / 4 : (counters_in_lds == 2 ? (K + 7) / 8 : 0));
    wide_levels = wide_orders && !counters_in_lds ? wide_level_launches(expected_levels, level_cap) : 0u;
    if (order_lds > p->max_lds) {
      return asp::set_error(ASP_ERR_TOO_LARGE, "%u levels x %u blocks do not fit the order kernel's LDS",
                            level_cap, block_cap);
    }
    return ASP_OK;
  }

  // Buffers of the attempt for `use_chunk` sweeps per chunk, status words zeroed (on the plan's
  // stream, p->ev[0] recorded behind that), argument templates filled.
  int plan_buffers(uint32_t use_chunk) {
    const asp::SaHostLayout &L = p->host;
    hipStream_t s = p->stream;
    chunk = use_chunk;
    const uint32_t chunks = num_sweeps ? (num_sweeps + chunk - 1) / chunk : 1;
    nsets = static_cast<int>(std::min<uint32_t>(kSets, chunks));
    nlanes = static_cast<int>(std::min<uint32_t>(kLanes, chunks));
    for (int i = 0; i < nlanes; ++i) {
      const uint64_t scratch = order_in_lds ? 1 : static_cast<uint64_t>(chunk) * K;  // (LDS: no HBM scratch)
      ASP_TRY(d_prio[i].ensure(scratch));
      ASP_TRY(d_indeg[i].ensure(scratch));
      ASP_TRY(d_order[i].ensure(scratch));
      if (wide_orders) {
        ASP_TRY(d_peel_ctl[i].ensure(static_cast<uint64_t>(chunk) * 8));
        ASP_TRY(d_level_start[i].ensure(static_cast<uint64_t>(chunk) * (level_cap + 2)));
        ASP_TRY(d_later[i].ensure(static_cast<uint64_t>(chunk) * p->rq_quads));
      }
    }
    for (int i = 0; i < nsets; ++i) {
      OrderSet &o = sets[i];
      ASP_TRY(o.level_block.ensure(static_cast<uint64_t>(chunk) * (level_cap + 1)));
      ASP_TRY(o.num_levels.ensure(chunk));
      ASP_TRY(o.block_meta.ensure(static_cast<uint64_t>(chunk) * block_cap));
      ASP_TRY(o.spin_of_pos.ensure((static_cast<uint64_t>(chunk) * block_cap) << log_s));
      ASP_TRY(o.stream.ensure(static_cast<uint64_t>(chunk) * stream_kib * 1024));
    }
    ASP_HIP_TRY(hipMemsetAsync(d_status.ptr, 0, (kStatusWords) * sizeof(uint32_t), s));
    ASP_HIP_TRY(hipEventRecord(p->ev[0], s));
    oa = OrderArgs{};
    oa.rq_ptr = p->rq_ptr.ptr;
    oa.rq_col = reinterpret_cast<const uint4 *>(p->rq_col.ptr);
    oa.rq_val = reinterpret_cast<const double2 *>(p->rq_val.ptr);
    oa.field = p->field_dev.ptr;
    oa.seed = seed;
    oa.num_spins = static_cast<uint32_t>(K);
    oa.level_cap = level_cap;
    oa.block_cap = block_cap;
    oa.quad_cap = quad_cap;
    oa.stream_kib = stream_kib;
    oa.lanes_per_row = lanes_per_row;
    oa.log_s = log_s;
    oa.threads = order_threads;
    oa.num_quads = p->rq_quads;
    oa.max_quads = p->rq_max_quads;
    oa.lds_arrays = order_in_lds ? 1u : 0u;
    oa.lds_counters = static_cast<uint32_t>(counters_in_lds);
    oa.finish_only = wide_orders ? 1u : 0u;
    oa.wide_levels = wide_levels;
    oa.col_shift = (layout == kWide || layout == kGlobal) ? 2u : 0u;  // (byte offsets of 32-bit spin words)
    oa.status = d_status.ptr;
    // (the layout may change between attempts — capacities grow —: the HBM form needs a word per spin)
    ASP_TRY(d_state.ensure(static_cast<uint64_t>(groups) * K * (layout == kGlobal ? 4 : 1)));
    sa = ShuffledArgs{};
    sa.status = d_status.ptr;
    sa.betas = d_betas.ptr;
    sa.x0 = x0 ? d_x0.ptr : nullptr;
    sa.state = d_state.ptr;
    sa.best = d_best.ptr;
    sa.e_cur = d_ecur.ptr;
    sa.e_best = d_ebest.ptr;
    sa.accepted = d_accepted.ptr;
    sa.seed = seed;
    sa.scale = std::ldexp(1.0, L.energy_scale_exp);
    sa.num_spins = static_cast<uint32_t>(K);
    sa.words = words;
    sa.level_cap = level_cap;
    sa.block_cap = block_cap;
    sa.stream_kib = stream_kib;
    sa.replica_first = replica_offset;
    sa.log_s = log_s;
    sa.groups_total = groups;
    sa.waves = waves;
    return ASP_OK;
  }

  // The arguments of chunk `turn` (sweeps done .. done + now): buffer set turn % nsets, scratch area
  // turn % nlanes.
  void chunk_args(uint32_t turn, uint32_t done, uint32_t now, bool first_launch, OrderArgs *o_out,
                  ShuffledArgs *s_out) const {
    const OrderSet &o = sets[turn % static_cast<uint32_t>(nsets)];
    const uint32_t lane = turn % static_cast<uint32_t>(nlanes);
    OrderArgs x = oa;
    x.prio = d_prio[lane].ptr;
    x.indeg = d_indeg[lane].ptr;
    x.order = d_order[lane].ptr;
    x.peel_ctl = d_peel_ctl[lane].ptr;
    x.level_start_g = d_level_start[lane].ptr;
    x.later = d_later[lane].ptr;
    x.first_sweep = done;
    x.count = now;
    x.level_block = o.level_block.ptr;
    x.num_levels = o.num_levels.ptr;
    x.block_meta = o.block_meta.ptr;
    x.spin_of_pos = o.spin_of_pos.ptr;
    x.stream = o.stream.ptr;
    *o_out = x;
    ShuffledArgs y = sa;
    y.level_block = o.level_block.ptr;
    y.num_levels = o.num_levels.ptr;
    y.block_meta = o.block_meta.ptr;
    y.stream = o.stream.ptr;
    y.first_sweep = done;
    y.chunk_sweeps = now;
    y.initialise = first_launch ? 1u : 0u;
    *s_out = y;
  }

  // Queues one whole attempt (all chunks) on the run's streams; returns without waiting.
  int enqueue() {
    if (trivial) return ASP_OK;
    hipStream_t s = p->stream;
    for (auto &o : order_stream) {
      if (!o.stream) ASP_TRY(o.acquire());
    }
    for (int i = 0; i < kSets; ++i) {
      if (!ordered[i]) ASP_HIP_TRY(hipEventCreateWithFlags(&ordered[i], hipEventDisableTiming));
      if (!swept[i]) ASP_HIP_TRY(hipEventCreateWithFlags(&swept[i], hipEventDisableTiming));
    }
    ASP_TRY(plan_sizes());
    kernel = shuffled_kernel_for(m, layout, teams, packed_lanes);
    if (!kernel) {  // (no two-team form of this width and layout)
      teams = 1;
      kernel = shuffled_kernel_for(m, layout, 1, packed_lanes);
    }
    if (lds > 64 * 1024) {
      ASP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    }
    if (order_lds > 64 * 1024) {
      ASP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(wide_orders ? k_shuffled_orders<true> : k_shuffled_orders<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(order_lds)));
    }
    ASP_TRY(plan_buffers(chunk));
    WidePartsMax wide;
    if (wide_orders) {
      // descriptors of every chunk on the device (the wide kernels read them from a table)
      h_oargs.clear();
      for (uint32_t done = 0, turn = 0; done < num_sweeps; done += chunk, ++turn) {
        OrderArgs o_args;
        ShuffledArgs unused;
        chunk_args(turn, done, std::min(chunk, num_sweeps - done), turn == 0, &o_args, &unused);
        h_oargs.push_back(o_args);
      }
      if (!h_oargs.empty()) {
        ASP_TRY(d_oargs.ensure(h_oargs.size()));
        ASP_TRY(d_oargs.upload(h_oargs.data(), h_oargs.size(), s));
        ASP_HIP_TRY(hipEventRecord(p->ev[0], s));  // (the order streams wait for this event below)
      }
      const WideParts parts = wide_parts(K, lanes_per_row, log_s);
      wide.prio = parts.prio;
      wide.counts = parts.counts;
      wide.level = parts.level;
      wide.stream = parts.stream;
      wide.levels = wide_levels;
    }
    for (int i = 0; i < nlanes; ++i) {
      ASP_HIP_TRY(hipStreamWaitEvent(order_stream[i].stream, p->ev[0], 0));  // status zeroed, buffers ours
    }
    bool first_launch = true;
    uint32_t turn = 0;
    for (uint32_t done = 0; done < num_sweeps || first_launch; done += chunk, ++turn) {
      const uint32_t now = num_sweeps > done ? std::min(chunk, num_sweeps - done) : 0u;
      const int which = static_cast<int>(turn % static_cast<uint32_t>(nsets));
      hipStream_t os = order_stream[turn % static_cast<uint32_t>(nlanes)].stream;
      OrderArgs o_args;
      ShuffledArgs s_args;
      chunk_args(turn, done, now, first_launch, &o_args, &s_args);
      if (now) {
        // the orders of this chunk: after the sweep kernel that last read this buffer set has let
        // go of it; chunks alternate between the two order streams (a scratch area each)
        if (turn >= static_cast<uint32_t>(nsets)) ASP_HIP_TRY(hipStreamWaitEvent(os, swept[which], 0));
        if (wide_orders) ASP_TRY(launch_wide_front(os, d_oargs.ptr + turn, 1, now, wide));
        if (wide_orders) {
          hipLaunchKernelGGL(k_shuffled_orders<true>, dim3(now), dim3(order_threads), order_lds, os, o_args);
        } else {
          hipLaunchKernelGGL(k_shuffled_orders<false>, dim3(now), dim3(order_threads), order_lds, os, o_args);
        }
        ASP_HIP_TRY(hipGetLastError());
        if (wide_orders) ASP_TRY(launch_wide_stream(os, d_oargs.ptr + turn, 1, now, wide));
        ASP_HIP_TRY(hipEventRecord(ordered[which], os));

        ASP_HIP_TRY(hipStreamWaitEvent(s, ordered[which], 0));
      }
      hipLaunchKernelGGL(kernel, dim3(wgs), dim3(waves * teams * 64), lds, s, s_args);
      ASP_HIP_TRY(hipGetLastError());
      ASP_HIP_TRY(hipEventRecord(swept[which], s));
      first_launch = false;
    }
    ASP_HIP_TRY(hipEventRecord(p->ev[2], s));
    ASP_HIP_TRY(hipMemcpyAsync(status, d_status.ptr, sizeof status, hipMemcpyDeviceToHost, s));
    return ASP_OK;
  }

  // Waits for the attempt; *again = true when the capacities had to grow and the attempt must be
  // repeated (chains restart from their initial configuration; results do not depend on the
  // capacities).
  int collect(bool *again) {
    *again = false;
    if (trivial) return ASP_OK;
    ASP_HIP_TRY(hipStreamSynchronize(p->stream));
    for (auto &o : order_stream) ASP_HIP_TRY(hipStreamSynchronize(o.stream));
    return grow(again);
  }

  // After an attempt whose status words are in `status`: nothing to do, or larger capacities.
  int grow(bool *again) {
    *again = false;
    if (status[kStatBad] == 0) return ASP_OK;
    if (++attempt > 4) {
      return asp::set_error(ASP_ERR_TOO_LARGE, "visiting orders of %u levels / %u quads per sweep do not fit",
                            status[kStatLevels], status[kStatQuads]);
    }
    if (status[kStatLevels] > level_cap) {
      level_cap = static_cast<uint32_t>(std::min<uint64_t>(K, 2ull * status[kStatLevels] + 16));
      quad_cap = 0;
    } else {
      quad_cap = static_cast<uint32_t>(std::min<uint64_t>(kQuadLimit, 2ull * std::max(status[kStatQuads], quad_cap)));
    }
    *again = true;
    return ASP_OK;
  }

  // Energies of §4.6 from the packed best configurations and the copies to the caller (queued).
  int finish_enqueue() {
    if (trivial) return ASP_OK;
    hipStream_t s = p->stream;
    p->last_shuffled_levels = static_cast<int>(status[kStatLevels]);
#if ASP_SHUF_TIMING
    {
      unsigned long long host_ticks[kTimingSlots * kTimingWaves + kOrderTimingSlots];
      ASP_HIP_TRY(hipMemcpy(host_ticks, d_status.ptr + kStatWords, sizeof host_ticks, hipMemcpyDeviceToHost));
      static const char *names[kTimingSlots] = {"row sums", "request", "accept", "barrier", "sweep end", "level head"};
      for (uint32_t w = 0; w < waves && w < kTimingWaves; ++w) {
        std::fprintf(stderr, "wave %u:", w);
        for (uint32_t k = 0; k < kTimingSlots; ++k) {
          std::fprintf(stderr, " %s %.3f Mcyc", names[k], static_cast<double>(host_ticks[w * kTimingSlots + k]) * 1e-6);
        }
        std::fprintf(stderr, "\n");
      }
      static const char *onames[kOrderTimingSlots] = {"priorities", "counts", "peel", "level blocks", "sort", "widths",
                                                      "stream", "tables"};
      std::fprintf(stderr, "order kernel, wavefront 0, all sweeps:");
      for (uint32_t k = 0; k < kOrderTimingSlots; ++k) {
        std::fprintf(stderr, " %s %.3f Mcyc", onames[k], static_cast<double>(host_ticks[kTimingSlots * kTimingWaves + k]) * 1e-6);
      }
      std::fprintf(stderr, "\n");
    }
#endif
    ASP_TRY(asp::sa_permute_bits(p, d_best.ptr, repetitions, d_perm.ptr));
    ASP_TRY(asp::sa_energies_of_perm(p, d_perm.ptr, repetitions, d_partial.ptr, d_e.ptr));
    ASP_HIP_TRY(hipEventRecord(p->ev[3], s));
    // hipMemcpyDefault: out_x / out_e may be host pointers (the usual call) or device pointers
    ASP_HIP_TRY(hipMemcpyAsync(out_x, d_best.ptr, static_cast<uint64_t>(repetitions) * words * sizeof(uint64_t),
                               hipMemcpyDefault, s));
    ASP_HIP_TRY(hipMemcpyAsync(out_e, d_e.ptr, repetitions * sizeof(double), hipMemcpyDefault, s));
    p->last_tracked.assign(repetitions, 0);
    p->last_accepted.assign(repetitions, 0);
    ASP_HIP_TRY(hipMemcpyAsync(p->last_tracked.data(), d_ebest.ptr, repetitions * sizeof(int64_t),
                               hipMemcpyDeviceToHost, s));
    ASP_HIP_TRY(hipMemcpyAsync(p->last_accepted.data(), d_accepted.ptr, repetitions * sizeof(uint64_t),
                               hipMemcpyDeviceToHost, s));
    return ASP_OK;
  }

  int finish_wait() {
    if (trivial) return ASP_OK;
    ASP_HIP_TRY(hipStreamSynchronize(p->stream));
    p->last_m = m;
    p->last_layout = 5;
    p->last_threads = static_cast<int>(waves * teams * 64);
    p->last_groups = static_cast<int>(groups);
    p->last_shuffled_log_s = log_s;
    p->last_shuffled_wgs = wgs;
    p->last_shuffled_blocks = status[kStatBlocks];
    p->last_shuffled_quads = status[kStatQuads];
    if (!batched) {
      ASP_HIP_TRY(hipEventElapsedTime(&p->last_sweep_ms, p->ev[0], p->ev[2]));
      ASP_HIP_TRY(hipEventElapsedTime(&p->last_total_ms, p->ev[0], p->ev[3]));
    }
    return ASP_OK;
  }
};

}  // namespace

namespace {

using ShuffledBatchKernel = void (*)(const ShuffledArgs *, const ShuffledSlot *);

ShuffledBatchKernel shuffled_batch_kernel_for(int m, int layout, bool packed_lanes) {
  if (packed_lanes) {
    if (layout != kWide) return nullptr;
    switch (m) {
      case 1: return k_sa_sweep_shuffled_batch<1, kWide, true>;
      case 2: return k_sa_sweep_shuffled_batch<2, kWide, true>;
      case 4: return k_sa_sweep_shuffled_batch<4, kWide, true>;
      default: return nullptr;
    }
  }
  if (layout == kWide) {
    switch (m) {
      case 1: return k_sa_sweep_shuffled_batch<1, kWide>;
      case 2: return k_sa_sweep_shuffled_batch<2, kWide>;
      case 4: return k_sa_sweep_shuffled_batch<4, kWide>;
      default: return nullptr;
    }
  }
  if (layout == kNibbles) {
    switch (m) {
      case 1: return k_sa_sweep_shuffled_batch<1, kNibbles>;
      case 2: return k_sa_sweep_shuffled_batch<2, kNibbles>;
      case 4: return k_sa_sweep_shuffled_batch<4, kNibbles>;
      default: return nullptr;
    }
  }
  if (layout == kBits) return m == 1 ? k_sa_sweep_shuffled_batch<1, kBits> : nullptr;
  if (layout == kGlobal) return m == 1 ? k_sa_sweep_shuffled_batch<1, kGlobal> : nullptr;
  switch (m) {
    case 1: return k_sa_sweep_shuffled_batch<1, kBytes>;
    case 2: return k_sa_sweep_shuffled_batch<2, kBytes>;
    case 4: return k_sa_sweep_shuffled_batch<4, kBytes>;
    case 8: return k_sa_sweep_shuffled_batch<8, kBytes>;
    default: return nullptr;
  }
}

struct EventPool {
  std::vector<hipEvent_t> events;
  ~EventPool() {
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
  }
  int make(hipEvent_t *out, bool timing = false) {
    ASP_HIP_TRY(hipEventCreateWithFlags(out, timing ? hipEventDefault : hipEventDisableTiming));
    events.push_back(*out);
    return ASP_OK;
  }
};

// Problems with the same number of sweeps in SHARED launches: per chunk one order launch over
// (problem, sweep) and one sweep launch per class of workgroup shape over (problem, group of
// chains) — descriptors of every problem and chunk in one device table, uploaded once.  A launch
// per problem and chunk does not overlap on the device however many streams it is spread over
// (64 clusters: 21 s, as long as one after the other); the chip needs the workgroups of many
// problems inside ONE grid.  Every chain is the one its own asp_sa_anneal_shuffled call produces.
int run_shuffled_group(std::vector<ShuffledRun *> &runs, float *sweep_ms) {
  const uint32_t P = static_cast<uint32_t>(runs.size());
  const uint32_t num_sweeps = runs[0]->num_sweeps;
  // one block size for the shared order launch: the largest any problem wants (the workgroups of
  // a smaller problem retire the wavefronts beyond its own OrderArgs::threads at once)
  uint32_t order_threads = 64;
  for (ShuffledRun *r : runs) order_threads = std::max(order_threads, r->order_threads);
  for (ShuffledRun *r : runs) {
    r->batched = true;
    r->teams = 1;
  }
  // (declared before the streams: released after the streams have been waited for, also on an
  // early error return)
  DeviceBuffer<OrderArgs> d_oargs, d_oargs_wide;
  DeviceBuffer<ShuffledArgs> d_sargs;
  asp::ScopedStream order_stream[ShuffledRun::kLanes];
  for (auto &o : order_stream) ASP_TRY(o.acquire());
  // Classes of KERNEL: (layout, lane packing).  Wavefronts per workgroup and spins per block are
  // per-problem arguments (a workgroup retires the wavefronts beyond its problem's own), so a
  // chunk is one order launch and one sweep launch per class — two or three launches, each on a
  // stream of its own: the device runs FOUR hardware queues, and with a class per (wavefronts,
  // block size) — eleven streams on the production mix — kernels waiting for their chunk's orders
  // blocked the queues of kernels that could have run (profiles/r04_shuffled_batch_trace.txt).
  struct Class {
    int layout, m, lds_bucket;
    uint32_t waves = 0;  // of the launch: the most any member wants
    bool packed_lanes;
    std::vector<uint32_t> members;
    DeviceBuffer<ShuffledSlot> slots;  // (before the stream: released after it has been waited for)
    asp::ScopedStream stream;
    uint32_t num_slots = 0;
    size_t lds = 0;
    hipEvent_t swept[ShuffledRun::kSets] = {};
  };
  EventPool events;
  hipEvent_t ordered[ShuffledRun::kSets], t_begin, t_end;
  for (auto &e : ordered) ASP_TRY(events.make(&e));
  ASP_TRY(events.make(&t_begin, true));
  ASP_TRY(events.make(&t_end, true));
  for (;;) {
    // ---- plan: capacities, one chunk length for all, buffers ----
    uint32_t chunk = 256;
    for (ShuffledRun *r : runs) {
      ASP_TRY(r->plan_sizes());
      chunk = std::min(chunk, r->chunk);
    }
    size_t order_lds = 0, order_lds_wide = 0;  // of the two order launches (k_shuffled_orders_batch)
    for (ShuffledRun *r : runs) {
      ASP_TRY(r->plan_buffers(chunk));
      size_t &of = r->wide_orders ? order_lds_wide : order_lds;
      of = std::max(of, r->order_lds);
    }
    for (ShuffledRun *r : runs) ASP_HIP_TRY(hipStreamSynchronize(r->p->stream));  // schedules up, status zeroed
    if (order_lds > 64 * 1024) {
      ASP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_shuffled_orders_batch<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(order_lds)));
    }
    if (order_lds_wide > 64 * 1024) {
      ASP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_shuffled_orders_batch<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(order_lds_wide)));
    }
    std::vector<std::unique_ptr<Class>> classes;
    int lds_buckets = 2;  // (scanned on the kagome_36 pipeline: 8.1 / 7.4 / 9.3 s per round with 1 / 2 / 3)
    if (const char *env = std::getenv("ASP_SHUFFLED_LDS_BUCKETS")) lds_buckets = std::atoi(env);
    for (uint32_t i = 0; i < P; ++i) {
      ShuffledRun *r = runs[i];
      Class *c = nullptr;
      // (a launch has ONE LDS size, the largest of its members: a 2 000-spin model in the launch of a
      // 200 000-spin one would hold a whole compute unit's LDS — three LDS sizes per kernel, so
      // four, two or one workgroup per compute unit)
      int lds_bucket = r->lds <= 40 * 1024 ? 0 : (r->lds <= 80 * 1024 ? 1 : 2);
      if (lds_buckets == 2) lds_bucket = r->lds <= 80 * 1024 ? 0 : 1;  // (one workgroup per compute unit, or more)
      if (lds_buckets <= 1) lds_bucket = 0;
      for (auto &k : classes) {
        if (k->layout == r->layout && k->m == r->m && k->packed_lanes == r->packed_lanes &&
            k->lds_bucket == lds_bucket) {
          c = k.get();
        }
      }
      if (!c) {
        classes.emplace_back(new Class());
        c = classes.back().get();
        c->layout = r->layout;
        c->m = r->m;
        c->lds_bucket = lds_bucket;
        c->packed_lanes = r->packed_lanes;
        ASP_TRY(c->stream.acquire());
        for (auto &e : c->swept) ASP_TRY(events.make(&e));
      }
      c->members.push_back(i);
      c->lds = std::max(c->lds, r->lds);
      c->waves = std::max(c->waves, r->waves);
    }
    // the launches of the large workgroups first: the small ones fill the compute units they leave
    std::stable_sort(classes.begin(), classes.end(),
                     [](const std::unique_ptr<Class> &x, const std::unique_ptr<Class> &y) { return x->lds > y->lds; });
    // (a kernel is keyed by (m, layout, packing), a class also by its wavefronts and block size:
    // classes may share a kernel, and its dynamic-LDS limit must cover the largest of them)
    std::vector<std::pair<ShuffledBatchKernel, size_t>> kernel_lds;
    for (auto &c : classes) {
      // the longest problems first: a workgroup's time is sweeps x levels x one block visit whatever
      // the cluster's size, but the large clusters have more levels and wider rows
      std::stable_sort(c->members.begin(), c->members.end(),
                       [&](uint32_t x, uint32_t y) { return runs[x]->K > runs[y]->K; });
      // XCD-aware slot table: workgroups are dealt round-robin over the chip's eight XCDs (blocks b
      // and b + 8 share one, each XCD with an L2 of its own), so the workgroups of ONE problem get
      // slots of equal index mod 8 — the sweep's coupling stream, read by every workgroup of the
      // problem, is then fetched from HBM by one L2 instead of eight (a placement for speed only:
      // nothing depends on it).  Problems go to the XCD with the fewest workgroups so far.
      std::vector<ShuffledSlot> per_xcd[8];
      for (uint32_t i : c->members) {
        int least = 0;
        for (int x = 1; x < 8; ++x) {
          if (per_xcd[x].size() < per_xcd[least].size()) least = x;
        }
        for (uint32_t g = 0; g < runs[i]->wgs; ++g) per_xcd[least].push_back(ShuffledSlot{i, g});
      }
      size_t longest = 0;
      for (auto &list : per_xcd) longest = std::max(longest, list.size());
      std::vector<ShuffledSlot> slots;
      for (size_t j = 0; j < longest; ++j) {
        for (auto &list : per_xcd) slots.push_back(j < list.size() ? list[j] : ShuffledSlot{0xFFFFFFFFu, 0u});
      }
      c->num_slots = static_cast<uint32_t>(slots.size());
      ASP_TRY(c->slots.alloc(slots.size()));
      ASP_TRY(c->slots.upload(slots.data(), slots.size(), c->stream.stream));
      ASP_HIP_TRY(hipStreamSynchronize(c->stream.stream));  // `slots` dies with this scope
      ShuffledBatchKernel kernel = shuffled_batch_kernel_for(c->m, c->layout, c->packed_lanes);
      if (!kernel) return asp::set_error(ASP_ERR_INVALID, "no batched shuffled kernel for %d chains per group", c->m);
      bool seen = false;
      for (auto &k : kernel_lds) {
        if (k.first == kernel) {
          k.second = std::max(k.second, c->lds);
          seen = true;
        }
      }
      if (!seen) kernel_lds.emplace_back(kernel, c->lds);
    }
    if (std::getenv("ASP_SHUFFLED_HOST_TIMING")) {
      for (auto &c : classes) {
        std::fprintf(stderr, "  shuffled batch: class M=%d layout=%d packed=%d: %zu problems, %u slots x %u wavefronts, %zu bytes of LDS\n",
                     c->m, c->layout, int(c->packed_lanes), c->members.size(), c->num_slots, c->waves, c->lds);
      }
    }
    for (auto &k : kernel_lds) {
      if (k.second > 64 * 1024) {
        ASP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k.first),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(k.second)));
      }
    }
    // the wide order launches (large clusters): shapes for the largest of them; problems of the
    // fused path leave these kernels at once
    WidePartsMax wide;
    std::vector<uint32_t> wide_members;
    for (uint32_t i = 0; i < P; ++i) {
      ShuffledRun *r = runs[i];
      if (!r->wide_orders) continue;
      wide_members.push_back(i);
      const ShuffledRun::WideParts parts = ShuffledRun::wide_parts(r->K, r->lanes_per_row, r->log_s);
      wide.prio = std::max(wide.prio, parts.prio);
      wide.counts = std::max(wide.counts, parts.counts);
      wide.level = std::max(wide.level, parts.level);
      wide.stream = std::max(wide.stream, parts.stream);
      wide.levels = std::max(wide.levels, r->wide_levels);
    }
    // (one number of level launches for the shared grids: every wide problem's finish workgroup
    // must take over exactly where the launches stop)
    for (uint32_t i : wide_members) runs[i]->oa.wide_levels = wide.levels;
    // ---- descriptors of every (chunk, problem) ----
    const uint32_t chunks = num_sweeps ? (num_sweeps + chunk - 1) / chunk : 1;
    std::vector<OrderArgs> oargs(static_cast<size_t>(chunks) * P);
    std::vector<ShuffledArgs> sargs(static_cast<size_t>(chunks) * P);
    for (uint32_t turn = 0; turn < chunks; ++turn) {
      const uint32_t done = turn * chunk;
      const uint32_t now = num_sweeps > done ? std::min(chunk, num_sweeps - done) : 0u;
      for (uint32_t i = 0; i < P; ++i) {
        runs[i]->chunk_args(turn, done, now, turn == 0, &oargs[static_cast<size_t>(turn) * P + i],
                            &sargs[static_cast<size_t>(turn) * P + i]);
      }
    }
    // Timing-only ablations (results are WRONG; tools/time_shuffled_batch_only.py): 1 = orders of
    // the first buffer sets only, every later chunk sweeps through stale ones (the cost of the
    // sweep kernels alone); 2 = no sweep launches after the first chunk (the order kernels alone)
    // (only in a build with -DASP_SHUF_ABLATE_ENV=1: the product ignores the variable)
    int ablate = 0;
#if ASP_SHUF_ABLATE_ENV
    if (const char *env = std::getenv("ASP_SHUFFLED_ABLATE")) ablate = std::atoi(env);
#endif
    // (the wide kernels read a table of the wide problems only: their grids are workgroups per
    // (problem, sweep), and two thirds of a pipeline round are small models of the fused path)
    const uint32_t Pw = static_cast<uint32_t>(wide_members.size());
    std::vector<OrderArgs> oargs_wide(static_cast<size_t>(chunks) * Pw);
    for (uint32_t turn = 0; turn < chunks; ++turn) {
      for (uint32_t k = 0; k < Pw; ++k) {
        oargs_wide[static_cast<size_t>(turn) * Pw + k] = oargs[static_cast<size_t>(turn) * P + wide_members[k]];
      }
    }
    hipStream_t os0 = order_stream[0].stream;
    ASP_TRY(d_oargs.ensure(oargs.size()));
    ASP_TRY(d_sargs.ensure(sargs.size()));
    ASP_TRY(d_oargs.upload(oargs.data(), oargs.size(), os0));
    ASP_TRY(d_sargs.upload(sargs.data(), sargs.size(), os0));
    if (Pw) {
      ASP_TRY(d_oargs_wide.ensure(oargs_wide.size()));
      ASP_TRY(d_oargs_wide.upload(oargs_wide.data(), oargs_wide.size(), os0));
    }
    ASP_HIP_TRY(hipStreamSynchronize(os0));
    // ---- the pipeline of runs' enqueue(), with shared launches ----
    const uint32_t nsets = static_cast<uint32_t>(runs[0]->nsets), nlanes = static_cast<uint32_t>(runs[0]->nlanes);
    ASP_HIP_TRY(hipEventRecord(t_begin, os0));
    for (uint32_t turn = 0; turn < chunks; ++turn) {
      const uint32_t done = turn * chunk;
      const uint32_t now = num_sweeps > done ? std::min(chunk, num_sweeps - done) : 0u;
      const uint32_t which = turn % nsets;
      hipStream_t os = order_stream[turn % nlanes].stream;
      if (now && !(ablate == 1 && turn >= nsets)) {
        if (turn >= nsets) {
          for (auto &c : classes) ASP_HIP_TRY(hipStreamWaitEvent(os, c->swept[which], 0));
        }
        const OrderArgs *chunk_problems = d_oargs.ptr + static_cast<size_t>(turn) * P;
        const OrderArgs *chunk_wide = Pw ? d_oargs_wide.ptr + static_cast<size_t>(turn) * Pw : nullptr;
        if (Pw) ASP_TRY(launch_wide_front(os, chunk_wide, Pw, now, wide));
        if (Pw < P) {
          hipLaunchKernelGGL(k_shuffled_orders_batch<false>, dim3(P * now), dim3(order_threads), order_lds, os,
                             chunk_problems, now);
        }
        if (Pw) {
          hipLaunchKernelGGL(k_shuffled_orders_batch<true>, dim3(Pw * now), dim3(order_threads), order_lds_wide, os,
                             chunk_wide, now);
        }
        ASP_HIP_TRY(hipGetLastError());
        if (Pw) ASP_TRY(launch_wide_stream(os, chunk_wide, Pw, now, wide));
        ASP_HIP_TRY(hipEventRecord(ordered[which], os));

      }
      for (auto &c : classes) {
        if (ablate == 2 && turn > 0) continue;
        hipStream_t cs = c->stream.stream;
        if (now) ASP_HIP_TRY(hipStreamWaitEvent(cs, ordered[which], 0));
        ShuffledBatchKernel kernel = shuffled_batch_kernel_for(c->m, c->layout, c->packed_lanes);
        hipLaunchKernelGGL(kernel, dim3(c->num_slots), dim3(c->waves * 64), c->lds, cs,
                           d_sargs.ptr + static_cast<size_t>(turn) * P, c->slots.ptr);
        ASP_HIP_TRY(hipGetLastError());
        ASP_HIP_TRY(hipEventRecord(c->swept[which], cs));
      }
    }
    for (auto &c : classes) ASP_HIP_TRY(hipStreamWaitEvent(os0, c->swept[(chunks - 1) % nsets], 0));
    ASP_HIP_TRY(hipEventRecord(t_end, os0));
    for (auto &o : order_stream) ASP_HIP_TRY(hipStreamSynchronize(o.stream));
    for (auto &c : classes) ASP_HIP_TRY(hipStreamSynchronize(c->stream.stream));
    // ---- status of every problem; larger capacities and once more if any ran out ----
    bool again = false;
    for (ShuffledRun *r : runs) {
      ASP_HIP_TRY(hipMemcpy(r->status, r->d_status.ptr, sizeof r->status, hipMemcpyDeviceToHost));
      bool mine = false;
      ASP_TRY(r->grow(&mine));
      again = again || mine;
    }
    if (!again) break;
  }
  float ms = 0.0f;
  ASP_HIP_TRY(hipEventElapsedTime(&ms, t_begin, t_end));
  if (sweep_ms) *sweep_ms += ms;
  for (ShuffledRun *r : runs) {
    r->p->last_sweep_ms = ms / static_cast<float>(P);
    r->p->last_total_ms = ms / static_cast<float>(P);
  }
  return ASP_OK;
}

}  // namespace

namespace asp {

// The shuffled items of asp_sa_anneal_batch (csrc/sa_sweep.hip): every item is exactly its own
// asp_sa_anneal_shuffled call.  Items are grouped by their number of sweeps; the problems of a
// group share launches (run_shuffled_group), a group of one takes the single-problem path.
int sa_shuffled_batch(asp_sa_batch_item const *items, const uint32_t *which, uint32_t count, float *sweep_ms) {
  // ASP_SHUFFLED_HOST_TIMING=1: wall-clock phases of this function on stderr (development aid)
  const bool host_timing = std::getenv("ASP_SHUFFLED_HOST_TIMING") != nullptr;
  auto phase_start = std::chrono::steady_clock::now();
  auto phase = [&](const char *name) {
    if (!host_timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "  shuffled batch: %-18s %8.2f ms\n", name,
                 std::chrono::duration<double, std::milli>(now - phase_start).count());
    phase_start = now;
  };
  std::vector<std::unique_ptr<ShuffledRun>> runs;
  runs.reserve(count);
  uint64_t budget = 16ull << 30;  // bytes of visiting orders for the whole batch, per buffer set
  if (const char *env = std::getenv("ASP_SHUFFLED_BATCH_BYTES")) budget = std::strtoull(env, nullptr, 10);
  // chains per workgroup for the whole batch: as many as still leave two workgroups per compute unit
  int m = 1;
  bool saturates = false;  // more groups of chains than the chip holds workgroups (two per compute unit)
  if (count > 1) {
    const int num_cus = items[which[0]].plan ? items[which[0]].plan->num_cus : 256;
    for (int cand : {4, 2}) {
      uint64_t groups = 0;
      for (uint32_t k = 0; k < count; ++k) groups += (items[which[k]].repetitions + cand - 1) / cand;
      if (groups >= 2ull * static_cast<uint64_t>(num_cus)) {
        m = cand;
        saturates = true;
        break;
      }
    }
  }
  if (const char *env = std::getenv("ASP_SHUFFLED_BATCH_M")) {  // (development: scan the chains per group of a batch)
    const int forced = std::atoi(env);
    if (forced == 1 || forced == 2 || forced == 4) m = forced;
  }
  int big_m = 8;  // chains per workgroup of the clusters beyond the word layout, in a batch that fills the chip
  if (const char *env = std::getenv("ASP_SHUFFLED_BATCH_BIG_M")) {
    const int forced = std::atoi(env);
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) big_m = forced;
  }
  for (uint32_t k = 0; k < count; ++k) {
    const asp_sa_batch_item &it = items[which[k]];
    runs.emplace_back(new ShuffledRun());
    ShuffledRun &r = *runs.back();
    r.p = it.plan;
    r.seed = it.seed;
    r.betas = it.betas;
    r.num_sweeps = it.num_sweeps;
    r.repetitions = it.repetitions;
    r.replica_offset = it.replica_offset;
    r.out_x = it.out_x;
    r.out_e = it.out_e;
    r.budget = std::max<uint64_t>(64ull << 20, std::min<uint64_t>(3ull << 30, budget / count));
    if (count > 1 && it.plan && !it.plan->shuffled_m) {
      r.forced_m = m;
      // A cluster beyond the word layout (a word per spin and four chains: ~3.8e4 spins) in a batch
      // that fills the chip: eight chains per workgroup in the byte layout (setup() falls back to
      // four / one where bytes do not fit either).  The sweep's coupling stream — 12 bytes per
      // coupling, written once and read by EVERY workgroup of the problem — is the traffic of such a
      // model: 350 bytes per spin and sweep over M chains; the real kagome_36 order-2 models (3.5e4 ..
      // 3e5 spins, 64 chains each) moved 87 bytes per flip at M = 4.
      if (saturates && it.plan->host.num_spins * 4ull > it.plan->max_lds * 15 / 16) r.forced_m = big_m;
    }
    r.batch_saturates = saturates;
    if (const char *env = std::getenv("ASP_SHUFFLED_ORDER_THREADS")) {  // (development: scanned 128 .. 1024, no effect)
      r.order_threads_cap = static_cast<uint32_t>(std::max(64l, std::min(1024l, std::strtol(env, nullptr, 10) / 64 * 64)));
    }
    ASP_TRY(r.setup());
  }
  phase("setup");
  // groups of equal ladder length (and equal chains per workgroup: a plan with a forced width
  // keeps it and runs alone)
  std::vector<bool> taken(runs.size(), false);
  for (size_t i = 0; i < runs.size(); ++i) {
    if (taken[i] || runs[i]->trivial) continue;
    std::vector<ShuffledRun *> group;
    for (size_t j = i; j < runs.size(); ++j) {
      if (!taken[j] && !runs[j]->trivial && runs[j]->num_sweeps == runs[i]->num_sweeps &&
          shuffled_batch_kernel_for(runs[j]->m, kBytes, false)) {
        group.push_back(runs[j].get());
        taken[j] = true;
      }
    }
    taken[i] = true;
    if (group.size() >= 2) {
      ASP_TRY(run_shuffled_group(group, sweep_ms));
    } else {
      ShuffledRun &r = *runs[i];
      bool again = true;
      while (again) {
        ASP_TRY(r.enqueue());
        ASP_TRY(r.collect(&again));
      }
    }
  }
  phase("launches + wait");
  for (auto &r : runs) ASP_TRY(r->finish_enqueue());
  for (auto &r : runs) {
    ASP_TRY(r->finish_wait());
    if (sweep_ms && r->p && !r->batched) *sweep_ms += r->p->last_sweep_ms;
  }
  phase("energies + copies");
  runs.clear();
  phase("release");
  return ASP_OK;
}

}  // namespace asp

extern "C" {

int asp_sa_set_shuffled_launch(asp_sa_plan *p, int chains_per_group, int wavefronts) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  if (chains_per_group != 0 && !shuffled_kernel_for(chains_per_group, kBytes)) {
    return asp::set_error(ASP_ERR_INVALID, "chains_per_group must be 0, 1, 2, 4 or 8");
  }
  if (wavefronts < 0 || wavefronts > 8) return asp::set_error(ASP_ERR_INVALID, "wavefronts must be 0..8");
  p->shuffled_m = chains_per_group;
  p->shuffled_waves = wavefronts;
  return ASP_OK;
}

int asp_sa_set_shuffled_teams(asp_sa_plan *p, int teams) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  if (teams < 0 || teams > 2) return asp::set_error(ASP_ERR_INVALID, "teams must be 0 (automatic), 1 or 2");
  p->shuffled_teams = teams;
  return ASP_OK;
}

int asp_sa_last_shuffled(asp_sa_plan const *p, uint32_t *levels, float *order_ms) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  if (levels) *levels = static_cast<uint32_t>(p->last_shuffled_levels);
  if (order_ms) *order_ms = p->last_order_ms;
  return ASP_OK;
}

int asp_sa_last_shuffled_blocks(asp_sa_plan const *p, uint32_t *spins_per_block, uint32_t *workgroups) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  if (spins_per_block) *spins_per_block = 1u << p->last_shuffled_log_s;
  if (workgroups) *workgroups = p->last_shuffled_wgs;
  return ASP_OK;
}

int asp_sa_last_shuffled_fill(asp_sa_plan const *p, double *lane_fill, double *row_fill) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  const double S = static_cast<double>(1u << p->last_shuffled_log_s);
  if (lane_fill) {
    *lane_fill = p->last_shuffled_blocks ? static_cast<double>(p->host.num_spins) / (p->last_shuffled_blocks * S) : 0.0;
  }
  if (row_fill) {
    *row_fill = p->last_shuffled_quads ? static_cast<double>(p->rq_quads) / (p->last_shuffled_quads * S) : 0.0;
  }
  return ASP_OK;
}

int asp_sa_anneal_shuffled(asp_sa_plan *p, uint64_t seed, double const *betas, uint32_t num_sweeps,
                           uint32_t repetitions, uint32_t replica_offset, uint64_t const *x0,
                           uint64_t *out_x, double *out_e) {
  asp_clear_error();
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  ASP_TRY(asp::bind_device());
  ShuffledRun run;
  run.p = p;
  run.seed = seed;
  run.betas = betas;
  run.num_sweeps = num_sweeps;
  run.repetitions = repetitions;
  run.replica_offset = replica_offset;
  run.x0 = x0;
  run.out_x = out_x;
  run.out_e = out_e;
  ASP_TRY(run.setup());
  bool again = true;
  while (again) {
    ASP_TRY(run.enqueue());
    ASP_TRY(run.collect(&again));
  }
  ASP_TRY(run.finish_enqueue());
  return run.finish_wait();
}

}  // extern "C"
