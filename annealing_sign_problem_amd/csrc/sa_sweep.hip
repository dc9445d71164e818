// Metropolis single-spin-flip annealing sweep on gfx950 (specification ASP-SA-1,
// DESIGN.md §4): replaces ising_glass_annealer.anneal at the reference's call
// sites annealing_sign_problem/common.py:242-248 and
// experiments/full_hilbert_space.py:212-218.
//
// Mapping to the machine
//   * one workgroup = one GROUP of M replicas (M in {1,2,4,8}); the whole anneal
//     (all sweeps) is ONE launch, the spins never leave LDS;
//   * LDS: one byte per (padded) spin position, bit m = sign bit of replica m
//     (1 means s = -1), so one ds_read_u8 serves all M replicas of a neighbour
//     (kBytes); a 32-bit word per position for M = 4 on small clusters, where
//     the sign of a term is one SDWA instruction (kWide); beyond the capacity of
//     bytes one BIT per position and one replica per workgroup, flips applied
//     by a wavefront ballot (kBits), and beyond 1.3e6 spins the same bit words
//     in HBM (kGlobal);
//   * frozen sweeps: row sums cached in HBM and re-used while no neighbour of a
//     block flipped (dirty bytes), blocks whose proposals are all certain
//     rejections skipped outright (inert bytes);
//   * few chains on a large cluster: k_sa_sweep_team spreads ONE chain over 2-8
//     workgroups that exchange 64-bit flip words behind a device-scope barrier;
//   * a wavefront owns a 64-row block of one colour class: lane = spin.  Same
//     colour means no couplings inside the block, so the 64 x M proposals of a
//     block are independent and are decided at once;
//   * couplings stream from the quad-interleaved sliced ELL: per four terms a
//     lane issues three 16-byte loads (every wavefront instruction covers 1 KiB
//     of contiguous memory), shared by M replicas, one quad prefetched ahead;
//   * dE is an f64 sum in fixed row order (bit-exact against the oracle), the
//     acceptance uses a counter-based Philox4x32-10 word per (spin, sweep,
//     replica) and a fixed-sequence exp reached through a hardware-exp filter,
//     accepted flips are XOR-ed into LDS;
//   * the running energy of each replica is tracked exactly in 2^-S fixed point
//     (integer adds commute, so the parallel reduction is deterministic); the
//     best configuration is snapshotted to HBM at sweep granularity;
//   * DESCENT instantiation: strict-descent sweeps without random numbers (the
//     greedy solver's relaxation).
// f64-VALU- and vector-memory-bound integer+f64 work: no MFMA.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <initializer_list>
#include <mutex>
#include <vector>

#include "asp_common.hpp"
#include "sa_device.hpp"
#include "sa_internal.hpp"
#include "sa_plan.hpp"

namespace {

using asp::DeviceBuffer;
using asp::kDummySpin;
using asp::upload_vector;
using namespace asp::dev;  // Philox, expneg, the accept filters, coupling quads (sa_device.hpp)

struct SweepArgs {
  const uint32_t *color_block_start;  // num_colors + 1
  const uint32_t *block_width;        // num_blocks
  const uint64_t *ell_off;            // num_blocks + 1 (slabs)
  const uint32_t *ell_col;
  const double *ell_val;
  const uint32_t *spin_of_pos;  // num_blocks * 64
  const double *field_pos;      // num_blocks * 64
  const double *betas;          // num_sweeps
  const uint64_t *x0_perm;      // num_blocks sign-bit words or nullptr
  uint64_t *best_perm;          // [groups * M][num_blocks] sign-bit words
  long long *tracked;           // [groups * M] best tracked energy (fixed point)
  unsigned long long *accepted;  // [groups * M] accepted flips
  uint64_t seed;
  double scale;  // 2^S
  uint32_t num_colors, num_blocks, num_sweeps, replica_first;
  // Field cache (nullptr = off): [group][block][m][lane] local fields (the row sums `acc`) of the
  // last evaluation of every block, valid while the block's dirty byte in LDS is clear.
  double *field_cache;
  uint32_t cache_enter_flips;  // switch the cache on after a sweep with fewer flips than this
  uint64_t *spin_words;        // kGlobal: [groups][num_blocks] sign-bit words in HBM
  long long *trace;            // nullptr or [groups * M][num_sweeps + 1] tracked energy per sweep
};

// One 64-spin word of the bit-packed layouts; device-scope accesses when it lives in HBM and
// other wavefronts of the workgroup read it after the colour barrier.
template <bool GLOBAL>
__device__ __forceinline__ uint64_t load_word(const uint64_t *p) {
  if constexpr (GLOBAL) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    return *p;
  }
}
template <bool GLOBAL>
__device__ __forceinline__ void store_word(uint64_t *p, uint64_t v) {
  if constexpr (GLOBAL) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    *p = v;
  }
}

// (`Args` is SweepArgs, or SweepArgs in the constant address space: see k_sa_sweep_batch)
template <int M, int LAYOUT, typename Args>
__device__ __forceinline__ void snapshot(const uint8_t *spins, const Args &a, uint32_t group,
                                         uint32_t mask) {
  if constexpr (LAYOUT == kWide) {  // a wavefront per block: one ballot per replica
    const uint32_t *wide = reinterpret_cast<const uint32_t *>(spins);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t b = threadIdx.x >> 6; b < a.num_blocks; b += blockDim.x >> 6) {
      const uint32_t w = wide[b * 64u + lane];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        if (!((mask >> m) & 1u)) continue;  // workgroup-uniform
        const uint64_t word = __ballot((w >> (8 * m + 7)) & 1u);
        if (lane == 0) {
          a.best_perm[(static_cast<uint64_t>(group) * M + m) * a.num_blocks + b] = word;
        }
      }
    }
    return;
  }
  if constexpr (LAYOUT == kNibbles) {  // a wavefront per block: one ballot per replica
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t b = threadIdx.x >> 6; b < a.num_blocks; b += blockDim.x >> 6) {
      const uint32_t nibble = static_cast<uint32_t>(spins[b * 32u + (lane >> 1)]) >> ((lane & 1u) << 2);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        if (!((mask >> m) & 1u)) continue;  // workgroup-uniform
        const uint64_t word = __ballot((nibble >> m) & 1u);
        if (lane == 0) {
          a.best_perm[(static_cast<uint64_t>(group) * M + m) * a.num_blocks + b] = word;
        }
      }
    }
    return;
  }
  if constexpr (LAYOUT == kBits || LAYOUT == kGlobal) {  // the words already are the sign bits
    const uint64_t *words = reinterpret_cast<const uint64_t *>(spins);
    for (uint32_t w = threadIdx.x; w < a.num_blocks; w += blockDim.x) {
      uint64_t word;
      if constexpr (LAYOUT == kGlobal) {
        word = __hip_atomic_load(words + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        word = words[w];
      }
      a.best_perm[static_cast<uint64_t>(group) * a.num_blocks + w] = word;
    }
    return;
  }
  for (uint32_t w = threadIdx.x; w < a.num_blocks; w += blockDim.x) {
    const uint4 *src = reinterpret_cast<const uint4 *>(spins + 64u * w);
    uint4 q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = src[j];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (!((mask >> m) & 1u)) continue;
      uint64_t word = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t nib = gather_bit4(q[j].x, m) | (gather_bit4(q[j].y, m) << 4) |
                             (gather_bit4(q[j].z, m) << 8) | (gather_bit4(q[j].w, m) << 12);
        word |= static_cast<uint64_t>(nib) << (16 * j);
      }
      a.best_perm[(static_cast<uint64_t>(group) * M + m) * a.num_blocks + w] = word;
    }
  }
}

// DESCENT = true: strict-descent sweeps (accept iff dE < 0, no random numbers), used by the
// greedy solver's relaxation; the final configuration is snapshotted after every sweep.
// The whole anneal of one group of M replicas by one workgroup; `group` = index of the group
// inside its problem (k_sa_sweep: the workgroup id; k_sa_sweep_batch: looked up in a table).
template <int M, bool DESCENT, int LAYOUT, typename Args>
__device__ __forceinline__ void sa_sweep_body(const Args &a, const uint32_t group) {
  constexpr bool GLOBAL = LAYOUT == kGlobal;
  constexpr bool PACKED = LAYOUT == kBits || GLOBAL;  // one bit per position
  constexpr bool WIDE = LAYOUT == kWide;
  constexpr bool NIBBLES = LAYOUT == kNibbles;
  static_assert(!PACKED || M == 1, "the bit-packed layouts hold one replica");
  static_assert(!NIBBLES || (M <= 4 && !DESCENT), "the nibble layout holds up to four replicas");
  static_assert(!WIDE || (M <= 4 && !DESCENT), "the wide layout holds up to four replicas");
  extern __shared__ __align__(16) uint8_t lds[];
  // kGlobal: this workgroup's bit words live in HBM, the LDS holds the bookkeeping only
  uint8_t *spins = GLOBAL ? reinterpret_cast<uint8_t *>(a.spin_words +
                                                       static_cast<uint64_t>(group) * a.num_blocks)
                          : lds;
  // bytes of the spin area per block: 64 (a byte per position), 32 (a nibble), 8 (a bit) or 256 (a word)
  const uint32_t P = GLOBAL ? 0u : a.num_blocks * (PACKED ? 8u : (WIDE ? 256u : (NIBBLES ? 32u : 64u)));
  // P is a multiple of 64.  Per replica m: delta[m] = energy change of the running
  // sweep, book[m] = current tracked energy, book[8+m] = best, book[16+m] = accepted flips
  long long *delta = reinterpret_cast<long long *>(lds + P);
  long long *book = delta + 8;
  uint32_t *improved_flag = reinterpret_cast<uint32_t *>(book + 24);
  uint2 *meta = reinterpret_cast<uint2 *>(book + 26);  // per block {first ELL slab, width}
  // cache control: [0] flips of the running sweep, [1] 1 while the field cache is in use,
  // [2] 1 when the cache was just switched on (dirty bytes must be set);
  // then one dirty byte per block (bit m: replica m's cached fields are stale)
  // (the bit-packed layout keeps no per-block arrays in LDS besides the spin words: block
  // metadata is read from HBM with scalar loads, the field cache is not available)
  uint32_t *cache_ctl = reinterpret_cast<uint32_t *>(meta + (PACKED ? 0u : a.num_blocks));
  uint8_t *dirty = reinterpret_cast<uint8_t *>(cache_ctl + 4);
  // one "inert" byte per block, meaningful while the dirty byte is clear: at the block's last
  // evaluation every proposal was a certain rejection (beta * dE >= 23 -> expneg = 0, or
  // dE >= 0 in descent mode) and beta has not decreased since, so the visit can be skipped
  uint8_t *inert = dirty + ((a.num_blocks + 15u) & ~15u);
  const bool cache_available = !PACKED && a.field_cache != nullptr;

#if ASP_ABS_LDS
  // accumulate_quad addresses the spin bytes absolutely: the dynamic LDS block must be the
  // first (this kernel declares no static LDS)
  if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t *)lds) != 0) {
    __builtin_trap();
  }
#endif
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const uint32_t waves = blockDim.x >> 6;
  const uint32_t r0 = a.replica_first + group * M;
  const uint32_t key0 = static_cast<uint32_t>(a.seed);
  const uint32_t key1 = static_cast<uint32_t>(a.seed >> 32);

  // ---- initial configuration ---- (a wavefront initialises whole 64-position blocks)
  for (uint32_t b0 = tid >> 6; b0 < a.num_blocks; b0 += blockDim.x >> 6) {
    const uint32_t p = b0 * 64u + (tid & 63u);
    const uint32_t spin = a.spin_of_pos[p];
    uint32_t byte = 0;
    if (spin != kDummySpin) {
      if (a.x0_perm != nullptr) {
        byte = ((a.x0_perm[p >> 6] >> (p & 63u)) & 1ull) ? ((1u << M) - 1u) : 0u;  // replica mask
      } else {
        Philox4 rnd{};
        uint32_t have = 0xFFFFFFFFu;
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const uint32_t r = r0 + m;
          if (m == 0 || (r >> 2) != have) {
            have = r >> 2;
            rnd = philox4x32_10(spin, 0xFFFFFFFFu, have, 0u, key0, key1);
          }
          const uint32_t up = pick_word(rnd, r & 3u) & 1u;  // 1 -> s = +1 -> sign bit 0
          byte |= (up ^ 1u) << m;
        }
      }
    }
    if constexpr (PACKED) {
      const uint64_t word = __ballot(byte & 1u);
      if ((tid & 63u) == 0) store_word<GLOBAL>(reinterpret_cast<uint64_t *>(spins) + b0, word);
    } else if constexpr (WIDE) {
      reinterpret_cast<uint32_t *>(spins)[p] = spread_mask(byte);
    } else if constexpr (NIBBLES) {
      const uint32_t upper = __shfl_xor(byte, 1);  // the odd lane's nibble, for the even lane
      if ((tid & 1u) == 0) spins[p >> 1] = static_cast<uint8_t>(byte | (upper << 4));
    } else {
      spins[p] = static_cast<uint8_t>(byte);
    }
  }
  if constexpr (!PACKED) {
    for (uint32_t b = tid; b < a.num_blocks; b += blockDim.x) {
      meta[b] = make_uint2(static_cast<uint32_t>(a.ell_off[b]), a.block_width[b]);
    }
  }
  if (tid < 32) delta[tid] = 0;  // delta[8] + book[24]
  if (tid == 0) {
    *improved_flag = 0;
    cache_ctl[0] = 0;
    cache_ctl[1] = 0;
    cache_ctl[2] = 0;
  }
  __syncthreads();
  snapshot<M, LAYOUT>(spins, a, group, (1u << M) - 1u);
  __syncthreads();

  if (a.trace != nullptr && tid < M) {
    a.trace[(static_cast<uint64_t>(group) * M + tid) * (a.num_sweeps + 1ull)] = 0;
  }
  uint32_t one_hi[4] = {0x3FF00000u, 0x3FF00000u, 0x3FF00000u, 0x3FF00000u};
  for (uint32_t t = 0; t < a.num_sweeps; ++t) {
    const double beta = a.betas[t];
    // wave-uniform: cached fields are in use during this sweep
    const bool cached = cache_available && __builtin_amdgcn_readfirstlane(cache_ctl[1]) != 0;
    if (cached && t > 0 && beta < a.betas[t - 1]) {
      // certain rejections are only certain for non-decreasing beta (workgroup-uniform branch)
      for (uint32_t b = tid; b < a.num_blocks; b += blockDim.x) inert[b] = 0;
      __syncthreads();
    }
    long long q_acc[M];
    uint32_t n_acc[M];  // accepted flips of this lane in this sweep (< 2^32 blocks per sweep)
#pragma unroll
    for (int m = 0; m < M; ++m) {
      q_acc[m] = 0;
      n_acc[m] = 0;
    }

    for (uint32_t c = 0; c < a.num_colors; ++c) {
      const uint32_t b_begin = a.color_block_start[c];
      const uint32_t b_end = a.color_block_start[c + 1];
#if ASP_SERPENTINE
      // Serpentine assignment: blocks of a colour are sorted by descending width, so a plain
      // round-robin would always hand wave 0 the widest block of every round.  Any assignment
      // gives the same bits (blocks of one colour are independent).
      for (uint32_t round = 0;; ++round) {
        const uint32_t slot = (round & 1u) ? (waves - 1u - wave) : wave;
        const uint32_t b = b_begin + round * waves + slot;
        if (round * waves >= b_end - b_begin) break;
        if (b >= b_end) continue;
#else
      for (uint32_t b = b_begin + wave; b < b_end; b += waves) {
#endif
        const uint32_t p = b * 64u + lane;
        bool reuse = false;
        if (cached) {
          reuse = (__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(dirty[b])) &
                   ((1u << M) - 1u)) == 0u;
#if ASP_INERT_SKIP
          // fields unchanged and every proposal certain to be rejected again: nothing to do
          if (reuse && __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(inert[b])) != 0u) {
            continue;
          }
#endif
        }
        // {first slab, width}: one broadcast LDS read (two scalar loads when bit-packed)
        const uint2 info =
            PACKED ? make_uint2(static_cast<uint32_t>(a.ell_off[b]), a.block_width[b]) : meta[b];
        // wave-uniform by construction; readfirstlane makes the loop control scalar
        const uint32_t quads = __builtin_amdgcn_readfirstlane(info.y) >> 2;
        // info.x = first slab of the block (a multiple of 4): quad index = slab / 4
        const uint64_t first_quad = __builtin_amdgcn_readfirstlane(info.x) >> 2;
        const uint4 *cptr = reinterpret_cast<const uint4 *>(a.ell_col) + first_quad * 64u + lane;
        const double2 *vptr =
            reinterpret_cast<const double2 *>(a.ell_val) + first_quad * 128u + lane;
        // issued now, consumed after the row sum: their latency hides under the k-loop
        const uint32_t spin = a.spin_of_pos[p];
        const double h = a.field_pos[p];
        double acc[M];
        // Field cache: when the workgroup is in cached mode and no neighbour of this block's
        // spins has flipped since the block was last evaluated (dirty byte clear), the row
        // sums are read back from HBM — the very same f64 values the k-loop would produce.
        double *cache_row = nullptr;
        if (cached) {
          cache_row = a.field_cache +
                      ((static_cast<uint64_t>(group) * a.num_blocks + b) * M) * 64u + lane;
        }
        if (reuse) {
#pragma unroll
          for (int m = 0; m < M; ++m) acc[m] = cache_row[m * 64];
        } else {
#pragma unroll
          for (int m = 0; m < M; ++m) acc[m] = 0.0;
#if ASP_ABL_NO_KLOOP
          const uint32_t quads_run = 0;
#else
          const uint32_t quads_run = quads;
#endif
          // k-loop, prefetch distance one: the next quad's three 16-byte loads are in flight
          // while the current quad is gathered from LDS and accumulated, in the oracle's order
          // k = 0, 1, 2, ...  Loop control is scalar and the body has no conditional loads
          // (hipcc would otherwise drain the queue with vmcnt(0) at the loop header);
          // sched_barrier keeps each load group ahead of the accumulate it overlaps.  For an
          // even quad count the last load reads one quad past the block — the next block's
          // first slabs or the tail padding the plan appends — and is never consumed.
          Quad qa, qb;
          load_quad(qa, cptr, vptr, 0);
          uint32_t i = 0;
          for (; i + 2 <= quads_run; i += 2) {
            load_quad(qb, cptr, vptr, i + 1);
            __builtin_amdgcn_sched_barrier(0);
            accumulate_quad<M, LAYOUT>(qa, spins, acc, one_hi);
            __builtin_amdgcn_sched_barrier(0);
            load_quad(qa, cptr, vptr, i + 2);
            __builtin_amdgcn_sched_barrier(0);
            accumulate_quad<M, LAYOUT>(qb, spins, acc, one_hi);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (i < quads_run) accumulate_quad<M, LAYOUT>(qa, spins, acc, one_hi);
          if (cached) {
#pragma unroll
            for (int m = 0; m < M; ++m) cache_row[m * 64] = acc[m];
            if (lane == 0) dirty[b] = 0;  // nobody marks a block during its own colour step
          }
        }
        const bool valid = spin != kDummySpin;
        uint32_t own;
        if constexpr (PACKED) {
          own = static_cast<uint32_t>(
              (load_word<GLOBAL>(reinterpret_cast<const uint64_t *>(spins) + b) >> lane) & 1ull);
        } else if constexpr (WIDE) {
          own = reinterpret_cast<const uint32_t *>(spins)[p];
        } else if constexpr (NIBBLES) {
          own = (static_cast<uint32_t>(spins[p >> 1]) >> ((p & 1u) << 2)) & 15u;
        } else {
          own = spins[p];
        }
        uint32_t flip = 0;
        bool open = false;  // some proposal of this lane is not a certain rejection
        bool need = false;  // some proposal of this lane needs a random number
        double de[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const double g = __dadd_rn(acc[m], h);
          const bool negative = (own >> (WIDE ? 8 * m + 7 : m)) & 1u;  // s = -1
          de[m] = __dmul_rn(negative ? 2.0 : -2.0, g);
          if constexpr (DESCENT) {
            open = open || (valid && de[m] < 0.0);
          } else {
            const bool maybe = valid && !(__dmul_rn(beta, de[m]) >= 23.0);  // not a certain rejection
            open = open || maybe;
#if ASP_EXPERIMENT_GLAUBER
            need = need || maybe;
#else
            need = need || (maybe && !(de[m] <= 0.0));
#endif
          }
        }
        // Random numbers only when some proposal of the block is undecided without one
        // (dE <= 0 is accepted, beta * dE >= 23 rejected, whatever the draw): a wave-uniform
        // branch around the 10 Philox rounds and the exp filter.  Counter-based RNG: skipping
        // a draw changes nothing downstream.
#if ASP_PHILOX_SKIP
        const bool draw = !DESCENT && __ballot(need) != 0ull;
#else
        const bool draw = !DESCENT;
#endif
        uint32_t accept_mask = 0;
        if (draw) {
          Philox4 rnd{};
          uint32_t have = 0xFFFFFFFFu;
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const uint32_t r = r0 + m;
            if (m == 0 || (r >> 2) != have) {
              have = r >> 2;
#if ASP_ABL_NO_PHILOX
              rnd = Philox4{{spin * 2654435761u ^ t, spin ^ (t * 40503u), spin + have, t ^ key0}};
#else
              rnd = philox4x32_10(spin, t, have, 0u, key0, key1);
#endif
            }
            const uint32_t word = pick_word(rnd, r & 3u);
            bool accept;
#if ASP_ABL_NO_ACCEPT
            asm volatile("" ::"v"(de[m]), "v"(word));
            accept = false;
#elif ASP_ABL_NO_EXP
            accept = valid && (de[m] <= 0.0 || word < static_cast<uint32_t>(__dmul_rn(beta, de[m])));
#elif ASP_EXPERIMENT_GLAUBER
            {
              const double x = __dmul_rn(beta, de[m]);
              const double uu = (static_cast<double>(word) + 0.5) * 0x1p-32;
              accept = valid && x < 23.0 && uu < 1.0 / (1.0 + exp(x));
            }
#elif ASP_EXP_FILTER == 2
            accept = valid && (de[m] <= 0.0 || metropolis_accept_word(word, __dmul_rn(beta, de[m])));
#else
            const double u = __dmul_rn(__dadd_rn(static_cast<double>(word), 0.5), 0x1p-32);
#if ASP_EXP_FILTER
            accept = valid && (de[m] <= 0.0 || metropolis_accept(u, __dmul_rn(beta, de[m])));
#else
            accept = valid && (de[m] <= 0.0 || u < expneg(__dmul_rn(beta, de[m])));
#endif
#endif
            accept_mask |= (accept ? 1u : 0u) << m;
          }
        } else {
#pragma unroll
          for (int m = 0; m < M; ++m) {
            // (no draw needed: DESCENT, or every proposal decided; with the heat-bath experiment a
            // skipped draw means every proposal was a certain rejection)
            const bool accept = valid && (DESCENT ? de[m] < 0.0
                                                  : (ASP_EXPERIMENT_GLAUBER ? false : de[m] <= 0.0));
            accept_mask |= (accept ? 1u : 0u) << m;
          }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
          if ((accept_mask >> m) & 1u) {
            flip |= 1u << m;
            // rint(dE * 2^S) as int64: |dE * 2^S| < 2^51 by the plan's choice of S, so adding
            // 1.5 * 2^52 leaves the rounded integer in the mantissa (ties to even, = rint)
#if ASP_MAGIC_RINT
            q_acc[m] += __double_as_longlong(__dadd_rn(__dmul_rn(de[m], a.scale), 0x1.8p52)) -
                        0x4338000000000000ll;
#else
            q_acc[m] += static_cast<long long>(__builtin_rint(__dmul_rn(de[m], a.scale)));
#endif
            n_acc[m] += 1;
          }
        }
        if constexpr (PACKED) {
          // the block's 64 proposals decided: one XOR of the ballot into the block's word
          const uint64_t flips = __ballot(flip != 0);
          if (lane == 0 && flips != 0) {
            uint64_t *word = reinterpret_cast<uint64_t *>(spins) + b;
            store_word<GLOBAL>(word, load_word<GLOBAL>(word) ^ flips);
          }
        } else if constexpr (WIDE) {
          if (flip) reinterpret_cast<uint32_t *>(spins)[p] = own ^ spread_mask(flip);
        } else if constexpr (NIBBLES) {
          // the neighbouring lane owns the other nibble of the byte and may flip in the same
          // instruction: an LDS atomic on the word (eight positions) instead of a byte store
          if (flip) atomicXor(reinterpret_cast<uint32_t *>(spins) + (p >> 3), flip << ((p & 7u) << 2));
        } else {
          if (flip) spins[p] = static_cast<uint8_t>(own ^ flip);
        }
#if ASP_INERT_SKIP
        if (cached) {
          const bool none_open = __ballot(open) == 0ull;
          if (lane == 0) inert[b] = none_open ? 1 : 0;
        }
#endif
        if (cached && __ballot(flip != 0) != 0ull) {
          // Every neighbour of a flipped spin sits in a block of ANOTHER colour: mark those
          // blocks stale for the replicas that flipped.  The row's columns are streamed again
          // (columns only); in cached mode flips are rare by construction.
          uint32_t *dirty_words = reinterpret_cast<uint32_t *>(dirty);
          for (uint32_t q = 0; q < quads; ++q) {
            const uint4 c4 = cptr[q * 64u];
            if (flip) {
              const uint32_t cols[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                // padding entries point at the lane itself; wide columns are byte addresses
                if (cols[j] == (WIDE ? p * 4u : p)) continue;
                const uint32_t blk = cols[j] >> (WIDE ? 8 : 6);
                atomicOr(&dirty_words[blk >> 2], flip << (8u * (blk & 3u)));
              }
            }
          }
        }
      }
#if !ASP_ABL_NO_BARRIER
      __syncthreads();
#endif
    }

    // ---- exact (integer) reduction of the sweep's energy change ----
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const long long v = wave_sum_i64(q_acc[m]);
      const long long n = wave_sum_i64(static_cast<long long>(n_acc[m]));
      if (lane == 0 && n != 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&delta[m]),
                  static_cast<unsigned long long>(v));
        atomicAdd(reinterpret_cast<unsigned long long *>(&book[16 + m]),
                  static_cast<unsigned long long>(n));
        if (cache_available) atomicAdd(&cache_ctl[0], static_cast<uint32_t>(n));
      }
    }
    __syncthreads();
    if (cache_available && tid == 0) {
      // Cached mode pays when the flips of a sweep dirty only a fraction of the blocks:
      // enter below `cache_enter_flips` flips per sweep, leave above twice that.
      const uint32_t flips = cache_ctl[0];
      cache_ctl[0] = 0;
      const bool was = cache_ctl[1] != 0;
      const bool now = was ? flips < 2u * a.cache_enter_flips : flips < a.cache_enter_flips;
      cache_ctl[1] = now ? 1u : 0u;
      cache_ctl[2] = (now && !was) ? 1u : 0u;  // entering: every block starts stale
    }
    if (tid < M) {
      const long long e = book[tid] + delta[tid];
      book[tid] = e;
      delta[tid] = 0;
      if (a.trace != nullptr) {
        a.trace[(static_cast<uint64_t>(group) * M + tid) * (a.num_sweeps + 1ull) + t + 1u] = e;
      }
      if (e < book[8 + tid]) {
        book[8 + tid] = e;
        atomicOr(improved_flag, 1u << tid);
      }
    }
    __syncthreads();
    const uint32_t improved = DESCENT ? ((1u << M) - 1u) : *improved_flag;
    if (improved) snapshot<M, LAYOUT>(spins, a, group, improved);
    if (cache_available && cache_ctl[2] != 0) {
      for (uint32_t b = tid; b < a.num_blocks; b += blockDim.x) dirty[b] = 0xFF;
    }
    __syncthreads();
    if (tid == 0) {
      *improved_flag = 0;  // next write to it is two barriers away
      if (cache_available) cache_ctl[2] = 0;
    }
  }

  if (tid < M) {
    a.tracked[static_cast<uint64_t>(group) * M + tid] = book[8 + tid];
    a.accepted[static_cast<uint64_t>(group) * M + tid] =
        static_cast<unsigned long long>(book[16 + tid]);
  }
}

template <int M, bool DESCENT, int LAYOUT>
__global__ __launch_bounds__(ASP_MAX_THREADS) void k_sa_sweep(SweepArgs a) {
  sa_sweep_body<M, DESCENT, LAYOUT>(a, blockIdx.x);
}

// Many PROBLEMS in one launch (asp_sa_anneal_batch): workgroup -> (problem, group of M replicas)
// through a slot table, the problem's SweepArgs through a descriptor table in HBM (scalar
// loads: the slot is workgroup-uniform).  Slots are laid out per XCD — workgroup i runs on XCD
// i mod 8 — so that the groups of one problem share that XCD's L2 copy of its couplings, and in
// descending order of work inside an XCD (longest first, the tail stays short).  Every chain is
// bit-identical to the one its own single-problem launch produces: the body is the same and
// results never depend on the launch geometry.
struct BatchSlot {
  uint32_t problem;  // 0xFFFFFFFF: padding slot
  uint32_t group;
};
struct BatchArgs {
  const SweepArgs *problems;
  const BatchSlot *slots;  // [8][slots_per_xcd]
  uint32_t slots_per_xcd;
};

template <int M, int LAYOUT>
__global__ __launch_bounds__(ASP_MAX_THREADS) void k_sa_sweep_batch(BatchArgs b) {
  const BatchSlot slot = b.slots[(blockIdx.x & 7u) * b.slots_per_xcd + (blockIdx.x >> 3)];
  const uint32_t problem = __builtin_amdgcn_readfirstlane(slot.problem);
  if (problem == 0xFFFFFFFFu) return;
  // The descriptor is read through the CONSTANT address space, like kernel arguments: the
  // compiler may then re-load a field where it needs it instead of keeping all forty of them in
  // registers (as a by-value copy it spilled SGPRs into VGPR lanes and VGPRs to scratch:
  // 128 VGPRs + 68 B of scratch against the single-problem kernel's 112 and none; +8 %).
  using ConstArgs = const SweepArgs __attribute__((address_space(4)));
  ConstArgs *a = reinterpret_cast<ConstArgs *>(reinterpret_cast<uintptr_t>(b.problems + problem));
  sa_sweep_body<M, false, LAYOUT>(*a, __builtin_amdgcn_readfirstlane(slot.group));
}

// ---------------------------------------------------------------------------
// Team sweep: ONE chain spread over G workgroups (few chains on a large cluster)
// ---------------------------------------------------------------------------
// With fewer chains than compute units a chain bound to one workgroup leaves most of the chip
// idle and pays ceil(blocks of a colour / 16) rounds per colour step.  Here the G workgroups of a
// team each keep the whole configuration (bit-packed, LDS), visit every G-th slice of a colour's
// blocks, publish the 64-bit flip word of each block they visited, meet at a device-scope
// barrier and XOR the other members' flip words into their own copy.  Energy bookkeeping is
// summed over the team through parity-buffered atomics.  All workgroups must be resident together
// (the launcher keeps the grid within the CU count and serialises team launches); the barrier
// carries a watchdog so that neither a bug nor a busy device can hang the GPU — on a timeout
// the call is repeated without teams.  Chains are bit-identical to k_sa_sweep's.

// team launches of the process that the barrier's watchdog cut short (asp_sa_team_watchdog_trips)
std::atomic<uint64_t> g_team_watchdog_trips{0};

struct TeamArgs {
  SweepArgs s;
  uint32_t team_size;            // G
  uint32_t num_teams;            // = chains of the launch
  unsigned long long *arrivals;  // [num_teams] barrier counters (monotone)
  uint64_t *flips;               // [num_teams][num_blocks] flip words of the running colour step
  long long *sums;               // [num_teams][3 rotating slots][2] {dq, accepted} of a sweep
  uint32_t *abort;               // set by the watchdog
  uint32_t spin_limit;           // barrier polls before the watchdog gives up
};

// Barrier polls before the watchdog gives up and the call is repeated without teams: ~ 5 s (a
// barrier normally completes in ~ 5 us; members can only be kept waiting by other kernels
// holding compute units — team launches themselves take turns).
constexpr uint32_t kTeamSpinLimit = 1u << 22;

__device__ __forceinline__ void team_barrier(const TeamArgs &ta, unsigned long long *counter,
                                             unsigned long long &target) {
  // The exchange buffers are fine-grained (uncached, coherent across XCDs) and only touched with
  // device-scope atomics, so no cache write-back/invalidate is needed — a release/acquire fence
  // at agent scope flushes the whole L2 of the XCD on this chip and costs ~20 us.  Ordering:
  // every wavefront first waits for the acknowledgement of its own outstanding stores and
  // atomics (a workgroup-scope barrier alone does not: within a CU the vector L1 is shared, so
  // hipcc emits no vmcnt wait for it), then the workgroup barrier, then thread 0 announces the
  // arrival.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  target += ta.team_size;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t polls = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (__hip_atomic_load(ta.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
      if (++polls > ta.spin_limit) {
        __hip_atomic_store(ta.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(ASP_TEAM_SLEEP);
    }
  }
  __syncthreads();
}

// DESCENT as in k_sa_sweep: accept iff dE < 0, no random numbers, snapshot after every sweep.
template <bool DESCENT>
__global__ __launch_bounds__(1024) void k_sa_sweep_team(TeamArgs ta) {
  const SweepArgs &a = ta.s;
  extern __shared__ __align__(16) uint8_t lds[];
  uint64_t *words = reinterpret_cast<uint64_t *>(lds);  // sign bits, one word per block
  // book: [0] current tracked energy, [1] best, [2] this workgroup's dq, [3] its accepted flips,
  // [4] accepted flips of the chain so far
  long long *book = reinterpret_cast<long long *>(lds + static_cast<size_t>(a.num_blocks) * 8u);
  uint32_t *improved = reinterpret_cast<uint32_t *>(book + 5);
  // ctl[0]: 1 while the team tracks which blocks are untouched ("tracking": few flips per
  // sweep), ctl[1]: 1 in the sweep tracking was switched on.  Then per block a dirty byte (a
  // neighbour flipped since the block's last evaluation) and an inert byte (its last evaluation
  // was all certain rejections) — the bookkeeping of k_sa_sweep's cached mode without the cached
  // fields: a clean inert block is skipped, everything else is evaluated in full.
  uint32_t *ctl = improved + 1;
  uint8_t *dirty = reinterpret_cast<uint8_t *>(ctl + 3);
  uint8_t *inert = dirty + ((a.num_blocks + 15u) & ~15u);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, waves = blockDim.x >> 6;
  const uint32_t G = ta.team_size;
  // members of a team are num_teams apart: the same XCD when num_teams is a multiple of 8
  const uint32_t team = blockIdx.x % ta.num_teams, member = blockIdx.x / ta.num_teams;
  const uint32_t r = a.replica_first + team;
  const uint32_t key0 = static_cast<uint32_t>(a.seed), key1 = static_cast<uint32_t>(a.seed >> 32);
  uint64_t *flipbuf = ta.flips + static_cast<uint64_t>(team) * a.num_blocks;
  unsigned long long *counter = ta.arrivals + team;
  unsigned long long target = 0;

  // every member builds the same initial configuration
  for (uint32_t b0 = wave; b0 < a.num_blocks; b0 += waves) {
    const uint32_t p = b0 * 64u + lane;
    const uint32_t spin = a.spin_of_pos[p];
    uint32_t negative = 0;
    if (spin != kDummySpin) {
      if (a.x0_perm != nullptr) {
        negative = static_cast<uint32_t>((a.x0_perm[p >> 6] >> (p & 63u)) & 1ull);
      } else {
        const Philox4 rnd = philox4x32_10(spin, 0xFFFFFFFFu, r >> 2, 0u, key0, key1);
        negative = (pick_word(rnd, r & 3u) & 1u) ^ 1u;
      }
    }
    const uint64_t word = __ballot(negative);
    if (lane == 0) words[b0] = word;
  }
  if (tid < 5) book[tid] = 0;
  if (tid == 0) {
    *improved = 0;
    ctl[0] = 0;
    ctl[1] = 0;
  }
  __syncthreads();
  if (member == 0) {
    for (uint32_t w = tid; w < a.num_blocks; w += blockDim.x) {
      a.best_perm[static_cast<uint64_t>(team) * a.num_blocks + w] = words[w];
    }
  }

  uint32_t one_hi[4] = {0x3FF00000u, 0x3FF00000u, 0x3FF00000u, 0x3FF00000u};
  for (uint32_t t = 0; t < a.num_sweeps; ++t) {
    const double beta = a.betas[t];
    const bool tracking = __builtin_amdgcn_readfirstlane(ctl[0]) != 0;
    if (tracking && t > 0 && beta < a.betas[t - 1]) {
      // certain rejections are only certain for non-decreasing beta (team-uniform branch)
      for (uint32_t b = tid; b < a.num_blocks; b += blockDim.x) inert[b] = 0;
      __syncthreads();
    }
    long long q_acc = 0;
    uint32_t n_acc = 0;
    for (uint32_t c = 0; c < a.num_colors; ++c) {
      const uint32_t b_begin = a.color_block_start[c];
      const uint32_t b_end = a.color_block_start[c + 1];
      for (uint32_t b = b_begin + member * waves + wave; b < b_end; b += G * waves) {
        if (tracking && __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(dirty[b])) == 0u &&
            __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(inert[b])) != 0u) {
          // nothing around this block moved and every proposal was a certain rejection
          if (lane == 0) {
            __hip_atomic_store(flipbuf + b, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          continue;
        }
        const uint32_t p = b * 64u + lane;
        const uint32_t quads = a.block_width[b] >> 2;
        const uint64_t first_quad = a.ell_off[b] >> 2;
        const uint4 *cptr = reinterpret_cast<const uint4 *>(a.ell_col) + first_quad * 64u + lane;
        const double2 *vptr =
            reinterpret_cast<const double2 *>(a.ell_val) + first_quad * 128u + lane;
        const uint32_t spin = a.spin_of_pos[p];
        const double h = a.field_pos[p];
        double acc[1] = {0.0};
        Quad qa, qb;
        load_quad(qa, cptr, vptr, 0);
        uint32_t i = 0;
        for (; i + 2 <= quads; i += 2) {
          load_quad(qb, cptr, vptr, i + 1);
          __builtin_amdgcn_sched_barrier(0);
          accumulate_quad<1, kBits>(qa, lds, acc, one_hi);
          __builtin_amdgcn_sched_barrier(0);
          load_quad(qa, cptr, vptr, i + 2);
          __builtin_amdgcn_sched_barrier(0);
          accumulate_quad<1, kBits>(qb, lds, acc, one_hi);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (i < quads) accumulate_quad<1, kBits>(qa, lds, acc, one_hi);
        const bool valid = spin != kDummySpin;
        const bool negative = (words[b] >> lane) & 1ull;
        const double g = __dadd_rn(acc[0], h);
        const double de = __dmul_rn(negative ? 2.0 : -2.0, g);
        bool accept;
        // this lane's proposal is not a certain rejection
        const bool open = DESCENT ? (valid && de < 0.0)
                                  : (valid && !(__dmul_rn(beta, de) >= 23.0));
        if constexpr (DESCENT) {
          accept = valid && de < 0.0;
        } else {
          const Philox4 rnd = philox4x32_10(spin, t, r >> 2, 0u, key0, key1);
          const uint32_t word = pick_word(rnd, r & 3u);
          accept = valid && (de <= 0.0 || metropolis_accept_word(word, __dmul_rn(beta, de)));
        }
        if (accept) {
          q_acc += __double_as_longlong(__dadd_rn(__dmul_rn(de, a.scale), 0x1.8p52)) -
                   0x4338000000000000ll;
          n_acc += 1;
        }
        const uint64_t flips = __ballot(accept);
        if (lane == 0) {
          if (flips != 0) words[b] ^= flips;
          __hip_atomic_store(flipbuf + b, flips, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tracking) {
          const bool none_open = __ballot(open) == 0ull;
          if (lane == 0) {
            dirty[b] = 0;  // nobody marks a block during its own colour step
            inert[b] = none_open ? 1 : 0;
          }
          if (flips != 0) {
            // the neighbours of a flipped spin sit in blocks of other colours: stale now
            for (uint32_t q = 0; q < quads; ++q) {
              const uint4 c4 = cptr[q * 64u];
              if (accept) {  // (padding entries point at the lane's own position)
                if (c4.x != p) dirty[c4.x >> 6] = 1;
                if (c4.y != p) dirty[c4.y >> 6] = 1;
                if (c4.z != p) dirty[c4.z >> 6] = 1;
                if (c4.w != p) dirty[c4.w >> 6] = 1;
              }
            }
          }
        }
      }
      if (c + 1u == a.num_colors) {
        // the sweep's energy change rides on the last colour's barrier (integers: order-free)
        const long long v = wave_sum_i64(q_acc);
        const long long n = wave_sum_i64(static_cast<long long>(n_acc));
        if (lane == 0 && n != 0) {
          atomicAdd(reinterpret_cast<unsigned long long *>(&book[2]), static_cast<unsigned long long>(v));
          atomicAdd(reinterpret_cast<unsigned long long *>(&book[3]), static_cast<unsigned long long>(n));
        }
        __syncthreads();
        if (tid == 0) {
          long long *mine = ta.sums + (static_cast<uint64_t>(team) * 3u + t % 3u) * 2u;
          if (book[3] != 0) {
            __hip_atomic_fetch_add(&mine[0], book[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&mine[1], book[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          book[2] = 0;
          book[3] = 0;
        }
      }
      team_barrier(ta, counter, target);
      // the other members' flips of this colour step: XOR them in and, when tracking, mark the
      // blocks of the flipped spins' neighbours (a wavefront per flipped block, lane = row)
      if (tracking) {
        for (uint32_t i = wave; i < b_end - b_begin; i += waves) {
          if ((i / waves) % G == member) continue;
          const uint32_t b = b_begin + i;
          const uint64_t flips =
              __hip_atomic_load(flipbuf + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (flips == 0) continue;  // wave-uniform
          if (lane == 0) words[b] ^= flips;
          const uint32_t quads = a.block_width[b] >> 2;
          const uint4 *cptr =
              reinterpret_cast<const uint4 *>(a.ell_col) + (a.ell_off[b] >> 2) * 64u + lane;
          const bool flipped = (flips >> lane) & 1ull;
          for (uint32_t q = 0; q < quads; ++q) {
            const uint4 c4 = cptr[q * 64u];
            if (flipped) {
              const uint32_t p = b * 64u + lane;
              if (c4.x != p) dirty[c4.x >> 6] = 1;
              if (c4.y != p) dirty[c4.y >> 6] = 1;
              if (c4.z != p) dirty[c4.z >> 6] = 1;
              if (c4.w != p) dirty[c4.w >> 6] = 1;
            }
          }
        }
      } else {
        for (uint32_t i = tid; i < b_end - b_begin; i += blockDim.x) {
          if ((i / waves) % G != member) {
            const uint64_t flips = __hip_atomic_load(flipbuf + b_begin + i, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
            if (flips != 0) words[b_begin + i] ^= flips;
          }
        }
      }
      __syncthreads();
    }

    // ---- bookkeeping of the sweep: the team sums were exchanged with the last colour ----
    long long *mine = ta.sums + (static_cast<uint64_t>(team) * 3u + t % 3u) * 2u;
    long long *other = ta.sums + (static_cast<uint64_t>(team) * 3u + (t + 2u) % 3u) * 2u;
    if (tid == 0) {
      const long long dq = __hip_atomic_load(&mine[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long dn = __hip_atomic_load(&mine[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long e = book[0] + dq;
      book[0] = e;
      book[4] += dn;
      const bool better = e < book[1];
      if (better) book[1] = e;
      *improved = (better || DESCENT) ? 1u : 0u;
      // tracking pays while the flips of a sweep touch a fraction of the blocks (the same
      // hysteresis as the field cache; dn is the team-wide count, so all members agree)
      const bool was = ctl[0] != 0;
      const bool now = was ? dn < 2ll * a.cache_enter_flips : dn < static_cast<long long>(a.cache_enter_flips);
      ctl[0] = now ? 1u : 0u;
      ctl[1] = (now && !was) ? 1u : 0u;
      // three rotating slots: the one cleared here was read a sweep ago and is next added to
      // two sweeps from now, with team barriers on either side; one member clears it
      if (member == 0) {
        __hip_atomic_store(&other[0], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&other[1], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    if (*improved != 0u && member == 0) {
      for (uint32_t w = tid; w < a.num_blocks; w += blockDim.x) {
        a.best_perm[static_cast<uint64_t>(team) * a.num_blocks + w] = words[w];
      }
    }
    if (ctl[1] != 0u) {  // tracking starts with the next sweep: every block is stale
      for (uint32_t b = tid; b < a.num_blocks; b += blockDim.x) {
        dirty[b] = 1;
        inert[b] = 0;
      }
    }
    __syncthreads();
  }
  if (member == 0 && tid == 0) {
    a.tracked[team] = book[1];
    a.accepted[team] = static_cast<unsigned long long>(book[4]);
  }
}

// ---------------------------------------------------------------------------
// Energy of packed configurations: E = D + T, T = radix-64
// pairwise tree over the blocks of t_p = s_p (A_p . s / 2 + h_p).
// ---------------------------------------------------------------------------

struct EnergyArgs {
  const uint32_t *block_width;
  const uint64_t *ell_off;
  const uint32_t *ell_col;
  const double *ell_val;
  const double *field_pos;
  const uint64_t *perm_words;  // [count][num_blocks] sign bits
  double *partial;             // [count][num_blocks]
  uint32_t num_blocks;
};

// STAGED: the configuration's sign words are copied to LDS first; otherwise (more blocks than the
// LDS holds) they are gathered from HBM/L2 directly.
template <bool STAGED>
__device__ __forceinline__ void energy_blocks_body(const EnergyArgs &a, const uint32_t r) {
  extern __shared__ __align__(16) uint8_t lds[];
  const uint64_t *mine = a.perm_words + static_cast<uint64_t>(r) * a.num_blocks;
  const uint64_t *bits = mine;
  if constexpr (STAGED) {
    uint64_t *staged = reinterpret_cast<uint64_t *>(lds);
    for (uint32_t w = threadIdx.x; w < a.num_blocks; w += blockDim.x) staged[w] = mine[w];
    __syncthreads();
    bits = staged;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t waves = blockDim.x >> 6;
  for (uint32_t b = threadIdx.x >> 6; b < a.num_blocks; b += waves) {
    const uint32_t quads = a.block_width[b] >> 2;
    const uint64_t first_quad = a.ell_off[b] >> 2;
    const uint4 *cptr = reinterpret_cast<const uint4 *>(a.ell_col) + first_quad * 64u + lane;
    const double2 *vptr = reinterpret_cast<const double2 *>(a.ell_val) + first_quad * 128u + lane;
    double acc = 0.0;
    for (uint32_t q = 0; q < quads; ++q) {
      const uint4 c = cptr[q * 64u];
      const double2 v01 = vptr[q * 128u];
      const double2 v23 = vptr[q * 128u + 64u];
      const uint32_t cs[4] = {c.x, c.y, c.z, c.w};
      const double vs[4] = {v01.x, v01.y, v23.x, v23.y};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t neg = static_cast<uint32_t>((bits[cs[j] >> 6] >> (cs[j] & 63u)) & 1ull);
        acc = __dadd_rn(acc, signed_coupling(vs[j], neg, 0));
      }
    }
    const double g = __dadd_rn(__dmul_rn(0.5, acc), a.field_pos[b * 64u + lane]);
    const bool negative = (bits[b] >> lane) & 1ull;
    const double total = wave_tree_sum_f64(negative ? -g : g);
    if (lane == 0) a.partial[static_cast<uint64_t>(r) * a.num_blocks + b] = total;
  }
}

template <bool STAGED>
__global__ __launch_bounds__(512) void k_sa_energy_blocks(EnergyArgs a) {
  energy_blocks_body<STAGED>(a, blockIdx.x);
}

// R configurations per workgroup: every coupling quad is loaded once and applied to the R staged
// configurations (the chains of one call share the ELL: R times less load traffic than one
// configuration per workgroup).  Per configuration the arithmetic — and so the bits — are those
// of energy_blocks_body.
template <int R>
__global__ __launch_bounds__(512) void k_sa_energy_blocks_multi(EnergyArgs a, uint32_t count) {
  extern __shared__ __align__(16) uint8_t lds[];
  uint64_t *staged = reinterpret_cast<uint64_t *>(lds);  // [R][num_blocks]
  const uint32_t r0 = blockIdx.x * R;
  const uint32_t live = count - r0 < static_cast<uint32_t>(R) ? count - r0 : R;
  const uint64_t *rows = a.perm_words + static_cast<uint64_t>(r0) * a.num_blocks;
  for (uint32_t w = threadIdx.x; w < live * a.num_blocks; w += blockDim.x) staged[w] = rows[w];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t waves = blockDim.x >> 6;
  for (uint32_t b = threadIdx.x >> 6; b < a.num_blocks; b += waves) {
    const uint32_t quads = a.block_width[b] >> 2;
    const uint64_t first_quad = a.ell_off[b] >> 2;
    const uint4 *cptr = reinterpret_cast<const uint4 *>(a.ell_col) + first_quad * 64u + lane;
    const double2 *vptr = reinterpret_cast<const double2 *>(a.ell_val) + first_quad * 128u + lane;
    double acc[R];
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = 0.0;
    for (uint32_t q = 0; q < quads; ++q) {
      const uint4 c = cptr[q * 64u];
      const double2 v01 = vptr[q * 128u];
      const double2 v23 = vptr[q * 128u + 64u];
      const uint32_t cs[4] = {c.x, c.y, c.z, c.w};
      const double vs[4] = {v01.x, v01.y, v23.x, v23.y};
      const uint32_t *halves = reinterpret_cast<const uint32_t *>(staged);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t word = cs[j] >> 5, bit = cs[j] & 31u;  // 32-bit halves: ds_read_b32
#pragma unroll
        for (int k = 0; k < R; ++k) {
          // (rows beyond `live` hold stale LDS: their sums are computed and never stored)
          const uint32_t neg = (halves[k * 2u * a.num_blocks + word] >> bit) & 1u;
          acc[k] = __dadd_rn(acc[k], signed_coupling(vs[j], neg, 0));
        }
      }
    }
    const double field = a.field_pos[b * 64u + lane];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (static_cast<uint32_t>(k) >= live) break;  // workgroup-uniform
      const double g = __dadd_rn(__dmul_rn(0.5, acc[k]), field);
      const bool negative = (staged[k * a.num_blocks + b] >> lane) & 1ull;
      const double total = wave_tree_sum_f64(negative ? -g : g);
      if (lane == 0) a.partial[static_cast<uint64_t>(r0 + k) * a.num_blocks + b] = total;
    }
  }
}

// One wavefront per configuration folds its block sums 64 at a time, in place.
__device__ __forceinline__ void energy_fold_body(double *partial, uint32_t num_blocks,
                                                 double diag_sum, double *out_e, const uint32_t r) {
  const uint32_t lane = threadIdx.x;
  double *level = partial + static_cast<uint64_t>(r) * num_blocks;
  uint32_t n = num_blocks;
  while (n > 1) {
    const uint32_t groups = (n + 63u) / 64u;
    for (uint32_t g = 0; g < groups; ++g) {
      const uint32_t i = g * 64u + lane;
      const double v = i < n ? level[i] : 0.0;
      const double s = wave_tree_sum_f64(v);
      if (lane == 0) level[g] = s;
    }
    n = groups;
  }
  if (lane == 0) out_e[r] = __dadd_rn(diag_sum, num_blocks ? level[0] : 0.0);
}

__global__ __launch_bounds__(64) void k_sa_energy_fold(double *partial, uint32_t num_blocks,
                                                      double diag_sum, double *out_e) {
  energy_fold_body(partial, num_blocks, diag_sum, out_e, blockIdx.x);
}

// The same three steps for the chains of MANY problems in one launch each (asp_sa_anneal_batch):
// workgroup -> (problem, chain) through a table, the problem's pointers through a descriptor.
struct PostProblem {
  EnergyArgs e;  // perm_words / partial: this problem's rows
  const uint32_t *pos_of_spin;
  uint64_t num_spins;
  uint32_t words;
  double diag_sum;
  double *out_e;    // [repetitions]
  uint64_t *out_x;  // [repetitions][words]
};

template <bool STAGED>
__global__ __launch_bounds__(512) void k_sa_energy_blocks_batch(const PostProblem *problems,
                                                               const BatchSlot *chains) {
  const BatchSlot c = chains[blockIdx.x];
  const EnergyArgs a = problems[__builtin_amdgcn_readfirstlane(c.problem)].e;
  energy_blocks_body<STAGED>(a, __builtin_amdgcn_readfirstlane(c.group));
}

__global__ __launch_bounds__(64) void k_sa_energy_fold_batch(const PostProblem *problems,
                                                            const BatchSlot *chains) {
  const BatchSlot c = chains[blockIdx.x];
  const PostProblem &pp = problems[c.problem];
  energy_fold_body(pp.e.partial, pp.e.num_blocks, pp.diag_sum, pp.out_e, c.group);
}

// Packed original-order configurations (bit = +1) -> permuted sign-bit words.
__global__ __launch_bounds__(256) void k_permute_bits(const uint64_t *__restrict__ x,
                                                     uint32_t words,
                                                     const uint32_t *__restrict__ spin_of_pos,
                                                     uint32_t num_blocks, uint32_t count,
                                                     uint64_t *__restrict__ perm_words) {
  const uint64_t idx = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<uint64_t>(count) * num_blocks) return;
  const uint32_t r = static_cast<uint32_t>(idx / num_blocks);
  const uint32_t b = static_cast<uint32_t>(idx % num_blocks);
  uint64_t word = 0;
  for (uint32_t l = 0; l < 64; ++l) {
    const uint32_t spin = spin_of_pos[b * 64u + l];
    if (spin == kDummySpin) continue;
    const uint64_t up = (x[static_cast<uint64_t>(r) * words + (spin >> 6)] >> (spin & 63u)) & 1ull;
    word |= (up ^ 1ull) << l;
  }
  perm_words[idx] = word;
}

// Permuted sign-bit words -> packed original-order configurations (bit = +1).  A wavefront per
// output word: lane j owns spin 64 w + j, its position is read once (coalesced) and used for
// kUnpermuteChains configurations; the word of each is one ballot.
constexpr uint32_t kUnpermuteChains = 32;
__global__ __launch_bounds__(256) void k_unpermute_bits(const uint64_t *__restrict__ perm_words,
                                                       uint32_t num_blocks,
                                                       const uint32_t *__restrict__ pos_of_spin,
                                                       uint64_t num_spins, uint32_t words,
                                                       uint32_t count, uint64_t *__restrict__ x) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= words) return;  // whole wavefront
  const uint64_t spin = static_cast<uint64_t>(w) * 64u + lane;
  const bool live = spin < num_spins;
  const uint32_t pos = live ? pos_of_spin[spin] : 0u;
  const uint32_t first = blockIdx.y * kUnpermuteChains;
  const uint32_t last = first + kUnpermuteChains < count ? first + kUnpermuteChains : count;
  for (uint32_t r = first; r < last; ++r) {
    const uint64_t neg =
        (perm_words[static_cast<uint64_t>(r) * num_blocks + (pos >> 6)] >> (pos & 63u)) & 1ull;
    const uint64_t word = __ballot(live && neg == 0ull);
    if (lane == 0) x[static_cast<uint64_t>(r) * words + w] = word;
  }
}

__global__ __launch_bounds__(256) void k_unpermute_bits_batch(const PostProblem *problems,
                                                             const BatchSlot *chains) {
  const BatchSlot c = chains[blockIdx.x];
  const PostProblem &pp = problems[c.problem];
  const uint64_t *perm = pp.e.perm_words + static_cast<uint64_t>(c.group) * pp.e.num_blocks;
  for (uint32_t w = threadIdx.x; w < pp.words; w += blockDim.x) {
    uint64_t word = 0;
    for (uint32_t j = 0; j < 64; ++j) {
      const uint64_t spin = static_cast<uint64_t>(w) * 64u + j;
      if (spin >= pp.num_spins) break;
      const uint32_t pos = pp.pos_of_spin[spin];
      const uint64_t neg = (perm[pos >> 6] >> (pos & 63u)) & 1ull;
      word |= (neg ^ 1ull) << j;
    }
    pp.out_x[static_cast<uint64_t>(c.group) * pp.words + w] = word;
  }
}

}  // namespace

// ---------------------------------------------------------------------------
// Plan object and C ABI
// ---------------------------------------------------------------------------

namespace {

using SweepKernel = void (*)(SweepArgs);

SweepKernel sweep_kernel_for(int m, bool descent, int layout) {
  if (layout == kBits) return descent ? k_sa_sweep<1, true, kBits> : k_sa_sweep<1, false, kBits>;
  if (layout == kGlobal) {
    return descent ? k_sa_sweep<1, true, kGlobal> : k_sa_sweep<1, false, kGlobal>;
  }
  if (layout == kWide) return m == 4 ? k_sa_sweep<4, false, kWide> : nullptr;
  if (layout == kNibbles) return m == 4 ? k_sa_sweep<4, false, kNibbles> : nullptr;
  switch (m) {
    case 1: return descent ? k_sa_sweep<1, true, kBytes> : k_sa_sweep<1, false, kBytes>;
    case 2: return k_sa_sweep<2, false, kBytes>;
    case 4: return k_sa_sweep<4, false, kBytes>;
    case 8: return k_sa_sweep<8, false, kBytes>;
    default: return nullptr;
  }
}

size_t sweep_lds_bytes(const asp::SaHostLayout &L, int layout) {
  // spins | delta[8] book[24] | flag (16 B) | meta[num_blocks]
  // ... | cache_ctl[4] | dirty[num_blocks] | inert[num_blocks] (each rounded up to 16 B)
  // bit-packed: spin words | delta book | flag | cache_ctl only
  if (layout == kGlobal) return 34 * sizeof(long long) + 32;
  if (layout == kBits) return static_cast<size_t>(L.num_blocks) * 8 + 34 * sizeof(long long) + 32;
  const size_t per_block = layout == kWide ? 256 : (layout == kNibbles ? 32 : 64);
  return static_cast<size_t>(L.num_blocks) * per_block + 34 * sizeof(long long) +
         static_cast<size_t>(L.num_blocks) * sizeof(uint2) + 16 +
         2 * (((static_cast<size_t>(L.num_blocks) + 15) / 16) * 16);
}

// k_sa_sweep_team: sign words | book, flags (64 B) | dirty[num_blocks] | inert[num_blocks]
size_t team_lds_bytes(const asp::SaHostLayout &L) {
  return static_cast<size_t>(L.num_blocks) * 8 + 64 +
         2 * (((static_cast<size_t>(L.num_blocks) + 15) / 16) * 16);
}

// Launch geometry: as many replicas per group as still leaves one group per CU,
// as many wavefronts as a colour class has blocks.
void choose_launch(const asp_sa_plan *p, uint32_t repetitions, int *m_out, int *threads_out) {
  uint32_t widest = 1;  // blocks of the largest colour class
  for (uint32_t c = 0; c < p->host.num_colors; ++c) {
    widest = std::max(widest, p->host.color_block_start[c + 1] - p->host.color_block_start[c]);
  }
  int m = 1;
  bool two_per_cu = false;
  if (p->force_m) {
    m = p->force_m;
  } else {
    for (int cand : {8, 4, 2}) {
      if ((repetitions + cand - 1) / cand >= static_cast<uint32_t>(p->num_cus)) {
        m = cand;
        break;
      }
    }
    // Small clusters with chains to spare: two workgroups of four replicas per CU, eight
    // wavefronts each, fill the thin rounds of one another (K = 1e4, 2048 chains: 209 -> 228
    // G flips/s against eight replicas in one workgroup; the reverse from ~50 blocks per colour)
    if (m == 8 && widest <= 32 && !p->force_threads) {
      m = 4;
      two_per_cu = true;
    }
  }
  int threads = p->force_threads;
  if (two_per_cu) threads = 512;
  if (!threads) {
    // one wavefront per block of the LARGEST colour class (DSATUR classes are skewed, the
    // first is the biggest), at most 16
    // (round 1 preferred 12 wavefronts below ~80 blocks per colour; with this round's accept
    // phase 16 are faster at every size: K = 1e4 180 -> 198 G flips/s, 3e4 238 -> 253, two boxes)
    const uint32_t most = 16u;
    threads = static_cast<int>(std::min<uint32_t>(widest, most)) * 64;
  }
  *m_out = m;
  *threads_out = threads;
}

int energies_of_perm(asp_sa_plan *p, const uint64_t *perm_words, uint32_t count, double *partial,
                     double *out_e) {
  const asp::SaHostLayout &L = p->host;
  if (count == 0) return ASP_OK;
  EnergyArgs ea{p->block_width.ptr, p->ell_off.ptr, p->ell_col.ptr, p->ell_val.ptr,
                p->field_pos.ptr,   perm_words,     partial,        L.num_blocks};
  const size_t lds = static_cast<size_t>(L.num_blocks) * sizeof(uint64_t);
  constexpr int kShare = 4;  // configurations per workgroup sharing the coupling loads
  if (count >= 2 * kShare && lds * kShare <= p->max_lds) {
    if (lds * kShare > 64 * 1024) {
      ASP_HIP_TRY(hipFuncSetAttribute(
          reinterpret_cast<const void *>(k_sa_energy_blocks_multi<kShare>),
          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds * kShare)));
    }
    hipLaunchKernelGGL(k_sa_energy_blocks_multi<kShare>, dim3((count + kShare - 1) / kShare),
                       dim3(512), lds * kShare, p->stream, ea, count);
  } else if (lds > p->max_lds) {
    hipLaunchKernelGGL(k_sa_energy_blocks<false>, dim3(count), dim3(512), 0, p->stream, ea);
  } else {
    if (lds > 64 * 1024) {
      ASP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sa_energy_blocks<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds)));
    }
    hipLaunchKernelGGL(k_sa_energy_blocks<true>, dim3(count), dim3(512), lds, p->stream, ea);
  }
  hipLaunchKernelGGL(k_sa_energy_fold, dim3(count), dim3(64), 0, p->stream, partial, L.num_blocks,
                     L.diag_sum, out_e);
  ASP_HIP_TRY(hipGetLastError());
  return ASP_OK;
}

}  // namespace

namespace asp {

int sa_permute_bits(asp_sa_plan *p, const uint64_t *x, uint32_t count, uint64_t *perm) {
  const SaHostLayout &L = p->host;
  const uint64_t total = static_cast<uint64_t>(count) * L.num_blocks;
  if (total == 0) return ASP_OK;
  const uint32_t words = static_cast<uint32_t>((L.num_spins + 63) / 64);
  hipLaunchKernelGGL(k_permute_bits, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                     p->stream, x, words, p->spin_of_pos.ptr, L.num_blocks, count, perm);
  ASP_HIP_TRY(hipGetLastError());
  return ASP_OK;
}

int sa_energies_of_perm(asp_sa_plan *p, const uint64_t *perm, uint32_t count, double *partial,
                        double *out_e) {
  return energies_of_perm(p, perm, count, partial, out_e);
}

}  // namespace asp

extern "C" {

asp_sa_plan *asp_sa_plan_create(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                                double const *data, double const *field) {
  asp_clear_error();
  if (asp::require_device() != ASP_OK) return nullptr;
  asp_sa_plan *p = new (std::nothrow) asp_sa_plan();
  if (!p) {
    asp::set_error(ASP_ERR_ALLOC, "out of host memory");
    return nullptr;
  }
  if (asp::build_sa_layout(num_spins, indptr, indices, data, field, &p->host) != ASP_OK) {
    delete p;
    return nullptr;
  }
  const asp::SaHostLayout &L = p->host;
  // ASP_SA_TEAM=0: no team sweeps in this process — several processes share the device (worker
  // processes or ranks of the pipeline on one GPU), and a team launch needs all its workgroups
  // resident together, which another process's kernels can prevent (the watchdog then costs
  // seconds before the rerun without teams)
  if (const char *env = std::getenv("ASP_SA_TEAM")) {
    const int team = std::atoi(env);
    if (team == 0 || team == 2 || team == 4 || team == 8) p->team_mode = team;
  }
  bool ok = asp::stream_acquire(&p->stream) == ASP_OK;
  for (auto &e : p->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
  if (!ok) asp::set_error(ASP_ERR_HIP, "could not create HIP stream/events");
  int num_cus = 0;
  size_t max_lds = 0;
  if (ok && asp::device_limits(&num_cus, &max_lds) == ASP_OK) {
    p->num_cus = num_cus;
    p->max_lds = max_lds;
  }
  ok = ok && upload_vector(p->color_block_start, L.color_block_start, p->stream) == ASP_OK &&
       upload_vector(p->block_width, L.block_width, p->stream) == ASP_OK &&
       upload_vector(p->ell_off, L.ell_off, p->stream) == ASP_OK &&
       upload_vector(p->ell_col, L.ell_col, p->stream) == ASP_OK &&
       upload_vector(p->ell_val, L.ell_val, p->stream) == ASP_OK &&
       upload_vector(p->spin_of_pos, L.spin_of_pos, p->stream) == ASP_OK &&
       upload_vector(p->pos_of_spin, L.pos_of_spin, p->stream) == ASP_OK &&
       upload_vector(p->field_pos, L.field_pos, p->stream) == ASP_OK;
  if (ok && sweep_lds_bytes(L, kWide) <= p->max_lds) {
    std::vector<uint32_t> scaled(L.ell_col.size());
    for (size_t i = 0; i < scaled.size(); ++i) scaled[i] = L.ell_col[i] * 4u;
    ok = upload_vector(p->ell_col4, scaled, p->stream) == ASP_OK &&
         hipStreamSynchronize(p->stream) == hipSuccess;  // `scaled` dies with this scope
  }
  if (ok && hipStreamSynchronize(p->stream) != hipSuccess) {
    asp::set_error(ASP_ERR_HIP, "plan upload failed");
    ok = false;
  }
  if (!ok) {
    asp_sa_plan_destroy(p);
    return nullptr;
  }
  return p;
}

void asp_sa_plan_destroy(asp_sa_plan *p) {
  if (!p) return;
  for (auto &e : p->ev) {
    if (e) (void)hipEventDestroy(e);
  }
  if (p->stream) {
    (void)hipStreamSynchronize(p->stream);  // idle before it is recycled
    asp::stream_release(p->stream);
  }
  delete p;
}

int asp_sa_plan_info(asp_sa_plan const *p, asp_sa_info *info) {
  if (!p || !info) return asp::set_error(ASP_ERR_INVALID, "null argument");
  const asp::SaHostLayout &L = p->host;
  info->num_spins = L.num_spins;
  info->nnz_offdiag = L.a_col.size();
  info->ell_entries = L.ell_off.back() * 64;
  info->num_colors = L.num_colors;
  info->num_blocks = L.num_blocks;
  info->max_degree = L.max_degree;
  info->energy_scale_exp = L.energy_scale_exp;
  info->diag_sum = L.diag_sum;
  info->beta0_auto = L.beta0_auto;
  info->beta1_auto = L.beta1_auto;
  return ASP_OK;
}

int asp_sa_set_launch(asp_sa_plan *p, int replicas_per_group, int threads) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  if (replicas_per_group != 0 && !sweep_kernel_for(replicas_per_group, false, false)) {
    return asp::set_error(ASP_ERR_INVALID, "replicas_per_group must be 0, 1, 2, 4 or 8");
  }
  if (threads != 0 && (threads < 64 || threads > 1024 || threads % 64 != 0)) {
    return asp::set_error(ASP_ERR_INVALID, "threads must be 0 or a multiple of 64 in [64, 1024]");
  }
  p->force_m = replicas_per_group;
  p->force_threads = threads;
  return ASP_OK;
}

int asp_sa_set_field_cache(asp_sa_plan *p, int enable) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  p->use_field_cache = enable != 0;
  return ASP_OK;
}

int asp_sa_team_watchdog_trips(asp_sa_plan const *p, uint32_t *of_plan, uint64_t *of_process) {
  if (of_plan) *of_plan = p ? p->team_watchdog_trips : 0u;
  if (of_process) *of_process = g_team_watchdog_trips.load(std::memory_order_relaxed);
  return ASP_OK;
}

int asp_sa_set_team(asp_sa_plan *p, int team) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  if (team != -1 && team != 0 && team != 2 && team != 4 && team != 8) {
    return asp::set_error(ASP_ERR_INVALID, "team must be -1 (auto), 0 (off), 2, 4 or 8");
  }
  p->team_mode = team;
  return ASP_OK;
}

int asp_sa_set_wide(asp_sa_plan *p, int allow) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  p->allow_wide = allow != 0;
  return ASP_OK;
}

int asp_sa_last_layout(asp_sa_plan const *p) { return p ? p->last_layout : -1; }

int asp_sa_set_packed(asp_sa_plan *p, int packed) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  p->force_packed = packed < 0 ? 0 : (packed > 2 ? 2 : packed);
  return ASP_OK;
}

}  // extern "C"

namespace {

// All chains of one call; descent = strict-descent sweeps (greedy relaxation).
int run_chains(asp_sa_plan *p, uint64_t seed, double const *betas, uint32_t num_sweeps,
               uint32_t repetitions, uint32_t replica_offset, uint64_t const *x0, bool descent,
               uint64_t *out_x, double *out_e, int64_t *out_trace = nullptr) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  ASP_TRY(asp::bind_device());
  if (repetitions == 0) return ASP_OK;
  if (!out_x || !out_e || (num_sweeps && !betas)) {
    return asp::set_error(ASP_ERR_INVALID, "null argument");
  }
  if (num_sweeps == 0xFFFFFFFFu) {
    return asp::set_error(ASP_ERR_INVALID, "num_sweeps 2^32-1 is reserved");
  }
  if (static_cast<uint64_t>(replica_offset) + repetitions + 8 > 0xFFFFFFFFull) {
    return asp::set_error(ASP_ERR_INVALID, "replica ids exceed 32 bits");
  }
  for (uint32_t t = 0; t < num_sweeps; ++t) {
    if (!(betas[t] >= 0.0)) return asp::set_error(ASP_ERR_INVALID, "betas[%u] is not >= 0", t);
  }
  const asp::SaHostLayout &L = p->host;
  const uint64_t K = L.num_spins;
  const uint32_t words = static_cast<uint32_t>((K + 63) / 64);
  p->last_sweep_ms = p->last_total_ms = 0.0f;
  if (K == 0) {
    for (uint32_t r = 0; r < repetitions; ++r) out_e[r] = 0.0;
    return ASP_OK;
  }
  int m = 1, threads = 64;
  choose_launch(p, repetitions, &m, &threads);
  if (descent) m = 1;
  // One byte per position when that fits the LDS; otherwise one BIT per position, one replica
  // per workgroup (flips applied by wavefront ballot) — 8x the capacity.
  bool packed = p->force_packed != 0;
  bool nibbles = false;
  if (!packed && sweep_lds_bytes(L, kBytes) > p->max_lds) {
    // ... or, with chains enough for four per workgroup, four bits per position: twice the
    // capacity of bytes and still four replicas sharing every coupling load
    if (!descent && !out_trace && m >= 4 && sweep_lds_bytes(L, kNibbles) <= p->max_lds) {
      nibbles = true;
      m = 4;
    } else {
      packed = true;
    }
  }
  // not even a bit per position fits the LDS: keep the words in HBM (no size limit, slow)
  const bool global = p->force_packed == 2 || (packed && sweep_lds_bytes(L, kBits) > p->max_lds);
  if (packed) m = 1;
  // A word per position (SDWA sign trick) when four replicas share the
  // workgroup and the words fit; results do not depend on the layout.
  // (measured: +2..12 % with four replicas per workgroup, nothing with two)
  const bool wide = !packed && !nibbles && !descent && p->allow_wide && m == 4 &&
                    p->ell_col4.ptr != nullptr && sweep_lds_bytes(L, kWide) <= p->max_lds;
  // Few chains on a large cluster: spread each chain over a team of workgroups (k_sa_sweep_team).
  uint32_t team = 0;
  {
    uint32_t widest = 1;
    for (uint32_t c = 0; c < L.num_colors; ++c) {
      widest = std::max(widest, L.color_block_start[c + 1] - L.color_block_start[c]);
    }
    const size_t team_lds = team_lds_bytes(L);
    // (one colour class only — a diagonal or field-only J —: the single barrier per sweep would
    // not separate a fast member's next publication from a slow member's read of this one)
    const bool possible = !out_trace && !global && !nibbles && p->force_packed == 0 && L.num_colors >= 2 &&
                          p->force_m == 0 && team_lds <= p->max_lds &&
                          static_cast<uint64_t>(repetitions) * 2 <= static_cast<uint64_t>(p->num_cus);
    if (possible && p->team_mode >= 2) {
      team = static_cast<uint32_t>(p->team_mode);  // forced (tests, measurements)
    } else if (possible && p->team_mode < 0 && widest >= 64) {
      // auto: as many workgroups per chain as the chip has to spare, up to 8
      const uint32_t spare = static_cast<uint32_t>(p->num_cus) / repetitions;
      team = spare >= 8 ? 8u : (spare >= 4 ? 4u : 2u);
      while (team > 2 && widest < 16u * team) team >>= 1;  // every member needs a full round
    }
    if (team * repetitions > static_cast<uint32_t>(p->num_cus)) team = 0;  // must be co-resident
  }
  if (team >= 2) m = 1;
  const int layout = team >= 2 ? kBits
                               : (global ? kGlobal
                                         : (packed ? kBits : (nibbles ? kNibbles : (wide ? kWide : kBytes))));
  const size_t lds = team >= 2 ? team_lds_bytes(L) : sweep_lds_bytes(L, layout);
  if (lds > p->max_lds) {
    return asp::set_error(ASP_ERR_TOO_LARGE, "%zu B of LDS needed, %zu B available", lds,
                          p->max_lds);
  }
  const uint32_t groups = (repetitions + m - 1) / m;
  const uint64_t padded = static_cast<uint64_t>(groups) * m;
  hipStream_t s = p->stream;
  // every exit below, error or not, first waits for what was queued on the stream: copies into
  // the caller's buffers and kernels using the plan's work buffers never outlive the call
  asp::StreamFence fence(s);
  if (global) ASP_TRY(p->w_spins.ensure(static_cast<uint64_t>(groups) * L.num_blocks));

  DeviceBuffer<double> &d_betas = p->w_betas, &d_partial = p->w_partial, &d_e = p->w_e;
  DeviceBuffer<uint64_t> &d_best = p->w_best, &d_x0 = p->w_x0, &d_x0_perm = p->w_x0_perm,
                         &d_x = p->w_x;
  DeviceBuffer<long long> &d_tracked = p->w_tracked;
  DeviceBuffer<unsigned long long> &d_accepted = p->w_accepted;
  ASP_TRY(d_betas.ensure(num_sweeps));
  ASP_TRY(d_best.ensure(padded * L.num_blocks));
  ASP_TRY(d_tracked.ensure(padded));
  ASP_TRY(d_accepted.ensure(padded));
  ASP_TRY(d_partial.ensure(static_cast<uint64_t>(repetitions) * L.num_blocks));
  ASP_TRY(d_e.ensure(repetitions));
  ASP_TRY(d_x.ensure(static_cast<uint64_t>(repetitions) * words));
  ASP_TRY(d_betas.upload(betas, num_sweeps, s));
  ASP_HIP_TRY(hipMemsetAsync(d_accepted.ptr, 0, padded * sizeof(unsigned long long), s));
  if (x0) {
    ASP_TRY(d_x0.ensure(words));
    ASP_TRY(d_x0_perm.ensure(L.num_blocks));
    ASP_TRY(d_x0.upload(x0, words, s));
    hipLaunchKernelGGL(k_permute_bits, dim3((L.num_blocks + 255) / 256), dim3(256), 0, s, d_x0.ptr,
                       words, p->spin_of_pos.ptr, L.num_blocks, 1u, d_x0_perm.ptr);
  }

  SweepArgs args{};
  args.color_block_start = p->color_block_start.ptr;
  args.block_width = p->block_width.ptr;
  args.ell_off = p->ell_off.ptr;
  args.ell_col = wide ? p->ell_col4.ptr : p->ell_col.ptr;
  args.ell_val = p->ell_val.ptr;
  args.spin_of_pos = p->spin_of_pos.ptr;
  args.field_pos = p->field_pos.ptr;
  args.betas = d_betas.ptr;
  args.x0_perm = x0 ? d_x0_perm.ptr : nullptr;
  args.best_perm = d_best.ptr;
  args.tracked = d_tracked.ptr;
  args.accepted = d_accepted.ptr;
  args.seed = seed;
  args.scale = std::ldexp(1.0, L.energy_scale_exp);
  args.num_colors = L.num_colors;
  args.num_blocks = L.num_blocks;
  args.num_sweeps = num_sweeps;
  args.replica_first = replica_offset;
  args.field_cache = nullptr;
  args.cache_enter_flips = 0;
  args.spin_words = global ? p->w_spins.ptr : nullptr;
  args.trace = nullptr;
  const uint64_t trace_elems = padded * (static_cast<uint64_t>(num_sweeps) + 1);
  if (out_trace) {
    ASP_TRY(p->w_trace.ensure(trace_elems));
    args.trace = p->w_trace.ptr;
  }
  if (p->use_field_cache && !packed && team < 2) {  // (both the byte and the wide layout)
    // 512 B per block and replica; skipped when it would not fit comfortably in HBM
    const uint64_t cache_elems = padded * L.num_blocks * 64ull;
    if (cache_elems * sizeof(double) <= (32ull << 30) && p->w_field_cache.ensure(cache_elems) == ASP_OK) {
      args.field_cache = p->w_field_cache.ptr;
      // a flip stales ~degree blocks: cached mode pays while that is a fraction of all blocks
      const double degree = std::max(1.0, static_cast<double>(L.a_col.size()) / static_cast<double>(K));
      double factor = 0.7;
      if (const char *env = std::getenv("ASP_CACHE_FACTOR")) factor = std::atof(env);  // tuning aid
      args.cache_enter_flips =
          static_cast<uint32_t>(std::max(1.0, factor * static_cast<double>(L.num_blocks) / degree));
    } else {
      asp_clear_error();  // the cache is an optimisation: run without it
    }
  }

  p->team_abort_host = 0;
  // Two team kernels resident at the same time could each hold CUs the other is waiting for:
  // team launches of one process take turns (from launch to completion).
  static std::mutex team_launches;
  std::unique_lock<std::mutex> team_turn(team_launches, std::defer_lock);
  if (team >= 2) {
    team_turn.lock();
    // wavefronts per member: the widest colour class split over the team, at most 16
    uint32_t widest = 1;
    for (uint32_t c = 0; c < L.num_colors; ++c) {
      widest = std::max(widest, L.color_block_start[c + 1] - L.color_block_start[c]);
    }
    threads = static_cast<int>(std::min<uint32_t>(16u, (widest + team - 1) / team)) * 64;
    if (p->use_field_cache) {
      // the team's "tracking" of untouched blocks uses the field cache's switch-over threshold
      const double degree = std::max(1.0, static_cast<double>(L.a_col.size()) / static_cast<double>(K));
      args.cache_enter_flips =
          static_cast<uint32_t>(std::max(1.0, 0.7 * static_cast<double>(L.num_blocks) / degree));
    }
    TeamArgs ta{};
    ta.s = args;
    ta.team_size = team;
    ta.num_teams = repetitions;
    ta.spin_limit = kTeamSpinLimit;
    if (const char *env = std::getenv("ASP_TEAM_SPIN_LIMIT")) {  // test hook: provoke the watchdog
      ta.spin_limit = static_cast<uint32_t>(std::strtoul(env, nullptr, 10));
    }
    const size_t head_bytes = static_cast<size_t>(repetitions) * 8 * 7 + 16;
    const size_t need = head_bytes + static_cast<size_t>(repetitions) * L.num_blocks * 8;
    if (need > p->team_area_bytes) {
      if (p->team_area) (void)hipFree(p->team_area);
      p->team_area = nullptr;
      p->team_area_bytes = 0;
      ASP_HIP_TRY(hipExtMallocWithFlags(&p->team_area, need, hipDeviceMallocFinegrained));
      p->team_area_bytes = need;
    }
    ASP_HIP_TRY(hipMemsetAsync(p->team_area, 0, head_bytes, s));
    uint8_t *area = static_cast<uint8_t *>(p->team_area);
    ta.arrivals = reinterpret_cast<unsigned long long *>(area);
    ta.sums = reinterpret_cast<long long *>(area + static_cast<size_t>(repetitions) * 8);
    ta.abort = reinterpret_cast<uint32_t *>(area + static_cast<size_t>(repetitions) * 8 * 7);
    ta.flips = reinterpret_cast<uint64_t *>(area + head_bytes);
    const void *team_kernel = descent ? reinterpret_cast<const void *>(k_sa_sweep_team<true>)
                                      : reinterpret_cast<const void *>(k_sa_sweep_team<false>);
    if (lds > 64 * 1024) {
      ASP_HIP_TRY(hipFuncSetAttribute(team_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds)));
    }
    ASP_HIP_TRY(hipEventRecord(p->ev[0], s));
    ASP_HIP_TRY(hipEventRecord(p->ev[1], s));
    // An ORDINARY launch: team * repetitions <= CUs workgroups, each fitting a CU by itself,
    // are all resident on an otherwise idle device, and the watchdog (with the rerun below)
    // covers a device that is not idle.  hipLaunchCooperativeKernel would check the same
    // occupancy bound, but a process that has used it once dies in the HIP runtime's exit
    // handler when rocprofv3 is attached (tools/exit_probe.hip: a 40-line program does;
    // profiles/r02_exit_probe.txt).
    if (descent) {
      hipLaunchKernelGGL(k_sa_sweep_team<true>, dim3(repetitions * team), dim3(threads), lds, s, ta);
    } else {
      hipLaunchKernelGGL(k_sa_sweep_team<false>, dim3(repetitions * team), dim3(threads), lds, s, ta);
    }
    const hipError_t launched = hipGetLastError();
    if (launched != hipSuccess) {
      // the configuration cannot be launched: one workgroup per chain instead
      ASP_HIP_TRY(hipStreamSynchronize(s));
      team_turn.unlock();
      p->team_mode = 0;
      return run_chains(p, seed, betas, num_sweeps, repetitions, replica_offset, x0, descent, out_x,
                        out_e, out_trace);
    }
    ASP_HIP_TRY(hipEventRecord(p->ev[2], s));
    ASP_HIP_TRY(hipMemcpyAsync(&p->team_abort_host, ta.abort, sizeof p->team_abort_host,
                               hipMemcpyDeviceToHost, s));
  } else {
    SweepKernel kernel = sweep_kernel_for(m, descent, layout);
    if (lds > 64 * 1024) {
      ASP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds)));
    }
    ASP_HIP_TRY(hipEventRecord(p->ev[0], s));
    ASP_HIP_TRY(hipEventRecord(p->ev[1], s));
    hipLaunchKernelGGL(kernel, dim3(groups), dim3(threads), lds, s, args);
    ASP_HIP_TRY(hipGetLastError());
    ASP_HIP_TRY(hipEventRecord(p->ev[2], s));
  }
  // the first `repetitions` rows of best_perm are the real replicas
  ASP_TRY(energies_of_perm(p, d_best.ptr, repetitions, d_partial.ptr, d_e.ptr));
  hipLaunchKernelGGL(k_unpermute_bits,
                     dim3((words + 3) / 4, (repetitions + kUnpermuteChains - 1) / kUnpermuteChains),
                     dim3(256), 0, s, d_best.ptr, L.num_blocks, p->pos_of_spin.ptr, K, words,
                     repetitions, d_x.ptr);
  ASP_HIP_TRY(hipGetLastError());
  ASP_HIP_TRY(hipEventRecord(p->ev[3], s));
  // hipMemcpyDefault: out_x / out_e may be host pointers (the usual call) or device pointers
  // (distributed.py hands over RCCL-ready tensors, so a gather needs no host round trip)
  ASP_HIP_TRY(hipMemcpyAsync(out_x, d_x.ptr, static_cast<uint64_t>(repetitions) * words * sizeof(uint64_t),
                             hipMemcpyDefault, s));
  ASP_HIP_TRY(hipMemcpyAsync(out_e, d_e.ptr, repetitions * sizeof(double), hipMemcpyDefault, s));
  p->last_tracked.assign(repetitions, 0);
  p->last_accepted.assign(repetitions, 0);
  ASP_HIP_TRY(hipMemcpyAsync(p->last_tracked.data(), d_tracked.ptr, repetitions * sizeof(int64_t),
                             hipMemcpyDeviceToHost, s));
  ASP_HIP_TRY(hipMemcpyAsync(p->last_accepted.data(), d_accepted.ptr,
                             repetitions * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  if (out_trace) {  // rows of the real replicas come first
    ASP_HIP_TRY(hipMemcpyAsync(out_trace, p->w_trace.ptr,
                               static_cast<uint64_t>(repetitions) * (num_sweeps + 1ull) *
                                   sizeof(int64_t),
                               hipMemcpyDeviceToHost, s));
  }
  ASP_HIP_TRY(hipStreamSynchronize(s));
  if (p->team_abort_host != 0) {
    // The members of a team were not resident together (another process or a long kernel holding
    // compute units — the launch mutex only orders this process's team launches): the partial
    // results are discarded and the call is repeated with one workgroup per chain, the fallback
    // of a refused launch.  Teams stay off for this plan.
    if (team_turn.owns_lock()) team_turn.unlock();
    p->team_mode = 0;
    p->team_watchdog_trips += 1;
    g_team_watchdog_trips.fetch_add(1, std::memory_order_relaxed);
    return run_chains(p, seed, betas, num_sweeps, repetitions, replica_offset, x0, descent, out_x,
                      out_e, out_trace);
  }
  p->last_m = m;
  p->last_layout = team >= 2 ? 4 : layout;
  p->last_threads = threads;
  p->last_groups = static_cast<int>(groups);
  ASP_HIP_TRY(hipEventElapsedTime(&p->last_sweep_ms, p->ev[1], p->ev[2]));
  ASP_HIP_TRY(hipEventElapsedTime(&p->last_total_ms, p->ev[0], p->ev[3]));
  return ASP_OK;
}

}  // namespace

extern "C" {

int asp_sa_anneal(asp_sa_plan *p, uint64_t seed, double const *betas, uint32_t num_sweeps,
                  uint32_t repetitions, uint32_t replica_offset, uint64_t const *x0,
                  uint64_t *out_x, double *out_e) {
  asp_clear_error();
  return run_chains(p, seed, betas, num_sweeps, repetitions, replica_offset, x0, false, out_x,
                    out_e);
}

int asp_sa_anneal_trace(asp_sa_plan *p, uint64_t seed, double const *betas, uint32_t num_sweeps,
                        uint32_t repetitions, uint32_t replica_offset, uint64_t const *x0,
                        uint64_t *out_x, double *out_e, int64_t *out_trace) {
  asp_clear_error();
  if (!out_trace) return asp::set_error(ASP_ERR_INVALID, "null trace pointer");
  return run_chains(p, seed, betas, num_sweeps, repetitions, replica_offset, x0, false, out_x,
                    out_e, out_trace);
}

int asp_sa_greedy(asp_sa_plan *p, uint32_t max_sweeps, uint64_t *out_x, double *out_e,
                  uint32_t *out_sweeps) {
  asp_clear_error();
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  if (!out_x || !out_e) return asp::set_error(ASP_ERR_INVALID, "null argument");
  const asp::SaHostLayout &L = p->host;
  const uint64_t K = L.num_spins;
  const uint32_t words = static_cast<uint32_t>((K + 63) / 64);
  if (out_sweeps) *out_sweeps = 0;
  if (K == 0) {
    *out_e = 0.0;
    return ASP_OK;
  }
  // 1. strongest-coupling-first cluster merging on the host (O(E log E))
  std::vector<uint64_t> x(words, 0);
  ASP_TRY(asp::greedy_tree_signs(L, x.data()));
  // 2. strict-descent relaxation on the device, in chunks, until a chunk flips nothing
  constexpr uint32_t kChunk = 8;
  std::vector<double> zeros(kChunk, 0.0);
  std::vector<uint64_t> next(words, 0);
  uint32_t done = 0;
  double energy = 0.0;
  if (max_sweeps == 0) {
    ASP_TRY(asp_sa_energy(p, 1, x.data(), &energy));
  }
  while (done < max_sweeps) {
    const uint32_t chunk = std::min(kChunk, max_sweeps - done);
    ASP_TRY(run_chains(p, 0, zeros.data(), chunk, 1, 0, x.data(), true, next.data(), &energy));
    done += chunk;
    x.swap(next);
    if (p->last_accepted.empty() || p->last_accepted[0] == 0) break;
  }
  std::copy(x.begin(), x.end(), out_x);
  *out_e = energy;
  if (out_sweeps) *out_sweeps = done;
  return ASP_OK;
}

int asp_sa_last_stats(asp_sa_plan const *p, uint32_t count, int64_t *tracked, uint64_t *accepted) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  if (count != p->last_tracked.size()) {
    return asp::set_error(ASP_ERR_INVALID, "count does not match the last anneal call");
  }
  if (tracked) std::copy(p->last_tracked.begin(), p->last_tracked.end(), tracked);
  if (accepted) std::copy(p->last_accepted.begin(), p->last_accepted.end(), accepted);
  return ASP_OK;
}

int asp_sa_last_launch(asp_sa_plan const *p, int *replicas_per_group, int *threads, int *groups) {
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  if (replicas_per_group) *replicas_per_group = p->last_m;
  if (threads) *threads = p->last_threads;
  if (groups) *groups = p->last_groups;
  return ASP_OK;
}

float asp_sa_last_sweep_ms(asp_sa_plan const *p) { return p ? p->last_sweep_ms : 0.0f; }
float asp_sa_last_total_ms(asp_sa_plan const *p) { return p ? p->last_total_ms : 0.0f; }

int asp_sa_energy(asp_sa_plan *p, uint32_t count, uint64_t const *x, double *out_e) {
  asp_clear_error();
  if (!p) return asp::set_error(ASP_ERR_INVALID, "null plan");
  ASP_TRY(asp::bind_device());
  if (count == 0) return ASP_OK;
  if (!x || !out_e) return asp::set_error(ASP_ERR_INVALID, "null argument");
  const asp::SaHostLayout &L = p->host;
  const uint64_t K = L.num_spins;
  if (K == 0) {
    for (uint32_t r = 0; r < count; ++r) out_e[r] = 0.0;
    return ASP_OK;
  }
  const uint32_t words = static_cast<uint32_t>((K + 63) / 64);
  hipStream_t s = p->stream;
  DeviceBuffer<uint64_t> d_x, d_perm;
  DeviceBuffer<double> d_partial, d_e;
  ASP_TRY(d_x.alloc(static_cast<uint64_t>(count) * words));
  ASP_TRY(d_perm.alloc(static_cast<uint64_t>(count) * L.num_blocks));
  ASP_TRY(d_partial.alloc(static_cast<uint64_t>(count) * L.num_blocks));
  ASP_TRY(d_e.alloc(count));
  ASP_TRY(d_x.upload(x, static_cast<uint64_t>(count) * words, s));
  const uint64_t total = static_cast<uint64_t>(count) * L.num_blocks;
  hipLaunchKernelGGL(k_permute_bits, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                     s, d_x.ptr, words, p->spin_of_pos.ptr, L.num_blocks, count, d_perm.ptr);
  ASP_HIP_TRY(hipGetLastError());
  ASP_TRY(energies_of_perm(p, d_perm.ptr, count, d_partial.ptr, d_e.ptr));
  ASP_TRY(d_e.download(out_e, count, s));
  ASP_HIP_TRY(hipStreamSynchronize(s));
  return ASP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Batched anneal: many problems, one launch per size class
// ---------------------------------------------------------------------------
// The reference's production job is 50 000 sampled clusters of 50-1000 states, extended twice,
// each solved with 64 chains x 5120 sweeps (Makefile:9,115-127,
// experiments/sampled_connected_components.py:764-767, common.py:236-239).  One such problem
// is 64 workgroups of one or two wavefronts that wait on L2 latency at every colour step: a
// launch per problem leaves > 90 % of the chip idle.  Here the groups of ALL problems of a batch
// are the workgroups of a few launches (one per wavefront count), so the chip holds hundreds
// of problems at once and the latency of one hides behind the others.

namespace {

thread_local float g_batch_sweep_ms = 0.0f;

uint32_t widest_color(const asp::SaHostLayout &L) {
  uint32_t widest = 1;
  for (uint32_t c = 0; c < L.num_colors; ++c) {
    widest = std::max(widest, L.color_block_start[c + 1] - L.color_block_start[c]);
  }
  return widest;
}

struct BatchEntry {
  uint32_t item;     // index into the caller's array
  uint32_t waves;    // wavefronts per workgroup this problem wants
  int layout;        // kWide, kBytes or kBits
  double work;       // ~ time of one group: sweeps * ELL slabs
  uint64_t beta_at;  // offset of its ladder in the concatenated betas
};

}  // namespace

extern "C" {

float asp_sa_batch_last_ms(void) { return g_batch_sweep_ms; }

int asp_sa_anneal_batch(asp_sa_batch_item const *items, uint32_t count) {
  asp_clear_error();
  g_batch_sweep_ms = 0.0f;
  if (count == 0) return ASP_OK;
  if (!items) return asp::set_error(ASP_ERR_INVALID, "null items");
  ASP_TRY(asp::bind_device());
  // ---- validation (the checks of asp_sa_anneal, for every item before anything runs) ----
  for (uint32_t i = 0; i < count; ++i) {
    const asp_sa_batch_item &it = items[i];
    if (!it.plan) return asp::set_error(ASP_ERR_INVALID, "item %u: null plan", i);
    if (it.repetitions == 0) continue;
    if (!it.out_x || !it.out_e || (it.num_sweeps && !it.betas)) {
      return asp::set_error(ASP_ERR_INVALID, "item %u: null argument", i);
    }
    if (it.num_sweeps == 0xFFFFFFFFu) {
      return asp::set_error(ASP_ERR_INVALID, "item %u: num_sweeps 2^32-1 is reserved", i);
    }
    if (it.flags & ~static_cast<uint32_t>(ASP_SA_BATCH_SHUFFLED)) {
      return asp::set_error(ASP_ERR_INVALID, "item %u: unknown flags 0x%x", i, it.flags);
    }
    if (static_cast<uint64_t>(it.replica_offset) + it.repetitions + 8 > 0xFFFFFFFFull) {
      return asp::set_error(ASP_ERR_INVALID, "item %u: replica ids exceed 32 bits", i);
    }
    for (uint32_t t = 0; t < it.num_sweeps; ++t) {
      if (!(it.betas[t] >= 0.0)) {
        return asp::set_error(ASP_ERR_INVALID, "item %u: betas[%u] is not >= 0", i, t);
      }
    }
    for (uint32_t j = 0; j < i; ++j) {
      if (items[j].plan == it.plan && items[j].repetitions) {
        return asp::set_error(ASP_ERR_INVALID, "items %u and %u share a plan", j, i);
      }
    }
  }
  // ---- which items go into the shared launches ----
  // Problems whose spins need the bit-packed layouts, plans with a forced launch geometry
  // (tests, measurements) and a batch of one keep the single-problem path (team sweep included);
  // items asking for a fresh visiting order every sweep run concurrently on their plans' own
  // streams (csrc/sa_shuffled.hip).
  std::vector<BatchEntry> entries;
  std::vector<uint32_t> alone, shuffled;
  bool batch_bits = false, batch_nibbles = true;
  if (const char *env = std::getenv("ASP_BATCH_BITS")) batch_bits = std::atoi(env) != 0;  // tests, measurements
  if (const char *env = std::getenv("ASP_BATCH_NIBBLES")) batch_nibbles = std::atoi(env) != 0;
  for (uint32_t i = 0; i < count; ++i) {
    const asp_sa_batch_item &it = items[i];
    if (it.repetitions == 0) continue;
    if (it.flags & ASP_SA_BATCH_SHUFFLED) {
      shuffled.push_back(i);
      continue;
    }
    const asp_sa_plan *p = it.plan;
    const asp::SaHostLayout &L = p->host;
    // A byte per position must fit the LDS, or (up to ~2.4e5 spins) four bits per position with
    // four chains per workgroup.  Beyond that a chain would be one workgroup with a bit per
    // position: as a class of the shared launches it was measured and
    // LOSES against running those problems one after the other as team sweeps, every chain spread
    // over four CUs (a round of 64 clusters of the pipeline, largest model 157 706 spins: 5.9 s
    // against 4.05 s; ASP_BATCH_BITS=1 puts them into the shared launches all the same).
    const bool bytes_fit = L.num_spins > 0 && sweep_lds_bytes(L, kBytes) <= p->max_lds;
    const bool nibbles_fit = L.num_spins > 0 && sweep_lds_bytes(L, kNibbles) <= p->max_lds && batch_nibbles;
    const bool bits_fit = L.num_spins > 0 && sweep_lds_bytes(L, kBits) <= p->max_lds && batch_bits;
    if (!(bytes_fit || nibbles_fit || bits_fit) || p->force_m || p->force_threads || p->force_packed) {
      alone.push_back(i);
      continue;
    }
    const uint32_t widest = widest_color(L);
    BatchEntry e{};
    e.item = i;
    // Layout and wavefronts per workgroup in a shared launch.  Small problems — several
    // workgroups fit a CU — take the word layout (one-instruction signs) and 4 wavefronts: small
    // workgroups interleave better than the 16 a lone problem wants (cap 16 / 8 / 4 on clusters of
    // 1e2..1e4 spins, 128 problems: 109 / 133 / 152 G flips/s, 512: 152 / 200 / 213;
    // tools/tune_batch.py).  Larger ones keep a byte per spin — the word layout would leave them
    // one workgroup per CU — and take 16 wavefronts (the sampled-cluster pipeline's order-2 models,
    // 1e4..2e5 spins, cap 4 / 8 / 16: 150 / 234 / 279 G flips/s; profiles/r03_batch_tune_real.txt).
    uint64_t wide_max = 10000, small_max = 10000;
    if (const char *env = std::getenv("ASP_BATCH_WIDE_MAX")) wide_max = std::strtoull(env, nullptr, 10);    // tuning aids
    if (const char *env = std::getenv("ASP_BATCH_SMALL_MAX")) small_max = std::strtoull(env, nullptr, 10);
    e.layout = !bytes_fit ? (nibbles_fit ? kNibbles : kBits)
                          : (p->allow_wide && p->ell_col4.ptr && L.num_spins <= wide_max &&
                                     sweep_lds_bytes(L, kWide) <= p->max_lds
                                 ? kWide
                                 : kBytes);
    uint32_t cap = L.num_spins <= small_max ? 4u : 16u;
    if (const char *env = std::getenv("ASP_BATCH_WAVES")) cap = static_cast<uint32_t>(std::atoi(env));  // tuning aid
    e.waves = std::min<uint32_t>(widest, std::max(1u, std::min(16u, cap)));
    e.work = static_cast<double>(it.num_sweeps) * static_cast<double>(L.ell_off.back() + L.num_blocks);
    entries.push_back(e);
  }
  if (entries.size() == 1) {
    alone.push_back(entries[0].item);
    entries.clear();
  }
  if (!shuffled.empty()) {
    ASP_TRY(asp::sa_shuffled_batch(items, shuffled.data(), static_cast<uint32_t>(shuffled.size()),
                                   &g_batch_sweep_ms));
  }
  for (uint32_t i : alone) {
    const asp_sa_batch_item &it = items[i];
    ASP_TRY(run_chains(it.plan, it.seed, it.betas, it.num_sweeps, it.repetitions, it.replica_offset,
                       nullptr, false, it.out_x, it.out_e));
    g_batch_sweep_ms += it.plan->last_sweep_ms;
  }
  if (entries.empty()) return ASP_OK;

  const asp_sa_plan *first = items[entries[0].item].plan;
  const int num_cus = first->num_cus;
  const size_t max_lds = first->max_lds;
  // ---- size classes: workgroups of one launch have one thread count ----
  // A launch class = (wavefront count, spin layout): kWaves[c / 4] wavefronts, layout c % 4.
  static const uint32_t kWaves[] = {1, 2, 3, 4, 6, 8, 12, 16};
  static const int kLayouts[] = {kWide, kBytes, kBits, kNibbles};
  constexpr int kNumWaves = sizeof kWaves / sizeof kWaves[0];
  constexpr int kNumClasses = 4 * kNumWaves;
  auto class_of = [&](const BatchEntry &e) {
    int w = kNumWaves - 1;
    for (int c = 0; c < kNumWaves; ++c) {
      if (e.waves <= kWaves[c]) {
        w = c;
        break;
      }
    }
    return 4 * w + (e.layout == kWide ? 0 : (e.layout == kBytes ? 1 : (e.layout == kBits ? 2 : 3)));
  };
  auto waves_of_class = [&](int c) { return kWaves[c / 4]; };
  auto layout_of_class = [&](int c) { return kLayouts[c % 4]; };
  // ---- replicas per workgroup, per class ----
  // Measured on the production mix (tools/tune_batch.py, K log-uniform in [1e2, 1e4], 64 chains x
  // 5120 sweeps): four replicas per workgroup — the word layout with its one-instruction signs —
  // is fastest from 64 problems (86 G flips/s against 73 with two) over 128 (127 against 101
  // with eight, 95 with two, 62 with one) to 512 (161, the same as eight); fewer replicas per
  // workgroup only when four would leave SIMDs without a wavefront.
  int m_of_class[kNumClasses];
  {
    int m = 1;
    for (int cand : {4, 2}) {
      uint64_t waves = 0;
      for (const BatchEntry &e : entries) {
        const uint32_t reps = items[e.item].repetitions;
        const int per_group = e.layout == kBits ? 1 : (e.layout == kNibbles ? 4 : cand);
        waves += static_cast<uint64_t>((reps + per_group - 1) / per_group) * waves_of_class(class_of(e));
      }
      if (waves >= static_cast<uint64_t>(num_cus) * 4u) {
        m = cand;
        break;
      }
    }
    for (int c = 0; c < kNumClasses; ++c) m_of_class[c] = m;
    // (fewer replicas per workgroup for the classes of the largest problems, to shorten the
    // batch's longest workgroup, was measured and LOSES: 123 -> 97 G flips/s at 128 problems,
    // 157 -> 120 at 512; the word layout's efficiency outweighs the shorter tail)
    if (const char *env = std::getenv("ASP_BATCH_BIG_M")) {  // tuning aid
      const int forced = std::atoi(env);
      if (forced == 1 || forced == 2 || forced == 4 || forced == 8) {
        for (int c = 0; c < kNumClasses; ++c) {
          if (waves_of_class(c) >= 8) m_of_class[c] = forced;
        }
      }
    }
    if (const char *env = std::getenv("ASP_BATCH_M")) {  // tuning aid
      const int forced = std::atoi(env);
      if (forced == 1 || forced == 2 || forced == 4 || forced == 8) {
        for (int c = 0; c < kNumClasses; ++c) m_of_class[c] = forced;
      }
    }
    for (int c = 0; c < kNumClasses; ++c) {
      if (layout_of_class(c) == kBits) m_of_class[c] = 1;  // a bit per position: one chain per workgroup
      if (layout_of_class(c) == kNibbles) m_of_class[c] = 4;
      // the word layout exists for four replicas: with another count its problems take bytes
    }
  }
  auto m_of = [&](const BatchEntry &e) { return m_of_class[class_of(e)]; };
  for (BatchEntry &e : entries) {
    if (e.layout == kWide && m_of(e) != 4) e.layout = kBytes;
  }
  // ---- per-problem buffer offsets ----
  struct Offsets {
    uint64_t best, stat, cache, partial, e, x, groups, padded;
  };
  std::vector<Offsets> off(entries.size());
  uint64_t n_best = 0, n_stat = 0, n_cache = 0, n_partial = 0, n_e = 0, n_x = 0, n_betas = 0;
  bool use_cache = true;
  for (size_t k = 0; k < entries.size(); ++k) {
    const asp_sa_batch_item &it = items[entries[k].item];
    const asp::SaHostLayout &L = it.plan->host;
    const uint64_t m = static_cast<uint64_t>(m_of(entries[k]));
    const uint64_t groups = (it.repetitions + m - 1) / m, padded = groups * m;
    const uint64_t words = (L.num_spins + 63) / 64;
    off[k] = Offsets{n_best, n_stat, n_cache, n_partial, n_e, n_x, groups, padded};
    n_best += padded * L.num_blocks;
    n_stat += padded;
    if (entries[k].layout != kBits) n_cache += padded * L.num_blocks * 64ull;
    n_partial += static_cast<uint64_t>(it.repetitions) * L.num_blocks;
    n_e += it.repetitions;
    n_x += static_cast<uint64_t>(it.repetitions) * words;
    entries[k].beta_at = n_betas;
    n_betas += it.num_sweeps;
    use_cache = use_cache && it.plan->use_field_cache;
  }
  if (n_cache * sizeof(double) > (32ull << 30)) use_cache = false;

  asp::ScopedStream main_stream;
  ASP_TRY(main_stream.acquire());
  hipStream_t s = main_stream.stream;
  asp::ScopedStream class_stream[kNumClasses];
  DeviceBuffer<double> d_betas, d_partial, d_e, d_cache;
  DeviceBuffer<uint64_t> d_best, d_x;
  DeviceBuffer<long long> d_tracked;
  DeviceBuffer<unsigned long long> d_accepted;
  DeviceBuffer<SweepArgs> d_problems;
  DeviceBuffer<PostProblem> d_post;
  DeviceBuffer<BatchSlot> d_slots, d_chains;
  asp::StreamFence fence(s);
  ASP_TRY(d_betas.alloc(n_betas));
  ASP_TRY(d_best.alloc(n_best));
  ASP_TRY(d_tracked.alloc(n_stat));
  ASP_TRY(d_accepted.alloc(n_stat));
  ASP_TRY(d_partial.alloc(n_partial));
  ASP_TRY(d_e.alloc(n_e));
  ASP_TRY(d_x.alloc(n_x));
  if (use_cache && d_cache.alloc(n_cache) != ASP_OK) {
    asp_clear_error();  // the cache is an optimisation: run without it
    use_cache = false;
  }
  // ---- descriptors ----
  std::vector<double> h_betas(n_betas);
  std::vector<SweepArgs> h_problems(entries.size());
  std::vector<PostProblem> h_post(entries.size());
  std::vector<BatchSlot> h_chains;
  h_chains.reserve(n_e);
  size_t energy_lds = 0;
  for (size_t k = 0; k < entries.size(); ++k) {
    const asp_sa_batch_item &it = items[entries[k].item];
    const asp_sa_plan *p = it.plan;
    const asp::SaHostLayout &L = p->host;
    const bool wide = entries[k].layout == kWide;
    const bool packed = entries[k].layout == kBits;
    std::copy(it.betas, it.betas + it.num_sweeps, h_betas.begin() + entries[k].beta_at);
    SweepArgs a{};
    a.color_block_start = p->color_block_start.ptr;
    a.block_width = p->block_width.ptr;
    a.ell_off = p->ell_off.ptr;
    a.ell_col = wide ? p->ell_col4.ptr : p->ell_col.ptr;
    a.ell_val = p->ell_val.ptr;
    a.spin_of_pos = p->spin_of_pos.ptr;
    a.field_pos = p->field_pos.ptr;
    a.betas = d_betas.ptr + entries[k].beta_at;
    a.x0_perm = nullptr;
    a.best_perm = d_best.ptr + off[k].best;
    a.tracked = d_tracked.ptr + off[k].stat;
    a.accepted = d_accepted.ptr + off[k].stat;
    a.seed = it.seed;
    a.scale = std::ldexp(1.0, L.energy_scale_exp);
    a.num_colors = L.num_colors;
    a.num_blocks = L.num_blocks;
    a.num_sweeps = it.num_sweeps;
    a.replica_first = it.replica_offset;
    // (the bit-packed layout runs without the field cache, as in the single-problem launch)
    a.field_cache = use_cache && !packed ? d_cache.ptr + off[k].cache : nullptr;
    const double degree =
        std::max(1.0, static_cast<double>(L.a_col.size()) / static_cast<double>(L.num_spins));
    a.cache_enter_flips =
        packed ? 0u : static_cast<uint32_t>(std::max(1.0, 0.7 * static_cast<double>(L.num_blocks) / degree));
    a.spin_words = nullptr;
    a.trace = nullptr;
    h_problems[k] = a;
    PostProblem pp{};
    pp.e = EnergyArgs{p->block_width.ptr, p->ell_off.ptr, p->ell_col.ptr, p->ell_val.ptr,
                      p->field_pos.ptr, d_best.ptr + off[k].best, d_partial.ptr + off[k].partial,
                      L.num_blocks};
    pp.pos_of_spin = p->pos_of_spin.ptr;
    pp.num_spins = L.num_spins;
    pp.words = static_cast<uint32_t>((L.num_spins + 63) / 64);
    pp.diag_sum = L.diag_sum;
    pp.out_e = d_e.ptr + off[k].e;
    pp.out_x = d_x.ptr + off[k].x;
    h_post[k] = pp;
    for (uint32_t r = 0; r < it.repetitions; ++r) {
      h_chains.push_back(BatchSlot{static_cast<uint32_t>(k), r});
    }
    energy_lds = std::max(energy_lds, static_cast<size_t>(L.num_blocks) * sizeof(uint64_t));
  }
  // ---- slot tables: per class, problems longest first, dealt round-robin to the 8 XCDs ----
  struct ClassLaunch {
    uint64_t slot_at = 0;
    uint32_t slots_per_xcd = 0;
    size_t lds = 0;
    bool used = false;
  };
  ClassLaunch launches[kNumClasses];
  std::vector<BatchSlot> h_slots;
  for (int c = 0; c < kNumClasses; ++c) {
    std::vector<size_t> members;
    for (size_t k = 0; k < entries.size(); ++k) {
      if (class_of(entries[k]) == c) members.push_back(k);
    }
    if (members.empty()) continue;
    std::stable_sort(members.begin(), members.end(),
                     [&](size_t a, size_t b) { return entries[a].work > entries[b].work; });
    std::vector<BatchSlot> per_xcd[8];
    uint64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t k : members) {
      // the XCD with the fewest groups so far (ties: lowest index): balanced and deterministic
      int x = 0;
      for (int j = 1; j < 8; ++j) {
        if (load[j] < load[x]) x = j;
      }
      for (uint32_t g = 0; g < off[k].groups; ++g) {
        per_xcd[x].push_back(BatchSlot{static_cast<uint32_t>(k), g});
      }
      load[x] += off[k].groups;
      launches[c].lds = std::max(
          launches[c].lds, sweep_lds_bytes(items[entries[k].item].plan->host, layout_of_class(c)));
    }
    uint32_t most = 0;
    for (int x = 0; x < 8; ++x) most = std::max<uint32_t>(most, static_cast<uint32_t>(per_xcd[x].size()));
    launches[c].used = true;
    launches[c].slot_at = h_slots.size();
    launches[c].slots_per_xcd = most;
    for (int x = 0; x < 8; ++x) {
      per_xcd[x].resize(most, BatchSlot{0xFFFFFFFFu, 0});
      h_slots.insert(h_slots.end(), per_xcd[x].begin(), per_xcd[x].end());
    }
  }
  ASP_TRY(d_problems.alloc(h_problems.size()));
  ASP_TRY(d_post.alloc(h_post.size()));
  ASP_TRY(d_slots.alloc(h_slots.size()));
  ASP_TRY(d_chains.alloc(h_chains.size()));
  ASP_TRY(d_betas.upload(h_betas.data(), h_betas.size(), s));
  ASP_TRY(d_problems.upload(h_problems.data(), h_problems.size(), s));
  ASP_TRY(d_post.upload(h_post.data(), h_post.size(), s));
  ASP_TRY(d_slots.upload(h_slots.data(), h_slots.size(), s));
  ASP_TRY(d_chains.upload(h_chains.data(), h_chains.size(), s));
  ASP_HIP_TRY(hipMemsetAsync(d_accepted.ptr, 0, n_stat * sizeof(unsigned long long), s));
  hipEvent_t ev[2 + kNumClasses] = {};
  struct EventGuard {
    hipEvent_t *ev;
    int n;
    ~EventGuard() {
      for (int i = 0; i < n; ++i) {
        if (ev[i]) (void)hipEventDestroy(ev[i]);
      }
    }
  } event_guard{ev, 2 + kNumClasses};
  for (auto &e : ev) ASP_HIP_TRY(hipEventCreate(&e));
  ASP_HIP_TRY(hipEventRecord(ev[0], s));
  // ---- one sweep launch per class, each on its own stream so that they share the chip ----
  for (int c = 0; c < kNumClasses; ++c) {
    if (!launches[c].used) continue;
    ASP_TRY(class_stream[c].acquire());
    hipStream_t cs = class_stream[c].stream;
    ASP_HIP_TRY(hipStreamWaitEvent(cs, ev[0], 0));
    BatchArgs b{d_problems.ptr, d_slots.ptr + launches[c].slot_at, launches[c].slots_per_xcd};
    using BatchKernel = void (*)(BatchArgs);
    BatchKernel kernel = nullptr;
    if (layout_of_class(c) == kWide) {
      kernel = k_sa_sweep_batch<4, kWide>;
    } else if (layout_of_class(c) == kBits) {
      kernel = k_sa_sweep_batch<1, kBits>;
    } else if (layout_of_class(c) == kNibbles) {
      kernel = k_sa_sweep_batch<4, kNibbles>;
    } else {
      switch (m_of_class[c]) {
        case 1: kernel = k_sa_sweep_batch<1, kBytes>; break;
        case 2: kernel = k_sa_sweep_batch<2, kBytes>; break;
        case 4: kernel = k_sa_sweep_batch<4, kBytes>; break;
        default: kernel = k_sa_sweep_batch<8, kBytes>; break;
      }
    }
    if (launches[c].lds > 64 * 1024) {
      ASP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(launches[c].lds)));
    }
    hipLaunchKernelGGL(kernel, dim3(8u * launches[c].slots_per_xcd), dim3(64u * waves_of_class(c)),
                       launches[c].lds, cs, b);
    ASP_HIP_TRY(hipGetLastError());
    ASP_HIP_TRY(hipEventRecord(ev[2 + c], cs));
    ASP_HIP_TRY(hipStreamWaitEvent(s, ev[2 + c], 0));
  }
  ASP_HIP_TRY(hipEventRecord(ev[1], s));
  // ---- energies and original-order bits of every chain's best configuration ----
  const unsigned chains = static_cast<unsigned>(h_chains.size());
  if (energy_lds > max_lds) {
    hipLaunchKernelGGL(k_sa_energy_blocks_batch<false>, dim3(chains), dim3(512), 0, s, d_post.ptr,
                       d_chains.ptr);
  } else {
    if (energy_lds > 64 * 1024) {
      ASP_HIP_TRY(hipFuncSetAttribute(
          reinterpret_cast<const void *>(k_sa_energy_blocks_batch<true>),
          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(energy_lds)));
    }
    hipLaunchKernelGGL(k_sa_energy_blocks_batch<true>, dim3(chains), dim3(512), energy_lds, s,
                       d_post.ptr, d_chains.ptr);
  }
  hipLaunchKernelGGL(k_sa_energy_fold_batch, dim3(chains), dim3(64), 0, s, d_post.ptr, d_chains.ptr);
  hipLaunchKernelGGL(k_unpermute_bits_batch, dim3(chains), dim3(256), 0, s, d_post.ptr, d_chains.ptr);
  ASP_HIP_TRY(hipGetLastError());
  std::vector<uint64_t> h_x(n_x);
  std::vector<double> h_e(n_e);
  std::vector<long long> h_tracked(n_stat);
  std::vector<unsigned long long> h_accepted(n_stat);
  ASP_TRY(d_x.download(h_x.data(), n_x, s));
  ASP_TRY(d_e.download(h_e.data(), n_e, s));
  ASP_TRY(d_tracked.download(h_tracked.data(), n_stat, s));
  ASP_TRY(d_accepted.download(h_accepted.data(), n_stat, s));
  ASP_HIP_TRY(hipStreamSynchronize(s));
  float ms = 0.0f;
  ASP_HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
  g_batch_sweep_ms += ms;
  for (size_t k = 0; k < entries.size(); ++k) {
    const asp_sa_batch_item &it = items[entries[k].item];
    asp_sa_plan *p = it.plan;
    const uint64_t words = (p->host.num_spins + 63) / 64;
    std::copy(h_x.begin() + off[k].x, h_x.begin() + off[k].x + it.repetitions * words, it.out_x);
    std::copy(h_e.begin() + off[k].e, h_e.begin() + off[k].e + it.repetitions, it.out_e);
    p->last_tracked.assign(h_tracked.begin() + off[k].stat,
                           h_tracked.begin() + off[k].stat + it.repetitions);
    p->last_accepted.assign(h_accepted.begin() + off[k].stat,
                            h_accepted.begin() + off[k].stat + it.repetitions);
    const int c = class_of(entries[k]);
    p->last_m = m_of_class[c];
    p->last_layout = layout_of_class(c);
    p->last_threads = static_cast<int>(64u * waves_of_class(c));
    p->last_groups = static_cast<int>(off[k].groups);
    p->last_sweep_ms = p->last_total_ms = 0.0f;  // shared launches: see asp_sa_batch_last_ms
  }
  return ASP_OK;
}

}  // extern "C"
