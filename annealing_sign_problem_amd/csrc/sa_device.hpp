// Device arithmetic and coupling-stream helpers shared by the annealing kernels
// (csrc/sa_sweep.hip: colour order; csrc/sa_shuffled.hip: a fresh order every sweep).
// Specification: DESIGN.md §4.3-4.5.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

// Experiment switches (tools/ab_sweep.sh builds tagged variants with -D...=0/1).
#ifndef ASP_SERPENTINE
#define ASP_SERPENTINE 0
#endif
#ifndef ASP_MAGIC_RINT
#define ASP_MAGIC_RINT 1
#endif
#ifndef ASP_EXP_FILTER
#define ASP_EXP_FILTER 2  // 0 exact rule only, 1 f64 filter on u, 2 integer filter on the word
#endif
#ifndef ASP_INERT_SKIP
#define ASP_INERT_SKIP 1
#endif
#ifndef ASP_TEAM_SLEEP
#define ASP_TEAM_SLEEP 4  // s_sleep argument between two polls of the team barrier (0/1/4/16/64 scanned)
#endif
#ifndef ASP_J_MAJOR
#define ASP_J_MAJOR 1
#endif
#ifndef ASP_ABS_LDS
#define ASP_ABS_LDS 1
#endif
#ifndef ASP_SIGN_SHR
#define ASP_SIGN_SHR 1  // byte layout: v_lshrrev (fast VOP2) + v_lshl_or instead of v_lshlrev + v_bfi
#endif
#ifndef ASP_PHILOX_SKIP
#define ASP_PHILOX_SKIP 1  // no random numbers for a block none of whose proposals needs one
#endif
#ifndef ASP_EXPERIMENT_GLAUBER
#define ASP_EXPERIMENT_GLAUBER 0  // analysis only (tools/schedule_probe.py): heat-bath acceptance
#endif                            // 1 / (1 + exp(beta dE)) instead of Metropolis; NOT the specification
#ifndef ASP_MAX_THREADS
#define ASP_MAX_THREADS 1024  // launch bound of the sweep kernel (VGPR budget = 512 / waves per SIMD)
#endif
// Timing-only ablations (results are WRONG when any is set; never set in the product build).
#ifndef ASP_ABL_NO_ACCEPT
#define ASP_ABL_NO_ACCEPT 0
#endif
#ifndef ASP_ABL_NO_KLOOP
#define ASP_ABL_NO_KLOOP 0
#endif
#ifndef ASP_ABL_NO_BARRIER
#define ASP_ABL_NO_BARRIER 0
#endif
#ifndef ASP_ABL_NO_LDS
#define ASP_ABL_NO_LDS 0
#endif
#ifndef ASP_ABL_NO_GLOAD
#define ASP_ABL_NO_GLOAD 0
#endif
#ifndef ASP_ABL_NO_PHILOX
#define ASP_ABL_NO_PHILOX 0
#endif
#ifndef ASP_ABL_NO_EXP
#define ASP_ABL_NO_EXP 0
#endif
#ifndef ASP_ABL_HALF_BYTES
#define ASP_ABL_HALF_BYTES 0
#endif
#ifndef ASP_ABL_NO_FMA
#define ASP_ABL_NO_FMA 0
#endif

namespace asp {
namespace dev {

struct Philox4 {
  uint32_t w[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                 uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = static_cast<uint32_t>(p1);
    const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = static_cast<uint32_t>(p0);
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}

__device__ __forceinline__ uint32_t pick_word(const Philox4 &p, uint32_t which) {
  const uint32_t lo = (which & 1u) ? p.w[1] : p.w[0];
  const uint32_t hi = (which & 1u) ? p.w[3] : p.w[2];
  return (which & 2u) ? hi : lo;
}

// exp(-x), x >= 0: a fixed sequence of IEEE operations (v_rndne_f64, v_fma_f64,
// v_mul_f64) so that the result is bit-identical to the CPU restatement.
__device__ __forceinline__ double expneg(double x) {
  if (!(x < 23.0)) return 0.0;
  const double y = -x;
  const double kf = __builtin_rint(__dmul_rn(y, 0x1.71547652b82fep+0));
  double r = __builtin_fma(kf, -0x1.62e42fee00000p-1, y);
  r = __builtin_fma(kf, -0x1.a39ef35793c76p-33, r);
  double p = 0x1.6124613a86d09p-33;
  p = __builtin_fma(p, r, 0x1.1eed8eff8d898p-29);
  p = __builtin_fma(p, r, 0x1.ae64567f544e4p-26);
  p = __builtin_fma(p, r, 0x1.27e4fb7789f5cp-22);
  p = __builtin_fma(p, r, 0x1.71de3a556c734p-19);
  p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-16);
  p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-13);
  p = __builtin_fma(p, r, 0x1.6c16c16c16c17p-10);
  p = __builtin_fma(p, r, 0x1.1111111111111p-7);
  p = __builtin_fma(p, r, 0x1.5555555555555p-5);
  p = __builtin_fma(p, r, 0x1.5555555555555p-3);
  p = __builtin_fma(p, r, 0x1.0000000000000p-1);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  const long long k = static_cast<long long>(kf);
  const double scale = __longlong_as_double((1023ll + k) << 52);
  return __dmul_rn(p, scale);
}

// +-1.0 with the sign taken from bit `m` of the neighbour's spin byte (1 -> -1.0).
// acc = fma(v, +-1.0, acc) is bit-identical to acc + (+-v): the product is exact, so
// the only rounding is the add's.  Three VALU ops per (term, replica) and no
// register-pair shuffling: the low dword of the multiplier is a constant zero.
__device__ __forceinline__ double spin_factor(uint32_t spin_byte, int m) {
  uint32_t hi;
  if (m == 0) {
    hi = (spin_byte << 31) | 0x3FF00000u;  // v_lshl_or_b32: nothing but bit 0 survives the shift
  } else {
#if ASP_SIGN_SHR
    // replica m's bit to bit 0 with a RIGHT shift — a plain VOP2, 2.5 SIMD cycles per wave64 on
    // this chip, where every left shift and every VOP3 costs 4.3-4.4
    // (profiles/r02_issue_rate_probe.txt) — then the m = 0 instruction: 12.1 cycles per term
    // and replica with the FMA instead of 13.3
    const uint32_t down = spin_byte >> m;
    asm("v_lshl_or_b32 %0, %1, 31, %2" : "=v"(hi) : "v"(down), "s"(0x3FF00000u));
#else
    // bit 31 from the shifted byte, everything else from 1.0's high word: one v_bfi_b32
    // (hipcc folds the constant mask and emits v_and + v_or instead)
    const uint32_t shifted = spin_byte << (31 - m);
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hi) : "s"(0x80000000u), "v"(shifted), "v"(0x3FF00000u));
#endif
  }
  return __hiloint2double(static_cast<int>(hi), 0);
}

// u < expneg(x), decided through a hardware-exp filter.  v_exp_f32 of the f32-rounded
// argument is within ~2e-6 (relative) of expneg(x) for 0 < x < 23: 1 ulp of the instruction
// plus |x| * log2(e) * 2^-24 * ln 2 from rounding x to f32.  Outside a +-1e-5 band around that
// estimate the comparison is settled; inside it (probability ~2e-5 per proposal, so a wavefront
// takes the branch about once per thousand blocks) the exact sequence of §4.4 decides.  The
// result therefore ALWAYS equals `u < expneg(x)` — same bits as the oracle — at a fraction of
// the sixteen dependent f64 FMAs.
__device__ __forceinline__ bool metropolis_accept(double u, double x) {
  if (!(x < 23.0)) return false;  // expneg(x) = 0 < u; also NaN
  const float estimate = __builtin_amdgcn_exp2f(static_cast<float>(x) * -1.44269504f);
  const double p = static_cast<double>(estimate);
  if (u < p * (1.0 - 1e-5)) return true;
  if (u > p * (1.0 + 1e-5)) return false;
  return u < expneg(x);
}

// The same decision taken on the random WORD: u = (word + 0.5) * 2^-32 < p  <=>  word + 0.5 <
// p * 2^32.  With est = v_exp_f32 estimate of p (|est / p - 1| <= 2.63e-6, measured) and the two
// f32 products lo = est * 2^32 (1 - 2e-5), hi = est * 2^32 (1 + 2e-5) (constant and product
// rounding <= 1.3e-7 together): word < trunc(lo) implies word + 0.5 < lo < p * 2^32 (accept),
// word > trunc(hi) implies word + 0.5 > hi > p * 2^32 (reject); in between (~4e-5 of the
// proposals) the exact rule decides.  Integer compares and f32 products replace the f64
// conversions, products and compares of metropolis_accept.
__device__ __forceinline__ bool metropolis_accept_word(uint32_t word, double x) {
  if (!(x < 23.0)) return false;  // expneg(x) = 0 < u; also NaN
  const float estimate = __builtin_amdgcn_exp2f(static_cast<float>(x) * -1.44269504f);
  const float lo = estimate * (4294967296.0f * (1.0f - 2e-5f));  // < 2^32: conversion in range
  if (word < static_cast<uint32_t>(lo)) return true;
  const float hi = estimate * (4294967296.0f * (1.0f + 2e-5f));
  if (hi < 4294967040.0f && word > static_cast<uint32_t>(hi)) return false;
  const double u = __dmul_rn(__dadd_rn(static_cast<double>(word), 0.5), 0x1p-32);
  return u < expneg(x);
}

// v with its sign flipped when bit 0 of `neg` is set (energy kernel, not hot).
__device__ __forceinline__ double signed_coupling(double v, uint32_t neg, int m) {
  const unsigned long long flip = static_cast<unsigned long long>((neg >> m) & 1u) << 63;
  return __longlong_as_double(__double_as_longlong(v) ^ static_cast<long long>(flip));
}

__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
  for (int step = 1; step < 64; step <<= 1) v += __shfl_xor(v, step, 64);
  return v;
}

// Butterfly sum over the 64 lanes; every lane ends with the balanced-tree total
// ((v0+v1)+(v2+v3))+... (f64 addition commutes, so all lanes agree bitwise).
__device__ __forceinline__ double wave_tree_sum_f64(double v) {
#pragma unroll
  for (int step = 1; step < 64; step <<= 1) v = __dadd_rn(v, __shfl_xor(v, step, 64));
  return v;
}

// Replica mask (bit m) -> wide spin word (byte m = 0x80): bits 0..3 to bits 7, 15, 23, 31.
__device__ __forceinline__ uint32_t spread_mask(uint32_t mask) {
  return ((mask & 0xFu) * 0x00204081u & 0x01010101u) << 7;
}

// Collect bit m of each of the four bytes of d into a nibble (byte 0 -> bit 0).
__device__ __forceinline__ uint32_t gather_bit4(uint32_t d, int m) {
  const uint32_t t = (d >> m) & 0x01010101u;
  return ((t * 0x00204081u) >> 21) & 0xFu;
}

// ---------------------------------------------------------------------------
// Sweep kernel
// ---------------------------------------------------------------------------

// Four consecutive ELL entries of one lane (one row), k = 4q .. 4q+3.
struct Quad {
  uint4 c;
  double2 v01, v23;
};

// Three 16-byte loads per lane; quad index `q` is relative to the block's first quad.
__device__ __forceinline__ void load_quad(Quad &q, const uint4 *__restrict__ cptr,
                                          const double2 *__restrict__ vptr, uint32_t quad) {
#if ASP_ABL_NO_GLOAD
  const uint32_t l = (threadIdx.x * 37u + quad * 101u) & 0x3FFFu;
  q.c = make_uint4(l, l + 1u, l + 2u, l + 3u);
  q.v01 = make_double2(1.0 + quad, 2.0);
  q.v23 = make_double2(3.0, 4.0 + quad);
#else
  q.c = cptr[quad * 64u];
  q.v01 = vptr[quad * 128u];
#if ASP_ABL_HALF_BYTES
  q.v23 = make_double2(q.v01.y, q.v01.x);  // timing only: skip one of the two value loads
#else
  q.v23 = vptr[quad * 128u + 64u];
#endif
#endif
}

// PACKED = false: one LDS byte per position, bit m = replica m.  PACKED = true (M = 1 only):
// one LDS bit per position, 64 positions (= one block) per u64 word.
using LdsByte = __attribute__((address_space(3))) const uint8_t;
using LdsWord = __attribute__((address_space(3))) const uint32_t;

// How a workgroup keeps its spins in LDS.
//   kBytes: one byte per position, bit m = sign bit of replica m (M <= 8);
//   kBits:  one bit per position, one replica (8x the capacity);
//   kWide:  one 32-bit word per position, byte m = 0x80 * sign bit of replica m (M <= 4; fits
//           up to ~4e4 spins): the +-1.0 multiplier of a term is then ONE SDWA instruction.
//   kGlobal: the bit words of kBits kept in HBM (one replica): no LDS limit on the size, every
//           neighbour gather is an L2 access — the slow path for clusters beyond ~1.3e6 spins.
//   kNibbles: four bits per position — two positions share a byte — for M <= 4 replicas: twice the
//           capacity of kBytes (~2.4e5 spins) at four replicas per workgroup instead of kBits'
//           one; a flip is an LDS atomic XOR on the word that holds the nibble.
constexpr int kBytes = 0, kBits = 1, kWide = 2, kGlobal = 3, kNibbles = 6;  // (4, 5: team / shuffled launches, as reported by asp_sa_last_layout)

// kWide: byte m of `word` (0x00 / 0x80) OR 0x3F becomes byte 3 of `hi`, whose lower three bytes
// keep 0xF00000 — i.e. hi = high word of +1.0 or -1.0 — in one v_or_b32_sdwa (byte select on
// the source, byte-3 write with the rest preserved).  Two VALU ops per (term, replica)
// instead of three.
__device__ __forceinline__ double wide_factor(uint32_t word, int m, uint32_t &hi) {
  const uint32_t top = 0x3Fu;
  switch (m) {
    case 0:
      asm("v_or_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD "
          "src1_sel:BYTE_0" : "+v"(hi) : "v"(top), "v"(word));
      break;
    case 1:
      asm("v_or_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD "
          "src1_sel:BYTE_1" : "+v"(hi) : "v"(top), "v"(word));
      break;
    case 2:
      asm("v_or_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD "
          "src1_sel:BYTE_2" : "+v"(hi) : "v"(top), "v"(word));
      break;
    default:
      asm("v_or_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD "
          "src1_sel:BYTE_3" : "+v"(hi) : "v"(top), "v"(word));
      break;
  }
  return __hiloint2double(static_cast<int>(hi), 0);
}

// `one_hi`: four registers holding the high word of 1.0 (kWide rewrites their top byte).
template <int M, int LAYOUT>
__device__ __forceinline__ void accumulate_quad(const Quad &q, const uint8_t *spins,
                                                double (&acc)[M], uint32_t (&one_hi)[4]) {
  constexpr bool PACKED = LAYOUT == kBits;
  uint32_t s[4];
  const uint32_t cs[4] = {q.c.x, q.c.y, q.c.z, q.c.w};
#if ASP_ABL_NO_LDS
#pragma unroll
  for (int j = 0; j < 4; ++j) s[j] = cs[j] & 15u;
#else
  if constexpr (LAYOUT == kGlobal) {
    // words written by other wavefronts of the workgroup during earlier colour steps: read at
    // device scope (past the CU's vector L1)
    const uint32_t *words = reinterpret_cast<const uint32_t *>(spins);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t w = __hip_atomic_load(words + (cs[j] >> 5), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
      s[j] = (w >> (cs[j] & 31u)) & 1u;
    }
  } else if constexpr (LAYOUT == kWide) {
    // columns of the wide plan are LDS byte addresses (position * 4)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = *reinterpret_cast<LdsWord *>(static_cast<uintptr_t>(cs[j]));
    }
  } else if constexpr (PACKED) {
    const uint32_t *words = reinterpret_cast<const uint32_t *>(spins);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = (words[cs[j] >> 5] >> (cs[j] & 31u)) & 1u;
  } else if constexpr (LAYOUT == kNibbles) {
    // position c: byte c / 2, nibble c % 2; the bits above replica m's are ignored by spin_factor
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#if ASP_ABS_LDS
      const uint32_t byte = *reinterpret_cast<LdsByte *>(static_cast<uintptr_t>(cs[j] >> 1));
#else
      const uint32_t byte = spins[cs[j] >> 1];
#endif
      s[j] = byte >> ((cs[j] & 1u) << 2);
    }
  } else {
#if ASP_ABS_LDS
    // the spin bytes start at LDS address 0 (checked in the kernel prologue), so a position IS
    // its LDS address: no base add in front of every ds_read_u8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = *reinterpret_cast<LdsByte *>(static_cast<uintptr_t>(cs[j]));
    }
#else
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = spins[cs[j]];
#endif
  }
#endif
#if ASP_ABL_NO_FMA
  asm volatile("" ::"v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(q.v01.x), "v"(q.v01.y),
               "v"(q.v23.x), "v"(q.v23.y));
  return;
#endif
#if ASP_J_MAJOR
  // neighbour-major: consecutive FMAs go to different accumulators (each acc[m] still receives
  // its terms in ascending k)
  const double vs[4] = {q.v01.x, q.v01.y, q.v23.x, q.v23.y};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if constexpr (LAYOUT == kWide) {
        acc[m] = __builtin_fma(vs[j], wide_factor(s[j], m, one_hi[m & 3]), acc[m]);
      } else {
        acc[m] = __builtin_fma(vs[j], spin_factor(s[j], m), acc[m]);
      }
    }
  }
  return;
#endif
#pragma unroll
  for (int m = 0; m < M; ++m) {
    double x = acc[m];
    if constexpr (LAYOUT == kWide) {
      x = __builtin_fma(q.v01.x, wide_factor(s[0], m, one_hi[0]), x);
      x = __builtin_fma(q.v01.y, wide_factor(s[1], m, one_hi[1]), x);
      x = __builtin_fma(q.v23.x, wide_factor(s[2], m, one_hi[2]), x);
      x = __builtin_fma(q.v23.y, wide_factor(s[3], m, one_hi[3]), x);
    } else {
      x = __builtin_fma(q.v01.x, spin_factor(s[0], m), x);
      x = __builtin_fma(q.v01.y, spin_factor(s[1], m), x);
      x = __builtin_fma(q.v23.x, spin_factor(s[2], m), x);
      x = __builtin_fma(q.v23.y, spin_factor(s[3], m), x);
    }
    acc[m] = x;
  }
}

}  // namespace dev
}  // namespace asp
