// Development probe: throughput of returning device-scope atomics on per-"pair" counter arrays, as the
// wide order build's level kernel issues them.  hipcc --offload-arch=gfx950 -O3 tools/atomics_probe.hip
//   mapping 0: the parts of a pair are consecutive blocks (spread over the eight XCDs)
//   mapping 1: the parts of a pair are blocks of equal index mod 8 (one XCD, if blocks are dealt round-robin)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <bool RET>
__global__ __launch_bounds__(256) void k_probe(uint32_t *counters, uint32_t words_per_pair, uint32_t pairs, uint32_t parts,
                                               uint32_t rounds, int mapping, uint32_t *sink) {
  uint32_t pair, part;
  if (mapping == 0) {
    pair = blockIdx.x / parts;
    part = blockIdx.x % parts;
  } else {
    const uint32_t xcd = blockIdx.x & 7u, y = blockIdx.x >> 3;
    pair = (y / parts) * 8u + xcd;
    part = y % parts;
  }
  if (pair >= pairs) return;
  uint32_t *mine = counters + static_cast<uint64_t>(pair) * words_per_pair;
  uint32_t x = (part * 256u + threadIdx.x) * 2654435761u + pair * 40503u + 12345u, acc = 0;
  for (uint32_t r = 0; r < rounds; ++r) {
    x = x * 1664525u + 1013904223u;
    const uint32_t at = (x >> 8) % words_per_pair;
    if (RET) {
      acc += atomicSub(mine + at, 1u);
    } else {
      __hip_atomic_fetch_sub(mine + at, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (acc == 0x12345u) sink[0] = acc;
}

int main() {
  const uint32_t pairs = 480, parts = 32, rounds = 64;
  uint32_t *sink;
  CHECK(hipMalloc(&sink, 64));
  for (uint32_t kib : {86u, 344u, 800u}) {
    const uint32_t words = kib * 256u;
    uint32_t *counters;
    CHECK(hipMalloc(&counters, static_cast<size_t>(pairs) * words * 4));
    CHECK(hipMemset(counters, 0x7f, static_cast<size_t>(pairs) * words * 4));
    for (int ret = 1; ret >= 0; --ret) {
      for (int mapping = 0; mapping < 2; ++mapping) {
        for (uint32_t np : {480u, 64u, 8u}) {
          const uint32_t blocks = mapping == 0 ? np * parts : ((np + 7u) & ~7u) * parts;
          hipEvent_t a, b;
          CHECK(hipEventCreate(&a));
          CHECK(hipEventCreate(&b));
          float best = 1e9f;
          for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(a, 0));
            if (ret) {
              hipLaunchKernelGGL(k_probe<true>, dim3(blocks), dim3(256), 0, 0, counters, words, np, parts, rounds, mapping, sink);
            } else {
              hipLaunchKernelGGL(k_probe<false>, dim3(blocks), dim3(256), 0, 0, counters, words, np, parts, rounds, mapping, sink);
            }
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
          }
          const double atomics = static_cast<double>(np) * parts * 256 * rounds;
          std::printf("%4u KiB per pair, %3u pairs x %u parts, %s, mapping %d: %8.3f ms = %7.1f G atomics/s\n", kib, np, parts,
                      ret ? "returning" : "no return", mapping, best, atomics / best * 1e-6);
        }
      }
    }
    CHECK(hipFree(counters));
  }
  return 0;
}
