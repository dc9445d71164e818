// Internals of asp_operator shared by operator_apply.hip and sector_basis.hip.
#pragma once

#include <cstdint>
#include <vector>

#include "asp_common.hpp"

namespace asp {

struct Bond {
  double m[16];  // m[dst * 4 + src]
  uint32_t a, b;
  uint64_t flip[4];  // flip[x] = key bits toggled by a transition with src ^ dst == x
};

struct SymmetryArgs {
  const uint8_t *table;  // [num_permutations][64]
  uint32_t num_permutations;
  uint32_t number_spins;
  int32_t inversion;  // 0, +1, -1
  uint64_t mask;      // low number_spins bits
};

}  // namespace asp

struct asp_operator {
  uint32_t number_spins = 0;
  uint32_t num_bonds = 0;
  uint32_t max_connections = 1;
  bool unique_targets = true;
  std::vector<asp::Bond> bonds;
  asp::DeviceBuffer<asp::Bond> d_bonds;
  // symmetry-adapted basis (asp_operator_set_symmetry); num_permutations == 0: plain basis
  uint32_t num_permutations = 0;
  int32_t inversion = 0;
  asp::DeviceBuffer<uint8_t> d_table;
  asp::SymmetryArgs symmetry() const {
    return asp::SymmetryArgs{d_table.ptr, num_permutations, number_spins, inversion,
                        number_spins >= 64 ? ~0ull : ((1ull << number_spins) - 1ull)};
  }
};

