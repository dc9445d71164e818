// Greedy sign assignment (specification "ASP-GREEDY-1"): the host half of
// asp_sa_greedy.  Replaces ising_glass_annealer.greedy_solve (call site
// annealing_sign_problem/common.py:250); the reference's only in-tree description of the
// algorithm is the commented prototype strongest_coupling_greedy_color at common.py:298-438,
// whose three cases (both spins new / one new / two clusters) are followed here with a
// union-find that carries the sign of every spin relative to its cluster root.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "asp_common.hpp"
#include "sa_plan.hpp"

namespace asp {

namespace {

struct Bond {
  uint32_t i, j;  // i < j
  double weight;  // A_ij
};

struct SignedForest {
  std::vector<int32_t> parent;  // -1: spin not assigned to any cluster yet
  std::vector<uint32_t> size;
  std::vector<uint8_t> flip;    // sign relative to parent: 1 = opposite

  explicit SignedForest(size_t n) : parent(n, -1), size(n, 1), flip(n, 0) {}

  bool assigned(uint32_t v) const { return parent[v] >= 0; }
  void start(uint32_t v) { parent[v] = static_cast<int32_t>(v); }

  // Root of v and the sign of v relative to it (0 = same), compressing the path.
  std::pair<uint32_t, uint8_t> find(uint32_t v) {
    uint32_t root = v;
    uint8_t sign = 0;
    while (static_cast<uint32_t>(parent[root]) != root) {
      sign ^= flip[root];
      root = static_cast<uint32_t>(parent[root]);
    }
    uint8_t carried = sign;
    while (static_cast<uint32_t>(parent[v]) != root && v != root) {
      const uint32_t up = static_cast<uint32_t>(parent[v]);
      const uint8_t mine = flip[v];
      parent[v] = static_cast<int32_t>(root);
      flip[v] = carried;
      carried ^= mine;
      v = up;
    }
    return {root, sign};
  }
};

}  // namespace

int greedy_tree_signs(const SaHostLayout &L, uint64_t *x) {
  const size_t n = L.num_spins;
  std::vector<Bond> bonds;
  bonds.reserve(L.a_col.size() / 2);
  for (size_t i = 0; i < n; ++i) {
    for (int64_t k = L.a_ptr[i]; k < L.a_ptr[i + 1]; ++k) {
      const uint32_t j = static_cast<uint32_t>(L.a_col[k]);
      if (j > i) bonds.push_back(Bond{static_cast<uint32_t>(i), j, L.a_val[k]});
    }
  }
  // strongest first; ties keep (i, j) ascending (the generation order).  The bit pattern of
  // |w| orders like |w|, so this is a stable LSD radix sort (four 16-bit digits) of the
  // complemented patterns — the order std::stable_sort with `|a| > |b|` gives, several times
  // faster on the ~1e6 bonds of a large cluster.
  {
    const size_t m = bonds.size();
    std::vector<uint64_t> key(m), key_tmp(m);
    std::vector<uint32_t> idx(m), idx_tmp(m);
    for (size_t b = 0; b < m; ++b) {
      uint64_t bits;
      const double magnitude = std::fabs(bonds[b].weight);
      std::memcpy(&bits, &magnitude, sizeof bits);
      key[b] = ~bits;
      idx[b] = static_cast<uint32_t>(b);
    }
    std::vector<size_t> count(65536);
    for (int digit = 0; digit < 4; ++digit) {
      const int shift = 16 * digit;
      std::fill(count.begin(), count.end(), 0);
      for (size_t b = 0; b < m; ++b) count[(key[b] >> shift) & 0xFFFFu]++;
      size_t running = 0;
      for (size_t d = 0; d < 65536; ++d) {
        const size_t c = count[d];
        count[d] = running;
        running += c;
      }
      for (size_t b = 0; b < m; ++b) {
        const size_t at = count[(key[b] >> shift) & 0xFFFFu]++;
        key_tmp[at] = key[b];
        idx_tmp[at] = idx[b];
      }
      key.swap(key_tmp);
      idx.swap(idx_tmp);
    }
    std::vector<Bond> sorted(m);
    for (size_t b = 0; b < m; ++b) sorted[b] = bonds[idx[b]];
    bonds.swap(sorted);
  }

  SignedForest forest(n);
  for (const Bond &bond : bonds) {
    const bool has_i = forest.assigned(bond.i), has_j = forest.assigned(bond.j);
    if (!has_i && !has_j) {
      // new cluster {i: +1, j: -sign(w)}: the bond is satisfied (common.py:397-403)
      forest.start(bond.i);
      forest.parent[bond.j] = static_cast<int32_t>(bond.i);
      forest.flip[bond.j] = bond.weight > 0.0 ? 1 : 0;
      forest.size[bond.i] = 2;
    } else if (has_i != has_j) {
      // a new spin joins a cluster with the sign that lowers the energy of ALL its bonds
      // into that cluster (common.py:377-395)
      const uint32_t fresh = has_i ? bond.j : bond.i;
      const uint32_t root = forest.find(has_i ? bond.i : bond.j).first;
      double energy = 0.0;
      for (int64_t k = L.a_ptr[fresh]; k < L.a_ptr[fresh + 1]; ++k) {
        const uint32_t other = static_cast<uint32_t>(L.a_col[k]);
        if (!forest.assigned(other)) continue;
        const auto [r, sign] = forest.find(other);
        if (r != root) continue;
        energy = energy + (sign ? -L.a_val[k] : L.a_val[k]);
      }
      forest.parent[fresh] = static_cast<int32_t>(root);
      forest.flip[fresh] = energy > 0.0 ? 1 : 0;
      forest.size[root] += 1;
    } else {
      auto [ri, si] = forest.find(bond.i);
      auto [rj, sj] = forest.find(bond.j);
      if (ri == rj) continue;  // all earlier bonds were stronger (common.py:354-358)
      // flip the second cluster iff the bond is frustrated as the clusters stand
      // (common.py:362-365): s_i s_j w > 0
      const bool frustrated = (si == sj) == (bond.weight > 0.0);
      uint32_t keep = ri, gone = rj;
      if (forest.size[rj] > forest.size[ri]) std::swap(keep, gone);
      forest.parent[gone] = static_cast<int32_t>(keep);
      forest.flip[gone] = frustrated ? 1 : 0;
      forest.size[keep] += forest.size[gone];
    }
  }

  // signs relative to the roots; spins never touched by a bond are +1 clusters of their own
  std::vector<uint8_t> down(n, 0);
  std::vector<uint32_t> root_of(n, 0);
  for (size_t v = 0; v < n; ++v) {
    if (!forest.assigned(static_cast<uint32_t>(v))) {
      root_of[v] = static_cast<uint32_t>(v);
      continue;
    }
    const auto [r, sign] = forest.find(static_cast<uint32_t>(v));
    root_of[v] = r;
    down[v] = sign;
  }
  // orientation of every cluster: the one that lowers sum_i h_i s_i (no-op without a field)
  {
    std::vector<double> field_energy(n, 0.0);
    for (size_t v = 0; v < n; ++v) {
      const double h = L.field_pos[L.pos_of_spin[v]];
      field_energy[root_of[v]] = field_energy[root_of[v]] + (down[v] ? -h : h);
    }
    for (size_t v = 0; v < n; ++v) {
      if (field_energy[root_of[v]] > 0.0) down[v] ^= 1;
    }
  }
  const size_t words = (n + 63) / 64;
  std::fill(x, x + words, 0ull);
  for (size_t v = 0; v < n; ++v) {
    if (!down[v]) x[v / 64] |= 1ull << (v % 64);
  }
  return ASP_OK;
}

}  // namespace asp

// Host-only: the cluster-merging half of asp_sa_greedy without a device (inspection, CPU tests).
extern "C" int asp_sa_greedy_tree_host(uint64_t num_spins, int64_t const *indptr,
                                       int32_t const *indices, double const *data,
                                       double const *field, uint64_t *out_x) {
  asp_clear_error();
  if (num_spins && !out_x) return asp::set_error(ASP_ERR_INVALID, "null output");
  asp::SaHostLayout layout;
  ASP_TRY(asp::build_sa_layout(num_spins, indptr, indices, data, field, &layout));
  if (num_spins == 0) return ASP_OK;
  return asp::greedy_tree_signs(layout, out_x);
}
