// Host preprocessing for the annealing sweep (see sa_plan.hpp).
// Replaces what ising_glass_annealer.Hamiltonian does on construction
// (call sites annealing_sign_problem/common.py:204,681).
#include "sa_plan.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <numeric>
#include <queue>
#include <thread>

#include "asp_common.hpp"

namespace asp {

namespace {

struct Entry {
  int32_t col;
  double val;
};

// f(begin, end, part) over `parts` contiguous ranges of [0, n), on that many host threads.
// Everything run through this writes disjoint memory; anything whose floating-point result
// depends on the order (sums) stays sequential elsewhere.
template <typename F>
void parallel_ranges(uint64_t n, unsigned parts, F f) {
  if (parts <= 1 || n < 2 * static_cast<uint64_t>(parts)) {
    f(static_cast<uint64_t>(0), n, 0u);
    return;
  }
  std::vector<std::thread> pool;
  pool.reserve(parts - 1);
  const uint64_t step = (n + parts - 1) / parts;
  for (unsigned part = 1; part < parts; ++part) {
    const uint64_t begin = std::min<uint64_t>(n, part * step), end = std::min<uint64_t>(n, begin + step);
    pool.emplace_back([=] { f(begin, end, part); });
  }
  f(static_cast<uint64_t>(0), std::min<uint64_t>(n, step), 0u);
  for (auto &t : pool) t.join();
}

unsigned host_threads(uint64_t n) {
  if (n < 20000) return 1;  // thread start-up would cost more than the loops
  if (const char *env = std::getenv("ASP_HOST_THREADS")) {
    const int forced = std::atoi(env);
    if (forced >= 1) return static_cast<unsigned>(std::min(forced, 64));
  }
  const unsigned hw = std::thread::hardware_concurrency();
  // the loops are memory-latency bound: four threads take what there is to take (measured on
  // the GPU box: 92.6 -> 68.3 ms at K = 1e5, dbar = 23; eight or sixteen are no faster)
  return std::max(1u, std::min(4u, hw ? hw : 1u));
}

}  // namespace

int build_sa_layout(uint64_t num_spins, const int64_t *indptr, const int32_t *indices,
                    const double *data, const double *field, SaHostLayout *out) {
  const uint64_t n = num_spins;
  // ASP_PLAN_TIMING=1: stage times of this function on stderr (development aid)
  const bool timing = std::getenv("ASP_PLAN_TIMING") != nullptr;
  auto clock_now = [] { return std::chrono::steady_clock::now(); };
  auto stage_start = clock_now();
  auto stage = [&](const char *name) {
    if (!timing) return;
    const auto now = clock_now();
    std::fprintf(stderr, "  plan stage %-12s %8.3f ms\n", name,
                 std::chrono::duration<double, std::milli>(now - stage_start).count());
    stage_start = now;
  };
  if (n >= (1ull << 31)) {
    return set_error(ASP_ERR_TOO_LARGE, "%llu spins exceed the 2^31 limit", (unsigned long long)n);
  }
  if (n > 0 && (!indptr || !field)) return set_error(ASP_ERR_INVALID, "null indptr/field");
  SaHostLayout &L = *out;
  L = SaHostLayout();
  L.num_spins = n;
  if (n == 0) {
    L.a_ptr.assign(1, 0);
    L.color_block_start.assign(1, 0);
    L.ell_off.assign(1, 0);
    L.beta0_auto = L.beta1_auto = 1.0;
    return ASP_OK;
  }
  if (indptr[0] != 0) return set_error(ASP_ERR_INVALID, "indptr[0] must be 0");
  const int64_t nnz = indptr[n];
  if (nnz > 0 && (!indices || !data)) return set_error(ASP_ERR_INVALID, "null indices/data");
  for (uint64_t i = 0; i < n; ++i) {
    if (indptr[i + 1] < indptr[i]) return set_error(ASP_ERR_INVALID, "indptr is not monotone");
    for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
      if (indices[k] < 0 || static_cast<uint64_t>(indices[k]) >= n) {
        return set_error(ASP_ERR_INVALID, "column index %d out of range in row %llu", indices[k],
                         (unsigned long long)i);
      }
      if (k > indptr[i] && indices[k - 1] >= indices[k]) {
        return set_error(ASP_ERR_INVALID,
                         "row %llu is not in canonical CSR form (sorted, duplicate-free columns)",
                         (unsigned long long)i);
      }
      if (!std::isfinite(data[k])) {
        return set_error(ASP_ERR_INVALID, "non-finite coupling in row %llu", (unsigned long long)i);
      }
    }
    if (!std::isfinite(field[i])) {
      return set_error(ASP_ERR_INVALID, "non-finite field at %llu", (unsigned long long)i);
    }
  }

  stage("validate");
  // ---- is J exactly symmetric?  (What asp_operator_ising and asp_sparsify_component deliver: the
  //      sampled-cluster pipeline's every model.)  Then J^T = J, row i of A is row i of J without its
  //      diagonal and with x + x for x + y — the same bits as the merge below produces — and the
  //      transposition and the merge, 40 % of this function, are not needed.  Decided by two 64-bit
  //      multiset hashes of the entries above and below the diagonal, keyed by (min, max, bits of
  //      the value): equal multisets cancel exactly; unequal ones cancel in BOTH sums with
  //      probability 2^-128, and only then would a matrix be taken for symmetric that is not.
  bool symmetric = true;
  {
    auto mix = [](uint64_t x) {  // splitmix64 finaliser
      x ^= x >> 30;
      x *= 0xBF58476D1CE4E5B9ull;
      x ^= x >> 27;
      x *= 0x94D049BB133111EBull;
      x ^= x >> 31;
      return x;
    };
    uint64_t sum_a = 0, sum_b = 0;
    for (uint64_t i = 0; i < n; ++i) {
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
        const uint64_t j = static_cast<uint64_t>(indices[k]);
        if (j == i) continue;
        uint64_t bits;
        std::memcpy(&bits, &data[k], sizeof bits);
        const uint64_t lo = i < j ? i : j, hi = i < j ? j : i;
        const uint64_t pair = mix((lo << 32) | hi);
        const uint64_t a = mix(pair ^ bits), b = mix((pair + 0x9E3779B97F4A7C15ull) ^ (bits * 3 + 1));
        if (i < j) {
          sum_a += a;
          sum_b += b;
        } else {
          sum_a -= a;
          sum_b -= b;
        }
      }
    }
    symmetric = sum_a == 0 && sum_b == 0;
    // test hook: pretend the hashes collided, so that the exact confirmation below decides alone
    if (std::getenv("ASP_PLAN_HASHES_SAY_SYMMETRIC")) symmetric = true;
  }
  const unsigned parts = host_threads(n);
  if (symmetric) {
    // The hashes are the fast NEGATIVE test; a yes is confirmed exactly before the short cut is
    // taken (a false yes would silently build A = 2 J from the upper rows): every entry above the
    // diagonal finds its mirror with the same bits by a binary search in the mirror's row, and
    // there are as many entries below the diagonal as above (rows are duplicate-free, so the map
    // entry -> mirror is injective and the counts make it onto).  O(nnz log d), rows in parallel.
    std::vector<uint8_t> part_ok(parts, 1);
    std::vector<int64_t> part_upper(parts, 0), part_lower(parts, 0);
    parallel_ranges(n, parts, [&](uint64_t begin, uint64_t end, unsigned part) {
      int64_t upper = 0, lower = 0;
      bool ok = true;
      for (uint64_t i = begin; i < end && ok; ++i) {
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
          const uint64_t j = static_cast<uint64_t>(indices[k]);
          if (j < i) {
            ++lower;
          } else if (j > i) {
            ++upper;
            const int32_t *row = indices + indptr[j], *row_end = indices + indptr[j + 1];
            const int32_t *at = std::lower_bound(row, row_end, static_cast<int32_t>(i));
            if (at == row_end || static_cast<uint64_t>(*at) != i ||
                std::memcmp(&data[indptr[j] + (at - row)], &data[k], sizeof(double)) != 0) {
              ok = false;
              break;
            }
          }
        }
      }
      part_ok[part] = ok ? 1 : 0;
      part_upper[part] = upper;
      part_lower[part] = lower;
    });
    int64_t upper = 0, lower = 0;
    for (unsigned part = 0; part < parts; ++part) {
      symmetric = symmetric && part_ok[part];
      upper += part_upper[part];
      lower += part_lower[part];
    }
    symmetric = symmetric && upper == lower;
  }
  stage("symmetry");
  // ---- J^T by rows (bucket the entries by column; rows are visited in order so
  //      every bucket ends up sorted by original row) --------------------------
  std::vector<int64_t> t_ptr(symmetric ? 0 : n + 1, 0);
  std::vector<Entry> t_entries(symmetric ? 0 : static_cast<size_t>(nnz));
  if (!symmetric) {
    // parallel counting sort by column: part p owns a contiguous range of rows, so writing the
    // parts' entries of a column one after the other keeps every column sorted by row
    std::vector<int64_t> cursor(static_cast<size_t>(parts) * n, 0);
    parallel_ranges(n, parts, [&](uint64_t begin, uint64_t end, unsigned part) {
      int64_t *mine = cursor.data() + static_cast<size_t>(part) * n;
      for (int64_t k = indptr[begin]; k < indptr[end]; ++k) mine[indices[k]]++;
    });
    int64_t running = 0;
    for (uint64_t j = 0; j < n; ++j) {
      t_ptr[j] = running;
      for (unsigned part = 0; part < parts; ++part) {
        int64_t &slot = cursor[static_cast<size_t>(part) * n + j];
        const int64_t count = slot;
        slot = running;
        running += count;
      }
    }
    t_ptr[n] = running;
    parallel_ranges(n, parts, [&](uint64_t begin, uint64_t end, unsigned part) {
      int64_t *mine = cursor.data() + static_cast<size_t>(part) * n;
      for (uint64_t i = begin; i < end; ++i) {
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
          t_entries[mine[indices[k]]++] = Entry{static_cast<int32_t>(i), data[k]};
        }
      }
    });
  }

  stage("transpose");
  // ---- A = offdiag(J + J^T), D = sum_i J_ii ----------------------------------
  // Row i of A is the merge of row i of J and row i of J^T; two passes (count, fill) so that
  // the rows can be produced in parallel.  emit(col, value) is called in column order.
  auto merge_row = [&](uint64_t i, auto emit, double *diagonal) {
    if (symmetric) {  // J_ji is J_ij
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
        if (static_cast<uint64_t>(indices[k]) == i) {
          if (diagonal) *diagonal = data[k];
          continue;
        }
        const double v = data[k] + data[k];
        if (v != 0.0) emit(indices[k], v);
      }
      return;
    }
    const Entry *t = t_entries.data() + t_ptr[i];
    const Entry *t_end = t_entries.data() + t_ptr[i + 1];
    int64_t k = indptr[i];
    const int64_t k_end = indptr[i + 1];
    while (k < k_end || t < t_end) {
      const bool take_j = k < k_end && (t == t_end || indices[k] <= t->col);
      const bool take_t = t < t_end && (k == k_end || t->col <= indices[k]);
      const int32_t col = take_j ? indices[k] : t->col;
      const double x = take_j ? data[k] : 0.0;  // J_ij or +0
      const double y = take_t ? t->val : 0.0;   // J_ji or +0
      if (take_j) ++k;
      if (take_t) ++t;
      if (static_cast<uint64_t>(col) == i) {
        if (take_j && diagonal) *diagonal = x;
        continue;
      }
      const double v = x + y;
      if (v != 0.0) emit(col, v);
    }
  };
  L.a_ptr.assign(n + 1, 0);
  std::vector<double> diagonal(n, 0.0);
  std::vector<uint8_t> has_diagonal(n, 0);
  parallel_ranges(n, parts, [&](uint64_t begin, uint64_t end, unsigned) {
    for (uint64_t i = begin; i < end; ++i) {
      int64_t count = 0;
      merge_row(i, [&](int32_t, double) { ++count; }, nullptr);
      L.a_ptr[i + 1] = count;
    }
  });
  for (uint64_t i = 0; i < n; ++i) L.a_ptr[i + 1] += L.a_ptr[i];
  L.a_col.resize(static_cast<size_t>(L.a_ptr[n]));
  L.a_val.resize(static_cast<size_t>(L.a_ptr[n]));
  parallel_ranges(n, parts, [&](uint64_t begin, uint64_t end, unsigned) {
    for (uint64_t i = begin; i < end; ++i) {
      int64_t at = L.a_ptr[i];
      double d = 0.0;
      bool seen = false;
      // the diagonal element of J (stored or not) — rows are duplicate-free, so at most one
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
        if (static_cast<uint64_t>(indices[k]) == i) {
          d = data[k];
          seen = true;
        }
      }
      merge_row(i, [&](int32_t col, double v) {
        L.a_col[at] = col;
        L.a_val[at] = v;
        ++at;
      }, nullptr);
      diagonal[i] = d;
      has_diagonal[i] = seen ? 1 : 0;
    }
  });
  double diag = 0.0;  // D in row order, as the oracle sums it
  for (uint64_t i = 0; i < n; ++i) {
    if (has_diagonal[i]) diag = diag + diagonal[i];
  }
  L.diag_sum = diag;
  std::vector<Entry>().swap(t_entries);

  stage("merge A");
  // ---- DSATUR colouring (Brelaz 1979) --------------------------------
  // Repeatedly colour the uncoloured spin with the most DISTINCT colours among its
  // neighbours (ties: larger degree, then smaller index) with the smallest colour none of
  // its neighbours has.  About a third fewer colours than first-fit in index order, i.e. a
  // third fewer barriers per sweep.  Per-spin bitset of the colours seen so far.
  L.color.assign(n, -1);
  {
    uint32_t max_deg = 0;
    for (uint64_t i = 0; i < n; ++i) {
      max_deg = std::max<uint32_t>(max_deg, static_cast<uint32_t>(L.a_ptr[i + 1] - L.a_ptr[i]));
    }
    const size_t words = (static_cast<size_t>(max_deg) + 2 + 63) / 64;  // colours <= max_deg + 1
    std::vector<uint64_t> seen(n * words, 0);
    std::vector<uint32_t> saturation(n, 0);
    // Priority = (saturation desc, degree desc, index asc).  The last two never change: every
    // spin gets its RANK in (degree desc, index asc) once (a counting sort), and the queue is one
    // set of ranks per saturation value — a bit set with 64-ary summary levels, so "smallest
    // rank" is a few count-trailing-zeros and moving a spin up one saturation level a few word
    // operations on arrays that stay in cache (n / 8 bytes per level).
    auto degree_of = [&](uint32_t i) { return static_cast<uint32_t>(L.a_ptr[i + 1] - L.a_ptr[i]); };
    std::vector<uint32_t> rank_of(n), spin_at(n);
    {
      std::vector<uint64_t> start(static_cast<size_t>(max_deg) + 2, 0);
      for (uint64_t i = 0; i < n; ++i) ++start[max_deg - degree_of(static_cast<uint32_t>(i)) + 1];
      for (size_t d = 0; d + 1 < start.size(); ++d) start[d + 1] += start[d];
      for (uint64_t i = 0; i < n; ++i) {  // ascending index inside a degree class
        const uint64_t r = start[max_deg - degree_of(static_cast<uint32_t>(i))]++;
        rank_of[i] = static_cast<uint32_t>(r);
        spin_at[r] = static_cast<uint32_t>(i);
      }
    }
    struct RankSet {
      std::vector<std::vector<uint64_t>> level;  // level[0]: members; level[k + 1]: non-empty words of level[k]
      uint64_t count = 0;
      void init(uint64_t size) {
        uint64_t m = size;
        do {
          m = (m + 63) / 64;
          level.emplace_back(m, 0);
        } while (m > 1);
      }
      void insert(uint64_t i) {
        ++count;
        for (auto &l : level) {
          uint64_t &w = l[i >> 6];
          const bool was_empty = w == 0;
          w |= 1ull << (i & 63);
          if (!was_empty) return;
          i >>= 6;
        }
      }
      void erase(uint64_t i) {
        --count;
        for (auto &l : level) {
          uint64_t &w = l[i >> 6];
          w &= ~(1ull << (i & 63));
          if (w != 0) return;
          i >>= 6;
        }
      }
      uint64_t smallest() const {  // count > 0
        uint64_t i = 0;
        for (size_t k = level.size(); k-- > 0;) i = i * 64 + static_cast<uint64_t>(__builtin_ctzll(level[k][i]));
        return i;
      }
    };
    std::vector<RankSet> waiting(1);
    waiting[0].init(n);
    for (uint64_t r = 0; r < n; ++r) waiting[0].insert(r);
    size_t highest = 0;
    uint32_t ncol = 0;
    for (uint64_t remaining = n; remaining > 0; --remaining) {
      while (waiting[highest].count == 0) --highest;  // remaining > 0: some level holds a spin
      const uint64_t r = waiting[highest].smallest();
      waiting[highest].erase(r);
      const uint32_t v = spin_at[r];
      const uint64_t *mine = &seen[static_cast<size_t>(v) * words];
      uint32_t c = 0;
      while ((mine[c >> 6] >> (c & 63)) & 1ull) ++c;
      L.color[v] = static_cast<int32_t>(c);
      ncol = std::max(ncol, c + 1);
      for (int64_t k = L.a_ptr[v]; k < L.a_ptr[v + 1]; ++k) {
        const uint32_t u = static_cast<uint32_t>(L.a_col[k]);
        if (L.color[u] >= 0) continue;
        uint64_t &word = seen[static_cast<size_t>(u) * words + (c >> 6)];
        if ((word >> (c & 63)) & 1ull) continue;
        word |= 1ull << (c & 63);
        const size_t from = saturation[u]++;
        if (waiting.size() <= from + 1) {
          waiting.emplace_back();
          waiting.back().init(n);
        }
        waiting[from].erase(rank_of[u]);
        waiting[from + 1].insert(rank_of[u]);
        highest = std::max(highest, from + 1);
      }
    }
    L.num_colors = ncol;
  }

  stage("colouring");
  // ---- permutation: (colour asc, degree desc, index asc) ----------------------
  std::vector<uint32_t> order(n);
  std::iota(order.begin(), order.end(), 0u);
  auto degree = [&](uint32_t i) { return L.a_ptr[i + 1] - L.a_ptr[i]; };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    if (L.color[a] != L.color[b]) return L.color[a] < L.color[b];
    return degree(a) > degree(b);
  });
  for (uint64_t i = 0; i < n; ++i) {
    L.max_degree = std::max<uint32_t>(L.max_degree, static_cast<uint32_t>(degree(i)));
  }

  stage("permutation");
  // ---- blocks ------------------------------------------------------------------
  L.color_block_start.assign(L.num_colors + 1, 0);
  L.pos_of_spin.assign(n, 0);
  {
    size_t q = 0;
    uint32_t block = 0;
    for (uint32_t c = 0; c < L.num_colors; ++c) {
      L.color_block_start[c] = block;
      size_t q_end = q;
      while (q_end < n && static_cast<uint32_t>(L.color[order[q_end]]) == c) ++q_end;
      for (size_t q0 = q; q0 < q_end; q0 += 64) {
        uint32_t width = 0;
        for (size_t l = 0; l < 64; ++l) {
          const size_t qq = q0 + l;
          if (qq < q_end) {
            const uint32_t spin = order[qq];
            L.spin_of_pos.push_back(spin);
            L.pos_of_spin[spin] = block * 64u + static_cast<uint32_t>(l);
            L.field_pos.push_back(field[spin]);
            width = std::max<uint32_t>(width, static_cast<uint32_t>(degree(spin)));
          } else {
            L.spin_of_pos.push_back(kDummySpin);
            L.field_pos.push_back(0.0);
          }
        }
        width = (width + kWidthAlign - 1) / kWidthAlign * kWidthAlign;
        L.block_width.push_back(width);
        ++block;
      }
      q = q_end;
    }
    L.color_block_start[L.num_colors] = block;
    L.num_blocks = block;
  }

  stage("blocks");
  // ---- sliced ELL ----------------------------------------------------------------
  L.ell_off.assign(L.num_blocks + 1, 0);
  for (uint32_t b = 0; b < L.num_blocks; ++b) L.ell_off[b + 1] = L.ell_off[b] + L.block_width[b];
  const uint64_t slabs = L.ell_off[L.num_blocks];
  if (slabs >= (1ull << 32)) {
    return set_error(ASP_ERR_TOO_LARGE, "%llu ELL slabs exceed the 32-bit slab index",
                     (unsigned long long)slabs);
  }
  // + kEllTailSlabs of zeros: the sweep kernel's prefetch may read one quad past the end
  L.ell_col.assign((slabs + kEllTailSlabs) * 64, 0);
  L.ell_val.assign((slabs + kEllTailSlabs) * 64, 0.0);
  // Quad-interleaved storage: entries k = 4Q .. 4Q+3 of a lane are adjacent, so the kernel
  // fetches them with three 16-byte loads per lane (one uint4 of columns, two double2 of
  // values), each wavefront instruction covering 1 KiB of contiguous memory.
  //   ell_col[((Q * 64) + lane) * 4 + j]              column of k = 4Q + j
  //   ell_val[(((Q * 2 + h) * 64) + lane) * 2 + j]    value  of k = 4Q + 2h + j
  // with Q = ell_off[b] / 4 + q the global quad index (block widths are multiples of 4).
  parallel_ranges(L.num_blocks, parts, [&](uint64_t b_begin, uint64_t b_end, unsigned) {
  for (uint32_t b = static_cast<uint32_t>(b_begin); b < static_cast<uint32_t>(b_end); ++b) {
    for (uint32_t l = 0; l < 64; ++l) {
      const uint32_t pos = b * 64u + l;
      const uint32_t spin = L.spin_of_pos[pos];
      const int64_t begin = spin == kDummySpin ? 0 : L.a_ptr[spin];
      const int64_t deg = spin == kDummySpin ? 0 : degree(spin);
      for (uint32_t k = 0; k < L.block_width[b]; ++k) {
        const uint64_t quad = (L.ell_off[b] + k) / 4;
        const uint32_t j = k % 4;
        const uint64_t at_col = (quad * 64 + l) * 4 + j;
        const uint64_t at_val = ((quad * 2 + j / 2) * 64 + l) * 2 + (j % 2);
        if (static_cast<int64_t>(k) < deg) {
          L.ell_col[at_col] = L.pos_of_spin[L.a_col[begin + k]];
          L.ell_val[at_val] = L.a_val[begin + k];
        } else {
          L.ell_col[at_col] = pos;  // padding: own position, +0.0
          L.ell_val[at_val] = 0.0;
        }
      }
    }
  }
  });

  stage("ELL fill");
  // ---- energy scale and automatic beta range ----------------------------------------
  double bound = 0.0;      // B = 1/2 sum|A| + sum|h|  (E - D ranges within +-B)
  double max_delta = 0.0;  // max_i 2 (sum_j |A_ij| + |h_i|)
  double min_delta = std::numeric_limits<double>::infinity();
  for (uint64_t i = 0; i < n; ++i) {
    double row = 0.0;
    for (int64_t k = L.a_ptr[i]; k < L.a_ptr[i + 1]; ++k) {
      const double a = std::fabs(L.a_val[k]);
      row += a;
      if (a > 0.0) min_delta = std::min(min_delta, 2.0 * a);
    }
    const double h = std::fabs(field[i]);
    if (h > 0.0) min_delta = std::min(min_delta, 2.0 * h);
    bound += 0.5 * row + h;
    max_delta = std::max(max_delta, 2.0 * (row + h));
  }
  if (!std::isfinite(bound)) return set_error(ASP_ERR_INVALID, "couplings overflow double range");
  if (bound > 0.0) {
    // S: the whole tracked range 2B stays below 2^60 and a single |dE| <= max_delta stays
    // below 2^50, so the kernel can round dE * 2^S to int64 with one f64 add (|x| < 2^51)
    int e = 0, e_single = 0;
    (void)std::frexp(2.0 * bound, &e);    // 2B < 2^e
    (void)std::frexp(max_delta, &e_single);  // max_delta < 2^e_single
    L.energy_scale_exp = std::clamp(std::min(60 - e, 50 - e_single), -1000, 1000);
  }
  if (max_delta > 0.0 && std::isfinite(min_delta)) {
    L.beta0_auto = std::log(2.0) / max_delta;
    L.beta1_auto = std::log(100.0) / min_delta;
  } else {
    L.beta0_auto = L.beta1_auto = 1.0;
  }
  stage("scales");
  return ASP_OK;
}

// ---------------------------------------------------------------------------
// shuffled visiting orders
// ---------------------------------------------------------------------------

namespace {

uint32_t philox_word0(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
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
  return c0;
}

}  // namespace

void shuffled_orders(const SaHostLayout &L, uint64_t seed, uint32_t first, uint32_t count,
                     uint32_t *order, std::vector<uint32_t> *level_start, uint32_t *cap_out,
                     uint32_t *num_levels) {
  const uint64_t K = L.num_spins;
  const uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
  const unsigned parts = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  // (the number of levels is the longest descending-priority path: not bounded by the degree)
  std::vector<std::vector<uint32_t>> starts_of(count);
  parallel_ranges(count, parts, [&](uint64_t begin, uint64_t end, unsigned) {
    std::vector<uint64_t> keys(K);
    std::vector<uint32_t> level(K), fill;
    for (uint64_t s = begin; s < end; ++s) {
      const uint32_t t = first + static_cast<uint32_t>(s);
      for (uint64_t i = 0; i < K; ++i) {
        keys[i] = (static_cast<uint64_t>(philox_word0(static_cast<uint32_t>(i), t, 0xFFFFFFFEu, 0u, k0, k1)) << 32) | i;
      }
      std::sort(keys.begin(), keys.end());
      std::fill(level.begin(), level.end(), 0u);  // 0 = not visited yet
      uint32_t deepest = 0;
      for (uint64_t q = 0; q < K; ++q) {
        const uint32_t i = static_cast<uint32_t>(keys[q]);
        uint32_t above = 0;
        for (int64_t k = L.a_ptr[i]; k < L.a_ptr[i + 1]; ++k) above = std::max(above, level[L.a_col[k]]);
        level[i] = above + 1;
        deepest = std::max(deepest, above + 1);
      }
      // counting sort by level, the order of `keys` kept inside a level
      fill.assign(deepest + 2, 0u);
      for (uint64_t i = 0; i < K; ++i) ++fill[level[i]];  // levels are 1..deepest
      std::vector<uint32_t> &starts = starts_of[s];
      starts.assign(deepest + 1, 0u);
      uint32_t at = 0;
      for (uint32_t l = 1; l <= deepest; ++l) {
        starts[l - 1] = at;
        const uint32_t n = fill[l];
        fill[l] = at;
        at += n;
      }
      starts[deepest] = at;
      num_levels[s] = deepest;
      uint32_t *out = order + s * K;
      for (uint64_t q = 0; q < K; ++q) {
        const uint32_t i = static_cast<uint32_t>(keys[q]);
        out[fill[level[i]]++] = i;
      }
    }
  });
  uint32_t cap = 1;
  for (uint32_t s = 0; s < count; ++s) cap = std::max<uint32_t>(cap, static_cast<uint32_t>(starts_of[s].size()));
  level_start->assign(static_cast<size_t>(count) * cap, static_cast<uint32_t>(K));
  for (uint32_t s = 0; s < count; ++s) {
    std::copy(starts_of[s].begin(), starts_of[s].end(), level_start->begin() + static_cast<size_t>(s) * cap);
  }
  *cap_out = cap;
}

int build_row_quads(const SaHostLayout &L, RowQuads *out) {
  const uint64_t K = L.num_spins;
  RowQuads &R = *out;
  R = RowQuads();
  R.quad_ptr.assign(K + 1, 0);
  uint64_t quads = 0;
  for (uint64_t i = 0; i < K; ++i) {
    const uint64_t q = static_cast<uint64_t>(L.a_ptr[i + 1] - L.a_ptr[i] + 3) / 4;
    R.max_quads = std::max<uint32_t>(R.max_quads, static_cast<uint32_t>(q));
    quads += q;
    if (quads >= (1ull << 30)) {
      return set_error(ASP_ERR_TOO_LARGE, "%llu row quads exceed the 32-bit quad index",
                       (unsigned long long)quads);
    }
    R.quad_ptr[i + 1] = static_cast<uint32_t>(quads);
  }
  R.col.resize(quads * 4);
  R.val.resize(quads * 4);
  parallel_ranges(K, host_threads(K), [&](uint64_t begin, uint64_t end, unsigned) {
    for (uint64_t i = begin; i < end; ++i) {
      const int64_t row = L.a_ptr[i], deg = L.a_ptr[i + 1] - row;
      const uint64_t first = static_cast<uint64_t>(R.quad_ptr[i]) * 4;
      const uint64_t padded = static_cast<uint64_t>(R.quad_ptr[i + 1] - R.quad_ptr[i]) * 4;
      for (uint64_t k = 0; k < padded; ++k) {
        const bool real = static_cast<int64_t>(k) < deg;
        R.col[first + k] = real ? static_cast<uint32_t>(L.a_col[row + k]) : static_cast<uint32_t>(i);
        R.val[first + k] = real ? L.a_val[row + k] : 0.0;
      }
    }
  });
  return ASP_OK;
}

}  // namespace asp

extern "C" int asp_sa_shuffled_order_host(uint64_t num_spins, int64_t const *indptr,
                                          int32_t const *indices, double const *data,
                                          double const *field, uint64_t seed, uint32_t sweep,
                                          uint32_t *order, uint32_t *level_of_position,
                                          uint32_t *num_levels) {
  asp_clear_error();
  asp::SaHostLayout L;
  ASP_TRY(asp::build_sa_layout(num_spins, indptr, indices, data, field, &L));
  std::vector<uint32_t> starts, ord(num_spins ? num_spins : 1);
  uint32_t levels = 0, cap = 0;
  asp::shuffled_orders(L, seed, sweep, 1, ord.data(), &starts, &cap, &levels);
  if (order) std::copy(ord.begin(), ord.begin() + num_spins, order);
  if (level_of_position) {
    for (uint32_t l = 0; l < levels; ++l) {
      for (uint32_t q = starts[l]; q < starts[l + 1]; ++q) level_of_position[q] = l;
    }
  }
  if (num_levels) *num_levels = levels;
  return ASP_OK;
}

static void fill_info(const asp::SaHostLayout &L, asp_sa_info *info) {
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
}

extern "C" int asp_sa_layout_host(uint64_t num_spins, int64_t const *indptr,
                                  int32_t const *indices, double const *data,
                                  double const *field, asp_sa_info *info, int32_t *colors,
                                  uint32_t *position) {
  asp_clear_error();
  asp::SaHostLayout L;
  ASP_TRY(asp::build_sa_layout(num_spins, indptr, indices, data, field, &L));
  if (info) fill_info(L, info);
  if (colors) std::copy(L.color.begin(), L.color.end(), colors);
  if (position) std::copy(L.pos_of_spin.begin(), L.pos_of_spin.end(), position);
  return ASP_OK;
}
