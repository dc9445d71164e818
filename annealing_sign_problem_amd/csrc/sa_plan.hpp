// Sweep plan: the host-side preprocessing of an Ising Hamiltonian into the
// layout the gfx950 sweep kernel streams.
//
//   A = offdiag(J + J^T) (exact zeros dropped)          -> dE_i = -2 s_i (sum_j A_ij s_j + h_i)
//   DSATUR colouring of A's graph                        -> same-colour spins are independent
//   permutation by (colour, degree desc, index)          -> a colour class is a run of 64-row blocks
//   sliced ELL: block b has width w_b (multiple of 4) and starts at slab ell_off[b]; entries
//   are stored quad-interleaved (four consecutive k of a lane adjacent, see sa_plan.cpp) so a
//   lane fetches a quad with three 16-byte loads and every wavefront load instruction covers
//   1 KiB of contiguous memory.  Columns are padded POSITIONS (block * 64 + lane), i.e. direct
//   indices into the LDS spin array.  Padding entries point at the lane's own position with
//   value +0.0 (adding +-0 never changes a sum).
#pragma once

#include <cstdint>
#include <vector>

#include "asp.h"

namespace asp {

constexpr uint32_t kDummySpin = 0xFFFFFFFFu;  // spin_of_pos of a padding lane
constexpr int kWidthAlign = 4;                // block widths are rounded up to this
constexpr int kEllTailSlabs = 8;              // zero slabs after the last block (prefetch over-read)

struct SaHostLayout {
  uint64_t num_spins = 0;
  // A in CSR over ORIGINAL indices (sorted columns)
  std::vector<int64_t> a_ptr;
  std::vector<int32_t> a_col;
  std::vector<double> a_val;
  double diag_sum = 0.0;
  std::vector<int32_t> color;
  uint32_t num_colors = 0;
  uint32_t max_degree = 0;
  // permuted / padded structure
  uint32_t num_blocks = 0;
  std::vector<uint32_t> color_block_start;  // num_colors + 1
  std::vector<uint32_t> block_width;        // num_blocks
  std::vector<uint64_t> ell_off;            // num_blocks + 1, in 64-entry slabs
  std::vector<uint32_t> spin_of_pos;        // num_blocks * 64, kDummySpin for padding lanes
  std::vector<uint32_t> pos_of_spin;        // num_spins
  std::vector<double> field_pos;            // num_blocks * 64
  std::vector<uint32_t> ell_col;            // (ell_off.back() + kEllTailSlabs) * 64
  std::vector<double> ell_val;
  int32_t energy_scale_exp = 0;
  double beta0_auto = 0.0, beta1_auto = 0.0;
};

// Returns ASP_OK or records an error (non-canonical CSR, index out of range ...).
int build_sa_layout(uint64_t num_spins, const int64_t *indptr, const int32_t *indices,
                    const double *data, const double *field, SaHostLayout *out);

// Visiting orders of the "shuffled" sweep: sweep t visits the spins in ascending
// (priority, index), priority_t(i) = word 0 of Philox4x32-10(counter (i, t, 0xFFFFFFFE, 0), key
// seed).  A sequential sweep in that order equals visiting the LEVELS of the priority graph one
// after another (level(i) = 1 + max level of the neighbours that come before i; spins of a level
// are pairwise non-adjacent), which is what the device does.  For sweeps first..first+count-1:
// order[s * K + k] = k-th spin of sweep s in level-major order, level_start[s * cap + l] = first
// position of level l (l = 0..num_levels[s]; later entries = K), cap = the chunk's largest number
// of levels + 1 (the longest descending-priority path; not bounded by the degree).
void shuffled_orders(const SaHostLayout &layout, uint64_t seed, uint32_t first, uint32_t count,
                     uint32_t *order, std::vector<uint32_t> *level_start, uint32_t *cap,
                     uint32_t *num_levels);

// Rows of A over ORIGINAL indices padded to whole quads, for the shuffled sweep's order kernel
// (csrc/sa_shuffled.hip): row i is quads quad_ptr[i] .. quad_ptr[i + 1], entry k of the row is
// col[(quad_ptr[i] + k / 4) * 4 + k % 4] / val[same]; padding entries carry the row's own index
// and +0.0 (A has no diagonal, so "column == row" identifies padding).
struct RowQuads {
  std::vector<uint32_t> quad_ptr;  // num_spins + 1
  std::vector<uint32_t> col;       // quads * 4
  std::vector<double> val;         // quads * 4
  uint32_t max_quads = 0;          // longest row
};
int build_row_quads(const SaHostLayout &layout, RowQuads *out);

// Greedy sign assignment before relaxation (specification DESIGN.md §4.8): packed
// configuration (bit = +1), ceil(K/64) words.
int greedy_tree_signs(const SaHostLayout &layout, uint64_t *x);

}  // namespace asp
