/* ORACLE — test infrastructure only; never linked into or called by the product.
 *
 * CPU restatement of the Hamiltonian action + coupling build chain of the reference's
 * make_ising_model (annealing_sign_problem/common.py:131-208) for symmetry-free two-site
 * operators, written the way the reference does it — materialise every connection, search,
 * build the CSR matrix M with explicit zeros, then J = 0.5 * (M + M^T) with scipy's
 * accumulate-then-prune semantics — so that it checks the HIP path's shortcuts (no
 * materialisation, no transpose) rather than repeating them.
 *
 *   action            lattice_symmetries' batched_apply (third party, not in tree): unpinned;
 *                     call contract at common.py:96-103 (diagonal entry + off-diagonals)
 *   search + clip     common.py:116-128, 172-173
 *   elements          common.py:71-82
 *   symmetrise        common.py:190-196 (scipy csr + csr: sums per row, zeros dropped)
 *   extension         common.py:516-522
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

static uint64_t flip_mask(uint32_t a, uint32_t b, uint32_t x) {
  return ((x & 2u) ? (1ull << a) : 0ull) | ((x & 1u) ? (1ull << b) : 0ull);
}

/* Entries of one key in the order (diagonal, then bond, then dst).  Returns the count. */
static uint64_t apply_one(uint32_t num_bonds, const uint8_t *site_a, const uint8_t *site_b,
                          const double *m, uint64_t key, uint64_t *other, double *coeff) {
  uint64_t n = 1;
  double diagonal = 0.0;
  for (uint32_t k = 0; k < num_bonds; ++k) {
    const double *mk = m + (size_t)k * 16;
    const uint32_t a = site_a[k], b = site_b[k];
    const uint32_t src = (uint32_t)(((key >> a) & 1ull) * 2ull + ((key >> b) & 1ull));
    diagonal = diagonal + mk[src * 4 + src];
    for (uint32_t dst = 0; dst < 4; ++dst) {
      if (dst == src || mk[dst * 4 + src] == 0.0) continue;
      other[n] = key ^ flip_mask(a, b, src ^ dst);
      coeff[n] = mk[dst * 4 + src];
      ++n;
    }
  }
  other[0] = key;
  coeff[0] = diagonal;
  return n;
}

/* Flat batched_apply.  other/coeff need n * (1 + 3 * num_bonds) entries. */
uint64_t oracle_operator_apply(uint32_t num_bonds, const uint8_t *site_a, const uint8_t *site_b,
                               const double *m, uint64_t n, const uint64_t *keys,
                               uint64_t *other, double *coeff, int64_t *counts) {
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t c = apply_one(num_bonds, site_a, site_b, m, keys[i], other + total, coeff + total);
    counts[i] = (int64_t)c;
    total += c;
  }
  return total;
}

static uint64_t lower_bound(const uint64_t *keys, uint64_t n, uint64_t x) {
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (keys[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

static int cmp_i64(const void *a, const void *b) {
  const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* J = 0.5 * (M + M^T) as sorted COO; row/col/val need capacity for every connection.
 * Returns nnz, or UINT64_MAX when out of memory. */
uint64_t oracle_operator_ising(uint32_t num_bonds, const uint8_t *site_a, const uint8_t *site_b,
                               const double *m, uint64_t K, const uint64_t *keys,
                               const double *psi, int32_t *row, int32_t *col, double *val) {
  if (K == 0) return 0;
  const uint64_t per = 1 + 3ull * num_bonds;
  uint64_t *other = malloc(sizeof(uint64_t) * K * per);
  double *elem = malloc(sizeof(double) * K * per);
  int64_t *idx = malloc(sizeof(int64_t) * K * per);
  int64_t *offsets = malloc(sizeof(int64_t) * (K + 1));
  int64_t *counts = malloc(sizeof(int64_t) * K);
  if (!other || !elem || !idx || !offsets || !counts) return UINT64_MAX;
  const uint64_t N = oracle_operator_apply(num_bonds, site_a, site_b, m, K, keys, other, elem, counts);
  offsets[0] = 0;
  for (uint64_t r = 0; r < K; ++r) offsets[r + 1] = offsets[r] + counts[r];
  /* search, clip, membership, elements (two separately rounded products) */
  for (uint64_t r = 0; r < K; ++r) {
    for (int64_t e = offsets[r]; e < offsets[r + 1]; ++e) {
      uint64_t at = lower_bound(keys, K, other[e]);
      if (at > K - 1) at = K - 1;
      const int member = keys[at] == other[e];
      const double other_psi = member ? psi[at] : 0.0;
      idx[e] = (int64_t)at;
      elem[e] = (elem[e] * fabs(other_psi)) * fabs(psi[r]);
    }
  }
  /* transpose of M as CSR: entries of column c in ascending row order */
  int64_t *t_start = calloc(K + 1, sizeof(int64_t));
  int64_t *t_row = malloc(sizeof(int64_t) * (N ? N : 1));
  double *t_val = malloc(sizeof(double) * (N ? N : 1));
  int64_t *cursor = malloc(sizeof(int64_t) * K);
  double *sums = calloc(K, sizeof(double));   /* row of M */
  double *sums_t = calloc(K, sizeof(double)); /* row of M^T */
  uint8_t *seen = calloc(K, 1);
  int64_t *touched = malloc(sizeof(int64_t) * K);
  if (!t_start || !t_row || !t_val || !cursor || !sums || !sums_t || !seen || !touched) return UINT64_MAX;
  for (uint64_t e = 0; e < N; ++e) t_start[idx[e] + 1] += 1;
  for (uint64_t c = 0; c < K; ++c) t_start[c + 1] += t_start[c];
  memcpy(cursor, t_start, sizeof(int64_t) * K);
  for (uint64_t r = 0; r < K; ++r) {
    for (int64_t e = offsets[r]; e < offsets[r + 1]; ++e) {
      const int64_t at = cursor[idx[e]]++;
      t_row[at] = (int64_t)r;
      t_val[at] = elem[e];
    }
  }
  /* row by row: accumulate M's entries and M^T's entries separately (scipy's csr_binop_csr), keep
   * the columns whose two sums do not add to zero, sort by column */
  uint64_t nnz = 0;
  for (uint64_t r = 0; r < K; ++r) {
    uint64_t nt = 0;
    for (int64_t e = offsets[r]; e < offsets[r + 1]; ++e) {
      const int64_t c = idx[e];
      if (!seen[c]) { seen[c] = 1; touched[nt++] = c; }
      sums[c] = sums[c] + elem[e];
    }
    for (int64_t e = t_start[r]; e < t_start[r + 1]; ++e) {
      const int64_t c = t_row[e];
      if (!seen[c]) { seen[c] = 1; touched[nt++] = c; }
      sums_t[c] = sums_t[c] + t_val[e];
    }
    qsort(touched, nt, sizeof(int64_t), cmp_i64);
    for (uint64_t k = 0; k < nt; ++k) {
      const int64_t c = touched[k];
      const double both = sums[c] + sums_t[c]; /* scipy: op(A_row[j], B_row[j]), kept if != 0 */
      if (both != 0.0) {
        row[nnz] = (int32_t)r;
        col[nnz] = (int32_t)c;
        val[nnz] = 0.5 * both; /* `0.5 * matrix` scales the stored data, nothing is pruned */
        ++nnz;
      }
      sums[c] = 0.0;
      sums_t[c] = 0.0;
      seen[c] = 0;
    }
  }
  free(other); free(elem); free(idx); free(offsets); free(counts);
  free(t_start); free(t_row); free(t_val); free(cursor); free(sums); free(sums_t); free(seen); free(touched);
  return nnz;
}

static int cmp_u64(const void *a, const void *b) {
  const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* Sorted unique union of all targets; out needs n * (1 + 3 * num_bonds) entries. */
uint64_t oracle_operator_extend(uint32_t num_bonds, const uint8_t *site_a, const uint8_t *site_b,
                                const double *m, uint64_t n, const uint64_t *keys, uint64_t *out) {
  const uint64_t per = 1 + 3ull * num_bonds;
  double *coeff = malloc(sizeof(double) * (n ? n : 1) * per);
  int64_t *counts = malloc(sizeof(int64_t) * (n ? n : 1));
  if (!coeff || !counts) return UINT64_MAX;
  const uint64_t N = oracle_operator_apply(num_bonds, site_a, site_b, m, n, keys, out, coeff, counts);
  free(coeff);
  free(counts);
  qsort(out, N, sizeof(uint64_t), cmp_u64);
  uint64_t u = 0;
  for (uint64_t i = 0; i < N; ++i) {
    if (i == 0 || out[i] != out[i - 1]) out[u++] = out[i];
  }
  return u;
}
