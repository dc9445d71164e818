/* ORACLE — test infrastructure only (see oracle/README.md).
 *
 * CPU restatement of the greedy solver, specification "ASP-GREEDY-1" (DESIGN.md §4.8).
 * PARITY UNPINNED against the reference: the reference calls
 * ising_glass_annealer.greedy_solve (annealing_sign_problem/common.py:250), third-party and
 * unavailable; the only in-tree description is the commented prototype
 * strongest_coupling_greedy_color (common.py:298-438), whose case analysis this file follows:
 *   both spins new            -> new cluster, bond satisfied          common.py:397-403
 *   one spin new              -> joins with the sign that lowers the energy of all its
 *                                bonds into that cluster              common.py:377-395
 *   two different clusters    -> second one flipped iff the bond is frustrated
 *                                                                     common.py:359-376
 *   same cluster              -> nothing                              common.py:354-358
 * followed by strict-descent sweeps until no spin flips (common.py:417-433), here in the
 * colour order of the annealer (sa_oracle.c).
 *
 * Deliberately a different data structure from the product (explicit member lists with
 * small-into-large relabelling instead of a parity union-find).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* from sa_oracle.c */
int oracle_sa_layout(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                     double const *data, int32_t *colors, int64_t *order, int32_t *num_colors,
                     int64_t *nnz_offdiag, double *diag_sum);
int oracle_sa_energy(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                     double const *data, double const *field, uint32_t count, uint64_t const *x,
                     double *out_e);

typedef struct bond {
  int32_t i, j;
  double w;
} bond;

static int bond_cmp(void const *pa, void const *pb) {
  bond const *a = pa, *b = pb;
  double const x = fabs(a->w), y = fabs(b->w);
  if (x != y) return x > y ? -1 : 1;
  if (a->i != b->i) return a->i < b->i ? -1 : 1;
  if (a->j != b->j) return a->j < b->j ? -1 : 1;
  return 0;
}

/* A = offdiag(J + J^T) as dense-free CSR (same definition as sa_oracle.c). */
static void build_a(uint64_t n, int64_t const *indptr, int32_t const *indices, double const *data,
                    int64_t **a_ptr, int32_t **a_col, double **a_val) {
  int64_t const nnz = indptr[n];
  int64_t *cnt = calloc(n + 1, sizeof *cnt);
  /* upper bound of row sizes: own row + column occurrences */
  for (uint64_t i = 0; i < n; ++i) cnt[i] += indptr[i + 1] - indptr[i];
  for (int64_t k = 0; k < nnz; ++k) cnt[indices[k]]++;
  int64_t *start = calloc(n + 1, sizeof *start);
  for (uint64_t i = 0; i < n; ++i) start[i + 1] = start[i] + cnt[i];
  int32_t *col = malloc((size_t)(start[n] + 1) * sizeof *col);
  double *val = malloc((size_t)(start[n] + 1) * sizeof *val);
  uint8_t *from_t = malloc((size_t)(start[n] + 1));
  int64_t *fill = calloc(n + 1, sizeof *fill);
  /* scatter (i,j,J_ij) into row i and row j; then sort each row by column and combine */
  for (uint64_t i = 0; i < n; ++i) {
    for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
      int32_t const j = indices[k];
      if ((uint64_t)j == i) continue;
      int64_t p = start[i] + fill[i]++;
      col[p] = j; val[p] = data[k]; from_t[p] = 0;
      p = start[j] + fill[j]++;
      col[p] = (int32_t)i; val[p] = data[k]; from_t[p] = 1;
    }
  }
  *a_ptr = calloc(n + 1, sizeof **a_ptr);
  int64_t out = 0;
  for (uint64_t i = 0; i < n; ++i) {
    int64_t const lo = start[i], len = fill[i];
    /* insertion sort by (column, from_t): rows are short */
    for (int64_t a = 1; a < len; ++a) {
      int32_t c = col[lo + a]; double v = val[lo + a]; uint8_t t = from_t[lo + a];
      int64_t b = a - 1;
      while (b >= 0 && (col[lo + b] > c || (col[lo + b] == c && from_t[lo + b] > t))) {
        col[lo + b + 1] = col[lo + b]; val[lo + b + 1] = val[lo + b]; from_t[lo + b + 1] = from_t[lo + b];
        --b;
      }
      col[lo + b + 1] = c; val[lo + b + 1] = v; from_t[lo + b + 1] = t;
    }
    int64_t a = 0;
    while (a < len) {
      int32_t const c = col[lo + a];
      double x = 0.0, y = 0.0; /* J_ic, J_ci */
      while (a < len && col[lo + a] == c) {
        if (from_t[lo + a]) y = val[lo + a]; else x = val[lo + a];
        ++a;
      }
      double const v = x + y;
      if (v != 0.0) { col[out] = c; val[out] = v; ++out; }
    }
    (*a_ptr)[i + 1] = out;
  }
  *a_col = col; *a_val = val;
  free(cnt); free(start); free(fill); free(from_t);
}

/* Strict-descent sweeps in colour order until nothing flips (or max_sweeps). */
static uint32_t descend(uint64_t n, int64_t const *a_ptr, int32_t const *a_col,
                        double const *a_val, double const *field, int64_t const *order,
                        int8_t *s, uint32_t max_sweeps) {
  uint32_t sweeps = 0;
  while (sweeps < max_sweeps) {
    int flipped = 0;
    for (uint64_t q = 0; q < n; ++q) {
      int64_t const i = order[q];
      double acc = 0.0;
      for (int64_t k = a_ptr[i]; k < a_ptr[i + 1]; ++k) {
        double const a = a_val[k];
        acc = acc + (s[a_col[k]] > 0 ? a : -a);
      }
      double const g = acc + field[i];
      double const de = s[i] > 0 ? -2.0 * g : 2.0 * g;
      if (de < 0.0) { s[i] = (int8_t)-s[i]; flipped = 1; }
    }
    ++sweeps;
    if (!flipped) break;
  }
  return sweeps;
}

int oracle_greedy_solve(uint64_t n, int64_t const *indptr, int32_t const *indices,
                        double const *data, double const *field, uint32_t max_sweeps,
                        int relax, uint64_t *out_x, double *out_e) {
  int64_t *a_ptr; int32_t *a_col; double *a_val;
  build_a(n, indptr, indices, data, &a_ptr, &a_col, &a_val);
  int64_t const m2 = a_ptr[n];
  bond *bonds = malloc((size_t)(m2 / 2 + 1) * sizeof *bonds);
  int64_t nb = 0;
  for (uint64_t i = 0; i < n; ++i)
    for (int64_t k = a_ptr[i]; k < a_ptr[i + 1]; ++k)
      if ((uint64_t)a_col[k] > i) { bonds[nb].i = (int32_t)i; bonds[nb].j = a_col[k]; bonds[nb].w = a_val[k]; ++nb; }
  qsort(bonds, (size_t)nb, sizeof *bonds, bond_cmp);

  /* clusters as linked member lists; cluster[v] = -1 while v is unassigned */
  int32_t *cluster = malloc((n + 1) * sizeof *cluster);
  int32_t *next = malloc((n + 1) * sizeof *next);   /* member list links */
  int32_t *head = malloc((n + 1) * sizeof *head), *tail = malloc((n + 1) * sizeof *tail);
  int32_t *count = calloc(n + 1, sizeof *count);
  int8_t *s = malloc(n + 1);
  for (uint64_t v = 0; v < n; ++v) { cluster[v] = -1; next[v] = -1; s[v] = 1; }
  for (int64_t e = 0; e < nb; ++e) {
    int32_t const i = bonds[e].i, j = bonds[e].j;
    double const w = bonds[e].w;
    if (cluster[i] < 0 && cluster[j] < 0) {
      cluster[i] = cluster[j] = i;
      head[i] = i; next[i] = j; tail[i] = j; next[j] = -1; count[i] = 2;
      s[i] = 1; s[j] = (int8_t)(w > 0.0 ? -1 : 1);
    } else if ((cluster[i] < 0) != (cluster[j] < 0)) {
      int32_t const fresh = cluster[i] < 0 ? i : j;
      int32_t const c = cluster[i] < 0 ? cluster[j] : cluster[i];
      double energy = 0.0;
      for (int64_t k = a_ptr[fresh]; k < a_ptr[fresh + 1]; ++k)
        if (cluster[a_col[k]] == c) energy = energy + (s[a_col[k]] > 0 ? a_val[k] : -a_val[k]);
      s[fresh] = (int8_t)(energy > 0.0 ? -1 : 1);
      cluster[fresh] = c; next[tail[c]] = fresh; tail[c] = fresh; next[fresh] = -1; count[c]++;
    } else if (cluster[i] != cluster[j]) {
      int32_t keep = cluster[i], gone = cluster[j];
      if (count[gone] > count[keep]) { int32_t t = keep; keep = gone; gone = t; }
      int const frustrated = (double)s[i] * (double)s[j] * w > 0.0;
      for (int32_t v = head[gone]; v >= 0; v = next[v]) {
        cluster[v] = keep;
        if (frustrated) s[v] = (int8_t)-s[v];
      }
      next[tail[keep]] = head[gone]; tail[keep] = tail[gone]; count[keep] += count[gone];
    }
  }
  /* orientation: per cluster, the sign of sum_i h_i s_i decides (isolated spins: own cluster) */
  double *fe = calloc(n + 1, sizeof *fe);
  for (uint64_t v = 0; v < n; ++v) {
    int32_t const c = cluster[v] < 0 ? (int32_t)v : cluster[v];
    fe[c] = fe[c] + (s[v] > 0 ? field[v] : -field[v]);
  }
  for (uint64_t v = 0; v < n; ++v) {
    int32_t const c = cluster[v] < 0 ? (int32_t)v : cluster[v];
    if (fe[c] > 0.0) s[v] = (int8_t)-s[v];
  }
  if (relax) {
    int64_t *order = malloc((n + 1) * sizeof *order);
    oracle_sa_layout(n, indptr, indices, data, NULL, order, NULL, NULL, NULL);
    descend(n, a_ptr, a_col, a_val, field, order, s, max_sweeps);
    free(order);
  }
  uint64_t const words = (n + 63) / 64;
  for (uint64_t w = 0; w < words; ++w) out_x[w] = 0;
  for (uint64_t v = 0; v < n; ++v) if (s[v] > 0) out_x[v / 64] |= (uint64_t)1 << (v % 64);
  oracle_sa_energy(n, indptr, indices, data, field, 1, out_x, out_e);
  free(a_ptr); free(a_col); free(a_val); free(bonds); free(cluster); free(next); free(head);
  free(tail); free(count); free(s); free(fe);
  return 0;
}
