/* ORACLE — test infrastructure only.  Nothing under annealing_sign_problem_amd/
 * may include, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * CPU restatement of the annealing sweep (specification "ASP-SA-1").
 *
 * PARITY UNPINNED against the reference's annealer: the reference calls
 * ising_glass_annealer 0.4.1.2 (conda-annealing.yml:8;
 * annealing_sign_problem/common.py:8,242-248), a third-party Haskell/C package
 * that is not in /root/reference and not installable offline.  Its RNG stream,
 * beta schedule and sweep order are unknown, so this file restates the
 * published algorithm (Metropolis single-spin-flip annealing of
 * E(s) = sum_ij J_ij s_i s_j + sum_i h_i s_i, energy convention pinned by
 * common.py:757-760 and experiments/full_hilbert_space.py:142-145; bit
 * convention pinned by cbits/build_matrix.c:72-74) with every free choice
 * fixed by DESIGN.md §4.  What IS pinned: the Philox4x32-10 generator against
 * the Random123 known-answer vectors (tests/test_oracle_sa.py), the energy
 * against numpy s^T J s + h^T s, and the exponential against libm.
 *
 * Everything here is deliberately written as plain sequential loops over the
 * ORIGINAL spin order; the HIP product (csrc/sa_sweep.hip) reaches the same
 * bits through a permuted sliced-ELL layout with 64 spins per wavefront.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11; Random123)                           */
/* ------------------------------------------------------------------------ */

void oracle_philox4x32_10(uint32_t const ctr[4], uint32_t const key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; ++round) {
    uint64_t const p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t const p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t const n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t const n1 = (uint32_t)p1;
    uint32_t const n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t const n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Random word of (spin i, sweep t, global replica r): DESIGN.md §4.3.
 * Four consecutive replicas share one Philox call. */
static uint32_t sa_random_word(uint64_t seed, uint32_t i, uint32_t t, uint32_t r) {
  uint32_t const ctr[4] = {i, t, r >> 2, 0u};
  uint32_t const key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t out[4];
  oracle_philox4x32_10(ctr, key, out);
  return out[r & 3u];
}

/* ------------------------------------------------------------------------ */
/* exp(-x) for the Metropolis test: a fixed sequence of IEEE operations      */
/*  so that CPU and GPU agree bit for bit.                   */
/* ------------------------------------------------------------------------ */

double oracle_expneg(double x) {
  if (!(x < 23.0)) {
    return 0.0; /* below the smallest uniform 2^-33; also catches NaN */
  }
  double const y = -x;
  double const kf = rint(y * 0x1.71547652b82fep+0);
  double r = fma(kf, -0x1.62e42fee00000p-1, y);
  r = fma(kf, -0x1.a39ef35793c76p-33, r);
  double p = 0x1.6124613a86d09p-33; /* 1/13! */
  p = fma(p, r, 0x1.1eed8eff8d898p-29);
  p = fma(p, r, 0x1.ae64567f544e4p-26);
  p = fma(p, r, 0x1.27e4fb7789f5cp-22);
  p = fma(p, r, 0x1.71de3a556c734p-19);
  p = fma(p, r, 0x1.a01a01a01a01ap-16);
  p = fma(p, r, 0x1.a01a01a01a01ap-13);
  p = fma(p, r, 0x1.6c16c16c16c17p-10);
  p = fma(p, r, 0x1.1111111111111p-7);
  p = fma(p, r, 0x1.5555555555555p-5);
  p = fma(p, r, 0x1.5555555555555p-3);
  p = fma(p, r, 0x1.0000000000000p-1);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  int64_t const k = (int64_t)kf; /* in [-34, 0] */
  uint64_t const bits = (uint64_t)(1023 + k) << 52;
  double scale;
  memcpy(&scale, &bits, sizeof scale);
  return p * scale;
}

/* ------------------------------------------------------------------------ */
/* Problem preparation                                                       */
/* ------------------------------------------------------------------------ */

typedef struct sa_problem {
  uint64_t n;
  int64_t *a_ptr;   /* CSR of A = offdiag(J + J^T), zeros dropped */
  int32_t *a_col;
  double *a_val;
  double diag_sum;  /* sum_i J_ii */
  int32_t *color;   /* DSATUR colour of every spin */
  int32_t num_colors;
  int64_t *order;   /* permutation: position -> spin; key (colour, -degree, index) */
  int64_t *color_start; /* num_colors + 1 offsets into `order` */
} sa_problem;

static void sa_problem_free(sa_problem *p) {
  free(p->a_ptr); free(p->a_col); free(p->a_val); free(p->color);
  free(p->order); free(p->color_start);
  memset(p, 0, sizeof *p);
}

/* J given as canonical CSR (sorted unique columns). */
static int sa_problem_init(sa_problem *p, uint64_t n, int64_t const *indptr,
                           int32_t const *indices, double const *data) {
  memset(p, 0, sizeof *p);
  p->n = n;
  int64_t const nnz = indptr[n];
  /* transpose */
  int64_t *t_ptr = calloc(n + 2, sizeof *t_ptr);
  int32_t *t_col = malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof *t_col);
  double *t_val = malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof *t_val);
  for (int64_t k = 0; k < nnz; ++k) t_ptr[indices[k] + 2]++;
  for (uint64_t i = 0; i < n; ++i) t_ptr[i + 2] += t_ptr[i + 1];
  for (uint64_t i = 0; i < n; ++i) {
    for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
      int64_t const dst = t_ptr[indices[k] + 1]++;
      t_col[dst] = (int32_t)i;
      t_val[dst] = data[k];
    }
  }
  /* now row j of the transpose is t_ptr[j] .. t_ptr[j+1] */
  p->a_ptr = calloc(n + 1, sizeof *p->a_ptr);
  p->a_col = malloc((size_t)(2 * nnz + 1) * sizeof *p->a_col);
  p->a_val = malloc((size_t)(2 * nnz + 1) * sizeof *p->a_val);
  int64_t out = 0;
  double diag = 0.0;
  for (uint64_t i = 0; i < n; ++i) {
    int64_t a = indptr[i], a_end = indptr[i + 1];
    int64_t b = t_ptr[i], b_end = t_ptr[i + 1];
    while (a < a_end || b < b_end) {
      int32_t ca = a < a_end ? indices[a] : INT32_MAX;
      int32_t cb = b < b_end ? t_col[b] : INT32_MAX;
      int32_t const c = ca < cb ? ca : cb;
      double const x = (ca == c) ? data[a] : 0.0;   /* J_ic or +0 */
      double const y = (cb == c) ? t_val[b] : 0.0;  /* J_ci or +0 */
      if (ca == c) ++a;
      if (cb == c) ++b;
      if ((uint64_t)c == i) {
        if (ca == c) diag = diag + x; /* J_ii, each row once, in row order */
        continue;
      }
      double const v = x + y;
      if (v != 0.0) {
        p->a_col[out] = c;
        p->a_val[out] = v;
        ++out;
      }
    }
    p->a_ptr[i + 1] = out;
  }
  p->diag_sum = diag;
  free(t_ptr); free(t_col); free(t_val);

  /* DSATUR colouring: next spin = most distinct neighbour colours, then
   * larger degree, then smaller index; colour = smallest one no neighbour has.  Own binary
   * heap of (saturation, degree, index) records with lazy invalidation. */
  p->color = malloc((size_t)(n ? n : 1) * sizeof *p->color);
  int32_t ncol = 0;
  {
    int64_t maxd = 0;
    for (uint64_t i = 0; i < n; ++i) {
      int64_t const d = p->a_ptr[i + 1] - p->a_ptr[i];
      if (d > maxd) maxd = d;
    }
    size_t const cap_colors = (size_t)maxd + 2;
    /* has[v * cap_colors + c] = 1 when a neighbour of v already has colour c */
    uint8_t *has = calloc((size_t)(n ? n : 1) * cap_colors, 1);
    int32_t *sat = calloc(n + 1, sizeof *sat);
    typedef struct { int32_t sat, deg, idx; } rec;
    size_t heap_cap = (size_t)(n + p->a_ptr[n] + 8), heap_n = 0;
    rec *heap = malloc(heap_cap * sizeof *heap);
#define REC_ABOVE(x, y) ((x).sat != (y).sat ? (x).sat > (y).sat : ((x).deg != (y).deg ? (x).deg > (y).deg : (x).idx < (y).idx))
#define HEAP_PUSH(r) do { size_t c_ = heap_n++; heap[c_] = (r); \
      while (c_ > 0) { size_t u_ = (c_ - 1) / 2; if (!REC_ABOVE(heap[c_], heap[u_])) break; \
        rec t_ = heap[c_]; heap[c_] = heap[u_]; heap[u_] = t_; c_ = u_; } } while (0)
    for (uint64_t i = 0; i < n; ++i) {
      p->color[i] = -1;
      rec r = {0, (int32_t)(p->a_ptr[i + 1] - p->a_ptr[i]), (int32_t)i};
      HEAP_PUSH(r);
    }
    while (heap_n > 0) {
      rec const top = heap[0];
      heap[0] = heap[--heap_n];
      for (size_t c_ = 0;;) { /* sift down */
        size_t l_ = 2 * c_ + 1, r_ = l_ + 1, m_ = c_;
        if (l_ < heap_n && REC_ABOVE(heap[l_], heap[m_])) m_ = l_;
        if (r_ < heap_n && REC_ABOVE(heap[r_], heap[m_])) m_ = r_;
        if (m_ == c_) break;
        rec t_ = heap[c_]; heap[c_] = heap[m_]; heap[m_] = t_; c_ = m_;
      }
      int32_t const v = top.idx;
      if (p->color[v] >= 0 || top.sat != sat[v]) continue;
      int32_t c = 0;
      while (has[(size_t)v * cap_colors + (size_t)c]) ++c;
      p->color[v] = c;
      if (c + 1 > ncol) ncol = c + 1;
      for (int64_t k = p->a_ptr[v]; k < p->a_ptr[v + 1]; ++k) {
        int32_t const u = p->a_col[k];
        if (p->color[u] >= 0 || has[(size_t)u * cap_colors + (size_t)c]) continue;
        has[(size_t)u * cap_colors + (size_t)c] = 1;
        sat[u] += 1;
        rec r = {sat[u], (int32_t)(p->a_ptr[u + 1] - p->a_ptr[u]), u};
        HEAP_PUSH(r);
      }
    }
#undef HEAP_PUSH
#undef REC_ABOVE
    free(heap); free(sat); free(has);
  }
  p->num_colors = ncol;

  /* permutation: colour ascending, degree descending, index ascending.
   * Counting sort on (colour, degree) keeps index order stable. */
  p->order = malloc((size_t)(n ? n : 1) * sizeof *p->order);
  p->color_start = calloc((size_t)ncol + 1, sizeof *p->color_start);
  int64_t max_deg = 0;
  for (uint64_t i = 0; i < n; ++i) {
    int64_t const d = p->a_ptr[i + 1] - p->a_ptr[i];
    if (d > max_deg) max_deg = d;
  }
  {
    /* stable sort by degree descending, then stable by colour ascending */
    int64_t *tmp = malloc((size_t)(n ? n : 1) * sizeof *tmp);
    int64_t *cnt = calloc((size_t)max_deg + 2, sizeof *cnt);
    for (uint64_t i = 0; i < n; ++i) cnt[max_deg - (p->a_ptr[i + 1] - p->a_ptr[i]) + 1]++;
    for (int64_t d = 0; d <= max_deg; ++d) cnt[d + 1] += cnt[d];
    for (uint64_t i = 0; i < n; ++i) tmp[cnt[max_deg - (p->a_ptr[i + 1] - p->a_ptr[i])]++] = (int64_t)i;
    free(cnt);
    int64_t *ccnt = calloc((size_t)ncol + 2, sizeof *ccnt);
    for (uint64_t i = 0; i < n; ++i) ccnt[p->color[i] + 1]++;
    for (int32_t c = 0; c < ncol; ++c) ccnt[c + 1] += ccnt[c];
    for (int32_t c = 0; c <= ncol; ++c) p->color_start[c] = ccnt[c];
    for (uint64_t q = 0; q < n; ++q) {
      int64_t const i = tmp[q];
      p->order[ccnt[p->color[i]]++] = i;
    }
    free(ccnt); free(tmp);
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Energy: E = D + T, T = radix-64 pairwise tree over the permuted blocks    */
/*                                                           */
/* ------------------------------------------------------------------------ */

static double tree64(double v[64]) {
  for (int step = 1; step < 64; step <<= 1) {
    for (int l = 0; l < 64; l += 2 * step) v[l] = v[l] + v[l + step];
  }
  return v[0];
}

static double sa_energy(sa_problem const *p, double const *field, int8_t const *s) {
  /* level 0: one partial per 64-lane block; colour classes are padded */
  int64_t nblocks = 0;
  for (int32_t c = 0; c < p->num_colors; ++c) {
    nblocks += (p->color_start[c + 1] - p->color_start[c] + 63) / 64;
  }
  int64_t cap = nblocks > 0 ? nblocks : 1;
  double *level = malloc((size_t)((cap + 63) / 64 * 64) * sizeof *level);
  int64_t b = 0;
  for (int32_t c = 0; c < p->num_colors; ++c) {
    for (int64_t q0 = p->color_start[c]; q0 < p->color_start[c + 1]; q0 += 64) {
      double v[64];
      for (int l = 0; l < 64; ++l) {
        int64_t const q = q0 + l;
        if (q >= p->color_start[c + 1]) { v[l] = 0.0; continue; }
        int64_t const i = p->order[q];
        double acc = 0.0;
        for (int64_t k = p->a_ptr[i]; k < p->a_ptr[i + 1]; ++k) {
          double const a = p->a_val[k];
          acc = acc + (s[p->a_col[k]] > 0 ? a : -a);
        }
        double const g = 0.5 * acc + field[i];
        v[l] = s[i] > 0 ? g : -g;
      }
      level[b++] = tree64(v);
    }
  }
  int64_t n = nblocks;
  if (n == 0) { free(level); return p->diag_sum + 0.0; }
  while (n > 1) {
    int64_t const groups = (n + 63) / 64;
    for (int64_t g = 0; g < groups; ++g) {
      double v[64];
      for (int l = 0; l < 64; ++l) v[l] = (g * 64 + l < n) ? level[g * 64 + l] : 0.0;
      level[g] = tree64(v);
    }
    n = groups;
  }
  double const total = p->diag_sum + level[0];
  free(level);
  return total;
}

/* ------------------------------------------------------------------------ */
/* The annealing chains                                                      */
/* ------------------------------------------------------------------------ */

/* One proposal on spin i; returns the rounded fixed-point dE when accepted,
 * sets *accepted. */
static int64_t sa_propose(sa_problem const *p, double const *field, uint64_t seed, double beta,
                          uint32_t t, uint32_t replica, double scale, int8_t *s, int64_t i,
                          int *accepted) {
  double acc = 0.0;
  for (int64_t k = p->a_ptr[i]; k < p->a_ptr[i + 1]; ++k) {
    double const a = p->a_val[k];
    acc = acc + (s[p->a_col[k]] > 0 ? a : -a);
  }
  double const g = acc + field[i];
  double const de = s[i] > 0 ? -2.0 * g : 2.0 * g;
  int accept = de <= 0.0;
  if (!accept) {
    uint32_t const w = sa_random_word(seed, (uint32_t)i, t, replica);
    double const u = ((double)w + 0.5) * 0x1p-32;
    accept = u < oracle_expneg(beta * de);
  }
  *accepted = accept;
  if (!accept) return 0;
  s[i] = (int8_t)-s[i];
  return (int64_t)rint(de * scale);
}

/* Visiting order of sweep t in the SHUFFLED variant: ascending (priority, index)
 * with priority = word 0 of Philox(counter (i, t, 0xFFFFFFFE, 0), key seed) — the same for every
 * chain.  keys[] receives (priority << 32 | index), sorted. */
static int cmp_u64(void const *a, void const *b) {
  uint64_t const x = *(uint64_t const *)a, y = *(uint64_t const *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}
static void sa_shuffled_order(uint64_t n, uint64_t seed, uint32_t t, uint64_t *keys) {
  uint32_t const key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (uint64_t i = 0; i < n; ++i) {
    uint32_t const ctr[4] = {(uint32_t)i, t, 0xFFFFFFFEu, 0u};
    uint32_t out[4];
    oracle_philox4x32_10(ctr, key, out);
    keys[i] = ((uint64_t)out[0] << 32) | i;
  }
  qsort(keys, n, sizeof(uint64_t), cmp_u64);
}

static void sa_run_chain(sa_problem const *p, double const *field, uint64_t seed,
                         double const *betas, uint32_t num_sweeps, uint32_t replica,
                         uint64_t const *x0, double scale, int8_t *s, int8_t *best,
                         int64_t *tracked_best, uint64_t *accepted_total, int64_t *trace,
                         int shuffled) {
  uint64_t const n = p->n;
  uint64_t *keys = shuffled ? malloc(sizeof(uint64_t) * (n ? n : 1)) : NULL;
  for (uint64_t i = 0; i < n; ++i) {
    if (x0 != NULL) {
      s[i] = ((x0[i / 64] >> (i % 64)) & 1u) ? 1 : -1;
    } else {
      s[i] = (sa_random_word(seed, (uint32_t)i, 0xFFFFFFFFu, replica) & 1u) ? 1 : -1;
    }
  }
  memcpy(best, s, n);
  int64_t e_cur = 0, e_best = 0;
  uint64_t accepted = 0;
  if (trace != NULL) trace[0] = 0;
  for (uint32_t t = 0; t < num_sweeps; ++t) {
    double const beta = betas[t];
    int64_t q_sweep = 0;
    if (shuffled) {
      sa_shuffled_order(n, seed, t, keys);
      for (uint64_t q = 0; q < n; ++q) {
        int accept;
        q_sweep += sa_propose(p, field, seed, beta, t, replica, scale, s, (int64_t)(uint32_t)keys[q],
                              &accept);
        accepted += (uint64_t)accept;
      }
    } else {
      for (int32_t c = 0; c < p->num_colors; ++c) {
        for (int64_t q = p->color_start[c]; q < p->color_start[c + 1]; ++q) {
          int accept;
          q_sweep += sa_propose(p, field, seed, beta, t, replica, scale, s, p->order[q], &accept);
          accepted += (uint64_t)accept;
        }
      }
    }
    e_cur += q_sweep;
    if (trace != NULL) trace[t + 1] = e_cur;
    if (e_cur < e_best) {
      e_best = e_cur;
      memcpy(best, s, n);
    }
  }
  *tracked_best = e_best;
  *accepted_total = accepted;
  free(keys);
}

/* Returns 0 on success.  out_x: R * ceil(K/64) words; out_e: R doubles;
 * out_tracked (optional): R fixed-point best energies relative to the start;
 * out_accepted (optional): R counts of accepted flips. */
static int anneal_impl(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                       double const *data, double const *field, uint64_t seed,
                       double const *betas, uint32_t num_sweeps, uint32_t repetitions,
                       uint32_t replica_offset, uint64_t const *x0, int32_t energy_scale_exp,
                       uint64_t *out_x, double *out_e, int64_t *out_tracked,
                       uint64_t *out_accepted, int64_t *out_trace, int num_threads, int shuffled) {
  sa_problem p;
  if (sa_problem_init(&p, num_spins, indptr, indices, data) != 0) return -1;
  uint64_t const words = (num_spins + 63) / 64;
  double const scale = ldexp(1.0, energy_scale_exp);
  if (num_threads < 1) num_threads = 1;
#pragma omp parallel num_threads(num_threads)
  {
    int8_t *s = malloc(num_spins ? num_spins : 1);
    int8_t *best = malloc(num_spins ? num_spins : 1);
#pragma omp for schedule(dynamic, 1)
    for (uint32_t rr = 0; rr < repetitions; ++rr) {
      int64_t tracked = 0;
      uint64_t accepted = 0;
      sa_run_chain(&p, field, seed, betas, num_sweeps, replica_offset + rr, x0, scale, s,
                   best, &tracked, &accepted,
                   out_trace ? out_trace + (uint64_t)rr * ((uint64_t)num_sweeps + 1) : NULL, shuffled);
      uint64_t *x = out_x + (uint64_t)rr * words;
      for (uint64_t w = 0; w < words; ++w) x[w] = 0;
      for (uint64_t i = 0; i < num_spins; ++i) {
        if (best[i] > 0) x[i / 64] |= (uint64_t)1 << (i % 64);
      }
      out_e[rr] = sa_energy(&p, field, best);
      if (out_tracked) out_tracked[rr] = tracked;
      if (out_accepted) out_accepted[rr] = accepted;
    }
    free(s);
    free(best);
  }
  sa_problem_free(&p);
  return 0;
}

int oracle_sa_anneal(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                     double const *data, double const *field, uint64_t seed,
                     double const *betas, uint32_t num_sweeps, uint32_t repetitions,
                     uint32_t replica_offset, uint64_t const *x0, int32_t energy_scale_exp,
                     uint64_t *out_x, double *out_e, int64_t *out_tracked,
                     uint64_t *out_accepted, int num_threads) {
  return anneal_impl(num_spins, indptr, indices, data, field, seed, betas, num_sweeps,
                     repetitions, replica_offset, x0, energy_scale_exp, out_x, out_e, out_tracked,
                     out_accepted, NULL, num_threads, 0);
}

/* The SHUFFLED variant: identical in everything but the visiting order. */
int oracle_sa_anneal_shuffled(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                              double const *data, double const *field, uint64_t seed,
                              double const *betas, uint32_t num_sweeps, uint32_t repetitions,
                              uint32_t replica_offset, uint64_t const *x0, int32_t energy_scale_exp,
                              uint64_t *out_x, double *out_e, int64_t *out_tracked,
                              uint64_t *out_accepted, int num_threads) {
  return anneal_impl(num_spins, indptr, indices, data, field, seed, betas, num_sweeps,
                     repetitions, replica_offset, x0, energy_scale_exp, out_x, out_e, out_tracked,
                     out_accepted, NULL, num_threads, 1);
}

/* Same, and out_trace[r * (T + 1) + t] = tracked energy (fixed point, relative to the initial
 * configuration) of chain r after t sweeps. */
int oracle_sa_anneal_trace(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                           double const *data, double const *field, uint64_t seed,
                           double const *betas, uint32_t num_sweeps, uint32_t repetitions,
                           uint32_t replica_offset, uint64_t const *x0, int32_t energy_scale_exp,
                           uint64_t *out_x, double *out_e, int64_t *out_trace, int num_threads) {
  return anneal_impl(num_spins, indptr, indices, data, field, seed, betas, num_sweeps,
                     repetitions, replica_offset, x0, energy_scale_exp, out_x, out_e, NULL, NULL,
                     out_trace, num_threads, 0);
}

/* E(x) for `count` packed configurations. */
int oracle_sa_energy(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                     double const *data, double const *field, uint32_t count,
                     uint64_t const *x, double *out_e) {
  sa_problem p;
  if (sa_problem_init(&p, num_spins, indptr, indices, data) != 0) return -1;
  uint64_t const words = (num_spins + 63) / 64;
  int8_t *s = malloc(num_spins ? num_spins : 1);
  for (uint32_t r = 0; r < count; ++r) {
    for (uint64_t i = 0; i < num_spins; ++i) {
      s[i] = ((x[r * words + i / 64] >> (i % 64)) & 1u) ? 1 : -1;
    }
    out_e[r] = sa_energy(&p, field, s);
  }
  free(s);
  sa_problem_free(&p);
  return 0;
}

/* Colouring and permutation, exported so the tests can compare the product's
 * plan (host logic) with this restatement.  colors: K ints; order: K positions. */
int oracle_sa_layout(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                     double const *data, int32_t *colors, int64_t *order,
                     int32_t *num_colors, int64_t *nnz_offdiag, double *diag_sum) {
  sa_problem p;
  if (sa_problem_init(&p, num_spins, indptr, indices, data) != 0) return -1;
  if (colors) memcpy(colors, p.color, num_spins * sizeof *colors);
  if (order) memcpy(order, p.order, num_spins * sizeof *order);
  if (num_colors) *num_colors = p.num_colors;
  if (nnz_offdiag) *nnz_offdiag = p.a_ptr[num_spins];
  if (diag_sum) *diag_sum = p.diag_sum;
  sa_problem_free(&p);
  return 0;
}
