// Analysis only: Metropolis SA on a CSR Ising model with (a) typewriter order, (b) random-site
// selection, (c) random permutation per sweep; geometric ladder.  E = sum_ij J_ij s_i s_j (full
// double sum, symmetric J): dE_i = -4 s_i sum_{j != i} J_ij s_j.
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
static inline uint64_t splitmix(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static inline double uni(uint64_t *s) { return ((splitmix(s) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
// mode 3: colour classes (colour_start[c]..colour_start[c+1] in `colour_order`, a list of spins
// grouped by colour) visited in a fresh random order of the COLOURS every sweep
void order_probe(int64_t n, const int64_t *indptr, const int32_t *indices, const double *data, int mode,
                 const double *betas, int sweeps, int reps, uint64_t seed, int8_t *out_spins, double *out_e,
                 int num_colours, const int64_t *colour_start, const int32_t *colour_order, int shared_order) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int r = 0; r < reps; ++r) {
    uint64_t st = seed * 1000003ull + (uint64_t)r * 7919ull + 12345ull;
    int8_t *s = out_spins + (int64_t)r * n;
    int8_t *best = malloc(n);
    double *f = malloc(sizeof(double) * n);  // local field sum_{j != i} J_ij s_j
    int32_t *perm = malloc(sizeof(int32_t) * n);
    for (int64_t i = 0; i < n; ++i) { s[i] = (splitmix(&st) & 1) ? 1 : -1; perm[i] = (int32_t)i; }
    for (int64_t i = 0; i < n; ++i) { double a = 0; for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) if (indices[k] != i) a += data[k] * s[indices[k]]; f[i] = a; }
    double e = 0; for (int64_t i = 0; i < n; ++i) { e += 2.0 * 0.5 * s[i] * f[i]; for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) if (indices[k] == i) e += data[k]; }
    double ebest = e; for (int64_t i = 0; i < n; ++i) best[i] = s[i];
    for (int t = 0; t < sweeps; ++t) {
      const double beta = betas[t];
      if (mode == 2) for (int64_t i = n - 1; i > 0; --i) { int64_t j = (int64_t)(splitmix(&st) % (uint64_t)(i + 1)); int32_t tmp = perm[i]; perm[i] = perm[j]; perm[j] = tmp; }
      if (mode == 3) {
        int cperm[64]; for (int c = 0; c < num_colours; ++c) cperm[c] = c;
        uint64_t shared = seed * 77ull + (uint64_t)t * 1315423911ull;  // same for every chain when shared_order
        uint64_t *src = shared_order ? &shared : &st;
        for (int c = num_colours - 1; c > 0; --c) { int j = (int)(splitmix(src) % (uint64_t)(c + 1)); int tmp = cperm[c]; cperm[c] = cperm[j]; cperm[j] = tmp; }
        int64_t at = 0;
        for (int c = 0; c < num_colours; ++c) for (int64_t k = colour_start[cperm[c]]; k < colour_start[cperm[c] + 1]; ++k) perm[at++] = colour_order[k];
      }
      for (int64_t v = 0; v < n; ++v) {
        const int64_t i = mode == 0 ? v : (mode == 1 ? (int64_t)(splitmix(&st) % (uint64_t)n) : perm[v]);  // modes 2, 3: perm
        const double de = -4.0 * s[i] * f[i];
        if (de <= 0.0 || uni(&st) < exp(-beta * de)) {
          s[i] = -s[i];
          e += de;
          for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) if (indices[k] != i) f[indices[k]] += 2.0 * data[k] * s[i];
        }
      }
      if (e < ebest) { ebest = e; for (int64_t i = 0; i < n; ++i) best[i] = s[i]; }
    }
    for (int64_t i = 0; i < n; ++i) s[i] = best[i];
    out_e[r] = ebest;
    free(best); free(f); free(perm);
  }
}
