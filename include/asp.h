/* asp.h — C ABI of libasp_hip.so, the MI355X (gfx950) implementation of the
 * sign-optimisation hot path of twesterhout/annealing-sign-problem.
 *
 * Plain C: pointers and sizes only, no torch / C++ types.  Every entry point
 * names the reference interface it replaces (paths relative to the reference
 * checkout).  Unless a function says "device", pointers are HOST pointers and
 * the callee stages them through HBM itself.
 *
 * Error model: the two drop-in symbols keep the reference's signatures (no
 * error return, cbits/build_matrix.h:7-14); every failure — no GPU, HIP error,
 * violated precondition that was detected — is recorded and can be read with
 * asp_last_error().  All other functions return 0 on success and a negative
 * asp_status otherwise.  Nothing in this library falls back to a CPU path.
 */
#ifndef ASP_H
#define ASP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* Status / errors                                                           */
/* ------------------------------------------------------------------------- */

typedef enum asp_status {
  ASP_OK = 0,
  ASP_ERR_NO_DEVICE = -1,  /* no HIP device visible                          */
  ASP_ERR_HIP = -2,        /* a HIP runtime call failed                      */
  ASP_ERR_INVALID = -3,    /* invalid argument / violated precondition       */
  ASP_ERR_TOO_LARGE = -4,  /* problem does not fit this implementation       */
  ASP_ERR_ALLOC = -5       /* host or device allocation failed               */
} asp_status;

/* Last error recorded on the calling thread ("" if none). */
const char *asp_last_error(void);
int asp_last_error_code(void);
void asp_clear_error(void);

/* Number of HIP devices (>= 0) or a negative asp_status. */
int asp_device_count(void);
/* 1 once this process has made a HIP call THROUGH THIS LIBRARY (any entry point that needs the
 * device), 0 before.  It knows nothing of HIP state created elsewhere in the process (a
 * profiler's preloaded tool library, the host's own HIP or torch.cuda calls): 0 means "not by
 * this library", and only a process whose GPU is untouched by anybody may fork workers that each
 * open the device (sampled_components --workers checks both). */
int asp_device_touched(void);
/* Select the device used by this library in the whole process (every entry point binds
 * its calling thread to it). */
int asp_set_device(int device);
/* The device this library computes on: the asp_set_device choice, else HIP's current device
 * of the calling thread; negative asp_status on failure. */
int asp_get_device(void);
/* Waits for the device(s) and releases what the library keeps alive BETWEEN calls: pooled
 * device memory and pooled streams.  Meant to be called once, after every handle (plans,
 * operators, builds) has been destroyed and before the process starts to exit, so that no HIP
 * call is left to static destructors or interpreter teardown, where the runtime or a profiler
 * attached to it may already be gone (the Python binding registers it with atexit and
 * destroys its live handles first).  The library stays usable afterwards, without pooling. */
int asp_shutdown(void);
/* Library version, "major.minor.patch". */
const char *asp_version(void);

/* ------------------------------------------------------------------------- */
/* (1) Drop-in replacements for cbits/build_matrix.h:3-14                    */
/*     (cffi cdef duplicate: annealing_sign_problem/build_extension.py:5-20) */
/* ------------------------------------------------------------------------- */

typedef struct ls_bits512 {
  uint64_t words[8];
} ls_bits512;

/* Replaces build_matrix (cbits/build_matrix.c:22-65).
 *
 * For every row r and each of its other_counts[r] connections e (flat,
 * row-major): look other_spins[e] up in the table spins[0..num_spins), which
 * is sorted ascending under the lexicographic order of words[0..7]
 * (cbits/build_matrix.c:7-20) and unique.
 *   hit  -> append (r, position, counts[r]*other_coeffs[e]*|psi[r]|*|other_psi[e]|)
 *           to (row_indices, col_indices, elements), preserving input order;
 *   miss -> field[r] += counts[r]*other_coeffs[e]*|psi[r]|*other_psi[e]
 *           (signed other_psi, left-to-right accumulation in row order).
 * Products are evaluated left to right without FMA contraction, so results are
 * bit-identical to the reference compiled without -ffast-math.
 * Caller allocates all outputs (capacity sum(other_counts) for the COO triple,
 * num_spins for field).  Returns the number of COO entries written; on failure
 * returns 0, leaves outputs untouched and records an error.
 */
uint64_t build_matrix(uint64_t num_spins, ls_bits512 const spins[],
                      int64_t const *counts, double const *psi,
                      ls_bits512 const *other_spins, double const *other_coeffs,
                      int64_t const *other_counts, double const *other_psi,
                      uint32_t *row_indices, uint32_t *col_indices,
                      double *elements, double *field);

/* Replaces extract_signs (cbits/build_matrix.c:67-76): bit i of signs[i/64]
 * is set iff psi[i] > 0 (zero and NaN clear it); ceil(num_spins/64) words are
 * fully overwritten. */
void extract_signs(uint64_t num_spins, double const *psi, uint64_t *signs);

/* ------------------------------------------------------------------------- */
/* (2) Device-resident form of the same coupling build (what bench.py times) */
/* ------------------------------------------------------------------------- */

typedef struct asp_build asp_build;

/* Allocate HBM for a build with num_spins rows and num_other = sum(other_counts)
 * connections.  NULL on failure. */
asp_build *asp_build_create(uint64_t num_spins, uint64_t num_other);
/* Host -> HBM copy of the seven input arrays of build_matrix. */
int asp_build_upload(asp_build *b, ls_bits512 const *spins, int64_t const *counts,
                     double const *psi, ls_bits512 const *other_spins,
                     double const *other_coeffs, int64_t const *other_counts,
                     double const *other_psi);
/* Run the kernels on resident inputs; *nnz receives the COO length.  The
 * device time of the launch sequence is measured with HIP events on the
 * library's stream and returned by asp_build_last_ms(). */
int asp_build_run(asp_build *b, uint64_t *nnz);
float asp_build_last_ms(asp_build const *b);
/* HBM -> host copy of the outputs of the last run (any pointer may be NULL). */
int asp_build_download(asp_build *b, uint32_t *row_indices, uint32_t *col_indices,
                       double *elements, double *field);
void asp_build_destroy(asp_build *b);

/* ------------------------------------------------------------------------- */
/* (3) Live coupling build: the two numba kernels of common.make_ising_model */
/* ------------------------------------------------------------------------- */

/* Replaces _clipped_search_sorted (annealing_sign_problem/common.py:116-128)
 * fused with the membership test (common.py:173) and
 * _make_ising_model_compute_elements (common.py:71-82), for 64-bit keys
 * (number_spins <= 64, asserted by the reference at common.py:86).
 *
 *   other_indices[e] = clip(searchsorted_left(keys, other_keys[e]), 0, K-1)
 *   member[e]        = other_keys[e] == keys[other_indices[e]]
 *   elements[e]      = (other_coeffs[e] * |member ? psi[idx] : 0|) * |psi[row(e)]|
 *   offsets          = [0, cumsum(other_counts)]
 * keys must be sorted ascending (np.unique output).  Any output may be NULL.
 */
int asp_ising_elements(uint64_t num_spins, uint64_t const *keys, double const *psi,
                       uint64_t num_other, uint64_t const *other_keys,
                       double const *other_coeffs, int64_t const *other_counts,
                       int64_t *other_indices, uint8_t *member, double *elements,
                       int64_t *offsets);

/* Device time (ms, HIP events) of the scan + kernel of this thread's last
 * asp_ising_elements call, without the host<->HBM copies. */
float asp_ising_elements_last_ms(void);

/* ------------------------------------------------------------------------- */
/* (3b) Hamiltonian action on bit-packed basis states, and the coupling      */
/*      build fused with it                                                  */
/* ------------------------------------------------------------------------- */

/* A sum of two-site terms on <= 64 spin-1/2 sites without lattice symmetries:
 * what the reference obtains from lattice_symmetries' ls.Operator built from
 * physical_systems/<model>.yaml (`terms: [{matrix, sites}]`, e.g.
 * heisenberg_kagome_16.yaml:5-12; call sites annealing_sign_problem/common.py:96,
 * 283,516-522).  Bit i of a key is site i; a 4x4 matrix acts on |b_a b_b> with
 * row/column index 2*b_a + b_b, `matrices[bond*16 + dst*4 + src]`.  Bonds are
 * listed in term order, then site-pair order.  Matrices are real (the reference
 * rejects |Im| > 1e-6, common.py:99-101). */
typedef struct asp_operator asp_operator;

int asp_operator_create(uint32_t number_spins, uint32_t num_bonds, uint8_t const *site_a,
                        uint8_t const *site_b, double const *matrices, asp_operator **out);
void asp_operator_destroy(asp_operator *op);

/* Symmetry-adapted basis in the trivial sector of a group of site permutations, with optional
 * global spin inversion of character spin_inversion = +1 / -1 (0: none) — the bases of
 * physical_systems/heisenberg_kagome_36.yaml:7-29, heisenberg_pyrochlore_2x2x2.yaml:1-17 and
 * heisenberg_kagome_18.yaml:4, which the reference gets from lattice_symmetries.
 * table[e * 64 + i] = destination site of site i under element e, for ALL elements of the group
 * (closed under composition; element 0 the identity).  Afterwards keys are orbit representatives
 * (the smallest state of an orbit), asp_operator_apply returns for every connection the
 * representative of the target and the coefficient c * chi(g) * norm(target) / norm(source),
 * norm(s)^2 = (sum of the stabiliser's characters) / |G|, and asp_operator_extend the sorted
 * unique representatives.  Equal targets within a row are not merged (asp_operator_ising
 * then runs its duplicate-keeping variant). */
int asp_operator_set_symmetry(asp_operator *op, uint32_t num_permutations, uint8_t const *table,
                              int32_t spin_inversion);
/* (representative, character of a group element mapping the key onto it, norm) of n keys; any
 * output may be NULL.  norm 0: the key's orbit is not part of the sector. */
int asp_operator_state_info(asp_operator const *op, uint64_t n, uint64_t const *keys,
                            uint64_t *representatives, double *characters, double *norms);

/* y = H x, matrix-free, in a fixed-magnetisation basis without lattice symmetries
 * (csrc/plain_basis.hip): sk_32_1.yaml's C(32,16) = 6.0e8 states x 496 bonds are too many matrix
 * elements to keep resident (asp_sector_rows) but have a closed-form index.  States ascending;
 * asp_plain_basis_states writes them (device pointer, `dimension` entries).  Bonds must conserve
 * the magnetisation (off-diagonal weight on 01 <-> 10 only); 2..36 spins. */
typedef struct asp_plain_basis asp_plain_basis;
int asp_plain_basis_create(asp_operator const *op, int32_t hamming_weight, asp_plain_basis **out);
void asp_plain_basis_destroy(asp_plain_basis *pb);
uint64_t asp_plain_basis_dimension(asp_plain_basis const *pb);
int asp_plain_basis_states(asp_plain_basis const *pb, uint64_t *states_dev);
int asp_plain_matvec(asp_plain_basis const *pb, double const *x_dev, double *y_dev);

/* Positions of states in a long ascending list kept on the device (csrc/key_table.hip): what
 * `basis.batched_index(spins)` (common.py:813-818) is for a basis of tens of millions of
 * representatives.  index[q] = position of queries[q], or -1.  Host pointers; thread-safe. */
typedef struct asp_table asp_table;
int asp_table_create(uint64_t n, uint64_t const *sorted_keys, asp_table **out);
void asp_table_destroy(asp_table *t);
int asp_table_index(asp_table const *t, uint64_t m, uint64_t const *queries, int64_t *index);

/* A whole symmetry sector on the device (csrc/sector_basis.hip) — what the reference reads from
 * SpinED's output (common.py:783-803: /basis/representatives, /hamiltonian/eigenvectors) and
 * what those absent files would hold for heisenberg_kagome_36.yaml: 31.5 million representatives.
 * Every *_dev pointer is DEVICE memory of the bound device; the calls run on a stream of their
 * own and return when the work is done (the caller synchronises ITS streams before calling).
 *
 * asp_sector_enumerate: all states of `number_spins` spins with `hamming_weight` bits set (all
 * states when negative) that are the smallest member of their orbit and lie in the sector
 * (norm > 0), ascending, with their norms (norms_dev may be NULL).  *count receives the number
 * found; nothing is written when it exceeds `capacity` (capacity 0: sizing call), which then
 * fails with ASP_ERR_TOO_LARGE.  Up to 48 spins; works without symmetries too (every state its
 * own representative, norm 1). */
int asp_sector_enumerate(asp_operator const *op, int32_t hamming_weight, uint64_t capacity,
                         uint64_t *reps_dev, double *norms_dev, uint64_t *count);
/* Entries per row of the sector's matrix (off-diagonal transitions a state can take). */
uint32_t asp_sector_width(asp_operator const *op);
/* The operator in the basis `reps_dev` (sorted representatives, their norms) as an ELL matrix:
 * slot k of row i at [k * n + i]; idx = row index of the target's representative, val =
 * c * chi * norm(target) / norm(source) (the coefficient asp_operator_apply returns); unused
 * slots hold (i, 0.0); diag = the diagonal, summed bond by bond.  Targets outside `reps_dev`
 * (another magnetisation, norm 0) are dropped.  width >= asp_sector_width(op). */
int asp_sector_rows(asp_operator const *op, uint64_t n, uint64_t const *reps_dev,
                    double const *norms_dev, uint32_t width, uint32_t *idx_dev, double *val_dev,
                    double *diag_dev);
/* y = H x over those arrays: y[i] = diag[i] x[i] + sum_k val[k n + i] x[idx[k n + i]], k ascending. */
int asp_sector_matvec(uint64_t n, uint32_t width, uint32_t const *idx_dev, double const *val_dev,
                      double const *diag_dev, double const *x_dev, double *y_dev);

/* 1 when every row's targets are pairwise distinct for every input state (distinct flip
 * masks), which asp_operator_ising requires; 0 otherwise. */
int asp_operator_unique_targets(asp_operator const *op);
/* Upper bound of other_counts[i] (diagonal entry included). */
uint32_t asp_operator_max_connections(asp_operator const *op);

/* Replaces `hamiltonian.batched_apply` + the flattening of _batched_apply
 * (common.py:85-106) for n keys: per key one diagonal entry (the sum of the
 * bonds' diagonal matrix elements, added in bond order), then one entry per
 * non-zero off-diagonal element m[dst][src] in (bond, dst) order:
 *   other_keys[e] = key ^ flip(src ^ dst), other_coeffs[e] = m[dst][src].
 * other_counts[i] receives the number of entries of key i; *total their sum.
 * Fails with ASP_ERR_INVALID when *total > capacity (nothing is written then
 * except other_counts and *total). */
int asp_operator_apply(asp_operator const *op, uint64_t n, uint64_t const *keys,
                       uint64_t capacity, uint64_t *other_keys, double *other_coeffs,
                       int64_t *other_counts, uint64_t *total);

/* make_ising_model's arithmetic (common.py:131-208) without materialising the
 * connections: for sorted unique keys[K] and amplitudes psi[K] (already
 * L2-normalised over the cluster) computes
 *   M_ij = (H_ij * |psi_j|) * |psi_i|   for i, j both in the cluster,
 *   J    = 0.5 * (M + M^T), zeros dropped, as COO sorted by (row, col)
 * — the matrix `0.5 * (matrix + matrix.T); sort_indices(); tocoo()` of
 * common.py:194-196, bit for bit.  *nnz receives the length; row/col/val need
 * capacity >= *nnz (call with capacity 0 and NULL outputs to size them).
 * Operators whose rows reach pairwise distinct states (asp_operator_unique_targets) take one
 * fused pass.  Otherwise — symmetry-adapted bases, single-site flips — several connections of a
 * row may end in the same state, and the variant that keeps the reference's arithmetic for
 * them runs: duplicates of a row summed in connection order from 0 (scipy's csr + csr on
 * non-canonical input), then 0.5 * (Mhat_rj + Mhat_jr), entries pruned iff that sum is 0.  That
 * variant needs every coupling to have its mirror (Mhat_rj present iff Mhat_jr present: any
 * operator with a symmetric pattern); otherwise it fails with ASP_ERR_INVALID and
 * make_ising_model takes the asp_ising_elements route. */
int asp_operator_ising(asp_operator const *op, uint64_t num_spins, uint64_t const *keys,
                       double const *psi, uint64_t capacity, int32_t *row, int32_t *col,
                       double *val, uint64_t *nnz);
/* The same matrix as canonical CSR: indptr[num_spins + 1] instead of the row of every entry —
 * what asp_sa_plan_create and asp_sparsify_component take, without a counting pass over the
 * rows on the host (the sampled-cluster pipeline's form). */
int asp_operator_ising_csr(asp_operator const *op, uint64_t num_spins, uint64_t const *keys,
                           double const *psi, uint64_t capacity, int64_t *indptr, int32_t *col,
                           double *val, uint64_t *nnz);

/* make_hamiltonian_extension's state set (common.py:516-522): the sorted unique
 * union of every key's targets (its own diagonal entry included).  *count receives
 * the size; out needs capacity >= *count (capacity 0 / NULL sizes it). */
int asp_operator_extend(asp_operator const *op, uint64_t n, uint64_t const *keys,
                        uint64_t capacity, uint64_t *out, uint64_t *count);

/* Device time (ms, HIP events, no host<->HBM copies) of this thread's last
 * asp_operator_apply / _ising / _extend call. */
float asp_operator_last_ms(void);

/* ------------------------------------------------------------------------- */
/* (3c) Global-cutoff sparsification + component extraction                  */
/* ------------------------------------------------------------------------- */

/* sparsify_using_global_cutoff (annealing_sign_problem/common.py:634-692) on a CSR matrix
 * (rows sorted by column, no duplicates):
 *   M'_ij = 0 where |M_ij| < reltol * max|M| unless spins i and j are both frozen   (:634-643)
 *   graph = non-zeros of 0.5 * (M' + M'^T)                                         (:660-662)
 *   keep[i] = 1 iff i is connected to `anchor` in that graph                        (:664-668)
 *   out_* = M[keep][:, keep], the UN-pruned block, as CSR with remapped columns    (:674)
 * Fails with ASP_ERR_INVALID when a frozen spin is not in the anchor's component (the
 * reference asserts it, :666).  *kept_spins and *out_nnz always receive the sizes; pass
 * capacity 0 and NULL out_* to get the mask only.  out_indptr needs *kept_spins + 1 entries. */
int asp_sparsify_component(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                           double const *data, uint8_t const *is_frozen, double reltol,
                           uint64_t anchor, uint8_t *keep, uint64_t *kept_spins,
                           uint64_t capacity, int64_t *out_indptr, int32_t *out_indices,
                           double *out_data, uint64_t *out_nnz);
/* Device time (ms) of this thread's last asp_sparsify_component call, copies excluded. */
float asp_sparsify_last_ms(void);

/* ------------------------------------------------------------------------- */
/* (4) Annealer: replaces ising_glass_annealer.{Hamiltonian,anneal}          */
/*     call sites: common.py:204,242-248; full_hilbert_space.py:212-218      */
/* ------------------------------------------------------------------------- */

typedef struct asp_sa_plan asp_sa_plan;

/* Build the device-resident sweep plan for E(s) = sum_ij J_ij s_i s_j +
 * sum_i h_i s_i.  J is a square CSR matrix with sorted, duplicate-free column
 * indices per row (scipy "canonical format"); it may carry a diagonal and need
 * not be symmetric (the sweep uses J + J^T).  Host preprocessing: symmetrised
 * off-diagonal part, DSATUR colouring, colour-major permutation,
 * 64-row sliced-ELL slabs; all uploaded once.  NULL on failure. */
asp_sa_plan *asp_sa_plan_create(uint64_t num_spins, int64_t const *indptr,
                                int32_t const *indices, double const *data,
                                double const *field);
void asp_sa_plan_destroy(asp_sa_plan *p);

typedef struct asp_sa_info {
  uint64_t num_spins;
  uint64_t nnz_offdiag;     /* entries of offdiag(J + J^T) after dropping zeros */
  uint64_t ell_entries;     /* slab entries incl. padding (64 * sum of widths)  */
  uint32_t num_colors;
  uint32_t num_blocks;      /* 64-row blocks (colour classes padded)            */
  uint32_t max_degree;
  int32_t energy_scale_exp; /* S: tracked energies are in units of 2^-S         */
  double diag_sum;          /* sum_i J_ii                                       */
  double beta0_auto;        /* ln 2 / max_i dE_max(i)                           */
  double beta1_auto;        /* ln 100 / min over non-zero couplings             */
} asp_sa_info;
int asp_sa_plan_info(asp_sa_plan const *p, asp_sa_info *info);

/* The same host preprocessing WITHOUT touching a device (inspection, CPU tests):
 * fills *info and, when non-NULL, colors[K] (greedy colour of every spin) and
 * position[K] (index of the spin in the padded, colour-major order). */
int asp_sa_layout_host(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                       double const *data, double const *field, asp_sa_info *info,
                       int32_t *colors, uint32_t *position);

/* Host-only: the visiting order of sweep `sweep` of the SHUFFLED variant (asp_sa_anneal_shuffled):
 * order[k] = k-th spin visited (level-major), level_of_position[k] = its level, *num_levels the
 * number of levels.  Any output may be NULL. */
int asp_sa_shuffled_order_host(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                               double const *data, double const *field, uint64_t seed,
                               uint32_t sweep, uint32_t *order, uint32_t *level_of_position,
                               uint32_t *num_levels);

/* Launch geometry override (0 = choose automatically).
 * replicas_per_group in {1,2,4,8}; threads multiple of 64, <= 1024. */
int asp_sa_set_launch(asp_sa_plan *p, int replicas_per_group, int threads);

/* Layout of the spins: by default one LDS byte per spin position (bit m = replica m of the
 * workgroup); automatically one LDS BIT per position with one replica per workgroup when the
 * bytes do not fit (K beyond ~1.4e5), and the same bit words kept in HBM when not even the
 * bits fit (K beyond ~1.3e6; slow, but no size limit).  packed = 1 / 2 forces the LDS-bit /
 * HBM-bit layout (tests, measurements), 0 restores the automatic choice.  Results never
 * depend on the layout. */
int asp_sa_set_packed(asp_sa_plan *p, int packed);

/* With four replicas per workgroup and up to ~4e4 spins the kernel keeps a 32-bit word
 * per position (one byte per replica), which makes the sign of a coupling term a single SDWA
 * instruction.  allow = 0 keeps the byte layout (tests, measurements); default 1.
 * Beyond the capacity of a byte per position (~1.4e5 spins) and with chains enough for four per
 * workgroup, four bits per position (up to ~2.4e5 spins; flips are LDS atomics); otherwise a bit
 * per position and one chain per workgroup.
 * asp_sa_last_layout: 0 = bytes, 1 = bits in LDS, 2 = words, 3 = bits in HBM, 6 = nibbles, for
 * the last anneal/greedy call. */
int asp_sa_set_wide(asp_sa_plan *p, int allow);
int asp_sa_last_layout(asp_sa_plan const *p);

/* Few chains on a large cluster (chains <= CUs / 2, e.g. the reference's default of 64
 * repetitions): each chain is spread over a TEAM of 2, 4 or 8 workgroups that split every colour
 * class, exchange one flip word per block and meet at a device-scope barrier per colour
 * (all of a team's workgroups must be resident together: the grid stays within the CU count,
 * team launches of one process take turns, and if the device is shared and a barrier times out the
 * call is repeated without teams — one process per GPU is assumed).  team = -1 chooses automatically (default), 0 never, 2/4/8 forces that
 * team size when the chains fit (tests, measurements).  Chains are bit-identical either way;
 * asp_sa_last_layout reports 4 for a team launch. */
int asp_sa_set_team(asp_sa_plan *p, int team);
/* How often the team barrier's watchdog gave up (the members of a team were not resident together:
 * another process or a long kernel held compute units): each trip costs the ~5 s the watchdog waits
 * plus the repeat of the call without teams, and used to be silent.  of_plan: calls of this plan
 * (NULL plan: 0); of_process: all plans of the process.  Either pointer may be NULL. */
int asp_sa_team_watchdog_trips(asp_sa_plan const *p, uint32_t *of_plan, uint64_t *of_process);

/* Field cache (default on): once a sweep flips few spins, a workgroup keeps the local fields
 * of every block in HBM and re-evaluates a block only after one of its neighbours flipped.
 * Pure optimisation of frozen sweeps; results are identical with it on or off. */
int asp_sa_set_field_cache(asp_sa_plan *p, int enable);

/* Run `repetitions` independent annealing chains (global replica ids
 * replica_offset .. replica_offset+repetitions-1) of num_sweeps sweeps, sweep t
 * at inverse temperature betas[t].  x0 == NULL: random initial spins from the
 * counter RNG; otherwise every chain starts from the packed configuration x0
 * (ceil(K/64) words, bit set = +1).  Outputs (host): out_x[repetitions *
 * ceil(K/64)] best configuration of each chain, out_e[repetitions] its energy.
 * The result depends only on (J, h, seed, betas, global replica id), not on the
 * launch geometry or the number of GPUs.  out_x / out_e may also be DEVICE pointers on the
 * library's device (the copies use hipMemcpyDefault): the multi-GPU layer gathers them over
 * RCCL without a host round trip. */
int asp_sa_anneal(asp_sa_plan *p, uint64_t seed, double const *betas,
                  uint32_t num_sweeps, uint32_t repetitions, uint32_t replica_offset,
                  uint64_t const *x0, uint64_t *out_x, double *out_e);
/* asp_sa_anneal that also returns every chain's energy after each sweep — the per-sweep
 * traces the older annealer API handed back as `(x, e_current[], e_best[]) = anneal(h, x0, seed,
 * number_sweeps, beta0, beta1)` (annealing_sign_problem/train.py:238-245,297).
 * out_trace[r * (num_sweeps + 1) + t] = tracked energy of chain r after t sweeps in units of
 * 2^-energy_scale_exp, relative to its initial configuration (entry 0 is 0); exact integer
 * bookkeeping of the accepted dE.  The best-so-far trace is its running
 * minimum. */
int asp_sa_anneal_trace(asp_sa_plan *p, uint64_t seed, double const *betas, uint32_t num_sweeps,
                        uint32_t repetitions, uint32_t replica_offset, uint64_t const *x0,
                        uint64_t *out_x, double *out_e, int64_t *out_trace);

/* asp_sa_anneal with a fresh visiting order every sweep — what the reference's annealer almost
 * certainly does (its published success probabilities are reproduced by this order and by no
 * fixed one: DESIGN.md §6.1), and therefore the order the Python entry points use by default.
 * Sweep t visits the spins in ascending (priority, index), priority = word 0 of
 * Philox4x32-10(counter (i, t, 0xFFFFFFFE, 0), key seed); every chain uses the same order.
 * Proposal arithmetic, random words, energy bookkeeping, outputs and determinism are those of
 * asp_sa_anneal; only the order differs.  The orders are built ON THE DEVICE, a chunk of sweeps at
 * a time (csrc/sa_shuffled.hip: priorities, levels of the priority graph, the sweep's couplings
 * re-laid level by level in blocks of 4 .. 64 spins), beside the sweep kernel of the previous
 * chunk: a workgroup per sweep with its arrays in LDS for clusters up to ~1.2e4 spins, grids over
 * all sweeps of the chunk with one launch per level beyond.  Spins stay in LDS in original order
 * (a word, a byte, four bits or one bit per spin: up to ~6e5 spins) and in HBM beyond that.
 * out_x / out_e may be host or device pointers.  asp_sa_last_layout reports 5. */
int asp_sa_anneal_shuffled(asp_sa_plan *p, uint64_t seed, double const *betas, uint32_t num_sweeps,
                           uint32_t repetitions, uint32_t replica_offset, uint64_t const *x0,
                           uint64_t *out_x, double *out_e);
/* Launch geometry of the shuffled sweep (0 = automatic): chains per workgroup in {1,2,4,8} and
 * wavefronts per workgroup in 1..8.  Results never depend on it. */
int asp_sa_set_shuffled_launch(asp_sa_plan *p, int chains_per_group, int wavefronts);
/* Teams of a shuffled-sweep workgroup (0 = automatic): with 2 the wavefronts split into two teams
 * that visit the same blocks for one half of the group's chains each (`wavefronts` above is then
 * per team, at most 4).  Results never depend on it. */
int asp_sa_set_shuffled_teams(asp_sa_plan *p, int teams);
/* Of the last asp_sa_anneal_shuffled call: the largest number of levels of a sweep. */
int asp_sa_last_shuffled(asp_sa_plan const *p, uint32_t *levels, float *order_ms);
/* Of the last shuffled call or batch item of this plan: spins per block of the level-major coupling
 * stream (64, or 4 .. 32 with lane packing: a wavefront then visits a block for 64 / spins_per_block
 * groups of chains at once — small levels filled with chains instead of padding lanes) and the
 * workgroups of its sweep launches.  Results never depend on either; ASP_SHUFFLED_LOG_S=2..6 in
 * the environment forces the block size, ASP_SHUFFLED_NO_PACKING=1 keeps blocks of 64. */
int asp_sa_last_shuffled_blocks(asp_sa_plan const *p, uint32_t *spins_per_block, uint32_t *workgroups);
/* Of the last shuffled call or batch item of this plan, how full its level-major blocks were (the
 * sweep of the call with the most blocks / quads): lane_fill = spins / lane slots (K over blocks x
 * spins per block: what is lost to levels that do not fill their last block), row_fill = couplings
 * of the rows (in quads) / coupling slots (blocks x their width: what is lost to a block being as
 * wide as its longest row).  The exec-mask counters of a profiler do not see either: a padding lane
 * and a padding coupling execute like real ones. */
int asp_sa_last_shuffled_fill(asp_sa_plan const *p, double *lane_fill, double *row_fill);

/* MANY independent problems in one call — the shape of the reference's production job: tens of
 * thousands of sampled clusters, each solved with 64 repetitions x 5120 sweeps
 * (Makefile:9,115-127; experiments/sampled_connected_components.py:764-767; common.py:236-239).
 * Item i is exactly asp_sa_anneal(plan, seed, betas, num_sweeps, repetitions, replica_offset,
 * NULL, out_x, out_e): every chain is bit-identical to that call's.  The groups of all problems
 * become the workgroups of a few shared launches (one per wavefront count), so a batch of small
 * clusters fills the chip instead of leaving > 90 % of it idle launch by launch.  Plans must
 * be distinct.  Problems that need the bit-packed spin layouts, and a batch of one, take the
 * single-problem path inside the call.  Items with ASP_SA_BATCH_SHUFFLED in `flags` are
 * asp_sa_anneal_shuffled calls (a fresh visiting order every sweep): the items with the same
 * number of sweeps SHARE their launches — per chunk of sweeps one order build over (problem, sweep)
 * and one sweep launch per kernel class (chains per group, spin layout, lane packing) over
 * (problem, workgroup), workgroups of one problem on one XCD —, so a batch of small clusters fills
 * its wavefronts with chains (blocks of 4 .. 32 spins) and the chip with problems. */
#define ASP_SA_BATCH_SHUFFLED 1u
typedef struct asp_sa_batch_item {
  asp_sa_plan *plan;
  uint64_t seed;
  double const *betas;  /* num_sweeps values */
  uint32_t num_sweeps;
  uint32_t repetitions;
  uint32_t replica_offset;
  uint32_t flags;       /* 0, or ASP_SA_BATCH_SHUFFLED: the item is an asp_sa_anneal_shuffled call */
  uint64_t *out_x;      /* repetitions * ceil(K/64) words */
  double *out_e;        /* repetitions */
} asp_sa_batch_item;
int asp_sa_anneal_batch(asp_sa_batch_item const *items, uint32_t count);
/* Device time (ms) of the sweep launches of this thread's last asp_sa_anneal_batch call. */
float asp_sa_batch_last_ms(void);

/* Replaces ising_glass_annealer.greedy_solve (call site common.py:250; the only in-tree
 * description is the commented prototype at common.py:298-438): couplings are visited
 * strongest first and clusters of already-signed spins are merged so that the visited
 * coupling is satisfied (host, union-find with parity); then strict-descent sweeps
 * (flip iff dE < 0, the sweep kernel without random numbers) run on the device until no
 * spin flips or max_sweeps is reached.  Deterministic.  out_x: ceil(K/64) words. */
int asp_sa_greedy(asp_sa_plan *p, uint32_t max_sweeps, uint64_t *out_x, double *out_e,
                  uint32_t *out_sweeps);

/* Host-only: the cluster-merging half of asp_sa_greedy (no relaxation, no device). */
int asp_sa_greedy_tree_host(uint64_t num_spins, int64_t const *indptr, int32_t const *indices,
                            double const *data, double const *field, uint64_t *out_x);

/* Device time (ms, HIP events on the launch stream) of the sweep kernel of the
 * last asp_sa_anneal call, and of everything device-side in that call. */
float asp_sa_last_sweep_ms(asp_sa_plan const *p);
float asp_sa_last_total_ms(asp_sa_plan const *p);

/* Diagnostics of the last asp_sa_anneal call: per chain the best tracked energy
 * (fixed point, units of 2^-S, relative to the chain's start) and the number of
 * accepted flips.  `count` = that call's repetitions.  Either may be NULL. */
int asp_sa_last_stats(asp_sa_plan const *p, uint32_t count, int64_t *tracked,
                      uint64_t *accepted);
/* Launch geometry used by the last asp_sa_anneal call. */
int asp_sa_last_launch(asp_sa_plan const *p, int *replicas_per_group, int *threads,
                       int *groups);

/* E(x) for `count` packed configurations (host in, host out). */
int asp_sa_energy(asp_sa_plan *p, uint32_t count, uint64_t const *x, double *out_e);

#ifdef __cplusplus
}
#endif
#endif /* ASP_H */
