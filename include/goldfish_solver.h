/* goldfish_solver.h -- C ABI of libgoldfish_solver.so: sparse direct solves with K on the device.
 *
 * Replaces GOLDFISH/utils/opt_utils.py:156-209 (solve_Ax_b / solve_ATx_b: a MUMPS factorisation of a copy of K on every
 * call, GOLDFISH/operations/disp_imop.py:38-44, 130-142).  K never leaves the device: its values are read in place from
 * libgoldfish_hip's buffer (gf_device_ptr(h, GF_BUF_VAL_K)).
 *
 * Method (hand-written HIP, no library dependency): the control points are renumbered by a bandwidth-reducing ordering
 * computed once on the host (reverse Cuthill-McKee on the neighbour graph; the pattern never changes during an
 * optimisation), K is scattered into block-banded storage (64 x 64 tiles, lower triangle) and factorised as
 * P K P^T = L D L^T, right-looking over block columns: diagonal tile (LDL^T + inverse of its unit-triangular factor, one
 * workgroup), panel tiles L_ik = A_ik L_kk^-T D_k^-1 and trailing tiles A_ij -= (L_ik D_k) L_jk^T as 64^3 products on
 * v_mfma_f64_16x16x4.  No pivoting (K is symmetric; a vanishing pivot is reported).  Solves are block forward / backward
 * substitutions with the inverted diagonal tiles, followed by iterative refinement with the block-CSR K itself.
 * K is symmetric, so K^T x = b is the same solve.
 * All functions return 0 on success; gfs_last_error() describes the last failure of the calling thread. */
#ifndef GOLDFISH_SOLVER_H
#define GOLDFISH_SOLVER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gfs_handle gfs_handle;

const char* gfs_last_error(void);

/* ncp control points, 3 dofs each (dof = 3 * cp + component).  nb_ptr / nb (host): control-point-level neighbour lists =
 * the block pattern of K as gf_pattern(GF_MAT_K) returns it (K's values: per control point a the three dof rows, each
 * [neighbour k][j], i.e. value of entry ((a, i), (nb[k], j)) at 9 * nb_ptr[a] + i * 3 * deg(a) + 3 * k + j).
 * new_index (host, ncp): position of every control point in the factorisation order (a permutation).
 * d_valK: DEVICE pointer to K's values, borrowed and re-read by every gfs_refactor. */
int gfs_create(int device, int64_t ncp, const int64_t* nb_ptr, const int32_t* nb, const int32_t* new_index, const double* d_valK, gfs_handle** out);
/* Nested-dissection multifrontal factorisation instead of the skyline (large models: memory O(n log n) and work O(n^1.5) for a shell instead of
 * n x bandwidth and n x bandwidth^2).  The fronts come from the caller's symbolic phase (goldfish_amd/_nd.py: recursive coordinate bisection with
 * vertex separators), all host arrays: elim [ncp] / elim_off [nfronts + 1] = control points eliminated per front, fronts in post-order; bnd / bnd_off =
 * boundary control points per front in ascending elimination order; parent [nfronts] (-1 = root); order [ncp] = elimination position of a control
 * point; front_of [ncp]; pmap [bnd_off[nfronts]] = position of each boundary control point in the PARENT front's numbering (eliminated first).
 * gfs_refactor / gfs_solve / gfs_solve_dev / gfs_info / gfs_destroy work on the handle as for the skyline. */
int gfs_create_nd(int device, int64_t ncp, const int64_t* nb_ptr, const int32_t* nb, const double* d_valK, int64_t nfronts, const int64_t* elim, const int64_t* elim_off,
                  const int64_t* bnd, const int64_t* bnd_off, const int64_t* parent, const int64_t* order, const int64_t* front_of, const int64_t* pmap, gfs_handle** out);
void gfs_destroy(gfs_handle* h);
/* numeric factorisation of the values currently in d_valK */
int gfs_refactor(gfs_handle* h);
/* Optional, before a gfs_refactor: start clearing the factor storage now (asynchronously, on the handle's stream; C4: 57 GB, 9.4 - 11.5 ms of pure HBM writes, 5 % of a
 * factorisation).  The factors at hand are gone from this call on; gfs_refactor skips its own clearing once.  For a caller whose device is IDLE before the next
 * factorisation (host-side work between two solves).  Measured and NOT used by the Newton loop (tools/time_prepare_overlap.py, profiles/r05_prepare_overlap.txt): beside the
 * assembly pass the fill kernel takes the element kernel's issue slots -- the pass takes 21.3 instead of 11.5 ms, the factorisation 208.7 instead of 218.9 ms, the step the same. */
int gfs_prepare_refactor(gfs_handle* h);
/* x = K^{-1} b with the current factors and up to max_refine steps of iterative refinement (stops when the residual no longer
 * decreases); b, x: 3 * ncp doubles, host pointers (gfs_solve) or device pointers (gfs_solve_dev).
 * rel_residual (may be NULL): |b - K x| / |b| of the returned solution (2-norm). */
int gfs_solve(gfs_handle* h, const double* b, double* x, int max_refine, double* rel_residual);
int gfs_solve_dev(gfs_handle* h, const double* d_b, double* d_x, int max_refine, double* rel_residual);
/* General mode for a K that is not symmetric (the load stiffness of a follower pressure, GOLDFISH's solve_ATx_b next to solve_Ax_b,
 * utils/opt_utils.py:156-209): gfs_set_general(h, 1) builds the reverse index of the (symmetric) block pattern once; from then on gfs_refactor
 * factors the SYMMETRIC PART (K + K^T) / 2 and the solves use it as the preconditioner of the iterative refinement, whose residual is taken with
 * K itself (gfs_solve*: K x = b) or with K^T (gfs_solve_transposed*: K^T x = b).  Converges when the skew part is small against the symmetric
 * part (refinement contracts by |S^-1 (K - S)|); the returned residual and gfs_info's backward error say whether it did -- give max_refine
 * room (40: the loop ends by itself when the residual no longer drops).  With general mode off (the default) the transposed solves are the plain ones. */
int gfs_set_general(gfs_handle* h, int nonsymmetric);
int gfs_solve_transposed(gfs_handle* h, const double* b, double* x, int max_refine, double* rel_residual);
int gfs_solve_transposed_dev(gfs_handle* h, const double* d_b, double* d_x, int max_refine, double* rel_residual);
/* Several right-hand sides in one call (the adjoints of several functionals -- internal energy, volume, aggregated stress -- share K^T): b, x hold nrhs
 * vectors of 3 * ncp doubles one after the other; nrhs <= 8; rel_residual (may be NULL): nrhs values; transpose != 0: K^T x = b (general mode).  In the
 * nested-dissection mode groups of three right-hand sides share ONE pass over the factors (the sweeps are bound by the factor bytes: every tile entry is loaded once
 * and multiplied into three sums, each in the order of the single solve, so x is bitwise what gfs_solve returns), the groups run next to each other on their own
 * streams, the refinement in lockstep rounds; in the skyline mode (small models) the right-hand sides run one after the other.  gfs_info's backward error is then
 * the largest of the nrhs solves. */
int gfs_solve_multi(gfs_handle* h, int nrhs, const double* b, double* x, int max_refine, double* rel_residual, int transpose);
int gfs_solve_multi_dev(gfs_handle* h, int nrhs, const double* d_b, double* d_x, int max_refine, double* rel_residual, int transpose);
/* ---- Symbolic phase on the host (no device call): nested dissection of the control-point graph by recursive coordinate bisection with vertex separators, every cut
 * at the rank within cut_window (a fraction of the region's size) of the median that gives the smallest separator; the fronts in post-order with their boundaries and the
 * extend-add maps -- the arguments gfs_create_nd takes.  nb_ptr / nb: neighbour lists as for gfs_create; coords: ncp x dim doubles; leaf: regions of at most that many
 * control points are not split; threads: the two halves of the large regions run on up to that many threads.  The result is what goldfish_amd/_nd.py computes (the tests
 * compare the two entry by entry).  gfs_symbolic_sizes gives the number of fronts and the length of bnd / pmap; gfs_symbolic_copy fills caller-allocated arrays: elim [ncp],
 * elim_off [nfronts + 1], bnd [nbnd], bnd_off [nfronts + 1], parent [nfronts], order [ncp], front_of [ncp], pmap [nbnd] (any of them may be NULL). */
typedef struct gfs_symbolic gfs_symbolic;
int gfs_symbolic_create(int64_t ncp, const int64_t* nb_ptr, const int32_t* nb, const double* coords, int dim, int64_t leaf, double cut_window, int threads, gfs_symbolic** out);
void gfs_symbolic_sizes(const gfs_symbolic* s, int64_t* nfronts, int64_t* nbnd);
void gfs_symbolic_copy(const gfs_symbolic* s, int64_t* elim, int64_t* elim_off, int64_t* bnd, int64_t* bnd_off, int64_t* parent, int64_t* order, int64_t* front_of, int64_t* pmap);
void gfs_symbolic_destroy(gfs_symbolic* s);
/* ---- Partial handles: the pieces of a factorisation distributed over several GPUs (goldfish_amd/_dsolver.py; one process per GPU, K's values replicated).
 * gfs_create_nd_partial takes a SUB-FOREST of the elimination tree with the arguments of gfs_create_nd and these differences: elim / elim_off cover only the control
 * points this handle eliminates; front_of[cp] = -1 for the others; order[cp] = position in elim for the handle's own control points, larger than all of those for
 * control points eliminated LATER by another handle (the boundaries of this handle's root fronts), smaller than zero for control points eliminated EARLIER by another
 * handle; a root front may have a boundary (its Schur complement is what gfs_export_schur packs); a front may eliminate nothing -- a STUB that stands for a subtree
 * factored by another handle: its tiles are that subtree root's Schur complement, copied in by every gfs_refactor from the device buffer registered with
 * gfs_set_schur_source (gfs_schur_doubles(h, front) doubles: the lower triangle of 64 x 64 tiles, row I at I (I + 1) / 2, tile (I, J) at offset I - J), its boundary
 * contribution to the forward sweep is what gfs_set_fbnd stored (3 doubles per boundary control point, in the order of the front's boundary list).
 * The sweeps of a partial handle run in halves: gfs_forward_dev (right-hand side -> y of the handle's dofs, boundary contributions of every front: gfs_get_fbnd of a
 * root front is what the owner of the tree above needs), gfs_backward_dev (x of the handle's dofs into the vector at gfs_x_ptr, 3 * ncp doubles in the original
 * numbering, whose entries at the root fronts' boundary control points the caller has written before). */
int gfs_create_nd_partial(int device, int64_t ncp, const int64_t* nb_ptr, const int32_t* nb, const double* d_valK, int64_t nfronts, const int64_t* elim, const int64_t* elim_off,
                          const int64_t* bnd, const int64_t* bnd_off, const int64_t* parent, const int64_t* order, const int64_t* front_of, const int64_t* pmap, gfs_handle** out);
int64_t gfs_schur_doubles(gfs_handle* h, int64_t front);
int gfs_export_schur(gfs_handle* h, int64_t front, double* d_buf);
int gfs_set_schur_source(gfs_handle* h, int64_t front, const double* d_buf);
int gfs_get_fbnd(gfs_handle* h, int64_t front, double* d_out);
int gfs_set_fbnd(gfs_handle* h, int64_t front, const double* d_in);
/* the same for n fronts in one call: their boundary contributions packed one after the other in d_out / d_in (one synchronisation instead of one blocking copy per front) */
int gfs_get_fbnd_packed(gfs_handle* h, int64_t n, const int64_t* fronts, double* d_out);
int gfs_set_fbnd_packed(gfs_handle* h, int64_t n, const int64_t* fronts, const double* d_in);
double* gfs_x_ptr(gfs_handle* h);
/* A handle on ONE RANK'S K of a sharded model (local numbering: owned control points, then ghosts): mask [ncp] (host), 1 = the control point's rows hold values.
 * gfs_refactor then reads the block of a pair whose later control point has no row from the earlier one's row, transposed (K symmetric): the subtrees a rank
 * eliminates never need a row another rank assembled -- no replicated K (goldfish_amd/_dsolver.py, round 5).  NULL: every row holds values (the default). */
int gfs_set_row_mask(gfs_handle* h, const unsigned char* mask);
int gfs_forward_dev(gfs_handle* h, const double* d_b);
int gfs_backward_dev(gfs_handle* h);
/* info[0] = half bandwidth (dofs), [1] = block columns, [2] = band tiles per block row, [3] = device bytes,
 * [4] = flops of one factorisation, [5] = 1 if the last factorisation met a pivot below 1e-14 * max |diag|, else 0,
 * [6] = normwise backward error |b - K x| / (|K|_F |x| + |b|) of the last solve (what a backward-stable solve keeps at round-off
 * level whatever cond(K); |b - K x| / |b| alone has a floor of eps cond(K)), [7] = |K|_F of the factored matrix */
int gfs_info(gfs_handle* h, double info[8]);

#ifdef __cplusplus
}
#endif
#endif
