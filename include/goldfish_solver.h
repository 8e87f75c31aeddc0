/* goldfish_solver.h -- C ABI of libgoldfish_solver.so: sparse direct solves with K on the device.
 *
 * Replaces GOLDFISH/utils/opt_utils.py:156-209 (solve_Ax_b / solve_ATx_b: a MUMPS factorisation of a copy of K on every
 * call, GOLDFISH/operations/disp_imop.py:130-142) for everything after the first solve: the sparsity pattern of K never
 * changes during an optimisation, so the symbolic work and the fill-reducing ordering are done ONCE on the host
 * (SuperLU through scipy, symmetric mode) and every later Newton step / adjoint solve is a numeric re-factorisation
 * (rocSOLVER csrrf_refactlu) plus triangular solves (csrrf_solve) on the GPU, reading K's values in place from
 * libgoldfish_hip's buffer (gf_device_ptr(h, GF_BUF_VAL_K)).  K is symmetric, so K^T x = b is the same solve.
 * All functions return 0 on success; gfs_last_error() describes the last failure of the calling thread. */
#ifndef GOLDFISH_SOLVER_H
#define GOLDFISH_SOLVER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gfs_handle gfs_handle;

const char* gfs_last_error(void);

/* n: matrix order.  ptrA/indA (host, CSR pattern of K, sorted) and d_valA (DEVICE pointer to K's values, borrowed and
 * re-read by every gfs_refactor).  ptrT/indT/valT (host): T = (L - I) + U of a factorisation P K Q = L U of the CURRENT K;
 * pivP/pivQ (host): row i of P K is row pivP[i] of K, column j of K Q is column pivQ[j] of K. */
int gfs_create(int device, int64_t n, int64_t nnzA, const int32_t* ptrA, const int32_t* indA, const double* d_valA,
               int64_t nnzT, const int32_t* ptrT, const int32_t* indT, const double* valT,
               const int32_t* pivP, const int32_t* pivQ, gfs_handle** out);
void gfs_destroy(gfs_handle* h);
/* numeric re-factorisation with the values currently in d_valA (same pattern) */
int gfs_refactor(gfs_handle* h);
/* x = K^{-1} b (host pointers, n doubles each) with the current factors */
int gfs_solve(gfs_handle* h, const double* b, double* x);
/* fill-in and memory of the factors */
int64_t gfs_nnz_factors(gfs_handle* h);
int64_t gfs_device_bytes(gfs_handle* h);

#ifdef __cplusplus
}
#endif
#endif
