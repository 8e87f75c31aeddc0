/* goldfish_hip.h -- C ABI of libgoldfish_hip.so, the MI355X (gfx950) implementation of
 * GOLDFISH's shell assembly + sensitivity hot path.
 *
 * The reference has NO FFI for this path (it is Python -> pybind dolfin / petsc4py,
 * SURVEY.md 8(b)); every entry point below therefore cites the Python method of
 * GOLDFISH/nonmatching_opt.py or GOLDFISH/operations/<name>.py whose work it replaces.
 * The Python host layer (goldfish_amd/nonmatching_opt.py) binds these with ctypes;
 * INTEGRATION.md shows the stub a GOLDFISH maintainer would add.
 *
 * Ownership: host arrays belong to the caller; device buffers belong to the handle;
 * pointers returned by gf_device_ptr are borrowed for the handle's lifetime.
 * Errors: 0 = ok, non-zero = error code, message via gf_last_error() (thread-local).
 * Threading: one handle per host thread; calls on one handle are not re-entrant.
 * All work of a handle is issued on one HIP stream; host-pointer calls synchronise
 * before returning, *_async / device-pointer calls do not (use gf_sync).
 */
#ifndef GOLDFISH_HIP_H
#define GOLDFISH_HIP_H

#include <stdint.h>
#include "goldfish_model.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gf_handle gf_handle;

/* assemble flags */
enum { GF_ASM_R = 1, GF_ASM_K = 2, GF_ASM_DRDCP = 4, GF_ASM_DRDH = 8, GF_ASM_ALL = 15 };
/* device buffers exposed through gf_device_ptr */
enum {
    GF_BUF_CP = 0,      /* double[total_cp][4] (c_x, c_y, c_z, w)           */
    GF_BUF_U = 1,       /* double[ndof]                                      */
    GF_BUF_H = 2,       /* double[total_cp]                                  */
    GF_BUF_R = 3,       /* double[ndof] residual                             */
    GF_BUF_VAL_K = 4,   /* CSR values, see gf_pattern                        */
    GF_BUF_VAL_C0 = 5, GF_BUF_VAL_C1 = 6, GF_BUF_VAL_C2 = 7,
    GF_BUF_VAL_H = 8
};

int         gf_device_count(void);
const char* gf_last_error(void);

/* replaces NonMatchingOpt.__init__ + mortar_meshes_setup + set_residuals for the device path
 * (nonmatching_opt.py:12-127, 422-452): builds element/quadrature tables, static CSR patterns,
 * coupling lists, and allocates all device state on `device`. */
int  gf_create(const gf_model_desc* desc, int device, gf_handle** out);
void gf_destroy(gf_handle* h);

int64_t gf_total_cp(const gf_handle* h);
int64_t gf_num_dofs(const gf_handle* h);
int64_t gf_num_elements(const gf_handle* h);
int64_t gf_num_gauss_points(const gf_handle* h);
int64_t gf_num_mortar_points(const gf_handle* h);
int64_t gf_device_bytes(const gf_handle* h);

/* update_CPIGA(cp_array_iga, field) nonmatching_opt.py:495-506 (homogeneous coordinate `field`) */
int gf_set_cp(gf_handle* h, int field, const double* cp, int64_t n);
/* update_h_th_IGA nonmatching_opt.py:516-525 (one value per control point) */
int gf_set_thickness(gf_handle* h, const double* hth, int64_t n);
/* update_uIGA nonmatching_opt.py:474-484 */
int gf_set_u(gf_handle* h, const double* u, int64_t n);

/* static CSR patterns (row = vector dof).  which: GF_MAT_* of goldfish_model.h */
int64_t gf_nnz(const gf_handle* h, int which);
int     gf_pattern(const gf_handle* h, int which, int64_t* rowptr, int32_t* col);
/* control-point-level pattern of K: nb_ptr [total_cp + 1], nb [gf_cp_graph_size]: the control points coupled to control point a (itself included), ascending --
 * the block pattern of K (every entry stands for a 3 x 3 block) that goldfish_solver.h's gfs_create / gfs_create_nd / gfs_symbolic_create take */
int64_t gf_cp_graph_size(const gf_handle* h);
int     gf_cp_graph(const gf_handle* h, int64_t* nb_ptr, int32_t* nb);

/* One pass of the hot path over the current state, results left in HBM:
 *   GF_ASM_R     RIGA()          nonmatching_opt.py:941-948  (assemble_RFE :726-770 + BCs)
 *   GF_ASM_K     dRIGAduIGA()    nonmatching_opt.py:950-959  (assemble_dRFEduFE :772-841)
 *   GF_ASM_DRDCP dRIGAdCPIGA(f)  nonmatching_opt.py:992-1004 (assemble_dRFEdCPFE :843-926), all 3 fields
 *   GF_ASM_DRDH  dRIGAdh_th()    nonmatching_opt.py:1006-1015
 * i.e. DispImOpeartion.apply_nonlinear + linearize (operations/disp_imop.py:33-56). Asynchronous. */
int gf_assemble(gf_handle* h, int flags);
int gf_sync(gf_handle* h);

/* host copies of results (synchronising) */
int gf_get_residual(gf_handle* h, double* R, int64_t n);
int gf_get_values(gf_handle* h, int which, double* vals, int64_t n);

/* y += A x (transpose = 0) or y += A^T x (transpose = 1), A = matrix `which` as last assembled:
 * DispImOpeartion.apply_linear_fwd / apply_linear_rev, operations/disp_imop.py:58-128. Host pointers. */
int gf_apply(gf_handle* h, int which, int transpose, const double* x, int64_t nx, double* y, int64_t ny);

/* IntEnergyExOperation.Wint/dWintduIGA/dWintdCPIGA/dWintdh_th (operations/int_energy_exop.py:55-107)
 * and VolumeExOperation.volume/dvoldCPIGA/dvoldh_th (operations/volume_exop.py:46-84).
 * out[0] = W_int, out[1] = volume, out[2] = penalty energy.  Gradient pointers may be NULL.
 * dWdu has Dirichlet rows zeroed when apply_bcs != 0. dWdcp/dVdcp: 3 arrays of total_cp. */
int gf_functionals(gf_handle* h, double out[3], double* dWdu, double* dWdcp, double* dWdh,
                   double* dVdcp, double* dVdh, int apply_bcs);

/* ComplianceExOperation.cpl/dcplduIGA/dcpldCPIGA (operations/compliance_exop.py:50-99): C = sum_s int forces[s] . u_hom dA
 * with forces = 3 values per patch; dCdu [ndof] (Dirichlet rows zeroed when apply_bcs), dCdcp 3 arrays of total_cp. */
/* one gradient field of the LAST gf_functionals call, copied on request (call gf_functionals with NULL gradient pointers, then fetch
 * what the caller actually uses: every field is a D2H copy of ndof or total_cp doubles): field 0 dWdu [ndof], 1 dWdcp [3 total_cp],
 * 2 dWdh [total_cp], 3 dVdcp [3 total_cp], 4 dVdh [total_cp].  Fails if another functional entry (gf_compliance, gf_stress_forms,
 * gf_shape_regu) has used the gradient buffer since. */
int gf_get_functional_gradient(gf_handle* h, int field, double* out, int64_t n);
/* per-patch W_int and volume of the LAST gf_functionals call (the terms of the sums out[0], out[1]; ghost patches of a shard: 0) --
 * VolumeExOperation(vol_surf_inds = a subset of the patches) sums the listed ones (operations/volume_exop.py:9-27, 46-50).
 * Either pointer may be NULL; np = number of patches. */
int gf_functionals_per_patch(gf_handle* h, double* W_patch, double* V_patch, int64_t np);
int gf_compliance(gf_handle* h, const double* forces, int64_t nf, double* C, double* dCdu, double* dCdcp, int apply_bcs);

/* MaxvMStressExOperation (operations/max_vmstress_exop.py): the per-patch aggregation forms
 *   forms[s] = int g(sigma_vM) dA,  g = exp(rho (sigma - m_list[s]))   mode 0  (KS_symexp :167)
 *                                   g = (sigma / m_list[s])^rho        mode 1  (pnorm_symexp :170; induced_power :173
 *                                                                               = two calls, rho+1 and rho)
 * of the von Mises stress at the station xi2 = surf * h/2 (surf = +1 "top", -1 "bottom", 0 "middle", :29-36), which the
 * reference takes from PENGoLINS ShellStressSVK.vonMisesStress (:38-47); measure 0 = Cauchy stress, 1 = 2nd Piola-Kirchhoff.
 * vmax[s] = largest Gauss-point sigma_vM of patch s (compute_m / compute_max_vM :147-165 use an L2 projection onto
 * linears there).  Gradients of the forms (un-weighted; every control point belongs to one patch, so the caller applies
 * its per-patch chain-rule factors, :330-440): dIdu [ndof] (dmax_vMdu_forms :78-95; Dirichlet rows zeroed when
 * apply_bcs), dIdcp 3 arrays of total_cp (dmax_vMdcp_forms :98-118), dIdh [total_cp] (dmax_vMdh_th_forms :122-137).
 * forms, vmax: [n_patches] (0 for the ghost patches of a shard).  Any output pointer may be NULL. */
int gf_stress_forms(gf_handle* h, int mode, double rho, const double* m_list, int64_t nm, int surf, int measure,
                    double* forms, double* vmax, double* dIdu, double* dIdcp, double* dIdh, int apply_bcs);

/* Shape regularisation of demos_om/shape_opt/eVTOL/int_energy_regu_exop.py:30-38 (IntEnergyReguExOperation adds it to W_int):
 *   value = sum_s coef[s] int |grad_s(P_f - P_f^0)|^2 dA,   grad_s = surface gradient on the CURRENT geometry (spline.grad),
 * P_f = homogeneous coordinate `field` of the control net (cpFuncs[field]), P_f^0 = cp0 (its initial value, [total_cp]).
 * dcp (may be NULL): 3 arrays of total_cp, d value / d (homogeneous coordinates 0, 1, 2) -- the geometry enters through the
 * metric and the area element too.  No dependence on u or the thickness. */
int gf_shape_regu(gf_handle* h, int field, const double* cp0, int64_t ncp, const double* coef, int64_t nc, double* value, double* dcp);

/* Moving intersections: NonMatchingOpt.dRIGAdxi / dRIGAdxi_sub (nonmatching_opt.py:1042-1341) -- derivative of the penalty
 * residual with respect to the parametric coordinates of the mortar vertices, at the current u / CP / thickness.
 * blocks[v][dir][side'][a][i] (npts x 6 x 2 x (p+1)^2 x 3 doubles): d(residual entry i of the a-th support control point of
 * side') for dir 0..3 = d/d(xi of side dir/2, parametric direction dir%2) of vertex v itself and dir 4,5 = d/d(tau_0, tau_1),
 * the parametric curve tangent the model was created with (if_tau): the caller chains it with its own d(tau)/d(xi of the
 * neighbouring vertices).  windows (may be NULL): first support control-point indices (iu0, iv0) per vertex and side,
 * [npts][2][2]; the a-th support control point of a patch with nu control points in u is (iu0 + a % (p+1)) + (iv0 + a / (p+1)) * nu.
 * No Dirichlet treatment (the reference zeroes the rows afterwards, :1057-1062). */
int gf_penalty_dxi(gf_handle* h, double* blocks, int64_t n, int32_t* windows, int64_t nw);
/* the same for the mortar vertices v_first .. v_first + v_count - 1 only (blocks, windows sized for v_count vertices): the vertices
 * of the interfaces that actually move, instead of every vertex of the model (0.64 GB of blocks at C4) */
int gf_penalty_dxi_range(gf_handle* h, int64_t v_first, int64_t v_count, double* blocks, int64_t n, int32_t* windows, int64_t nw);
/* NonMatchingOpt.update_transfer_matrices (nonmatching_opt.py:567-600 rebuilds the mortar transfer matrices at new parametric coordinates) for ONE interface
 * without re-creating the model: xi [npts_if][2 sides][2], tau [npts_if][2], wt [npts_if] in gf_model_desc's layout (if_xi, if_tau, if_wt).  Returns 0 when
 * the vertex tables were patched (every vertex stayed in its knot spans, so the coupling pattern and all index tables are unchanged; the assembled matrices are
 * marked stale), 2 -- with nothing changed -- when a vertex crossed a knot line: the caller then creates a new handle; 1 on errors. */
int gf_update_interface(gf_handle* h, int iface, const double* xi, const double* tau, const double* wt, int64_t npts_if);
/* Reverse-mode product with those blocks WITHOUT moving them to the host (DispMintImOpeartion.apply_linear_rev, operations/disp_mi_imop.py:75-104:
 * d_xi += (dR/dxi)^T d_res): out[v][dir] = sum over side', a, i of blocks[v][dir][side'][a][i] * lam[dof], the rows this handle owns (owned patches of a shard)
 * with Dirichlet rows skipped (the reference zeroes them, nonmatching_opt.py:1057-1062); lam: ndof doubles (host), out: 6 per mortar vertex of the range (host).
 * The caller chains dir 4, 5 with d(tau)/d(xi) as for gf_penalty_dxi.  On a shard the results of the ranks add up (a cut interface is evaluated by both). */
int gf_penalty_dxi_rev(gf_handle* h, int64_t v_first, int64_t v_count, const double* lam, int64_t nlam, double* out, int64_t nout);

/* borrowed device pointer to one of the GF_BUF_* buffers (for zero-copy users: bench, RCCL exchange) */
void* gf_device_ptr(gf_handle* h, int which);
/* y_dev += A x_dev on device pointers (no host copies, asynchronous) */
int gf_apply_dev(gf_handle* h, int which, int transpose, const double* x_dev, double* y_dev);
/* DispImOpeartion.apply_linear_fwd / _rev (disp_imop.py:58-128) in ONE call: nmat products with one copy in and one copy out
 * per vector instead of gf_apply's three per product.
 *   transpose = 0:  ys[0] (ndof, in/out) += sum_m A_which[m] xs[m]      (xs[m]: ndof values for K, total_cp otherwise)
 *   transpose = 1:  ys[m] (in/out, ndof for K, total_cp otherwise) += A_which[m]^T xs[0]      (xs[0]: ndof values)
 * Host pointers; 1 <= nmat <= 5, every matrix must have been assembled. */
int gf_apply_many(gf_handle* h, int transpose, int nmat, const int* which, const double* const* xs, double* const* ys);
/* average duration (ms) of the dominant kernel (shell element kernel) over the launches since the
 * last call, measured with HIP events on the handle's stream; resets the accumulator. */
double gf_kernel_ms(gf_handle* h, int* n_launches);

/* the HIP stream (hipStream_t) the handle launches on: lets a caller order its own streams / collectives against the
 * library's work with events (torch.cuda.ExternalStream(ptr)) instead of gf_sync */
void* gf_stream(gf_handle* h);

/* which element path gf_assemble runs on this handle: 4 = walking MFMA kernel that stores row records + record gather (default for
 * p = 2, 3), 6 = p = 4 default: Newton passes (R, K) through row records (one walk, gf_element_rec4.hpp), passes with dR/dCP / dR/dh through
 * element blocks (each pass kind on the path that is faster for it), 5 = p = 4 with row records for every pass (three walks per full pass;
 * GF_ASSEMBLY=rec: half the memory and traffic), 0 = MFMA element kernel, one block per element + row gather (GF_ASSEMBLY=block: the
 * cross-check path), 3 = FP64-VALU element kernel + gather (GF_ELEMENT=valu) */
int gf_assembly_path(const gf_handle* h);

#ifdef __cplusplus
}
#endif
#endif
