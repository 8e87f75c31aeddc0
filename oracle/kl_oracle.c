/* kl_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).  Never linked, imported or
 * called by the product path (goldfish_amd/, libgoldfish_hip.so); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * PARITY UNPINNED against FEniCS/PENGoLINS: the reference's arithmetic for this path
 * lives in un-vendored, un-pinned third-party packages (PENGoLINS, ShNAPr, tIGAr,
 * legacy FEniCS 2019.x, PETSc -- SURVEY.md section 8(c)); the reference's tests hold no
 * golden vectors.  What pins this restatement instead:
 *   (1) tests/test_oracle_derivatives.py: every gradient/Hessian here vs torch.autograd
 *       of the bare energies in oracle/kl_energy_torch.py, and vs finite differences
 *       (the reference's own verification pattern, check_partials /
 *       GOLDFISH/nonmatching_opt.py:975-990 dRIGAdCPIGA_FD);
 *   (2) tests/test_known_answers.py: Scordelis-Lo roof -> 0.3006
 *       (GOLDFISH/tests/test_slr.py:42-50), patch tests, rigid-body invariance.
 *
 * What it restates (reference file:line, read as text):
 *   gfo_residual   <- NonMatchingOpt.assemble_RFE + RIGA        nonmatching_opt.py:726-770, 941-948
 *   gfo_assemble K <- assemble_dRFEduFE + dRIGAduIGA + BCs      nonmatching_opt.py:772-841, 950-959, 660-724
 *   gfo_assemble dRdCP_f <- assemble_dRFEdCPFE + dRIGAdCPIGA    nonmatching_opt.py:843-926, 992-1004;
 *                          utils/opt_utils.py:212-260 (penalty shape-derivative blocks)
 *   gfo_assemble dRdh <- assemble_dRFEdh_th + dRIGAdh_th        nonmatching_opt.py:928-938, 1006-1015
 *   gfo_functionals <- IntEnergyExOperation / VolumeExOperation operations/int_energy_exop.py:55-107,
 *                                                               operations/volume_exop.py:46-84
 * Formulation: SURVEY.md Appendix A (Kiendl 2009 KL shell, SVK, Herrema 2019 penalty).
 * The FEniCS detour (P6 triangles + M^T K M extraction) is NOT reproduced: assembly is
 * directly in IGA dofs with (p+1)x(q+1) Gauss points per non-empty knot span.
 *
 * Style: "classical" loops over basis-function pairs (r=(a,i), s=(b,j)) with the Kiendl
 * first/second strain variations -- deliberately different from the HIP kernels'
 * phi^T G phi pointwise-Hessian formulation, so agreement is between two derivations.
 * Penalty Hessians are obtained by complex-step differentiation of a hand-written
 * analytic gradient (exact to round-off), again independent of the HIP closed forms.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/goldfish_model.h"

#define MAXP 5
#define MAXNB ((MAXP + 1) * (MAXP + 1))

typedef struct {
    int p, q, nu, nv;
    double *ku, *kv;
    int nelu, nelv;          /* non-empty spans per direction */
    int *spanu, *spanv;      /* knot-span index of each element */
    int ngu, ngv;            /* Gauss points per direction (p+1, q+1) */
    double *bu, *bv;         /* [nel][ng][3][deg+1] 1-D basis values/1st/2nd derivatives */
    double *wu, *wv;         /* [nel][ng] Gauss weight * span length/2 */
    int64_t cp_off;
    double E, nu_, f[3], pd[3];   /* pd != 0: load per unit projected area (gf_model_desc.load_proj) */
    double press, et[4][3];       /* follower pressure; dead edge tractions, edge 2 d + side (gf_model_desc.pressure / edge_traction) */
} patch_t;

typedef struct {
    int pa, pb;
    int64_t npts;
    /* per point, per side: support base (iu0, iv0), rational basis value/d1/d2 */
    int *iu0, *iv0;          /* [npts*2] */
    double *R;               /* [npts][2][3][MAXNB] */
    double *tau, *wt;        /* [npts*2], [npts] */
    double ad, ar;
} iface_t;

typedef struct gfo_model {
    int np;
    patch_t* P;
    int64_t total_cp, ndof;
    double *cp, *w, *h, *u;  /* cp: [total_cp*3] homogeneous; u: [ndof] */
    unsigned char* zero;     /* [ndof] Dirichlet mask */
    int64_t npl; int64_t* pl_dof; double* pl_val;
    int ni; iface_t* ifs;
    /* CP-level neighbour lists: shell only (dRdh) and shell+coupling (K, dRdCP) */
    int64_t *nb_ptr_s, *nb_ptr_c; int32_t *nb_s, *nb_c;
    /* optional quadrature rule of the shell terms given on the unit square (gfo_set_quadrature); nq = 0: (p+1) x (q+1) Gauss points per span */
    int nq; double *qx, *qy, *qw;
} gfo_model;

/* ------------------------------------------------------------------ basics */
static void gauss_legendre(int n, double* x, double* w) {
    for (int i = 0; i < n; ++i) {
        double z = cos(M_PI * (i + 0.75) / (n + 0.5)), pp = 0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1, p2 = 0;
            for (int j = 1; j <= n; ++j) { double p3 = p2; p2 = p1; p1 = ((2.0 * j - 1) * z * p2 - (j - 1.0) * p3) / j; }
            pp = n * (z * p1 - p2) / (z * z - 1);
            double dz = p1 / pp; z -= dz;
            if (fabs(dz) < 1e-16) break;
        }
        x[n - 1 - i] = z; w[n - 1 - i] = 2 / ((1 - z * z) * pp * pp);
    }
}

static int find_span(int n, int p, const double* U, double xi) {
    /* n = number of basis functions; returns i with U[i] <= xi < U[i+1] (last non-empty span at the right end) */
    if (xi >= U[n]) { int i = n - 1; while (i > p && U[i] >= U[i + 1]) --i; return i; }
    if (xi <= U[p]) { int i = p; while (i < n - 1 && U[i] >= U[i + 1]) ++i; return i; }
    int lo = p, hi = n, mid = (lo + hi) / 2;
    while (xi < U[mid] || xi >= U[mid + 1]) { if (xi < U[mid]) hi = mid; else lo = mid; mid = (lo + hi) / 2; }
    return mid;
}

/* The NURBS Book A2.3: ders[k][j] = k-th derivative of N_{span-p+j,p} at xi, k = 0..2 */
static void basis_ders(int span, double xi, int p, const double* U, double ders[3][MAXP + 1]) {
    double ndu[MAXP + 1][MAXP + 1], left[MAXP + 1], right[MAXP + 1], a[2][MAXP + 1];
    ndu[0][0] = 1;
    for (int j = 1; j <= p; ++j) {
        left[j] = xi - U[span + 1 - j]; right[j] = U[span + j] - xi;
        double saved = 0;
        for (int r = 0; r < j; ++r) {
            ndu[j][r] = right[r + 1] + left[j - r];
            double temp = ndu[r][j - 1] / ndu[j][r];
            ndu[r][j] = saved + right[r + 1] * temp; saved = left[j - r] * temp;
        }
        ndu[j][j] = saved;
    }
    for (int j = 0; j <= p; ++j) ders[0][j] = ndu[j][p];
    for (int k = 1; k <= 2; ++k) for (int j = 0; j <= p; ++j) ders[k][j] = 0;
    int nd = p < 2 ? p : 2;
    for (int r = 0; r <= p; ++r) {
        int s1 = 0, s2 = 1; a[0][0] = 1;
        for (int k = 1; k <= nd; ++k) {
            double d = 0; int rk = r - k, pk = p - k;
            if (r >= k) { a[s2][0] = a[s1][0] / ndu[pk + 1][rk]; d = a[s2][0] * ndu[rk][pk]; }
            int j1 = rk >= -1 ? 1 : -rk, j2 = (r - 1 <= pk) ? k - 1 : p - r;
            for (int j = j1; j <= j2; ++j) { a[s2][j] = (a[s1][j] - a[s1][j - 1]) / ndu[pk + 1][rk + j]; d += a[s2][j] * ndu[rk + j][pk]; }
            if (r <= pk) { a[s2][k] = -a[s1][k - 1] / ndu[pk + 1][r]; d += a[s2][k] * ndu[r][pk]; }
            ders[k][r] = d; int t = s1; s1 = s2; s2 = t;
        }
    }
    double r = p;
    for (int k = 1; k <= nd; ++k) { for (int j = 0; j <= p; ++j) ders[k][j] *= r; r *= (p - k); }
}

static inline void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* Rational basis N_a/W and its parametric derivatives from tensor-product B-spline values.
 * in : Nb[6][nb] = (N, N_1, N_2, N_11, N_22, N_12), wl[nb] local weights
 * out: Rb[6][nb] same order for R~_a = N_a / W  (tIGAr "rationalize") */
static void rationalize(int nb, double Nb[6][MAXNB], const double* wl, double Rb[6][MAXNB]) {
    double W[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 6; ++k) for (int a = 0; a < nb; ++a) W[k] += Nb[k][a] * wl[a];
    double iW = 1.0 / W[0];
    for (int a = 0; a < nb; ++a) {
        double R = Nb[0][a] * iW;
        double R1 = (Nb[1][a] - R * W[1]) * iW, R2 = (Nb[2][a] - R * W[2]) * iW;
        Rb[0][a] = R; Rb[1][a] = R1; Rb[2][a] = R2;
        Rb[3][a] = (Nb[3][a] - 2 * R1 * W[1] - R * W[3]) * iW;
        Rb[4][a] = (Nb[4][a] - 2 * R2 * W[2] - R * W[4]) * iW;
        Rb[5][a] = (Nb[5][a] - R1 * W[2] - R2 * W[1] - R * W[5]) * iW;
    }
}

/* ------------------------------------------------------------------ model */
static void patch_tables(patch_t* P) {
    for (int d = 0; d < 2; ++d) {
        int p = d ? P->q : P->p, n = d ? P->nv : P->nu; const double* U = d ? P->kv : P->ku;
        int nel = 0; int* sp = (int*)malloc(sizeof(int) * (n + 1));
        for (int i = p; i < n; ++i) if (U[i + 1] > U[i]) sp[nel++] = i;
        int ng = p + 1; double gx[MAXP + 2], gw[MAXP + 2]; gauss_legendre(ng, gx, gw);
        double* tb = (double*)malloc(sizeof(double) * nel * ng * 3 * (p + 1));
        double* tw = (double*)malloc(sizeof(double) * nel * ng);
        for (int e = 0; e < nel; ++e) for (int g = 0; g < ng; ++g) {
            double a = U[sp[e]], b = U[sp[e] + 1], xi = 0.5 * (a + b) + 0.5 * (b - a) * gx[g];
            double ders[3][MAXP + 1]; basis_ders(sp[e], xi, p, U, ders);
            for (int k = 0; k < 3; ++k) for (int j = 0; j <= p; ++j) tb[((e * ng + g) * 3 + k) * (p + 1) + j] = ders[k][j];
            tw[e * ng + g] = 0.5 * (b - a) * gw[g];
        }
        if (d) { P->nelv = nel; P->spanv = sp; P->ngv = ng; P->bv = tb; P->wv = tw; }
        else   { P->nelu = nel; P->spanu = sp; P->ngu = ng; P->bu = tb; P->wu = tw; }
    }
}

static int cmp_i64(const void* a, const void* b) { int64_t x = *(const int64_t*)a, y = *(const int64_t*)b; return (x > y) - (x < y); }

static void build_nbrs(gfo_model* M, int with_coupling, int64_t** ptr_out, int32_t** nb_out) {
    /* collect (a,b) pairs as keys a*total_cp+b, sort, unique */
    int64_t cap = 0;
    for (int s = 0; s < M->np; ++s) { patch_t* P = &M->P[s]; cap += (int64_t)P->nelu * P->nelv * (P->p + 1) * (P->q + 1) * (P->p + 1) * (P->q + 1); }
    if (with_coupling) for (int i = 0; i < M->ni; ++i) {
        int na = (M->P[M->ifs[i].pa].p + 1) * (M->P[M->ifs[i].pa].q + 1), nb = (M->P[M->ifs[i].pb].p + 1) * (M->P[M->ifs[i].pb].q + 1);
        cap += M->ifs[i].npts * (int64_t)(na + nb) * (na + nb);
    }
    int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * (cap > 0 ? cap : 1)); int64_t nk = 0, T = M->total_cp;
    for (int s = 0; s < M->np; ++s) {
        patch_t* P = &M->P[s];
        for (int ev = 0; ev < P->nelv; ++ev) for (int eu = 0; eu < P->nelu; ++eu) {
            int iu0 = P->spanu[eu] - P->p, iv0 = P->spanv[ev] - P->q;
            for (int jv = 0; jv <= P->q; ++jv) for (int ju = 0; ju <= P->p; ++ju) {
                int64_t a = P->cp_off + (iu0 + ju) + (int64_t)(iv0 + jv) * P->nu;
                for (int kv = 0; kv <= P->q; ++kv) for (int ku = 0; ku <= P->p; ++ku)
                    keys[nk++] = a * T + P->cp_off + (iu0 + ku) + (int64_t)(iv0 + kv) * P->nu;
            }
        }
    }
    if (with_coupling) for (int i = 0; i < M->ni; ++i) {
        iface_t* F = &M->ifs[i]; int64_t loc[2 * MAXNB];
        for (int64_t v = 0; v < F->npts; ++v) {
            int n = 0;
            for (int sd = 0; sd < 2; ++sd) {
                patch_t* P = &M->P[sd ? F->pb : F->pa];
                for (int jv = 0; jv <= P->q; ++jv) for (int ju = 0; ju <= P->p; ++ju)
                    loc[n++] = P->cp_off + (F->iu0[2 * v + sd] + ju) + (int64_t)(F->iv0[2 * v + sd] + jv) * P->nu;
            }
            for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) keys[nk++] = loc[a] * T + loc[b];
        }
    }
    qsort(keys, nk, sizeof(int64_t), cmp_i64);
    int64_t nu_ = 0; for (int64_t k = 0; k < nk; ++k) if (k == 0 || keys[k] != keys[k - 1]) keys[nu_++] = keys[k];
    int64_t* ptr = (int64_t*)calloc(T + 1, sizeof(int64_t)); int32_t* nb = (int32_t*)malloc(sizeof(int32_t) * (nu_ > 0 ? nu_ : 1));
    for (int64_t k = 0; k < nu_; ++k) { ptr[keys[k] / T + 1]++; nb[k] = (int32_t)(keys[k] % T); }
    for (int64_t a = 0; a < T; ++a) ptr[a + 1] += ptr[a];
    free(keys); *ptr_out = ptr; *nb_out = nb;
}

gfo_model* gfo_create(const gf_model_desc* D) {
    gfo_model* M = (gfo_model*)calloc(1, sizeof(gfo_model));
    M->np = D->n_patches; M->P = (patch_t*)calloc(M->np, sizeof(patch_t));
    M->total_cp = D->cp_off[M->np]; M->ndof = 3 * M->total_cp;
    for (int s = 0; s < M->np; ++s) {
        patch_t* P = &M->P[s];
        P->p = D->degree[2 * s]; P->q = D->degree[2 * s + 1]; P->nu = D->ncp[2 * s]; P->nv = D->ncp[2 * s + 1];
        if (P->p > MAXP || P->q > MAXP || P->p < 1 || P->q < 1) { fprintf(stderr, "kl_oracle: degree out of range\n"); return NULL; }
        int64_t o0 = D->knot_off[2 * s], o1 = D->knot_off[2 * s + 1], o2 = D->knot_off[2 * s + 2];
        P->ku = (double*)malloc(sizeof(double) * (o1 - o0)); memcpy(P->ku, D->knots + o0, sizeof(double) * (o1 - o0));
        P->kv = (double*)malloc(sizeof(double) * (o2 - o1)); memcpy(P->kv, D->knots + o1, sizeof(double) * (o2 - o1));
        P->cp_off = D->cp_off[s]; P->E = D->young[s]; P->nu_ = D->poisson[s];
        for (int k = 0; k < 3; ++k) { P->f[k] = D->body_force ? D->body_force[3 * s + k] : 0.0; P->pd[k] = D->load_proj ? D->load_proj[3 * s + k] : 0.0; }
        P->press = D->pressure ? D->pressure[s] : 0.0;
        for (int e = 0; e < 4; ++e) for (int k = 0; k < 3; ++k) P->et[e][k] = D->edge_traction ? D->edge_traction[12 * s + 3 * e + k] : 0.0;
        patch_tables(P);
    }
    M->cp = (double*)calloc(M->ndof, sizeof(double)); M->u = (double*)calloc(M->ndof, sizeof(double));
    M->h = (double*)calloc(M->total_cp, sizeof(double)); M->w = (double*)malloc(sizeof(double) * M->total_cp);
    memcpy(M->w, D->weights, sizeof(double) * M->total_cp);
    M->zero = (unsigned char*)calloc(M->ndof, 1);
    for (int64_t k = 0; k < D->n_zero_dofs; ++k) M->zero[D->zero_dofs[k]] = 1;
    M->npl = D->n_point_loads; M->pl_dof = (int64_t*)malloc(sizeof(int64_t) * (M->npl + 1)); M->pl_val = (double*)malloc(sizeof(double) * (M->npl + 1));
    for (int64_t k = 0; k < M->npl; ++k) { M->pl_dof[k] = D->pl_dof[k]; M->pl_val[k] = D->pl_val[k]; }
    M->ni = D->n_interfaces; M->ifs = (iface_t*)calloc(M->ni > 0 ? M->ni : 1, sizeof(iface_t));
    for (int i = 0; i < M->ni; ++i) {
        iface_t* F = &M->ifs[i]; F->pa = D->if_patch[2 * i]; F->pb = D->if_patch[2 * i + 1];
        int64_t o = D->if_off[i]; F->npts = D->if_off[i + 1] - o;
        F->iu0 = (int*)malloc(sizeof(int) * 2 * F->npts); F->iv0 = (int*)malloc(sizeof(int) * 2 * F->npts);
        F->R = (double*)calloc((size_t)F->npts * 2 * 3 * MAXNB, sizeof(double));
        F->tau = (double*)malloc(sizeof(double) * 2 * F->npts); F->wt = (double*)malloc(sizeof(double) * F->npts);
        F->ad = D->if_alpha[2 * i]; F->ar = D->if_alpha[2 * i + 1];
        for (int64_t v = 0; v < F->npts; ++v) {
            F->tau[2 * v] = D->if_tau[2 * (o + v)]; F->tau[2 * v + 1] = D->if_tau[2 * (o + v) + 1]; F->wt[v] = D->if_wt[o + v];
            for (int sd = 0; sd < 2; ++sd) {
                patch_t* P = &M->P[sd ? F->pb : F->pa];
                double xu = D->if_xi[4 * (o + v) + 2 * sd], xv = D->if_xi[4 * (o + v) + 2 * sd + 1];
                int su = find_span(P->nu, P->p, P->ku, xu), sv = find_span(P->nv, P->q, P->kv, xv);
                double du[3][MAXP + 1], dv[3][MAXP + 1]; basis_ders(su, xu, P->p, P->ku, du); basis_ders(sv, xv, P->q, P->kv, dv);
                int nb = (P->p + 1) * (P->q + 1); double Nb[6][MAXNB], Rb[6][MAXNB], wl[MAXNB];
                for (int jv = 0; jv <= P->q; ++jv) for (int ju = 0; ju <= P->p; ++ju) {
                    int a = ju + jv * (P->p + 1);
                    Nb[0][a] = du[0][ju] * dv[0][jv]; Nb[1][a] = du[1][ju] * dv[0][jv]; Nb[2][a] = du[0][ju] * dv[1][jv];
                    Nb[3][a] = du[2][ju] * dv[0][jv]; Nb[4][a] = du[0][ju] * dv[2][jv]; Nb[5][a] = du[1][ju] * dv[1][jv];
                    wl[a] = M->w[P->cp_off + (su - P->p + ju) + (int64_t)(sv - P->q + jv) * P->nu];
                }
                rationalize(nb, Nb, wl, Rb);
                F->iu0[2 * v + sd] = su - P->p; F->iv0[2 * v + sd] = sv - P->q;
                for (int k = 0; k < 3; ++k) for (int a = 0; a < nb; ++a) F->R[((v * 2 + sd) * 3 + k) * MAXNB + a] = Rb[k][a];
            }
        }
    }
    build_nbrs(M, 0, &M->nb_ptr_s, &M->nb_s);
    build_nbrs(M, 1, &M->nb_ptr_c, &M->nb_c);
    return M;
}

void gfo_destroy(gfo_model* M) {
    if (!M) return;
    for (int s = 0; s < M->np; ++s) { patch_t* P = &M->P[s]; free(P->ku); free(P->kv); free(P->spanu); free(P->spanv); free(P->bu); free(P->bv); free(P->wu); free(P->wv); }
    for (int i = 0; i < M->ni; ++i) { iface_t* F = &M->ifs[i]; free(F->iu0); free(F->iv0); free(F->R); free(F->tau); free(F->wt); }
    free(M->P); free(M->ifs); free(M->cp); free(M->u); free(M->h); free(M->w); free(M->zero); free(M->pl_dof); free(M->pl_val);
    free(M->nb_ptr_s); free(M->nb_ptr_c); free(M->nb_s); free(M->nb_c); free(M->qx); free(M->qy); free(M->qw); free(M);
}

/* Quadrature rule of the shell integrals as n points (x, y, weight) on the unit square, mapped to every non-empty knot span
 * (n = 0: back to the tensor Gauss rule).  Used by tests/test_quadrature_gap.py to integrate with the rule FEniCS applies in the
 * reference -- two triangles per span, collapsed Gauss-Jacobi of degree quad_deg (GOLDFISH/tests/test_tbeam.py:31 quad_deg = 3p,
 * test_slr.py:37 2p, demos_om/thickness_opt/plate/plate_const_th_opt_wint.py:139-144 4p) -- and so bound the distance between the
 * tensor-Gauss assembly of this oracle / the HIP kernels and what the reference integrates.  Penalty terms (vertex quadrature) do not change. */
void gfo_set_quadrature(gfo_model* M, int n, const double* x, const double* y, const double* w) {
    free(M->qx); free(M->qy); free(M->qw); M->qx = M->qy = M->qw = NULL; M->nq = 0;
    if (n <= 0) return;
    M->qx = (double*)malloc(sizeof(double) * n); M->qy = (double*)malloc(sizeof(double) * n); M->qw = (double*)malloc(sizeof(double) * n);
    memcpy(M->qx, x, sizeof(double) * n); memcpy(M->qy, y, sizeof(double) * n); memcpy(M->qw, w, sizeof(double) * n); M->nq = n;
}

int64_t gfo_total_cp(const gfo_model* M) { return M->total_cp; }
int64_t gfo_num_gauss_points(const gfo_model* M) {
    int64_t n = 0; for (int s = 0; s < M->np; ++s) n += (int64_t)M->P[s].nelu * M->P[s].nelv * M->P[s].ngu * M->P[s].ngv; return n;
}
int64_t gfo_num_mortar_points(const gfo_model* M) { int64_t n = 0; for (int i = 0; i < M->ni; ++i) n += M->ifs[i].npts; return n; }
void gfo_set_cp(gfo_model* M, int field, const double* v) { for (int64_t a = 0; a < M->total_cp; ++a) M->cp[3 * a + field] = v[a]; }
void gfo_set_thickness(gfo_model* M, const double* v) { memcpy(M->h, v, sizeof(double) * M->total_cp); }
void gfo_set_u(gfo_model* M, const double* v) { memcpy(M->u, v, sizeof(double) * M->ndof); }

/* ---- CSR patterns (rows = vector dofs) ---------------------------------- */
int64_t gfo_nnz(const gfo_model* M, int which) {
    if (which == GF_MAT_K) return 9 * M->nb_ptr_c[M->total_cp];
    if (which == GF_MAT_DRDH) return 3 * M->nb_ptr_s[M->total_cp];
    return 3 * M->nb_ptr_c[M->total_cp];
}
void gfo_pattern(const gfo_model* M, int which, int64_t* rowptr, int32_t* col) {
    const int64_t* ptr = which == GF_MAT_DRDH ? M->nb_ptr_s : M->nb_ptr_c; const int32_t* nb = which == GF_MAT_DRDH ? M->nb_s : M->nb_c;
    int bw = which == GF_MAT_K ? 3 : 1; int64_t pos = 0; rowptr[0] = 0;
    for (int64_t a = 0; a < M->total_cp; ++a) for (int i = 0; i < 3; ++i) {
        for (int64_t k = ptr[a]; k < ptr[a + 1]; ++k) for (int j = 0; j < bw; ++j) col[pos++] = nb[k] * bw + j;
        rowptr[3 * a + i + 1] = pos;
    }
}
static inline int64_t nb_find(const int64_t* ptr, const int32_t* nb, int64_t a, int64_t b) {
    int64_t lo = ptr[a], hi = ptr[a + 1] - 1;
    while (lo <= hi) { int64_t mid = (lo + hi) / 2; if (nb[mid] == b) return mid - ptr[a]; if (nb[mid] < b) lo = mid + 1; else hi = mid - 1; }
    return -1;
}
/* position of entry (row dof 3a+i, col block b [, comp j]) in the CSR value array */
static inline int64_t pos_K(const gfo_model* M, int64_t a, int i, int64_t b, int j) {
    int64_t k = nb_find(M->nb_ptr_c, M->nb_c, a, b), deg = M->nb_ptr_c[a + 1] - M->nb_ptr_c[a];
    return 9 * M->nb_ptr_c[a] + i * 3 * deg + 3 * k + j;
}
static inline int64_t pos_C(const gfo_model* M, int64_t a, int i, int64_t b) {
    int64_t k = nb_find(M->nb_ptr_c, M->nb_c, a, b), deg = M->nb_ptr_c[a + 1] - M->nb_ptr_c[a];
    return 3 * M->nb_ptr_c[a] + i * deg + k;
}
static inline int64_t pos_H(const gfo_model* M, int64_t a, int i, int64_t b) {
    int64_t k = nb_find(M->nb_ptr_s, M->nb_s, a, b), deg = M->nb_ptr_s[a + 1] - M->nb_ptr_s[a];
    return 3 * M->nb_ptr_s[a] + i * deg + k;
}

/* ------------------------------------------------------------ shell element */
typedef struct {
    double *R, *K, *C[3], *H;          /* global outputs (any may be NULL) */
    double *dWdu, *dWdcp[3], *dWdh, *dVdcp[3], *dVdh; double Wint, Vol;
} out_t;

/* material tensor in curvilinear Voigt components and its variation wrt the metric */
static void material(double A11, double A22, double A12, double E, double nu, double C[3][3], double* J, double c[3]) {
    double det = A11 * A22 - A12 * A12, c11 = A22 / det, c22 = A11 / det, c12 = -A12 / det, Eb = E / (1 - nu * nu);
    C[0][0] = Eb * c11 * c11; C[1][1] = Eb * c22 * c22; C[0][1] = C[1][0] = Eb * (nu * c11 * c22 + (1 - nu) * c12 * c12);
    C[0][2] = C[2][0] = Eb * c11 * c12; C[1][2] = C[2][1] = Eb * c22 * c12; C[2][2] = Eb * 0.5 * ((1 - nu) * c11 * c22 + (1 + nu) * c12 * c12);
    *J = sqrt(det); c[0] = c11; c[1] = c22; c[2] = c12;
}
static void material_var(const double c[3], double dA11, double dA22, double dA12, double E, double nu, double dC[3][3]) {
    double c11 = c[0], c22 = c[1], c12 = c[2], Eb = E / (1 - nu * nu);
    double d11 = -c11 * c11 * dA11 - c12 * c12 * dA22 - 2 * c11 * c12 * dA12;
    double d22 = -c12 * c12 * dA11 - c22 * c22 * dA22 - 2 * c12 * c22 * dA12;
    double d12 = -c11 * c12 * dA11 - c12 * c22 * dA22 - (c11 * c22 + c12 * c12) * dA12;
    dC[0][0] = Eb * 2 * c11 * d11; dC[1][1] = Eb * 2 * c22 * d22;
    dC[0][1] = dC[1][0] = Eb * (nu * (d11 * c22 + c11 * d22) + 2 * (1 - nu) * c12 * d12);
    dC[0][2] = dC[2][0] = Eb * (d11 * c12 + c11 * d12); dC[1][2] = dC[2][1] = Eb * (d22 * c12 + c22 * d12);
    dC[2][2] = Eb * 0.5 * ((1 - nu) * (d11 * c22 + c11 * d22) + 2 * (1 + nu) * c12 * d12);
}
static inline void mv3(double A[3][3], const double* x, double* y) { for (int i = 0; i < 3; ++i) y[i] = A[i][0] * x[0] + A[i][1] * x[1] + A[i][2] * x[2]; }

/* first variation of the configuration quantities wrt coefficient (a, comp i):
 * dm[3] metric Voigt, db[3] curvature Voigt (f_k h_k.n), dnt = d(g1 x g2), dj, dn */
typedef struct { double dm[3], db[3], dnt[3], dj, dn[3]; } var1_t;
static void first_var(const double g[5][3], const double n[3], double j, const double Ra[6], int i, var1_t* V) {
    double R1 = Ra[1], R2 = Ra[2];
    V->dm[0] = 2 * R1 * g[0][i]; V->dm[1] = 2 * R2 * g[1][i]; V->dm[2] = 2 * (R1 * g[1][i] + R2 * g[0][i]);
    double e[3] = {0, 0, 0}; e[i] = 1; double t1[3], t2[3];
    cross3(e, g[1], t1); cross3(g[0], e, t2);
    for (int k = 0; k < 3; ++k) V->dnt[k] = R1 * t1[k] + R2 * t2[k];
    V->dj = dot3(n, V->dnt);
    for (int k = 0; k < 3; ++k) V->dn[k] = (V->dnt[k] - n[k] * V->dj) / j;
    V->db[0] = Ra[3] * n[i] + dot3(g[2], V->dn);
    V->db[1] = Ra[4] * n[i] + dot3(g[3], V->dn);
    V->db[2] = 2 * (Ra[5] * n[i] + dot3(g[4], V->dn));
}

/* How the strains are EVALUATED (the quantities are the same): 0 = as differences of the metric / curvature coefficients of the deformed and the reference
 * configuration, eps = (a - A) / 2, kappa = B - b -- the arithmetic of the reference's UFL forms (ShNAPr's surfaceEnergyDensitySVK via tIGAr) and the default;
 * 1 = from the displacement derivatives d = x - X without cancellation: eps_ab = (A_a . d_b + d_a . A_b + d_a . d_b) / 2,
 * kappa_ab = -(H_ab . (n - N) + d_ab . n) with n - N = (delta - N s / (j + J)) / j, delta = A_1 x d_2 + d_1 x A_2 + d_1 x d_2, s = 2 Nt . delta + delta . delta.
 * Mode 0 loses eps_machine |A|^2 of absolute accuracy in the strain, i.e. a residual floor of about eps_machine E h |A|^2 |grad N| per entry, however small the
 * load; mode 1 is what the HIP kernels evaluate since round 5 (tests/test_strain_evaluation.py compares the two where mode 0 is well conditioned). */
static int g_strain_mode = 0;
void gfo_set_strain_mode(int mode) { g_strain_mode = mode; }
int gfo_get_strain_mode(void) { return g_strain_mode; }

static void shell_element(const gfo_model* M, const patch_t* P, int eu, int ev, int want_mats, int want_fun, out_t* O) {
    const int p = P->p, q = P->q, nb = (p + 1) * (q + 1), nd = 3 * nb;
    const int iu0 = P->spanu[eu] - p, iv0 = P->spanv[ev] - q;
    int64_t gid[MAXNB]; double c[MAXNB][3], d[MAXNB][3], ul[MAXNB][3], hl[MAXNB], wl[MAXNB];
    for (int jv = 0; jv <= q; ++jv) for (int ju = 0; ju <= p; ++ju) {
        int a = ju + jv * (p + 1); int64_t g = P->cp_off + (iu0 + ju) + (int64_t)(iv0 + jv) * P->nu; gid[a] = g;
        for (int k = 0; k < 3; ++k) { c[a][k] = M->cp[3 * g + k]; ul[a][k] = M->u[3 * g + k]; d[a][k] = c[a][k] + ul[a][k]; }
        hl[a] = M->h[g]; wl[a] = M->w[g];
    }
    double* Re = (double*)calloc(nd, sizeof(double));
    double *Ke = NULL, *Ce = NULL, *He = NULL;
    if (want_mats) { Ke = (double*)calloc((size_t)nd * nd, sizeof(double)); Ce = (double*)calloc((size_t)nd * nd, sizeof(double)); He = (double*)calloc((size_t)nd * nb, sizeof(double)); }
    double *gWc = NULL, *gWh = NULL, *gVc = NULL, *gVh = NULL;
    if (want_fun) { gWc = (double*)calloc(nd, sizeof(double)); gWh = (double*)calloc(nb, sizeof(double)); gVc = (double*)calloc(nd, sizeof(double)); gVh = (double*)calloc(nb, sizeof(double)); }
    var1_t* Vd = (var1_t*)malloc(sizeof(var1_t) * nd); var1_t* Vr = (var1_t*)malloc(sizeof(var1_t) * nd);
    const double E = P->E, nu = P->nu_, f3[3] = {1, 1, 2};

    const int ngp = M->nq > 0 ? M->nq : P->ngu * P->ngv;
    for (int gp = 0; gp < ngp; ++gp) {
        const double *tu, *tv; double wq, du_[3][MAXP + 1], dv_[3][MAXP + 1];
        if (M->nq > 0) {                                   /* rule given on the unit square: evaluate the 1-D bases at the mapped point */
            const double a0 = P->ku[P->spanu[eu]], a1 = P->ku[P->spanu[eu] + 1], b0 = P->kv[P->spanv[ev]], b1 = P->kv[P->spanv[ev] + 1];
            basis_ders(P->spanu[eu], a0 + (a1 - a0) * M->qx[gp], p, P->ku, du_); basis_ders(P->spanv[ev], b0 + (b1 - b0) * M->qy[gp], q, P->kv, dv_);
            tu = &du_[0][0]; tv = &dv_[0][0]; wq = M->qw[gp] * (a1 - a0) * (b1 - b0);
        } else {
            const int gu = gp % P->ngu, gv = gp / P->ngu;
            tu = P->bu + (size_t)((eu * P->ngu + gu) * 3) * (p + 1); tv = P->bv + (size_t)((ev * P->ngv + gv) * 3) * (q + 1);
            wq = P->wu[eu * P->ngu + gu] * P->wv[ev * P->ngv + gv];
        }
        const int su_ = M->nq > 0 ? MAXP + 1 : p + 1, sv_ = M->nq > 0 ? MAXP + 1 : q + 1;      /* row stride of the 1-D tables */
        double Nb[6][MAXNB], Rb[6][MAXNB];
        for (int jv = 0; jv <= q; ++jv) for (int ju = 0; ju <= p; ++ju) {
            int a = ju + jv * (p + 1);
            double u0 = tu[ju], u1 = tu[su_ + ju], u2 = tu[2 * su_ + ju], v0 = tv[jv], v1 = tv[sv_ + jv], v2 = tv[2 * sv_ + jv];
            Nb[0][a] = u0 * v0; Nb[1][a] = u1 * v0; Nb[2][a] = u0 * v1; Nb[3][a] = u2 * v0; Nb[4][a] = u0 * v2; Nb[5][a] = u1 * v1;
        }
        rationalize(nb, Nb, wl, Rb);
        double G[5][3] = {{0}}, g[5][3] = {{0}}, th = 0;
        for (int a = 0; a < nb; ++a) { th += Nb[0][a] * hl[a]; for (int m = 0; m < 5; ++m) for (int k = 0; k < 3; ++k) { G[m][k] += Rb[m + 1][a] * c[a][k]; g[m][k] += Rb[m + 1][a] * d[a][k]; } }
        /* reference and deformed configuration */
        double Nt[3], nt[3]; cross3(G[0], G[1], Nt); cross3(g[0], g[1], nt);
        double Jn = sqrt(dot3(Nt, Nt)), jn = sqrt(dot3(nt, nt)), Nn[3], n[3];
        for (int k = 0; k < 3; ++k) { Nn[k] = Nt[k] / Jn; n[k] = nt[k] / jn; }
        double A11 = dot3(G[0], G[0]), A22 = dot3(G[1], G[1]), A12 = dot3(G[0], G[1]);
        double C[3][3], J, cc[3]; material(A11, A22, A12, E, nu, C, &J, cc);
        double eps[3] = {0.5 * (dot3(g[0], g[0]) - A11), 0.5 * (dot3(g[1], g[1]) - A22), dot3(g[0], g[1]) - A12};
        double kap[3];
        for (int k = 0; k < 3; ++k) kap[k] = f3[k] * (dot3(G[2 + k], Nn) - dot3(g[2 + k], n));
        if (g_strain_mode == 1) {                          /* displacement-based evaluation: the same strains without the cancellation */
            double dg[5][3] = {{0}};
            for (int a = 0; a < nb; ++a) for (int m = 0; m < 5; ++m) for (int k = 0; k < 3; ++k) dg[m][k] += Rb[m + 1][a] * ul[a][k];
            eps[0] = dot3(G[0], dg[0]) + 0.5 * dot3(dg[0], dg[0]);
            eps[1] = dot3(G[1], dg[1]) + 0.5 * dot3(dg[1], dg[1]);
            eps[2] = dot3(G[0], dg[1]) + dot3(dg[0], G[1]) + dot3(dg[0], dg[1]);
            double t1[3], t2[3], t3_[3], dl[3], dn[3];
            cross3(G[0], dg[1], t1); cross3(dg[0], G[1], t2); cross3(dg[0], dg[1], t3_);
            for (int k = 0; k < 3; ++k) dl[k] = t1[k] + t2[k] + t3_[k];
            const double sdl = 2.0 * dot3(Nt, dl) + dot3(dl, dl);
            for (int k = 0; k < 3; ++k) dn[k] = (dl[k] - Nn[k] * sdl / (jn + Jn)) / jn;      /* n - N */
            for (int k = 0; k < 3; ++k) kap[k] = -f3[k] * (dot3(G[2 + k], dn) + dot3(dg[2 + k], n));
        }
        double Ceps[3], Ckap[3]; mv3(C, eps, Ceps); mv3(C, kap, Ckap);
        double t3 = th * th * th / 12.0, nv[3], mo[3];
        for (int k = 0; k < 3; ++k) { nv[k] = th * Ceps[k]; mo[k] = t3 * Ckap[k]; }
        double psi = 0.5 * th * dot3(eps, Ceps) + 0.5 * t3 * dot3(kap, Ckap);
        O->Wint += wq * J * psi; O->Vol += wq * J * th;

        for (int a = 0; a < nb; ++a) for (int i = 0; i < 3; ++i) {
            double Ra[6]; for (int k = 0; k < 6; ++k) Ra[k] = Rb[k][a];
            first_var(g, n, jn, Ra, i, &Vd[3 * a + i]); first_var(G, Nn, Jn, Ra, i, &Vr[3 * a + i]);
        }
        /* residual: dPsi/dU_r - distributed load (per unit area: J, or per unit projected area: pd . (G1 x G2)) */
        const int proj = P->pd[0] != 0 || P->pd[1] != 0 || P->pd[2] != 0;
        const double sload = proj ? dot3(P->pd, Nt) : J;
        double rint[3 * MAXNB];
        for (int r = 0; r < nd; ++r) {
            double de[3] = {0.5 * Vd[r].dm[0], 0.5 * Vd[r].dm[1], 0.5 * Vd[r].dm[2]};
            double dk[3] = {-Vd[r].db[0], -Vd[r].db[1], -Vd[r].db[2]};
            rint[r] = dot3(nv, de) + dot3(mo, dk);
            /* follower pressure p sqrt(det a / det A) a2 . z dA = p (g1 x g2) . z dxi (tube_shape_opt_wint.py:303-324): no J */
            Re[r] += wq * (J * rint[r] - sload * P->f[r % 3] * Rb[0][r / 3] - P->press * nt[r % 3] * Rb[0][r / 3]);
        }
        if (want_fun) {
            for (int s = 0; s < nd; ++s) {
                /* dW/dc_s = dW/dU_s (deformed path) + reference path */
                double dA[3] = {Vr[s].dm[0], Vr[s].dm[1], Vr[s].dm[2]}, dB[3] = {Vr[s].db[0], Vr[s].db[1], Vr[s].db[2]};
                double dC[3][3]; material_var(cc, dA[0], dA[1], 0.5 * dA[2], E, nu, dC);
                double dCe[3], dCk[3]; mv3(dC, eps, dCe); mv3(dC, kap, dCk);
                double de_ref[3] = {-0.5 * dA[0], -0.5 * dA[1], -0.5 * dA[2]};
                double dpsi = th * dot3(Ceps, de_ref) + 0.5 * th * dot3(eps, dCe) + t3 * dot3(Ckap, dB) + 0.5 * t3 * dot3(kap, dCk);
                gWc[s] += wq * (J * rint[s] + Vr[s].dj * psi + J * dpsi);
                gVc[s] += wq * Vr[s].dj * th;
            }
            for (int b = 0; b < nb; ++b) { gWh[b] += wq * J * Nb[0][b] * (0.5 * dot3(eps, Ceps) + 0.125 * th * th * dot3(kap, Ckap)); gVh[b] += wq * J * Nb[0][b]; }
        }
        if (!want_mats) continue;
        /* dR/dh: d(rint)/dt * N_b */
        for (int r = 0; r < nd; ++r) {
            double de[3] = {0.5 * Vd[r].dm[0], 0.5 * Vd[r].dm[1], 0.5 * Vd[r].dm[2]}, dk[3] = {-Vd[r].db[0], -Vd[r].db[1], -Vd[r].db[2]};
            double rh = dot3(Ceps, de) + 0.25 * th * th * dot3(Ckap, dk);
            for (int b = 0; b < nb; ++b) He[r * nb + b] += wq * J * Nb[0][b] * rh;
        }
        /* tangent (Kiendl 2009 second variations) and mixed reference derivative */
        for (int s = 0; s < nd; ++s) {
            const int b = s / 3, js = s % 3;
            double des[3] = {0.5 * Vd[s].dm[0], 0.5 * Vd[s].dm[1], 0.5 * Vd[s].dm[2]}, dks[3] = {-Vd[s].db[0], -Vd[s].db[1], -Vd[s].db[2]};
            double Cde[3], Cdk[3]; mv3(C, des, Cde); mv3(C, dks, Cdk);
            /* reference-path variations for column s */
            double dA[3] = {Vr[s].dm[0], Vr[s].dm[1], Vr[s].dm[2]}, dB[3] = {Vr[s].db[0], Vr[s].db[1], Vr[s].db[2]};
            double dC[3][3]; material_var(cc, dA[0], dA[1], 0.5 * dA[2], E, nu, dC);
            double dCe[3], dCk[3], Cder[3], Cdkr[3]; mv3(dC, eps, dCe); mv3(dC, kap, dCk);
            double de_ref[3] = {-0.5 * dA[0], -0.5 * dA[1], -0.5 * dA[2]}; mv3(C, de_ref, Cder); mv3(C, dB, Cdkr);
            double dnv[3], dmo[3]; for (int k = 0; k < 3; ++k) { dnv[k] = th * (dCe[k] + Cder[k]); dmo[k] = t3 * (dCk[k] + Cdkr[k]); }
            const double dJ = Vr[s].dj, dsl = proj ? dot3(P->pd, Vr[s].dnt) : dJ;
            for (int r = 0; r < nd; ++r) {
                const int a = r / 3, ir = r % 3;
                double der[3] = {0.5 * Vd[r].dm[0], 0.5 * Vd[r].dm[1], 0.5 * Vd[r].dm[2]}, dkr[3] = {-Vd[r].db[0], -Vd[r].db[1], -Vd[r].db[2]};
                /* second variations */
                double dde[3] = {0, 0, 0};
                if (ir == js) { dde[0] = Rb[1][a] * Rb[1][b]; dde[1] = Rb[2][a] * Rb[2][b]; dde[2] = Rb[1][a] * Rb[2][b] + Rb[2][a] * Rb[1][b]; }
                double sk = Rb[1][a] * Rb[2][b] - Rb[2][a] * Rb[1][b], ei[3] = {0, 0, 0}, ej[3] = {0, 0, 0}, ddnt[3];
                ei[ir] = 1; ej[js] = 1; cross3(ei, ej, ddnt); for (int k = 0; k < 3; ++k) ddnt[k] *= sk;
                double ddj = dot3(Vd[s].dn, Vd[r].dnt) + dot3(n, ddnt), ddn[3];
                for (int k = 0; k < 3; ++k) ddn[k] = (ddnt[k] - Vd[s].dn[k] * Vd[r].dj - Vd[r].dn[k] * Vd[s].dj - n[k] * ddj) / jn;
                double ddk[3];
                for (int k = 0; k < 3; ++k) ddk[k] = -f3[k] * (Rb[3 + k][a] * Vd[s].dn[ir] + Rb[3 + k][b] * Vd[r].dn[js] + dot3(g[2 + k], ddn));
                double krs = th * dot3(der, Cde) + t3 * dot3(dkr, Cdk) + dot3(nv, dde) + dot3(mo, ddk);
                /* load stiffness of the follower pressure: d(g1 x g2)/dU_s = d(g1 x g2)/dc_s (the deformed tangents see c + U) */
                const double pk = -P->press * Rb[0][a] * Vd[s].dnt[ir];
                Ke[r * nd + s] += wq * (J * krs + pk);
                double phi21 = dJ * rint[r] + J * (dot3(dnv, der) + dot3(dmo, dkr)) - dsl * P->f[ir] * Rb[0][a];
                Ce[r * nd + s] += wq * (J * krs + pk + phi21);
            }
        }
    }
    /* scatter */
    for (int r = 0; r < nd; ++r) {
        int64_t a = gid[r / 3]; int i = r % 3;
        if (O->R) O->R[3 * a + i] += Re[r];
    }
    if (want_mats) for (int r = 0; r < nd; ++r) {
        int64_t a = gid[r / 3]; int i = r % 3;
        for (int s = 0; s < nd; ++s) {
            int64_t b = gid[s / 3]; int j = s % 3;
            if (O->K) O->K[pos_K(M, a, i, b, j)] += Ke[r * nd + s];
            if (O->C[j]) O->C[j][pos_C(M, a, i, b)] += Ce[r * nd + s];
        }
        if (O->H) for (int b = 0; b < nb; ++b) O->H[pos_H(M, a, i, gid[b])] += He[r * nb + b];
    }
    if (want_fun) for (int a = 0; a < nb; ++a) {
        for (int k = 0; k < 3; ++k) { if (O->dWdcp[k]) O->dWdcp[k][gid[a]] += gWc[3 * a + k]; if (O->dVdcp[k]) O->dVdcp[k][gid[a]] += gVc[3 * a + k]; }
        if (O->dWdh) O->dWdh[gid[a]] += gWh[a]; if (O->dVdh) O->dVdh[gid[a]] += gVh[a];
    }
    free(Re); free(Ke); free(Ce); free(He); free(gWc); free(gWh); free(gVc); free(gVh); free(Vd); free(Vr);
}

/* ------------------------------------------------------------------ penalty */
typedef double complex cplx;
static inline void ccross(const cplx* a, const cplx* b, cplx* c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; }
static inline cplx cdot(const cplx* a, const cplx* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; } /* NOT conjugated: analytic continuation */
static void cunit_normal(const cplx* g1, const cplx* g2, cplx* n, cplx* j) { cplx t[3]; ccross(g1, g2, t); *j = csqrt(cdot(t, t)); for (int k = 0; k < 3; ++k) n[k] = t[k] / *j; }
/* gradient of (n(g1,g2) . v) wrt g1, g2 with v held fixed */
static void normal_pullback(const cplx* g1, const cplx* g2, const cplx* n, cplx j, const cplx* v, cplx* o1, cplx* o2) {
    cplx nv = cdot(n, v), w[3]; for (int k = 0; k < 3; ++k) w[k] = (v[k] - n[k] * nv) / j;
    ccross(g2, w, o1); ccross(w, g1, o2);
}
/* U = X / |X|, dU = (X + dx) / |X + dx| - U without cancellation (analytic in X, dx) */
static void cunit_diff(const cplx* X, const cplx* dx, cplx* U, cplx* dU) {
    cplx L = csqrt(cdot(X, X)), xd[3] = {X[0] + dx[0], X[1] + dx[1], X[2] + dx[2]};
    cplx l = csqrt(cdot(xd, xd)), f = (2.0 * cdot(X, dx) + cdot(dx, dx)) / (l + L);
    for (int k = 0; k < 3; ++k) { U[k] = X[k] / L; dU[k] = (dx[k] - U[k] * f) / l; }
}
/* strain mode 1: the rotation measures e1 = nA.nB - NA.NB, e2 = (at x nA).nB - (At x NA).NB from the displacement tangents dY = y_tangents - Y (real parts given
 * separately, summed from the displacement coefficients; imaginary parts = the complex-step seeds of y and Y, which subtract exactly) */
static void pen_rot_measures_c(const cplx y[18], const cplx Y[12], const double* dYre, const double tau[2], cplx* e1, cplx* e2) {
    static const int tan_[12] = {3, 4, 5, 6, 7, 8, 12, 13, 14, 15, 16, 17};
    cplx dY[12], N[2][3], dn[2][3];
    for (int k = 0; k < 12; ++k) dY[k] = dYre[k] + I * (cimag(y[tan_[k]]) - cimag(Y[k]));
    for (int sd = 0; sd < 2; ++sd) {
        const cplx *G1 = Y + 6 * sd, *G2 = G1 + 3, *d1 = dY + 6 * sd, *d2 = d1 + 3;
        cplx Nt[3], a[3], b[3], c[3], dl[3];
        ccross(G1, G2, Nt); ccross(G1, d2, a); ccross(d1, G2, b); ccross(d1, d2, c);
        for (int k = 0; k < 3; ++k) dl[k] = a[k] + b[k] + c[k];
        cunit_diff(Nt, dl, N[sd], dn[sd]);
    }
    cplx tr[3], dt_[3], At[3], dat[3];
    for (int k = 0; k < 3; ++k) { tr[k] = tau[0] * Y[k] + tau[1] * Y[3 + k]; dt_[k] = tau[0] * dY[k] + tau[1] * dY[3 + k]; }
    cunit_diff(tr, dt_, At, dat);
    *e1 = cdot(N[0], dn[1]) + cdot(dn[0], N[1]) + cdot(dn[0], dn[1]);
    cplx An[3], x0[3], x1[3], x2[3], dan[3];
    ccross(At, N[0], An); ccross(At, dn[0], x0); ccross(dat, N[0], x1); ccross(dat, dn[0], x2);
    for (int k = 0; k < 3; ++k) dan[k] = x0[k] + x1[k] + x2[k];
    *e2 = cdot(An, dn[1]) + cdot(dan, N[1]) + cdot(dan, dn[1]);
}
/* y = (uA, gA1, gA2, uB, gB1, gB2), Y = (GA1, GA2, GB1, GB2); dYre: NULL, or (strain mode 1) the displacement tangents; returns energy, fills gradient wrt y */
static cplx pen_grad(const cplx y[18], const cplx Y[12], const double* dYre, const double tau[2], double ad, double ar, double dt, cplx gr[18]) {
    const cplx *uA = y, *gA1 = y + 3, *gA2 = y + 6, *uB = y + 9, *gB1 = y + 12, *gB2 = y + 15;
    const cplx *GA1 = Y, *GA2 = Y + 3, *GB1 = Y + 6, *GB2 = Y + 9;
    cplx tref[3], tdef[3], At[3], at[3];
    for (int k = 0; k < 3; ++k) { tref[k] = tau[0] * GA1[k] + tau[1] * GA2[k]; tdef[k] = tau[0] * gA1[k] + tau[1] * gA2[k]; }
    cplx L = csqrt(cdot(tref, tref)), lt = csqrt(cdot(tdef, tdef));
    for (int k = 0; k < 3; ++k) { At[k] = tref[k] / L; at[k] = tdef[k] / lt; }
    cplx nA[3], nB[3], NA[3], NB[3], jA, jB, JA, JB;
    cunit_normal(gA1, gA2, nA, &jA); cunit_normal(gB1, gB2, nB, &jB); cunit_normal(GA1, GA2, NA, &JA); cunit_normal(GB1, GB2, NB, &JB);
    cplx an[3], An[3]; ccross(at, nA, an); ccross(At, NA, An);
    cplx e1 = cdot(nA, nB) - cdot(NA, NB), e2 = cdot(an, nB) - cdot(An, NB);
    if (dYre && g_strain_mode == 1) pen_rot_measures_c(y, Y, dYre, tau, &e1, &e2);
    cplx dd[3] = {uA[0] - uB[0], uA[1] - uB[1], uA[2] - uB[2]}, c0 = dt * L;
    cplx en = c0 * (0.5 * ad * cdot(dd, dd) + 0.5 * ar * (e1 * e1 + e2 * e2));
    if (!gr) return en;
    for (int k = 0; k < 3; ++k) { gr[k] = c0 * ad * dd[k]; gr[9 + k] = -c0 * ad * dd[k]; }
    /* s1 = nA.nB */
    cplx s1A1[3], s1A2[3], s1B1[3], s1B2[3];
    normal_pullback(gA1, gA2, nA, jA, nB, s1A1, s1A2); normal_pullback(gB1, gB2, nB, jB, nA, s1B1, s1B2);
    /* s2 = at . (nA x nB) = nA . (nB x at) = nB . (at x nA) */
    cplx cx[3], v1[3], v2[3], s2A1[3], s2A2[3], s2B1[3], s2B2[3];
    ccross(nA, nB, cx); ccross(nB, at, v1); ccross(at, nA, v2);
    normal_pullback(gA1, gA2, nA, jA, v1, s2A1, s2A2); normal_pullback(gB1, gB2, nB, jB, v2, s2B1, s2B2);
    cplx ac = cdot(at, cx);
    for (int k = 0; k < 3; ++k) { cplx ptc = (cx[k] - at[k] * ac) / lt; s2A1[k] += tau[0] * ptc; s2A2[k] += tau[1] * ptc; }
    for (int k = 0; k < 3; ++k) {
        gr[3 + k] = c0 * ar * (e1 * s1A1[k] + e2 * s2A1[k]); gr[6 + k] = c0 * ar * (e1 * s1A2[k] + e2 * s2A2[k]);
        gr[12 + k] = c0 * ar * (e1 * s1B1[k] + e2 * s2B1[k]); gr[15 + k] = c0 * ar * (e1 * s1B2[k] + e2 * s2B2[k]);
    }
    return en;
}
/* exported for tests: energy, gradient, Hessians of one mortar vertex */
static void penalty_point_impl(const double y[18], const double Y[12], const double* dY, const double tau[2], double ad, double ar, double dt,
                               double* energy, double grad[18], double Hyy[18 * 18], double HyY[18 * 12]) {
    cplx yc[18], Yc[12], g[18]; const double hstep = 1e-30;
    for (int k = 0; k < 18; ++k) yc[k] = y[k]; for (int k = 0; k < 12; ++k) Yc[k] = Y[k];
    cplx en = pen_grad(yc, Yc, dY, tau, ad, ar, dt, g);
    if (energy) *energy = creal(en); if (grad) for (int k = 0; k < 18; ++k) grad[k] = creal(g[k]);
    if (Hyy) for (int c = 0; c < 18; ++c) { yc[c] = y[c] + hstep * I; pen_grad(yc, Yc, dY, tau, ad, ar, dt, g); yc[c] = y[c]; for (int r = 0; r < 18; ++r) Hyy[r * 18 + c] = cimag(g[r]) / hstep; }
    if (HyY) for (int c = 0; c < 12; ++c) { Yc[c] = Y[c] + hstep * I; pen_grad(yc, Yc, dY, tau, ad, ar, dt, g); Yc[c] = Y[c]; for (int r = 0; r < 18; ++r) HyY[r * 12 + c] = cimag(g[r]) / hstep; }
}
void gfo_penalty_point(const double y[18], const double Y[12], const double tau[2], double ad, double ar, double dt,
                       double* energy, double grad[18], double Hyy[18 * 18], double HyY[18 * 12]) {
    penalty_point_impl(y, Y, NULL, tau, ad, ar, dt, energy, grad, Hyy, HyY);
}

static void penalty_all(const gfo_model* M, int want_mats, out_t* O, double* Wpen) {
    for (int ii = 0; ii < M->ni; ++ii) {
        const iface_t* F = &M->ifs[ii]; const patch_t* PP[2] = {&M->P[F->pa], &M->P[F->pb]};
        for (int64_t v = 0; v < F->npts; ++v) {
            int nbs[2]; int64_t gid[2][MAXNB]; const double* Rv[2][3];
            double y[18] = {0}, Y[12] = {0}, dY[12] = {0};
            for (int sd = 0; sd < 2; ++sd) {
                const patch_t* P = PP[sd]; nbs[sd] = (P->p + 1) * (P->q + 1);
                for (int k = 0; k < 3; ++k) Rv[sd][k] = F->R + ((v * 2 + sd) * 3 + k) * MAXNB;
                for (int jv = 0; jv <= P->q; ++jv) for (int ju = 0; ju <= P->p; ++ju) {
                    int a = ju + jv * (P->p + 1); int64_t g = P->cp_off + (F->iu0[2 * v + sd] + ju) + (int64_t)(F->iv0[2 * v + sd] + jv) * P->nu; gid[sd][a] = g;
                    for (int k = 0; k < 3; ++k) {
                        double cc = M->cp[3 * g + k], uu = M->u[3 * g + k];
                        y[9 * sd + k] += Rv[sd][0][a] * uu;
                        y[9 * sd + 3 + k] += Rv[sd][1][a] * (cc + uu); y[9 * sd + 6 + k] += Rv[sd][2][a] * (cc + uu);
                        Y[6 * sd + k] += Rv[sd][1][a] * cc; Y[6 * sd + 3 + k] += Rv[sd][2][a] * cc;
                        dY[6 * sd + k] += Rv[sd][1][a] * uu; dY[6 * sd + 3 + k] += Rv[sd][2][a] * uu;
                    }
                }
            }
            double en, gr[18], Hyy[18 * 18], HyY[18 * 12];
            penalty_point_impl(y, Y, dY, F->tau + 2 * v, F->ad, F->ar, F->wt[v], &en, gr, want_mats ? Hyy : NULL, want_mats ? HyY : NULL);
            if (Wpen) *Wpen += en;
            for (int sd = 0; sd < 2; ++sd) for (int a = 0; a < nbs[sd]; ++a) for (int i = 0; i < 3; ++i) {
                double r = 0; for (int m = 0; m < 3; ++m) r += Rv[sd][m][a] * gr[9 * sd + 3 * m + i];
                if (O->R) O->R[3 * gid[sd][a] + i] += r;
                if (!want_mats) continue;
                for (int td = 0; td < 2; ++td) for (int b = 0; b < nbs[td]; ++b) for (int j = 0; j < 3; ++j) {
                    double kk = 0, cs = 0;
                    for (int m = 0; m < 3; ++m) for (int mm = 0; mm < 3; ++mm) {
                        double hyy = Hyy[(9 * sd + 3 * m + i) * 18 + 9 * td + 3 * mm + j];
                        kk += Rv[sd][m][a] * hyy * Rv[td][mm][b];
                        if (mm > 0) cs += Rv[sd][m][a] * (hyy + HyY[(9 * sd + 3 * m + i) * 12 + 6 * td + 3 * (mm - 1) + j]) * Rv[td][mm][b];
                    }
                    if (O->K) O->K[pos_K(M, gid[sd][a], i, gid[td][b], j)] += kk;
                    if (O->C[j]) O->C[j][pos_C(M, gid[sd][a], i, gid[td][b])] += cs;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ edge tractions
 * dWext = f . z |dX/dt| dt on the patch edge xi_d = side (t = the other parameter), p + 1 Gauss points per edge span:
 * R_(a,i) -= w f_i R_a |X_t|,  dR_(a,i)/dc_(b,k) -= w f_i R_a (X_t,k / |X_t|) R_b,t  (the measure depends on the geometry; no K term) */
static void edge_loads(const gfo_model* M, out_t* O, int want_mats) {
    for (int s = 0; s < M->np; ++s) {
        const patch_t* P = &M->P[s];
        for (int e = 0; e < 4; ++e) {
            const double* f = P->et[e];
            if (f[0] == 0 && f[1] == 0 && f[2] == 0) continue;
            const int d = e / 2, side = e % 2, td = 1 - d;                       /* t runs along direction td */
            const int pt = td ? P->q : P->p, pd_ = d ? P->q : P->p, nelt = td ? P->nelv : P->nelu, ng = td ? P->ngv : P->ngu;
            const double* kd = d ? P->kv : P->ku; const int nd_ = d ? P->nv : P->nu;
            const double xfix = side ? kd[nd_] : kd[pd_];
            const int sd = find_span(nd_, pd_, kd, xfix);
            double dd[3][MAXP + 1]; basis_ders(sd, xfix, pd_, kd, dd);
            const int* spt = td ? P->spanv : P->spanu; const double* tabt = td ? P->bv : P->bu; const double* wt = td ? P->wv : P->wu;
            const int nb = (P->p + 1) * (P->q + 1);
            for (int et = 0; et < nelt; ++et) for (int g = 0; g < ng; ++g) {
                const double* tt = tabt + (size_t)((et * ng + g) * 3) * (pt + 1);
                const double wq = wt[et * ng + g];
                double Nb[6][MAXNB], Rb[6][MAXNB], wl[MAXNB], c[MAXNB][3]; int64_t gid[MAXNB];
                const int iu0 = (d == 0 ? sd : spt[et]) - P->p, iv0 = (d == 1 ? sd : spt[et]) - P->q;
                for (int jv = 0; jv <= P->q; ++jv) for (int ju = 0; ju <= P->p; ++ju) {
                    const int a = ju + jv * (P->p + 1);
                    const double u0 = d == 0 ? dd[0][ju] : tt[ju], u1 = d == 0 ? dd[1][ju] : tt[(pt + 1) + ju], u2 = d == 0 ? dd[2][ju] : tt[2 * (pt + 1) + ju];
                    const double v0 = d == 1 ? dd[0][jv] : tt[jv], v1 = d == 1 ? dd[1][jv] : tt[(pt + 1) + jv], v2 = d == 1 ? dd[2][jv] : tt[2 * (pt + 1) + jv];
                    Nb[0][a] = u0 * v0; Nb[1][a] = u1 * v0; Nb[2][a] = u0 * v1; Nb[3][a] = u2 * v0; Nb[4][a] = u0 * v2; Nb[5][a] = u1 * v1;
                    const int64_t gg = P->cp_off + (iu0 + ju) + (int64_t)(iv0 + jv) * P->nu; gid[a] = gg; wl[a] = M->w[gg];
                    for (int k = 0; k < 3; ++k) c[a][k] = M->cp[3 * gg + k];
                }
                rationalize(nb, Nb, wl, Rb);
                const double* Rt = Rb[1 + td];                                  /* derivative along the edge */
                double Xt[3] = {0, 0, 0};
                for (int a = 0; a < nb; ++a) for (int k = 0; k < 3; ++k) Xt[k] += Rt[a] * c[a][k];
                const double len = sqrt(dot3(Xt, Xt));
                for (int a = 0; a < nb; ++a) for (int i = 0; i < 3; ++i) {
                    if (O->R) O->R[3 * gid[a] + i] -= wq * f[i] * Rb[0][a] * len;
                    if (want_mats) for (int b = 0; b < nb; ++b) for (int k = 0; k < 3; ++k)
                        if (O->C[k] && Rb[0][a] != 0.0 && Rt[b] != 0.0) O->C[k][pos_C(M, gid[a], i, gid[b])] -= wq * f[i] * Rb[0][a] * Xt[k] / len * Rt[b];
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ drivers */
static void run_shell(const gfo_model* M, int want_mats, int want_fun, out_t* O) {
    /* patches own disjoint shell rows/entries -> parallel over patches is race-free */
    double Wint = 0, Vol = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : Wint, Vol)
    for (int s = 0; s < M->np; ++s) {
        out_t L = *O; L.Wint = 0; L.Vol = 0; const patch_t* P = &M->P[s];
        for (int ev = 0; ev < P->nelv; ++ev) for (int eu = 0; eu < P->nelu; ++eu) shell_element(M, P, eu, ev, want_mats, want_fun, &L);
        Wint += L.Wint; Vol += L.Vol;
    }
    O->Wint = Wint; O->Vol = Vol;
}

void gfo_residual(const gfo_model* M, double* R) {
    out_t O; memset(&O, 0, sizeof(O)); memset(R, 0, sizeof(double) * M->ndof); O.R = R;
    run_shell(M, 0, 0, &O); penalty_all(M, 0, &O, NULL); edge_loads(M, &O, 0);
    for (int64_t k = 0; k < M->npl; ++k) R[M->pl_dof[k]] -= M->pl_val[k];
    for (int64_t r = 0; r < M->ndof; ++r) if (M->zero[r]) R[r] = 0;
}

/* CSR value arrays in the gfo_pattern layouts; any pointer may be NULL */
void gfo_assemble(const gfo_model* M, double* K, double* C0, double* C1, double* C2, double* H) {
    out_t O; memset(&O, 0, sizeof(O)); O.K = K; O.C[0] = C0; O.C[1] = C1; O.C[2] = C2; O.H = H;
    if (K) memset(K, 0, sizeof(double) * gfo_nnz(M, GF_MAT_K));
    for (int f = 0; f < 3; ++f) if (O.C[f]) memset(O.C[f], 0, sizeof(double) * gfo_nnz(M, GF_MAT_DRDCP0));
    if (H) memset(H, 0, sizeof(double) * gfo_nnz(M, GF_MAT_DRDH));
    run_shell(M, 1, 0, &O); penalty_all(M, 1, &O, NULL); edge_loads(M, &O, 1);
    /* Dirichlet: K rows+cols zero, diag 1; dRdCP rows zero; dRdh untouched (nonmatching_opt.py:1012-1014) */
    for (int64_t a = 0; a < M->total_cp; ++a) {
        int64_t deg = M->nb_ptr_c[a + 1] - M->nb_ptr_c[a];
        for (int i = 0; i < 3; ++i) {
            int64_t row = 3 * a + i;
            if (K) for (int64_t k = 0; k < deg; ++k) for (int j = 0; j < 3; ++j) {
                int64_t col = 3 * (int64_t)M->nb_c[M->nb_ptr_c[a] + k] + j, pos = 9 * M->nb_ptr_c[a] + i * 3 * deg + 3 * k + j;
                if (M->zero[row] || M->zero[col]) K[pos] = (row == col) ? 1.0 : 0.0;
            }
            if (M->zero[row]) for (int f = 0; f < 3; ++f) if (O.C[f]) for (int64_t k = 0; k < deg; ++k) O.C[f][3 * M->nb_ptr_c[a] + i * deg + k] = 0.0;
        }
    }
}

/* functionals: out[0]=Wint (shell strain energy), out[1]=volume, out[2]=penalty energy.
 * dWdu has Dirichlet rows zeroed when apply_bcs != 0 (int_energy_exop.py:61-66). */
void gfo_functionals(const gfo_model* M, double out[3], double* dWdu, double* dWdcp0, double* dWdcp1, double* dWdcp2, double* dWdh,
                     double* dVdcp0, double* dVdcp1, double* dVdcp2, double* dVdh, int apply_bcs) {
    out_t O; memset(&O, 0, sizeof(O));
    double* Rtmp = (double*)calloc(M->ndof, sizeof(double)); O.R = Rtmp;
    O.dWdcp[0] = dWdcp0; O.dWdcp[1] = dWdcp1; O.dWdcp[2] = dWdcp2; O.dWdh = dWdh; O.dVdcp[0] = dVdcp0; O.dVdcp[1] = dVdcp1; O.dVdcp[2] = dVdcp2; O.dVdh = dVdh;
    for (int f = 0; f < 3; ++f) { if (O.dWdcp[f]) memset(O.dWdcp[f], 0, sizeof(double) * M->total_cp); if (O.dVdcp[f]) memset(O.dVdcp[f], 0, sizeof(double) * M->total_cp); }
    if (dWdh) memset(dWdh, 0, sizeof(double) * M->total_cp); if (dVdh) memset(dVdh, 0, sizeof(double) * M->total_cp);
    /* the loads must not enter dWint/du: temporarily zero them */
    gfo_model* MM = (gfo_model*)M; double (*fs)[4] = (double (*)[4])malloc(sizeof(double[4]) * M->np);
    for (int s = 0; s < M->np; ++s) { for (int k = 0; k < 3; ++k) { fs[s][k] = MM->P[s].f[k]; MM->P[s].f[k] = 0; } fs[s][3] = MM->P[s].press; MM->P[s].press = 0; }
    run_shell(M, 0, 1, &O);
    for (int s = 0; s < M->np; ++s) { for (int k = 0; k < 3; ++k) MM->P[s].f[k] = fs[s][k]; MM->P[s].press = fs[s][3]; }
    free(fs);
    double Wpen = 0; out_t O2; memset(&O2, 0, sizeof(O2)); penalty_all(M, 0, &O2, &Wpen);
    out[0] = O.Wint; out[1] = O.Vol; out[2] = Wpen;
    if (dWdu) for (int64_t r = 0; r < M->ndof; ++r) dWdu[r] = (apply_bcs && M->zero[r]) ? 0.0 : Rtmp[r];
    free(Rtmp);
}

void gfo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
/* ComplianceExOperation (operations/compliance_exop.py:21-48): C = sum_s int forces[s] . u_hom dA with the
 * homogeneous (non-rationalised) displacement function; dC/du, dC/dCP_f.  forces: [3*n_patches]. */
void gfo_compliance(const gfo_model* M, const double* forces, double* Cout, double* dCdu, double* dCdcp0, double* dCdcp1, double* dCdcp2, int apply_bcs) {
    double* dC[3] = {dCdcp0, dCdcp1, dCdcp2}; double Ctot = 0;
    if (dCdu) memset(dCdu, 0, sizeof(double) * M->ndof);
    for (int f = 0; f < 3; ++f) if (dC[f]) memset(dC[f], 0, sizeof(double) * M->total_cp);
    for (int s = 0; s < M->np; ++s) {
        const patch_t* P = &M->P[s]; const int p = P->p, q = P->q, nb = (p + 1) * (q + 1); const double* fs = forces + 3 * s;
        for (int ev = 0; ev < P->nelv; ++ev) for (int eu = 0; eu < P->nelu; ++eu) {
            const int iu0 = P->spanu[eu] - p, iv0 = P->spanv[ev] - q; int64_t gid[MAXNB]; double wl[MAXNB];
            for (int jv = 0; jv <= q; ++jv) for (int ju = 0; ju <= p; ++ju) { int a = ju + jv * (p + 1); gid[a] = P->cp_off + (iu0 + ju) + (int64_t)(iv0 + jv) * P->nu; wl[a] = M->w[gid[a]]; }
            for (int gv = 0; gv < P->ngv; ++gv) for (int gu = 0; gu < P->ngu; ++gu) {
                const double* tu = P->bu + (size_t)((eu * P->ngu + gu) * 3) * (p + 1); const double* tv = P->bv + (size_t)((ev * P->ngv + gv) * 3) * (q + 1);
                const double wq = P->wu[eu * P->ngu + gu] * P->wv[ev * P->ngv + gv];
                double Nb[6][MAXNB], Rb[6][MAXNB];
                for (int jv = 0; jv <= q; ++jv) for (int ju = 0; ju <= p; ++ju) {
                    int a = ju + jv * (p + 1);
                    double u0 = tu[ju], u1 = tu[(p + 1) + ju], u2 = tu[2 * (p + 1) + ju], v0 = tv[jv], v1 = tv[(q + 1) + jv], v2 = tv[2 * (q + 1) + jv];
                    Nb[0][a] = u0 * v0; Nb[1][a] = u1 * v0; Nb[2][a] = u0 * v1; Nb[3][a] = u2 * v0; Nb[4][a] = u0 * v2; Nb[5][a] = u1 * v1;
                }
                rationalize(nb, Nb, wl, Rb);
                double G1[3] = {0, 0, 0}, G2[3] = {0, 0, 0}, Uh[3] = {0, 0, 0};
                for (int a = 0; a < nb; ++a) for (int k = 0; k < 3; ++k) { G1[k] += Rb[1][a] * M->cp[3 * gid[a] + k]; G2[k] += Rb[2][a] * M->cp[3 * gid[a] + k]; Uh[k] += Nb[0][a] * M->u[3 * gid[a] + k]; }
                double Nt[3]; cross3(G1, G2, Nt); const double J = sqrt(dot3(Nt, Nt)); double Nn[3] = {Nt[0] / J, Nt[1] / J, Nt[2] / J}, J1[3], J2[3];
                cross3(G2, Nn, J1); cross3(Nn, G1, J2);
                const double fu = dot3(fs, Uh);
                Ctot += wq * J * fu;
                for (int a = 0; a < nb; ++a) for (int k = 0; k < 3; ++k) {
                    if (dCdu) dCdu[3 * gid[a] + k] += wq * J * fs[k] * Nb[0][a];
                    if (dC[k]) dC[k][gid[a]] += wq * fu * (J1[k] * Rb[1][a] + J2[k] * Rb[2][a]);
                }
            }
        }
    }
    if (dCdu && apply_bcs) for (int64_t r = 0; r < M->ndof; ++r) if (M->zero[r]) dCdu[r] = 0;
    *Cout = Ctot;
}

/* ---- stress aggregation forms ---------------------------------------------------------------
 * MaxvMStressExOperation (operations/max_vmstress_exop.py:167-186): per patch s the form
 *     I_s = int g(sigma_vM) dA,   g = exp(rho (sigma - m_s))  [mode 0, KS_symexp :167]
 *                                 g = (sigma / m_s)^rho       [mode 1, pnorm_symexp :170 / induced_power :173]
 * sigma_vM = ShellStressSVK(...).vonMisesStress(xi2) of PENGoLINS (not vendored) at xi2 = sgn * h/2 (:29-36),
 * restated in the classical local-Cartesian route: strain E = eps + xi2 kappa, orthonormal basis from A1, A2,
 * plane-stress SVK law, and (measure 0) push-forward to the Cauchy stress sigma = F S F^T / Jr or (measure 1) the
 * 2nd Piola-Kirchhoff stress itself.  Every derivative is a complex-step derivative of the integrand J g(sigma). */
static cplx vm_integrand(const cplx* z, const cplx* Z, cplx t, double E, double nu, double sgn, int measure, int mode, double rho, double ms, cplx* sig_out) {
    const cplx *A1 = Z, *A2 = Z + 3, *a1 = z, *a2 = z + 3;
    cplx N[3], n[3], Jn, jn; cunit_normal(A1, A2, N, &Jn); cunit_normal(a1, a2, n, &jn);
    cplx Ev[3];
    Ev[0] = 0.5 * (cdot(a1, a1) - cdot(A1, A1)); Ev[1] = 0.5 * (cdot(a2, a2) - cdot(A2, A2)); Ev[2] = 0.5 * (cdot(a1, a2) - cdot(A1, A2));
    const cplx xi = 0.5 * sgn * t;
    for (int k = 0; k < 3; ++k) Ev[k] += xi * (cdot(Z + 6 + 3 * k, N) - cdot(z + 6 + 3 * k, n));     /* (E11, E22, E12) tensor components */
    cplx A11 = cdot(A1, A1), A22 = cdot(A2, A2), A12 = cdot(A1, A2), det = A11 * A22 - A12 * A12;
    cplx Ac1[3], Ac2[3], e1[3], e2[3];
    for (int k = 0; k < 3; ++k) { Ac1[k] = (A22 * A1[k] - A12 * A2[k]) / det; Ac2[k] = (A11 * A2[k] - A12 * A1[k]) / det; }
    cplx l1 = csqrt(A11); for (int k = 0; k < 3; ++k) e1[k] = A1[k] / l1;
    cplx pr = cdot(A2, e1); for (int k = 0; k < 3; ++k) e2[k] = A2[k] - pr * e1[k];
    cplx l2 = csqrt(cdot(e2, e2)); for (int k = 0; k < 3; ++k) e2[k] /= l2;
    cplx T[2][2] = {{cdot(Ac1, e1), cdot(Ac1, e2)}, {cdot(Ac2, e1), cdot(Ac2, e2)}};                   /* T[a][i] = A^a . e_i */
    cplx Ec[2][2] = {{Ev[0], Ev[2]}, {Ev[2], Ev[1]}}, Eb[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) { Eb[i][j] = 0; for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) Eb[i][j] += T[a][i] * Ec[a][b] * T[b][j]; }
    const double D = E / (1 - nu * nu);
    cplx S11 = D * (Eb[0][0] + nu * Eb[1][1]), S22 = D * (Eb[1][1] + nu * Eb[0][0]), S12 = D * (1 - nu) * Eb[0][1], sig;
    if (measure == 1) sig = csqrt(S11 * S11 - S11 * S22 + S22 * S22 + 3 * S12 * S12);
    else {
        cplx F1[3], F2[3], s[3][3], tr = 0, tr2 = 0;
        for (int k = 0; k < 3; ++k) { F1[k] = a1[k] * T[0][0] + a2[k] * T[1][0]; F2[k] = a1[k] * T[0][1] + a2[k] * T[1][1]; }
        cplx Jr = jn / Jn;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) s[i][j] = (S11 * F1[i] * F1[j] + S22 * F2[i] * F2[j] + S12 * (F1[i] * F2[j] + F2[i] * F1[j])) / Jr;
        for (int i = 0; i < 3; ++i) { tr += s[i][i]; for (int j = 0; j < 3; ++j) tr2 += s[i][j] * s[j][i]; }
        sig = csqrt(1.5 * tr2 - 0.5 * tr * tr);
    }
    if (sig_out) *sig_out = sig;
    cplx g = mode == 0 ? cexp(rho * (sig - ms)) : cpow(sig / ms, rho);
    return Jn * g;
}

/* Is_out[np], vmax[np] (largest Gauss-point sigma_vM per patch), dIdu [ndof], dIdcp_f [total_cp], dIdh [total_cp]; pointers may be NULL */
void gfo_stress_forms(const gfo_model* M, int mode, double rho, const double* m_list, double sgn, int measure, double* Is_out, double* vmax,
                      double* dIdu, double* dIdcp0, double* dIdcp1, double* dIdcp2, double* dIdh, int apply_bcs) {
    double* dC[3] = {dIdcp0, dIdcp1, dIdcp2}; const double hs = 1e-30;
    if (dIdu) memset(dIdu, 0, sizeof(double) * M->ndof);
    for (int f = 0; f < 3; ++f) if (dC[f]) memset(dC[f], 0, sizeof(double) * M->total_cp);
    if (dIdh) memset(dIdh, 0, sizeof(double) * M->total_cp);
    const int want_grad = dIdu || dIdcp0 || dIdcp1 || dIdcp2 || dIdh;
    for (int s = 0; s < M->np; ++s) {
        const patch_t* P = &M->P[s]; const int p = P->p, q = P->q, nb = (p + 1) * (q + 1);
        double Is = 0, vm = 0;
        for (int ev = 0; ev < P->nelv; ++ev) for (int eu = 0; eu < P->nelu; ++eu) {
            const int iu0 = P->spanu[eu] - p, iv0 = P->spanv[ev] - q; int64_t gid[MAXNB]; double wl[MAXNB];
            for (int jv = 0; jv <= q; ++jv) for (int ju = 0; ju <= p; ++ju) { int a = ju + jv * (p + 1); gid[a] = P->cp_off + (iu0 + ju) + (int64_t)(iv0 + jv) * P->nu; wl[a] = M->w[gid[a]]; }
            for (int gv = 0; gv < P->ngv; ++gv) for (int gu = 0; gu < P->ngu; ++gu) {
                const double* tu = P->bu + (size_t)((eu * P->ngu + gu) * 3) * (p + 1); const double* tv = P->bv + (size_t)((ev * P->ngv + gv) * 3) * (q + 1);
                const double wq = P->wu[eu * P->ngu + gu] * P->wv[ev * P->ngv + gv];
                double Nb[6][MAXNB], Rb[6][MAXNB];
                for (int jv = 0; jv <= q; ++jv) for (int ju = 0; ju <= p; ++ju) {
                    int a = ju + jv * (p + 1);
                    double u0 = tu[ju], u1 = tu[(p + 1) + ju], u2 = tu[2 * (p + 1) + ju], v0 = tv[jv], v1 = tv[(q + 1) + jv], v2 = tv[2 * (q + 1) + jv];
                    Nb[0][a] = u0 * v0; Nb[1][a] = u1 * v0; Nb[2][a] = u0 * v1; Nb[3][a] = u2 * v0; Nb[4][a] = u0 * v2; Nb[5][a] = u1 * v1;
                }
                rationalize(nb, Nb, wl, Rb);
                cplx z[15], Z[15], t = 0, sg;
                for (int k = 0; k < 15; ++k) { z[k] = 0; Z[k] = 0; }
                for (int a = 0; a < nb; ++a) {
                    t += Nb[0][a] * M->h[gid[a]];
                    for (int mm = 0; mm < 5; ++mm) for (int k = 0; k < 3; ++k) { Z[3 * mm + k] += Rb[mm + 1][a] * M->cp[3 * gid[a] + k]; z[3 * mm + k] += Rb[mm + 1][a] * (M->cp[3 * gid[a] + k] + M->u[3 * gid[a] + k]); }
                }
                cplx val = vm_integrand(z, Z, t, P->E, P->nu_, sgn, measure, mode, rho, m_list[s], &sg);
                Is += wq * creal(val); if (creal(sg) > vm) vm = creal(sg);
                if (!want_grad) continue;
                double gz[15], gZ[15], gt;
                for (int c = 0; c < 15; ++c) {
                    cplx keep = z[c]; z[c] = keep + hs * I; gz[c] = cimag(vm_integrand(z, Z, t, P->E, P->nu_, sgn, measure, mode, rho, m_list[s], NULL)) / hs; z[c] = keep;
                    keep = Z[c]; Z[c] = keep + hs * I; gZ[c] = cimag(vm_integrand(z, Z, t, P->E, P->nu_, sgn, measure, mode, rho, m_list[s], NULL)) / hs; Z[c] = keep;
                }
                gt = cimag(vm_integrand(z, Z, t + hs * I, P->E, P->nu_, sgn, measure, mode, rho, m_list[s], NULL)) / hs;
                for (int a = 0; a < nb; ++a) {
                    for (int k = 0; k < 3; ++k) {
                        double du = 0, dc = 0;
                        for (int mm = 0; mm < 5; ++mm) { du += Rb[mm + 1][a] * gz[3 * mm + k]; dc += Rb[mm + 1][a] * (gz[3 * mm + k] + gZ[3 * mm + k]); }
                        if (dIdu) dIdu[3 * gid[a] + k] += wq * du;
                        if (dC[k]) dC[k][gid[a]] += wq * dc;
                    }
                    if (dIdh) dIdh[gid[a]] += wq * Nb[0][a] * gt;
                }
            }
        }
        if (Is_out) Is_out[s] = Is;
        if (vmax) vmax[s] = vm;
    }
    if (dIdu && apply_bcs) for (int64_t r = 0; r < M->ndof; ++r) if (M->zero[r]) dIdu[r] = 0;
}

/* ---- shape regularisation (demos_om/shape_opt/eVTOL/int_energy_regu_exop.py:30-38) ------------------------------------
 * value = sum_s coef[s] int |grad_s(P_f - P_f^0)|^2 dA with spline.grad on the current geometry: grad_s D = D_,a A^ab G_b
 * (contravariant metric by explicit inversion), D_,a = non-rational derivatives of the homogeneous coordinate difference.
 * Gradient wrt the three homogeneous coordinate fields by complex step of the integrand. */
static cplx regu_integrand(const cplx* G1, const cplx* G2, cplx D1, cplx D2) {
    cplx A11 = cdot(G1, G1), A22 = cdot(G2, G2), A12 = cdot(G1, G2), det = A11 * A22 - A12 * A12;
    cplx c11 = A22 / det, c22 = A11 / det, c12 = -A12 / det, gr[3];
    for (int k = 0; k < 3; ++k) gr[k] = (D1 * c11 + D2 * c12) * G1[k] + (D1 * c12 + D2 * c22) * G2[k];
    return cdot(gr, gr) * csqrt(det);
}
void gfo_shape_regu(const gfo_model* M, int field, const double* cp0, const double* coef, double* value, double* dcp0, double* dcp1, double* dcp2) {
    double* dC[3] = {dcp0, dcp1, dcp2}; const double hs = 1e-30; double tot = 0;
    for (int f = 0; f < 3; ++f) if (dC[f]) memset(dC[f], 0, sizeof(double) * M->total_cp);
    for (int s = 0; s < M->np; ++s) {
        const patch_t* P = &M->P[s]; const int p = P->p, q = P->q, nb = (p + 1) * (q + 1);
        for (int ev = 0; ev < P->nelv; ++ev) for (int eu = 0; eu < P->nelu; ++eu) {
            const int iu0 = P->spanu[eu] - p, iv0 = P->spanv[ev] - q; int64_t gid[MAXNB]; double wl[MAXNB];
            for (int jv = 0; jv <= q; ++jv) for (int ju = 0; ju <= p; ++ju) { int a = ju + jv * (p + 1); gid[a] = P->cp_off + (iu0 + ju) + (int64_t)(iv0 + jv) * P->nu; wl[a] = M->w[gid[a]]; }
            for (int gv = 0; gv < P->ngv; ++gv) for (int gu = 0; gu < P->ngu; ++gu) {
                const double* tu = P->bu + (size_t)((eu * P->ngu + gu) * 3) * (p + 1); const double* tv = P->bv + (size_t)((ev * P->ngv + gv) * 3) * (q + 1);
                const double wq = P->wu[eu * P->ngu + gu] * P->wv[ev * P->ngv + gv];
                double Nb[6][MAXNB], Rb[6][MAXNB];
                for (int jv = 0; jv <= q; ++jv) for (int ju = 0; ju <= p; ++ju) {
                    int a = ju + jv * (p + 1);
                    double u0 = tu[ju], u1 = tu[(p + 1) + ju], u2 = tu[2 * (p + 1) + ju], v0 = tv[jv], v1 = tv[(q + 1) + jv], v2 = tv[2 * (q + 1) + jv];
                    Nb[0][a] = u0 * v0; Nb[1][a] = u1 * v0; Nb[2][a] = u0 * v1; Nb[3][a] = u2 * v0; Nb[4][a] = u0 * v2; Nb[5][a] = u1 * v1;
                }
                rationalize(nb, Nb, wl, Rb);
                cplx G[6], D[2] = {0, 0};
                for (int k = 0; k < 6; ++k) G[k] = 0;
                for (int a = 0; a < nb; ++a) {
                    for (int k = 0; k < 3; ++k) { G[k] += Rb[1][a] * M->cp[3 * gid[a] + k]; G[3 + k] += Rb[2][a] * M->cp[3 * gid[a] + k]; }
                    const double dc = M->cp[3 * gid[a] + field] - cp0[gid[a]];
                    D[0] += Nb[1][a] * dc; D[1] += Nb[2][a] * dc;
                }
                tot += wq * coef[s] * creal(regu_integrand(G, G + 3, D[0], D[1]));
                if (!dcp0 && !dcp1 && !dcp2) continue;
                double gG[6], gD[2];
                for (int c = 0; c < 6; ++c) { cplx keep = G[c]; G[c] = keep + hs * I; gG[c] = cimag(regu_integrand(G, G + 3, D[0], D[1])) / hs; G[c] = keep; }
                gD[0] = cimag(regu_integrand(G, G + 3, D[0] + hs * I, D[1])) / hs; gD[1] = cimag(regu_integrand(G, G + 3, D[0], D[1] + hs * I)) / hs;
                for (int a = 0; a < nb; ++a) for (int k = 0; k < 3; ++k) {
                    double g = Rb[1][a] * gG[k] + Rb[2][a] * gG[3 + k];
                    if (k == field) g += Nb[1][a] * gD[0] + Nb[2][a] * gD[1];
                    if (dC[k]) dC[k][gid[a]] += wq * coef[s] * g;
                }
            }
        }
    }
    *value = tot;
}

int gfo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* evaluate physical position/displacement at a parametric point (post-processing for known-answer tests) */
void gfo_eval_point(const gfo_model* M, int patch, double xu, double xv, double X[3], double U[3]) {
    const patch_t* P = &M->P[patch];
    int su = find_span(P->nu, P->p, P->ku, xu), sv = find_span(P->nv, P->q, P->kv, xv);
    double du[3][MAXP + 1], dv[3][MAXP + 1]; basis_ders(su, xu, P->p, P->ku, du); basis_ders(sv, xv, P->q, P->kv, dv);
    double W = 0; X[0] = X[1] = X[2] = U[0] = U[1] = U[2] = 0;
    for (int jv = 0; jv <= P->q; ++jv) for (int ju = 0; ju <= P->p; ++ju) {
        int64_t g = P->cp_off + (su - P->p + ju) + (int64_t)(sv - P->q + jv) * P->nu; double N = du[0][ju] * dv[0][jv];
        W += N * M->w[g]; for (int k = 0; k < 3; ++k) { X[k] += N * M->cp[3 * g + k]; U[k] += N * M->u[3 * g + k]; }
    }
    for (int k = 0; k < 3; ++k) { X[k] /= W; U[k] /= W; }
}
