// kl_point.hpp -- pointwise closed forms of the Kirchhoff-Love / SVK shell energy density
// Psi(z, Z, t) and of the Herrema penalty energy pi(y, Y), device side (gfx950, FP64 VALU).
//
// Replaces what the reference obtains from UFL `derivative()` of ShNAPr's
// surfaceEnergyDensitySVK / PENGoLINS' penalty_energy + FFC code generation
// (GOLDFISH/nonmatching_opt.py:433-452 set_residuals, :404-420 mortar_dRmdCPm_symexp;
// GOLDFISH/utils/opt_utils.py:212-228 dRmdcpm_sub).  Derivation and notation:
// DESIGN.md section 3; the same formulas are checked against autograd on the host in
// tests/test_pointwise_forms.py.
//
// z = (g1,g2,h11,h22,h12) deformed, Z = (G1,G2,H11,H22,H12) reference; flattened index 3*m+i.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#define GF_HD __host__ __device__

namespace gf {

// ---- layout of the per-Gauss-point intermediate record (doubles) -------------------
enum : int {
    IM_J = 0, IM_WQ = 1,
    IM_W = 2,        // [6]  1/W, then the five parametric derivatives of W (for the rational basis)
    IM_G = 8,        // [6]  g1, g2
    IM_N = 14,       // [3]  deformed unit normal
    IM_NB = 17,      // [3]  reference unit normal
    IM_BG = 20,      // [3][6] d beta_k / d(g1,g2)
    IM_CEZ = 38,     // [3][6] J t  C ez
    IM_CBG = 56,     // [3][6] J t3 C bg
    IM_CT3 = 74,     // [6]  J t3 C  (00,01,02,11,12,22)
    IM_JNV = 80,     // [3]  J * membrane resultants (Voigt)
    IM_HMN = 83,     // [21] J * Hessian of (M . n), symmetric 6x6, row-major upper triangle (hmn_idx)
    IM_DN = 104,     // [3][6] dn_i / d(g1,g2)
    IM_JMOF = 122,   // [3]  J mo_k f_k
    IM_PZ = 125,     // [15] dPsi/dz
    IM_JCE = 140,    // [3]  J C eps            } d2Psi/dz dt is rebuilt from these two
    IM_JCK4 = 143,   // [3]  J (t^2/4) C kappa  } (pzt_entry)
    IM_JZJ = 146,    // [6]  (dJ/dZ)/J
    IM_JDNV = 152,   // [3][6] J d(nv)/dZ
    IM_JDMO = 170,   // [3][6] J d(mo)/dZ (tangent columns)
    IM_SIZE = 188
};
GF_HD __forceinline__ int hmn_idx(int r, int s) { const int lo = r < s ? r : s, hi = r < s ? s : r; return 6 * lo - lo * (lo - 1) / 2 + hi - lo; }


GF_HD __forceinline__ void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
GF_HD __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// idx of symmetric 3x3 stored as (00,01,02,11,12,22)
GF_HD __forceinline__ int sym3(int a, int b) {
    const int lut[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
    return lut[3 * a + b];
}

// n = unit(g1 x g2), j = |g1 x g2|, Dn[i][c] = dn_i / d(g1,g2)_c  (c = 0..5)
GF_HD inline void normal_derivs(const double* g1, const double* g2, double* n, double& j, double Dn[3][6]) {
    double nt[3]; cross3(g1, g2, nt);
    j = sqrt(dot3(nt, nt));
    const double ij = 1.0 / j;
    for (int k = 0; k < 3; ++k) n[k] = nt[k] * ij;
    // B = [-skew(g2) | skew(g1)], column c: d(g1 x g2)/d(.)_c ; Dn = (I - n n^T)/j * B
    for (int c = 0; c < 6; ++c) {
        double e[3] = {0, 0, 0}, col[3];
        e[c % 3] = 1.0;
        if (c < 3) cross3(e, g2, col); else cross3(g1, e, col);
        const double nc = dot3(n, col);
        for (int i = 0; i < 3; ++i) Dn[i][c] = (col[i] - n[i] * nc) * ij;
    }
}

// H[6][6] = Hessian of (M . n(g1,g2)); see oracle/kl_point_numpy.py hess_M_dot_n
GF_HD inline void hess_M_dot_n(const double* g1, const double* g2, const double* n, double j, const double* M, double H[6][6]) {
    const double Mn = dot3(M, n), ij2 = 1.0 / (j * j);
    double Q[3][3], v[3];
    for (int a = 0; a < 3; ++a) {
        v[a] = (M[a] - Mn * n[a]) / j;
        for (int b = 0; b < 3; ++b) Q[a][b] = -(M[a] * n[b] + n[a] * M[b] + Mn * ((a == b ? 1.0 : 0.0) - 3.0 * n[a] * n[b])) * ij2;
    }
    double Bc[6][3], QB[6][3];
    for (int c = 0; c < 6; ++c) {
        double e[3] = {0, 0, 0}; e[c % 3] = 1.0;
        if (c < 3) cross3(e, g2, Bc[c]); else cross3(g1, e, Bc[c]);
        for (int a = 0; a < 3; ++a) QB[c][a] = Q[a][0] * Bc[c][0] + Q[a][1] * Bc[c][1] + Q[a][2] * Bc[c][2];
    }
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H[r][c] = dot3(Bc[r], QB[c]);
    // +- skew(v) on the (g1,g2) / (g2,g1) blocks
    const double S[3][3] = {{0, -v[2], v[1]}, {v[2], 0, -v[0]}, {-v[1], v[0], 0}};
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { H[a][3 + b] -= S[a][b]; H[3 + a][b] += S[a][b]; }
}

// Material tensor (curvilinear Voigt, symmetric 6-storage) and dC/dA_q, q = (A11, A22, A12)
GF_HD inline void material(const double* G1, const double* G2, double E, double nu, double C[6], double dC[3][6], double& J) {
    const double A11 = dot3(G1, G1), A22 = dot3(G2, G2), A12 = dot3(G1, G2);
    const double det = A11 * A22 - A12 * A12, id = 1.0 / det;
    const double c11 = A22 * id, c22 = A11 * id, c12 = -A12 * id, Eb = E / (1.0 - nu * nu);
    J = sqrt(det);
    C[0] = Eb * c11 * c11; C[1] = Eb * (nu * c11 * c22 + (1 - nu) * c12 * c12); C[2] = Eb * c11 * c12;
    C[3] = Eb * c22 * c22; C[4] = Eb * c22 * c12; C[5] = Eb * 0.5 * ((1 - nu) * c11 * c22 + (1 + nu) * c12 * c12);
    const double dc[3][3] = {{-c11 * c11, -c12 * c12, -2 * c11 * c12},
                             {-c12 * c12, -c22 * c22, -2 * c12 * c22},
                             {-c11 * c12, -c12 * c22, -(c11 * c22 + c12 * c12)}};   // rows (c11,c22,c12), cols q
    for (int q = 0; q < 3; ++q) {
        const double d11 = dc[0][q], d22 = dc[1][q], d12 = dc[2][q];
        dC[q][0] = Eb * 2 * c11 * d11;
        dC[q][1] = Eb * (nu * (d11 * c22 + c11 * d22) + 2 * (1 - nu) * c12 * d12);
        dC[q][2] = Eb * (d11 * c12 + c11 * d12);
        dC[q][3] = Eb * 2 * c22 * d22;
        dC[q][4] = Eb * (d22 * c12 + c22 * d12);
        dC[q][5] = Eb * 0.5 * ((1 - nu) * (d11 * c22 + c11 * d22) + 2 * (1 + nu) * c12 * d12);
    }
}
GF_HD __forceinline__ void symmv(const double* C, const double* x, double* y) {
    y[0] = C[0] * x[0] + C[1] * x[1] + C[2] * x[2];
    y[1] = C[1] * x[0] + C[3] * x[1] + C[4] * x[2];
    y[2] = C[2] * x[0] + C[4] * x[1] + C[5] * x[2];
}


// ---- strains from the DISPLACEMENT derivatives (round 5) ------------------------------------------------------------------------------------------
// The reference's forms (ShNAPr surfaceEnergyDensitySVK) evaluate eps = (a - A) / 2 and kappa = B - b as differences of the metric / curvature coefficients
// of the two configurations: an absolute error of eps_machine |A|^2 in a strain that is 1e-6 ... 1e-10 for the thin shells of the demos, i.e. a floor of the
// residual of about eps_machine E h |A|^2 |grad N| per entry however small the load (Newton stalled above the reference's rtol = 1e-3 on C4).
// With dz = z - Z formed from the displacement coefficients alone (the callers sum U instead of c + U: the rational basis is linear in the coefficients) the
// same strains follow without cancellation:
//     eps_ab   = (Z_a . dz_b + dz_a . Z_b + dz_a . dz_b) / 2
//     n - N    = (delta - N s / (j + J)) / j,   delta = Z_1 x dz_2 + dz_1 x Z_2 + dz_1 x dz_2,   s = j^2 - J^2 = 2 J N . delta + delta . delta
//     kappa_ab = H_ab . N - h_ab . n = -(H_ab . (n - N) + dh_ab . n)
// Everything else of the closed forms (first / second derivatives) multiplies these and is evaluated as before.  oracle/kl_oracle.c has both evaluations
// (gfo_set_strain_mode); tests/test_strain_evaluation.py compares them.
GF_HD __forceinline__ void kl_strains(const double* Z, const double* dz, const double* n, const double* N, double j, double Jn, double* eps, double* kap) {
    eps[0] = dot3(Z, dz) + 0.5 * dot3(dz, dz);
    eps[1] = dot3(Z + 3, dz + 3) + 0.5 * dot3(dz + 3, dz + 3);
    eps[2] = dot3(Z, dz + 3) + dot3(dz, Z + 3) + dot3(dz, dz + 3);
    double a[3], b[3], c[3], dl[3], dn[3];
    cross3(Z, dz + 3, a); cross3(dz, Z + 3, b); cross3(dz, dz + 3, c);
    for (int k = 0; k < 3; ++k) dl[k] = a[k] + b[k] + c[k];
    const double f = (2.0 * Jn * dot3(N, dl) + dot3(dl, dl)) / (j + Jn), ij = 1.0 / j;
    for (int k = 0; k < 3; ++k) dn[k] = (dl[k] - N[k] * f) * ij;
    kap[0] = -(dot3(Z + 6, dn) + dot3(dz + 6, n));
    kap[1] = -(dot3(Z + 9, dn) + dot3(dz + 9, n));
    kap[2] = -2.0 * (dot3(Z + 12, dn) + dot3(dz + 12, n));
}

// Everything a Gauss point contributes, in compact form.  z, Z: [15]; dz = z - Z from the displacement coefficients; out: IM record (W slots untouched).
// REF = false (passes without dR/dCP: the K walk of gf_element_rec4.hpp): the reference-configuration derivatives (IM_JZJ, IM_JDNV, IM_JDMO) are not produced.
template <bool REF = true>
GF_HD inline void shell_point(const double* z, const double* Z, const double* dz, double t, double E, double nu, double* im) {
    const double f3[3] = {1.0, 1.0, 2.0};
    double n[3], N[3], j, Jn, Dn[3][6], DN[3][6];
    normal_derivs(z, z + 3, n, j, Dn);
    normal_derivs(Z, Z + 3, N, Jn, DN);
    double C[6], dC[3][6], J;
    material(Z, Z + 3, E, nu, C, dC, J);
    double eps[3], kap[3];
    kl_strains(Z, dz, n, N, j, Jn, eps, kap);
    const double t3 = t * t * t / 12.0;
    double Ce[3], Ck[3], nv[3], mo[3];
    symmv(C, eps, Ce); symmv(C, kap, Ck);
    for (int k = 0; k < 3; ++k) { nv[k] = t * Ce[k]; mo[k] = t3 * Ck[k]; }
    im[IM_J] = J;
    for (int c = 0; c < 6; ++c) im[IM_G + c] = z[c];
    for (int k = 0; k < 3; ++k) { im[IM_N + k] = n[k]; im[IM_NB + k] = N[k]; im[IM_JNV + k] = J * nv[k]; im[IM_JMOF + k] = J * mo[k] * f3[k]; im[IM_JCE + k] = J * Ce[k]; im[IM_JCK4 + k] = J * 0.25 * t * t * Ck[k]; }
    // ez[k][c] (c<6): row0 [g1,0], row1 [0,g2], row2 [g2,g1];  bg[k][c] = f_k h_k . Dn[:,c]
    double ez[3][6], bg[3][6], eZ[3][6], bG[3][6];
    for (int c = 0; c < 3; ++c) {
        ez[0][c] = z[c]; ez[0][3 + c] = 0; ez[1][c] = 0; ez[1][3 + c] = z[3 + c]; ez[2][c] = z[3 + c]; ez[2][3 + c] = z[c];
        eZ[0][c] = -Z[c]; eZ[0][3 + c] = 0; eZ[1][c] = 0; eZ[1][3 + c] = -Z[3 + c]; eZ[2][c] = -Z[3 + c]; eZ[2][3 + c] = -Z[c];
    }
    for (int k = 0; k < 3; ++k) for (int c = 0; c < 6; ++c) {
        bg[k][c] = f3[k] * (z[6 + 3 * k] * Dn[0][c] + z[7 + 3 * k] * Dn[1][c] + z[8 + 3 * k] * Dn[2][c]);
        bG[k][c] = f3[k] * (Z[6 + 3 * k] * DN[0][c] + Z[7 + 3 * k] * DN[1][c] + Z[8 + 3 * k] * DN[2][c]);
        im[IM_BG + 6 * k + c] = bg[k][c];
    }
    for (int i = 0; i < 3; ++i) for (int c = 0; c < 6; ++c) im[IM_DN + 6 * i + c] = Dn[i][c];
    for (int c = 0; c < 6; ++c) {
        double a[3] = {ez[0][c], ez[1][c], ez[2][c]}, b[3] = {bg[0][c], bg[1][c], bg[2][c]}, ca[3], cb[3];
        symmv(C, a, ca); symmv(C, b, cb);
        for (int k = 0; k < 3; ++k) { im[IM_CEZ + 6 * k + c] = J * t * ca[k]; im[IM_CBG + 6 * k + c] = J * t3 * cb[k]; }
    }
    for (int k = 0; k < 6; ++k) im[IM_CT3 + k] = J * t3 * C[k];
    // Pz
    for (int c = 0; c < 6; ++c) {
        double pe = 0, pb = 0;
        for (int k = 0; k < 3; ++k) { pe += nv[k] * ez[k][c]; pb += mo[k] * bg[k][c]; }
        im[IM_PZ + c] = J * (pe - pb);
    }
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) im[IM_PZ + 6 + 3 * k + i] = -J * mo[k] * f3[k] * n[i];
    // geometric bending term: M = sum_k mo_k f_k h_k
    double M[3], H[6][6];
    for (int i = 0; i < 3; ++i) M[i] = mo[0] * z[6 + i] + mo[1] * z[9 + i] + 2.0 * mo[2] * z[12 + i];
    hess_M_dot_n(z, z + 3, n, j, M, H);
    for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) im[IM_HMN + hmn_idx(r, c)] = J * H[r][c];
    // reference path: JZ/J, J dnv/dZ, J dmo/dZ (tangent columns)
    if constexpr (!REF) return;
    double JZ[6];
    cross3(Z + 3, N, JZ); cross3(N, Z, JZ + 3);
    for (int c = 0; c < 6; ++c) im[IM_JZJ + c] = JZ[c] / J;
    double dCe[3][3], dCk[3][3];
    for (int q = 0; q < 3; ++q) { symmv(dC[q], eps, dCe[q]); symmv(dC[q], kap, dCk[q]); }
    for (int c = 0; c < 6; ++c) {
        // d(A11,A22,A12)/dZ_c
        const int ic = c % 3;
        const double a0 = c < 3 ? 2 * Z[ic] : 0.0, a1 = c < 3 ? 0.0 : 2 * Z[3 + ic], a2 = c < 3 ? Z[3 + ic] : Z[ic];
        double ev[3] = {eZ[0][c], eZ[1][c], eZ[2][c]}, bv[3] = {bG[0][c], bG[1][c], bG[2][c]}, ce[3], cb[3];
        symmv(C, ev, ce); symmv(C, bv, cb);
        for (int k = 0; k < 3; ++k) {
            im[IM_JDNV + 6 * k + c] = J * t * (dCe[0][k] * a0 + dCe[1][k] * a1 + dCe[2][k] * a2 + ce[k]);
            im[IM_JDMO + 6 * k + c] = J * t3 * (dCk[0][k] * a0 + dCk[1][k] * a1 + dCk[2][k] * a2 + cb[k]);
        }
    }
}

// shell_point split by columns: every record array indexed by a tangent column c = 0..5 (d/dg1_i for c = i, d/dg2_i for
// c = 3 + i) is produced for the two columns c = ic and c = 3 + ic only; the three callers ic = 0, 1, 2 (three lanes of the
// MFMA element kernel, which run this code in lock-step) fill the whole record.  d[i] = delta(i, ic) selects components
// arithmetically (no register-array indexing by ic).  The scalar part is computed by every caller (same instruction
// stream on the GPU) and written by the caller with lead = true.  Same formulas as shell_point above.
// ref = false (Newton pass: no dR/dCP): the reference-configuration derivatives (IM_JDNV, IM_JDMO) are not produced.
template <bool REF = true>
GF_HD inline void shell_point_cols(const double* z, const double* Z, const double* dz, double t, double E, double nu, int ic, const double* d, bool lead, double* im) {
    const double f3[3] = {1.0, 1.0, 2.0};
    double nt[3], n[3], Nt[3], N[3];
    cross3(z, z + 3, nt); const double j = sqrt(dot3(nt, nt)), ij = 1.0 / j;
    cross3(Z, Z + 3, Nt); const double Jn = sqrt(dot3(Nt, Nt)), iJn = 1.0 / Jn;
    for (int k = 0; k < 3; ++k) { n[k] = nt[k] * ij; N[k] = Nt[k] * iJn; }
    double C[6], dC[3][6], J;
    material(Z, Z + 3, E, nu, C, dC, J);
    double eps[3], kap[3];
    kl_strains(Z, dz, n, N, j, Jn, eps, kap);
    const double t3 = t * t * t / 12.0;
    double Ce[3], Ck[3], nv[3], mo[3];
    symmv(C, eps, Ce); symmv(C, kap, Ck);
    for (int k = 0; k < 3; ++k) { nv[k] = t * Ce[k]; mo[k] = t3 * Ck[k]; }
    if (lead) {
        im[IM_J] = J;
        for (int c = 0; c < 6; ++c) im[IM_G + c] = z[c];
        for (int k = 0; k < 3; ++k) { im[IM_N + k] = n[k]; im[IM_NB + k] = N[k]; im[IM_JNV + k] = J * nv[k]; im[IM_JMOF + k] = J * mo[k] * f3[k]; im[IM_JCE + k] = J * Ce[k]; im[IM_JCK4 + k] = J * 0.25 * t * t * Ck[k]; }
        for (int k = 0; k < 6; ++k) im[IM_CT3 + k] = J * t3 * C[k];
        for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) im[IM_PZ + 6 + 3 * k + i] = -J * mo[k] * f3[k] * n[i];
    }
    // quantities shared by the two columns
    double JZ[6];
    cross3(Z + 3, N, JZ); cross3(N, Z, JZ + 3);
    double dCe[3][3], dCk[3][3];
    if constexpr (REF) for (int q = 0; q < 3; ++q) { symmv(dC[q], eps, dCe[q]); symmv(dC[q], kap, dCk[q]); }
    double M[3];
    for (int i = 0; i < 3; ++i) M[i] = mo[0] * z[6 + i] + mo[1] * z[9 + i] + 2.0 * mo[2] * z[12 + i];
    const double Mn = dot3(M, n), ij2 = ij * ij;
    double Q[3][3], v[3];
    for (int a = 0; a < 3; ++a) {
        v[a] = (M[a] - Mn * n[a]) * ij;
        for (int b = 0; b < 3; ++b) Q[a][b] = -(M[a] * n[b] + n[a] * M[b] + Mn * ((a == b ? 1.0 : 0.0) - 3.0 * n[a] * n[b])) * ij2;
    }
    const double g1c = dot3(d, z), g2c = dot3(d, z + 3), G1c = dot3(d, Z), G2c = dot3(d, Z + 3);
    // skew(v)[a][ic]: S = [[0,-v2,v1],[v2,0,-v0],[-v1,v0,0]]
    const double Sic[3] = {d[1] * (-v[2]) + d[2] * v[1], d[0] * v[2] + d[2] * (-v[0]), d[0] * (-v[1]) + d[1] * v[0]};
    for (int cc = 0; cc < 2; ++cc) {
        const int c = ic + 3 * cc;
        double col[3], COL[3];
        if (cc == 0) { cross3(d, z + 3, col); cross3(d, Z + 3, COL); } else { cross3(z, d, col); cross3(Z, d, COL); }
        const double nc = dot3(n, col), NC = dot3(N, COL);
        double Dn[3], DN[3];
        for (int i = 0; i < 3; ++i) { Dn[i] = (col[i] - n[i] * nc) * ij; DN[i] = (COL[i] - N[i] * NC) * iJn; }
        double ez[3], eZ[3], bg[3], bG[3];
        if (cc == 0) { ez[0] = g1c; ez[1] = 0.0; ez[2] = g2c; eZ[0] = -G1c; eZ[1] = 0.0; eZ[2] = -G2c; }
        else { ez[0] = 0.0; ez[1] = g2c; ez[2] = g1c; eZ[0] = 0.0; eZ[1] = -G2c; eZ[2] = -G1c; }
        for (int k = 0; k < 3; ++k) {
            bg[k] = f3[k] * dot3(z + 6 + 3 * k, Dn);
            bG[k] = f3[k] * dot3(Z + 6 + 3 * k, DN);
            im[IM_BG + 6 * k + c] = bg[k];
        }
        for (int i = 0; i < 3; ++i) im[IM_DN + 6 * i + c] = Dn[i];
        double ca[3], cb[3];
        symmv(C, ez, ca); symmv(C, bg, cb);
        for (int k = 0; k < 3; ++k) { im[IM_CEZ + 6 * k + c] = J * t * ca[k]; im[IM_CBG + 6 * k + c] = J * t3 * cb[k]; }
        double pe = 0, pb = 0;
        for (int k = 0; k < 3; ++k) { pe += nv[k] * ez[k]; pb += mo[k] * bg[k]; }
        im[IM_PZ + c] = J * (pe - pb);
        im[IM_JZJ + c] = dot3(d, JZ + 3 * cc) / J;
        if constexpr (REF) {
            const double a0 = cc == 0 ? 2 * G1c : 0.0, a1 = cc == 0 ? 0.0 : 2 * G2c, a2 = cc == 0 ? G2c : G1c;
            double ce[3], cb2[3];
            symmv(C, eZ, ce); symmv(C, bG, cb2);
            for (int k = 0; k < 3; ++k) {
                im[IM_JDNV + 6 * k + c] = J * t * (dCe[0][k] * a0 + dCe[1][k] * a1 + dCe[2][k] * a2 + ce[k]);
                im[IM_JDMO + 6 * k + c] = J * t3 * (dCk[0][k] * a0 + dCk[1][k] * a1 + dCk[2][k] * a2 + cb2[k]);
            }
        }
        // Hessian of M . n, column c, rows r <= c (symmetric storage): H[r][c] = Bc[r] . (Q Bc[c]) -+ skew(v)
        double QB[3];
        for (int a = 0; a < 3; ++a) QB[a] = Q[a][0] * col[0] + Q[a][1] * col[1] + Q[a][2] * col[2];
        for (int r = 0; r < 6; ++r) {
            double e[3] = {0, 0, 0}, Br[3];
            e[r % 3] = 1.0;
            if (r < 3) cross3(e, z + 3, Br); else cross3(z, e, Br);
            double h = dot3(Br, QB);
            if (cc == 1 && r < 3) h -= Sic[r];
            if (r <= c) im[IM_HMN + 6 * r - r * (r - 1) / 2 - r + c] = J * h;
        }
    }
}

// ---- expansion of single entries of Pzz / PzZ from the intermediate record -----------
GF_HD __forceinline__ double ez_entry(const double* im, int k, int r) {   // r < 6
    const int m = r / 3, i = r - 3 * m;
    if (k == 2) return im[IM_G + 3 * (1 - m) + i];
    return (k == m) ? im[IM_G + r] : 0.0;
}
GF_HD __forceinline__ double bz_entry(const double* im, int k, int r) {
    const double f3[3] = {1.0, 1.0, 2.0};
    if (r < 6) return im[IM_BG + 6 * k + r];
    const int kk = (r - 6) / 3, i = (r - 6) - 3 * kk;
    return kk == k ? f3[k] * im[IM_N + i] : 0.0;
}
// d2Psi/dz dt [r] rebuilt from the compact record
GF_HD inline double pzt_entry(const double* im, int r) {
    double v = 0.0;
    for (int k = 0; k < 3; ++k) v -= im[IM_JCK4 + k] * bz_entry(im, k, r);
    if (r < 6) for (int k = 0; k < 3; ++k) v += im[IM_JCE + k] * ez_entry(im, k, r);
    return v;
}
GF_HD inline double pzz_entry(const double* im, int r, int s) {
    const double f3[3] = {1.0, 1.0, 2.0};
    double v = 0.0;
    if (r < 6 && s < 6) {
        for (int k = 0; k < 3; ++k) v += ez_entry(im, k, r) * im[IM_CEZ + 6 * k + s] + im[IM_BG + 6 * k + r] * im[IM_CBG + 6 * k + s];
        const int m = r / 3, mm = s / 3;
        if (r - 3 * m == s - 3 * mm) v += im[IM_JNV + (m == mm ? m : 2)];
        v -= im[IM_HMN + hmn_idx(r, s)];
    } else if (r >= 6 && s >= 6) {
        const int k = (r - 6) / 3, i = (r - 6) - 3 * k, kk = (s - 6) / 3, jj = (s - 6) - 3 * kk;
        v = f3[k] * im[IM_N + i] * im[IM_CT3 + sym3(k, kk)] * f3[kk] * im[IM_N + jj];
    } else {
        const int hi = r >= 6 ? r : s, lo = r >= 6 ? s : r;          // symmetric
        const int k = (hi - 6) / 3, i = (hi - 6) - 3 * k;
        v = f3[k] * im[IM_N + i] * im[IM_CBG + 6 * k + lo] - im[IM_JMOF + k] * im[IM_DN + 6 * i + lo];
    }
    return v;
}
GF_HD inline double pzZ_entry(const double* im, int r, int s) {
    const double f3[3] = {1.0, 1.0, 2.0};
    double v = 0.0;
    if (s < 6) {
        v = im[IM_PZ + r] * im[IM_JZJ + s];
        if (r < 6) for (int k = 0; k < 3; ++k) v += ez_entry(im, k, r) * im[IM_JDNV + 6 * k + s];
        for (int k = 0; k < 3; ++k) v -= bz_entry(im, k, r) * im[IM_JDMO + 6 * k + s];
    } else {
        const int kk = (s - 6) / 3, jj = (s - 6) - 3 * kk;
        for (int k = 0; k < 3; ++k) v -= bz_entry(im, k, r) * im[IM_CT3 + sym3(k, kk)] * f3[kk] * im[IM_NB + jj];
    }
    return v;
}

// ---- distributed load: per unit area (pd == 0) or per unit PROJECTED area (pd != 0; gf_model_desc.load_proj) ----------
// load scalar s = |G1 x G2| = J, or pd . (G1 x G2) = J (pd . N); reference derivative from the record alone:
//   ds/dG1 = G2 x pd = (pd.N)(G2 x N) - (pd.(G2 x N)) N,  ds/dG2 = pd x G1 = (pd.N)(N x G1) - (pd.(N x G1)) N,
// with G2 x N = J * JZJ[0..2], N x G1 = J * JZJ[3..5].  pd == 0 gives pn = 1, q = 0: s = J, ds/dZ = dJ/dZ.
struct LoadGeom { double pn, q0, q1; };
// pd is uniform over the workgroup (per patch): real branches, so that the plain body force pays (almost) nothing for the
// general case (measured at C4: 15.3 ms element kernel without the feature, 15.55 with these branches, 15.9 with selects,
// 16.1 with the flag and direction hoisted into registers by hand)
GF_HD __forceinline__ bool load_is_projected(const double* pd) { return pd[0] != 0.0 || pd[1] != 0.0 || pd[2] != 0.0; }
GF_HD __forceinline__ LoadGeom load_geom(const double* im, const double* pd) {
    LoadGeom g = {1.0, 0.0, 0.0};
    if (load_is_projected(pd)) { g.pn = dot3(pd, im + IM_NB); g.q0 = dot3(pd, im + IM_JZJ); g.q1 = dot3(pd, im + IM_JZJ + 3); }
    return g;
}
GF_HD __forceinline__ double load_scalar(const double* im, const double* pd) {
    return load_is_projected(pd) ? im[IM_J] * dot3(pd, im + IM_NB) : im[IM_J];
}
// (ds/dZ . phi_b)_f = ds/dG1_f phi_b,1 + ds/dG2_f phi_b,2
GF_HD __forceinline__ double load_dz_dot(const double* im, const double* pd, const LoadGeom& g, int f, double pb0, double pb1) {
    if (load_is_projected(pd))
        return im[IM_J] * ((g.pn * im[IM_JZJ + f] - g.q0 * im[IM_NB + f]) * pb0 + (g.pn * im[IM_JZJ + 3 + f] - g.q1 * im[IM_NB + f]) * pb1);
    return im[IM_J] * (im[IM_JZJ + f] * pb0 + im[IM_JZJ + 3 + f] * pb1);
}

// ---- energy functionals: first derivatives of Psi wrt z, Z, t and of the area Jacobian ----------
// out: [0] Psi, [1] J, [2] dPsi/dt, [3..17] dPsi/dz, [18..32] dPsi/dZ, [33..38] dJ/d(G1,G2)
enum : int { FE_PSI = 0, FE_J = 1, FE_PT = 2, FE_PZ = 3, FE_PZR = 18, FE_JZ = 33, FE_SIZE = 39 };
GF_HD inline void shell_energy_point(const double* z, const double* Z, const double* dz, double t, double E, double nu, double* out) {
    const double f3[3] = {1.0, 1.0, 2.0};
    double n[3], N[3], j, Jn, Dn[3][6], DN[3][6];
    normal_derivs(z, z + 3, n, j, Dn);
    normal_derivs(Z, Z + 3, N, Jn, DN);
    double C[6], dC[3][6], J;
    material(Z, Z + 3, E, nu, C, dC, J);
    double eps[3], kap[3];
    kl_strains(Z, dz, n, N, j, Jn, eps, kap);
    const double t3 = t * t * t / 12.0;
    double Ce[3], Ck[3];
    symmv(C, eps, Ce); symmv(C, kap, Ck);
    const double psi = 0.5 * t * dot3(eps, Ce) + 0.5 * t3 * dot3(kap, Ck);
    out[FE_PSI] = J * psi; out[FE_J] = J;
    out[FE_PT] = J * (0.5 * dot3(eps, Ce) + 0.125 * t * t * dot3(kap, Ck));
    double JZ[6];
    cross3(Z + 3, N, JZ); cross3(N, Z, JZ + 3);
    double qe[3], qk[3];                         // 0.5 eps.dC_q eps, 0.5 kap.dC_q kap
    for (int q = 0; q < 3; ++q) { double a[3], b[3]; symmv(dC[q], eps, a); symmv(dC[q], kap, b); qe[q] = 0.5 * dot3(eps, a); qk[q] = 0.5 * dot3(kap, b); }
    for (int c = 0; c < 6; ++c) {
        const int ic = c % 3;
        // deformed: d eps/dz, d beta/dz
        const double e0 = c < 3 ? z[ic] : 0.0, e1 = c < 3 ? 0.0 : z[3 + ic], e2 = c < 3 ? z[3 + ic] : z[ic];
        double bg[3], bG[3];
        for (int k = 0; k < 3; ++k) {
            bg[k] = f3[k] * (z[6 + 3 * k] * Dn[0][c] + z[7 + 3 * k] * Dn[1][c] + z[8 + 3 * k] * Dn[2][c]);
            bG[k] = f3[k] * (Z[6 + 3 * k] * DN[0][c] + Z[7 + 3 * k] * DN[1][c] + Z[8 + 3 * k] * DN[2][c]);
        }
        out[FE_PZ + c] = J * (t * (Ce[0] * e0 + Ce[1] * e1 + Ce[2] * e2) - t3 * dot3(Ck, bg));
        // reference: d eps/dZ = -(...), d kappa/dZ = +bG, metric chain for C and J
        const double E0 = c < 3 ? Z[ic] : 0.0, E1 = c < 3 ? 0.0 : Z[3 + ic], E2 = c < 3 ? Z[3 + ic] : Z[ic];
        const double a0 = c < 3 ? 2 * Z[ic] : 0.0, a1 = c < 3 ? 0.0 : 2 * Z[3 + ic], a2 = c < 3 ? Z[3 + ic] : Z[ic];
        out[FE_PZR + c] = psi * JZ[c] + J * (-t * (Ce[0] * E0 + Ce[1] * E1 + Ce[2] * E2) + t3 * dot3(Ck, bG)
                                             + t * (qe[0] * a0 + qe[1] * a1 + qe[2] * a2) + t3 * (qk[0] * a0 + qk[1] * a1 + qk[2] * a2));
        out[FE_JZ + c] = JZ[c];
    }
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) {
        out[FE_PZ + 6 + 3 * k + i] = -J * t3 * Ck[k] * f3[k] * n[i];
        out[FE_PZR + 6 + 3 * k + i] = J * t3 * Ck[k] * f3[k] * N[i];
    }
}

// ---- von Mises stress at a through-thickness station (max_vmstress_exop.py:17-47, PENGoLINS ShellStressSVK) -----
// Contravariant 2nd Piola-Kirchhoff components s = C (eps + xi kap) at xi = sgn * t/2 (sgn = +1 top, -1 bottom,
// 0 middle).  A plane-stress tensor has sigma_vM^2 = (tr)^2 - 3 det, both invariants written with the metric of
// the basis the components refer to, so no local Cartesian basis is needed:
//   measure 0 (Cauchy, sigma = F S F^T / Jr on the deformed tangents, metric a, Jr^2 = det a / det A):
//       q = [ (s:a)^2 / det a - 3 det s ] det A
//   measure 1 (2nd Piola-Kirchhoff, metric A):   q = (s:A)^2 - 3 det s det A
// out: [0] sigma_vM, [1] J, [2] d sigma/dt, [3..17] d sigma/dz, [18..32] d sigma/dZ, [33..38] dJ/d(G1,G2)
// (same slots as shell_energy_point, so the functional kernels share their contraction code).
GF_HD inline void shell_stress_point(const double* z, const double* Z, const double* dz, double t, double E, double nu, double sgn, int measure, double* out) {
    const double f3[3] = {1.0, 1.0, 2.0};
    double n[3], N[3], j, Jn, Dn[3][6], DN[3][6];
    normal_derivs(z, z + 3, n, j, Dn);
    normal_derivs(Z, Z + 3, N, Jn, DN);
    double C[6], dC[3][6], J;
    material(Z, Z + 3, E, nu, C, dC, J);
    const double A[3] = {dot3(Z, Z), dot3(Z + 3, Z + 3), dot3(Z, Z + 3)};
    const double a[3] = {dot3(z, z), dot3(z + 3, z + 3), dot3(z, z + 3)};
    double eps[3], kap[3], e[3], s[3];
    kl_strains(Z, dz, n, N, j, Jn, eps, kap);
    const double xi = 0.5 * sgn * t;
    for (int k = 0; k < 3; ++k) e[k] = eps[k] + xi * kap[k];
    symmv(C, e, s);
    const bool cau = measure == 0;
    const double mt[3] = {cau ? a[0] : A[0], cau ? a[1] : A[1], cau ? a[2] : A[2]};      // by value: a pointer select would push a[], A[] to scratch
    const double tr = s[0] * mt[0] + s[1] * mt[1] + 2.0 * s[2] * mt[2];
    const double dS = s[0] * s[1] - s[2] * s[2], dm = mt[0] * mt[1] - mt[2] * mt[2], dA = A[0] * A[1] - A[2] * A[2];
    const double r = measure == 0 ? dA / dm : 1.0;
    const double q = tr * tr * r - 3.0 * dS * dA;
    const double sig = q > 0.0 ? sqrt(q) : 0.0;
    out[0] = sig; out[1] = J;
    double JZ[6];
    cross3(Z + 3, N, JZ); cross3(N, Z, JZ + 3);
    // adjoints of sigma
    const double qb = sig > 0.0 ? 0.5 / sig : 0.0;
    const double trb = qb * 2.0 * tr * r, dSb = -3.0 * qb * dA;
    double dAb = -3.0 * qb * dS, dmb = 0.0;
    if (measure == 0) { const double rb = qb * tr * tr; dAb += rb / dm; dmb = -rb * r / dm; }
    const double sb[3] = {trb * mt[0] + dSb * s[1], trb * mt[1] + dSb * s[0], 2.0 * trb * mt[2] - 2.0 * dSb * s[2]};
    const double mb[3] = {trb * s[0] + dmb * mt[1], trb * s[1] + dmb * mt[0], 2.0 * trb * s[2] - 2.0 * dmb * mt[2]};
    double eb[3];
    symmv(C, sb, eb);
    const double kb[3] = {xi * eb[0], xi * eb[1], xi * eb[2]};
    out[2] = 0.5 * sgn * dot3(eb, kap);
    double Ab[3] = {dAb * A[1], dAb * A[0], -2.0 * dAb * A[2]};
    for (int qq = 0; qq < 3; ++qq) { double w[3]; symmv(dC[qq], e, w); Ab[qq] += dot3(sb, w); }
    double ab[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < 3; ++k) { if (measure == 0) ab[k] = mb[k]; else Ab[k] += mb[k]; }
    for (int c = 0; c < 6; ++c) {
        const int ic = c % 3;
        const double e0 = c < 3 ? z[ic] : 0.0, e1 = c < 3 ? 0.0 : z[3 + ic], e2 = c < 3 ? z[3 + ic] : z[ic];
        const double E0 = c < 3 ? Z[ic] : 0.0, E1 = c < 3 ? 0.0 : Z[3 + ic], E2 = c < 3 ? Z[3 + ic] : Z[ic];
        double bg = 0.0, bG = 0.0;
        for (int k = 0; k < 3; ++k) {
            bg += kb[k] * f3[k] * (z[6 + 3 * k] * Dn[0][c] + z[7 + 3 * k] * Dn[1][c] + z[8 + 3 * k] * Dn[2][c]);
            bG += kb[k] * f3[k] * (Z[6 + 3 * k] * DN[0][c] + Z[7 + 3 * k] * DN[1][c] + Z[8 + 3 * k] * DN[2][c]);
        }
        out[FE_PZ + c] = (eb[0] + 2.0 * ab[0]) * e0 + (eb[1] + 2.0 * ab[1]) * e1 + (eb[2] + ab[2]) * e2 - bg;
        out[FE_PZR + c] = (2.0 * Ab[0] - eb[0]) * E0 + (2.0 * Ab[1] - eb[1]) * E1 + (Ab[2] - eb[2]) * E2 + bG;
        out[FE_JZ + c] = JZ[c];
    }
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) {
        out[FE_PZ + 6 + 3 * k + i] = -kb[k] * f3[k] * n[i];
        out[FE_PZR + 6 + 3 * k + i] = kb[k] * f3[k] * N[i];
    }
}

// ---------------------------------------------------------------------------- penalty
// unit tangent at = unit(tau0 g1 + tau1 g2), Dt[i][c]
GF_HD inline void tangent_derivs(const double* g1, const double* g2, const double* tau, double* at, double& L, double Dt[3][6]) {
    double tt[3];
    for (int k = 0; k < 3; ++k) tt[k] = tau[0] * g1[k] + tau[1] * g2[k];
    L = sqrt(dot3(tt, tt));
    for (int k = 0; k < 3; ++k) at[k] = tt[k] / L;
    for (int c = 0; c < 6; ++c) {
        const double tc = tau[c / 3];
        for (int i = 0; i < 3; ++i) Dt[i][c] = tc * ((i == c % 3 ? 1.0 : 0.0) - at[i] * at[c % 3]) / L;
    }
}
GF_HD inline void hess_M_dot_t(const double* at, double L, const double* tau, const double* M, double H[6][6]) {
    const double Ma = dot3(M, at), iL2 = 1.0 / (L * L);
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
        const int a = r % 3, b = c % 3;
        const double Q = -(M[a] * at[b] + at[a] * M[b] + Ma * ((a == b ? 1.0 : 0.0) - 3.0 * at[a] * at[b])) * iL2;
        H[r][c] = tau[r / 3] * tau[c / 3] * Q;
    }
}
// s1 = nA.nB, s2 = at.(nA x nB): gradients (12) and, if H1/H2 != nullptr, Hessians (12x12)
GF_HD inline void s_terms(const double* gA, const double* gB, const double* tau, double& s1, double& s2,
                               double* g1, double* g2, double (*H1)[12], double (*H2)[12], double& L, double* at) {
    double nA[3], nB[3], jA, jB, DnA[3][6], DnB[3][6], Dt[3][6];
    normal_derivs(gA, gA + 3, nA, jA, DnA);
    normal_derivs(gB, gB + 3, nB, jB, DnB);
    tangent_derivs(gA, gA + 3, tau, at, L, Dt);
    double cAB[3], cBt[3], ctA[3];
    cross3(nA, nB, cAB); cross3(nB, at, cBt); cross3(at, nA, ctA);
    s1 = dot3(nA, nB); s2 = dot3(at, cAB);
    for (int c = 0; c < 6; ++c) {
        g1[c] = DnA[0][c] * nB[0] + DnA[1][c] * nB[1] + DnA[2][c] * nB[2];
        g1[6 + c] = DnB[0][c] * nA[0] + DnB[1][c] * nA[1] + DnB[2][c] * nA[2];
        g2[c] = Dt[0][c] * cAB[0] + Dt[1][c] * cAB[1] + Dt[2][c] * cAB[2] + DnA[0][c] * cBt[0] + DnA[1][c] * cBt[1] + DnA[2][c] * cBt[2];
        g2[6 + c] = DnB[0][c] * ctA[0] + DnB[1][c] * ctA[1] + DnB[2][c] * ctA[2];
    }
    if (!H1) return;
    double H[6][6];
    hess_M_dot_n(gA, gA + 3, nA, jA, nB, H);
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H1[r][c] = H[r][c];
    hess_M_dot_n(gB, gB + 3, nB, jB, nA, H);
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H1[6 + r][6 + c] = H[r][c];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
        const double v = DnA[0][r] * DnB[0][c] + DnA[1][r] * DnB[1][c] + DnA[2][r] * DnB[2][c];
        H1[r][6 + c] = v; H1[6 + c][r] = v;
    }
    // H2
    double Ht[6][6], Hn[6][6];
    hess_M_dot_t(at, L, tau, cAB, Ht);
    hess_M_dot_n(gA, gA + 3, nA, jA, cBt, Hn);
    // X = -Dt^T skew(nB) DnA ; (skew(v) w = v x w)
    double SB_DnA[3][6], SA_DnB[3][6], St_DnB[3][6];
    for (int c = 0; c < 6; ++c) {
        double colA[3] = {DnA[0][c], DnA[1][c], DnA[2][c]}, colB[3] = {DnB[0][c], DnB[1][c], DnB[2][c]}, t0[3], t1[3], t2[3];
        cross3(nB, colA, t0); cross3(nA, colB, t1); cross3(at, colB, t2);
        for (int i = 0; i < 3; ++i) { SB_DnA[i][c] = t0[i]; SA_DnB[i][c] = t1[i]; St_DnB[i][c] = t2[i]; }
    }
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
        const double X = -(Dt[0][r] * SB_DnA[0][c] + Dt[1][r] * SB_DnA[1][c] + Dt[2][r] * SB_DnA[2][c]);
        const double Xt = -(Dt[0][c] * SB_DnA[0][r] + Dt[1][c] * SB_DnA[1][r] + Dt[2][c] * SB_DnA[2][r]);
        H2[r][c] = Ht[r][c] + Hn[r][c] + X + Xt;
        const double ab = Dt[0][r] * SA_DnB[0][c] + Dt[1][r] * SA_DnB[1][c] + Dt[2][r] * SA_DnB[2][c]
                        - (DnA[0][r] * St_DnB[0][c] + DnA[1][r] * St_DnB[1][c] + DnA[2][r] * St_DnB[2][c]);
        H2[r][6 + c] = ab; H2[6 + c][r] = ab;
    }
    hess_M_dot_n(gB, gB + 3, nB, jB, ctA, H);
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H2[6 + r][6 + c] = H[r][c];
}

// The two rotation measures of the penalty energy, e1 = nA . nB - NA . NB and e2 = (at x nA) . nB - (At x NA) . NB, from the displacement tangents
// dY = (dA1, dA2, dB1, dB2) = y_tangents - Y without cancellation (same idea as kl_strains: the measures are 1e-6 ... 1e-10, their terms O(1), and they are
// multiplied by alpha_r ~ 1e8): with n = N + dn, at = At + dat (differences of unit vectors in closed form)
//     e1 = NA . dnB + dnA . NB + dnA . dnB,     e2 = An . dnB + dan . NB + dan . dnB,  dan = At x dnA + dat x NA + dat x dnA,  An = At x NA
GF_HD inline void unit_diff(const double* X, const double* dx, double* U, double* dU, double& L) {       // U = X / |X|, dU = (X + dx) / |X + dx| - U
    L = sqrt(dot3(X, X));
    const double xd[3] = {X[0] + dx[0], X[1] + dx[1], X[2] + dx[2]};
    const double l = sqrt(dot3(xd, xd)), f = (2.0 * dot3(X, dx) + dot3(dx, dx)) / (l + L);
    for (int k = 0; k < 3; ++k) { U[k] = X[k] / L; dU[k] = (dx[k] - U[k] * f) / l; }
}
GF_HD inline void pen_rot_measures(const double* Y, const double* dY, const double* tau, double& e1, double& e2) {
    double N[2][3], dn[2][3];
    for (int sd = 0; sd < 2; ++sd) {
        const double *G1 = Y + 6 * sd, *G2 = G1 + 3, *d1 = dY + 6 * sd, *d2 = d1 + 3;
        double Nt[3], a[3], b[3], c[3], dl[3], J;
        cross3(G1, G2, Nt); cross3(G1, d2, a); cross3(d1, G2, b); cross3(d1, d2, c);
        for (int k = 0; k < 3; ++k) dl[k] = a[k] + b[k] + c[k];
        unit_diff(Nt, dl, N[sd], dn[sd], J);
    }
    double tr[3], dt_[3], At[3], dat[3], L;
    for (int k = 0; k < 3; ++k) { tr[k] = tau[0] * Y[k] + tau[1] * Y[3 + k]; dt_[k] = tau[0] * dY[k] + tau[1] * dY[3 + k]; }
    unit_diff(tr, dt_, At, dat, L);
    e1 = dot3(N[0], dn[1]) + dot3(dn[0], N[1]) + dot3(dn[0], dn[1]);
    double An[3], x0[3], x1[3], x2[3], dan[3];
    cross3(At, N[0], An); cross3(At, dn[0], x0); cross3(dat, N[0], x1); cross3(dat, dn[0], x2);
    for (int k = 0; k < 3; ++k) dan[k] = x0[k] + x1[k] + x2[k];
    e2 = dot3(An, dn[1]) + dot3(dan, N[1]) + dot3(dan, dn[1]);
}

enum : int { PB_GRAD = 0, PB_HYY = 18, PB_HYY_END = 18 + 324, PB_HYC = 18 + 324, PB_SIZE = 18 + 324 + 216, PB_EN = PB_SIZE };
constexpr int PB_STRIDE = PB_SIZE + 2;   // + energy + pad

// One mortar vertex: y[18] = (uA, gA1, gA2, uB, gB1, gB2), Y[12] = (GA1, GA2, GB1, GB2), dY[12] = the displacement tangents (gA1 - GA1, ...) summed from the
// displacement coefficients (the rotation measures e1, e2 are evaluated from them: pen_rot_measures).
// out: grad[18], Hyy[18][18], HyC[18][12] where HyC = Hyy[:, tangent cols] + HyY (the dR/dCP operator).
// grad_only: energy and gradient only (residual-only assemblies, functionals); the Hessian slots are left untouched.
// want: bit 0 = store Hyy, bit 1 = store HyC (the Newton pass needs only Hyy, linearize right after it only HyC)
GF_HD inline void penalty_point(const double* y, const double* Y, const double* dY, const double* tau, double ad, double ar, double dt, double* out, bool grad_only = false, int want = 3) {
    const int tan[12] = {3, 4, 5, 6, 7, 8, 12, 13, 14, 15, 16, 17};
    double s1, s2, S1, S2, g1[12], g2[12], G1[12], G2[12], L, Lr, at[3], At[3];
    double H1[12][12], H2[12][12];
    s_terms(y + 3, y + 12, tau, s1, s2, g1, g2, grad_only ? nullptr : H1, grad_only ? nullptr : H2, L, at);
    s_terms(Y, Y + 6, tau, S1, S2, G1, G2, nullptr, nullptr, Lr, At);
    double e1, e2;
    pen_rot_measures(Y, dY, tau, e1, e2);                     // = s1 - S1, s2 - S2 without the cancellation
    const double c0 = dt * Lr;
    double d[3] = {y[0] - y[9], y[1] - y[10], y[2] - y[11]};
    out[PB_EN] = c0 * (0.5 * ad * dot3(d, d) + 0.5 * ar * (e1 * e1 + e2 * e2));
    double* grad = out + PB_GRAD; double* Hyy = out + PB_HYY; double* HyC = out + PB_HYC;
    double gl[18];                                           // local copy: out is global memory on the device
    for (int k = 0; k < 18; ++k) gl[k] = 0.0;
    for (int k = 0; k < 3; ++k) { gl[k] = c0 * ad * d[k]; gl[9 + k] = -c0 * ad * d[k]; }
    for (int k = 0; k < 12; ++k) gl[tan[k]] = c0 * ar * (e1 * g1[k] + e2 * g2[k]);
    for (int k = 0; k < 18; ++k) grad[k] = gl[k];
    if (grad_only) return;
    // every entry is computed and stored exactly once (out is global memory on the device: no zero-fill / read-modify-write)
    // slot kinds of y: 0..2 uA, 3..8 tangents of A, 9..11 uB, 12..17 tangents of B;  tslot[r] = index into the 12 tangent slots or -1
    const int tslot[18] = {-1, -1, -1, 0, 1, 2, 3, 4, 5, -1, -1, -1, 6, 7, 8, 9, 10, 11};
    for (int r = 0; r < 18; ++r) {
        const int tr = tslot[r];
        double row[18];
        for (int c = 0; c < 18; ++c) {
            const int tc = tslot[c];
            double v = 0.0;
            if (tr < 0 && tc < 0) { if (r % 9 == c % 9) v = (r == c) ? c0 * ad : -c0 * ad; }            // displacement block: +-alpha_d I
            else if (tr >= 0 && tc >= 0) v = c0 * ar * (g1[tr] * g1[tc] + e1 * H1[tr][tc] + g2[tr] * g2[tc] + e2 * H2[tr][tc]);
            row[c] = v;
            if (want & 1) Hyy[r * 18 + c] = v;
        }
        // HyC[r][c], c over Y slots (GA1,GA2,GB1,GB2) == tangent slots of y
        if (want & 2) for (int c = 0; c < 12; ++c) {
            const double c0Y = c < 6 ? dt * tau[c / 3] * At[c % 3] : 0.0;
            double v = row[tan[c]] + gl[r] * c0Y / c0;
            if (tr >= 0) v -= c0 * ar * (g1[tr] * G1[c] + g2[tr] * G2[c]);
            HyC[r * 12 + c] = v;
        }
    }
}

// ---- moving intersections: derivative of the vertex gradient along one direction -------------
// d(R_pen)/d(xi) (nonmatching_opt.py:1042-1341) needs d(grad_y psi)/d(direction) for directions that move y and Y
// (the vertex slides on a patch: seeds from the second derivatives of the basis) or the curve tangent tau (a
// neighbouring vertex moves).  One forward-mode pass of the gradient in dual numbers gives it without Hessians.
struct Dual { double v, d; };
GF_HD __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
GF_HD __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
GF_HD __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
GF_HD __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.v * b.d + a.d * b.v}; }
GF_HD __forceinline__ Dual operator*(double a, Dual b) { return {a * b.v, a * b.d}; }
GF_HD __forceinline__ Dual operator*(Dual a, double b) { return {a.v * b, a.d * b}; }
GF_HD __forceinline__ Dual operator/(Dual a, Dual b) { const double q = a.v / b.v; return {q, (a.d - q * b.d) / b.v}; }
GF_HD __forceinline__ Dual dsqrt(Dual a) { const double s = sqrt(a.v); return {s, 0.5 * a.d / s}; }
GF_HD __forceinline__ double dsqrt(double a) { return sqrt(a); }

template <class T> GF_HD __forceinline__ void cross3t(const T* a, const T* b, T* c) {
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
template <class T> GF_HD __forceinline__ T dot3t(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> GF_HD __forceinline__ void unit_normal_t(const T* g1, const T* g2, T* n, T& j) {
    T t[3]; cross3t(g1, g2, t); j = dsqrt(dot3t(t, t));
    for (int k = 0; k < 3; ++k) n[k] = t[k] / j;
}
// gradient of (n(g1, g2) . v) wrt g1, g2 with v held fixed: w = (v - n (n.v)) / j, (g2 x w, w x g1)
template <class T> GF_HD __forceinline__ void normal_pullback_t(const T* g1, const T* g2, const T* n, T j, const T* v, T* o1, T* o2) {
    const T nv = dot3t(n, v); T w[3];
    for (int k = 0; k < 3; ++k) w[k] = (v[k] - n[k] * nv) / j;
    cross3t(g2, w, o1); cross3t(w, g1, o2);
}
// gradient (18) of the vertex penalty energy wrt y = (uA, gA1, gA2, uB, gB1, gB2); Y = (GA1, GA2, GB1, GB2); same energy
// as penalty_point (Herrema 2019: displacement jump + the two rotation measures, line element dt |tau . G_A|)
template <class T> GF_HD inline void penalty_grad_t(const T* y, const T* Y, const T* tau, double ad, double ar, double dt, T* gr) {
    const T *uA = y, *gA1 = y + 3, *gA2 = y + 6, *uB = y + 9, *gB1 = y + 12, *gB2 = y + 15;
    const T *GA1 = Y, *GA2 = Y + 3, *GB1 = Y + 6, *GB2 = Y + 9;
    T tref[3], tdef[3], At[3], at[3];
    for (int k = 0; k < 3; ++k) { tref[k] = tau[0] * GA1[k] + tau[1] * GA2[k]; tdef[k] = tau[0] * gA1[k] + tau[1] * gA2[k]; }
    const T L = dsqrt(dot3t(tref, tref)), lt = dsqrt(dot3t(tdef, tdef));
    for (int k = 0; k < 3; ++k) { At[k] = tref[k] / L; at[k] = tdef[k] / lt; }
    T nA[3], nB[3], NA[3], NB[3], jA, jB, JA, JB;
    unit_normal_t(gA1, gA2, nA, jA); unit_normal_t(gB1, gB2, nB, jB); unit_normal_t(GA1, GA2, NA, JA); unit_normal_t(GB1, GB2, NB, JB);
    T an[3], An[3]; cross3t(at, nA, an); cross3t(At, NA, An);
    const T e1 = dot3t(nA, nB) - dot3t(NA, NB), e2 = dot3t(an, nB) - dot3t(An, NB);
    const T c0 = dt * L;
    for (int k = 0; k < 3; ++k) { const T dk = uA[k] - uB[k]; gr[k] = ad * (c0 * dk); gr[9 + k] = -(ad * (c0 * dk)); }
    T s1A1[3], s1A2[3], s1B1[3], s1B2[3];
    normal_pullback_t(gA1, gA2, nA, jA, nB, s1A1, s1A2); normal_pullback_t(gB1, gB2, nB, jB, nA, s1B1, s1B2);
    T cx[3], v1[3], v2[3], s2A1[3], s2A2[3], s2B1[3], s2B2[3];
    cross3t(nA, nB, cx); cross3t(nB, at, v1); cross3t(at, nA, v2);
    normal_pullback_t(gA1, gA2, nA, jA, v1, s2A1, s2A2); normal_pullback_t(gB1, gB2, nB, jB, v2, s2B1, s2B2);
    const T ac = dot3t(at, cx);
    for (int k = 0; k < 3; ++k) { const T ptc = (cx[k] - at[k] * ac) / lt; s2A1[k] = s2A1[k] + tau[0] * ptc; s2A2[k] = s2A2[k] + tau[1] * ptc; }
    const T c1 = ar * (c0 * e1), c2 = ar * (c0 * e2);
    for (int k = 0; k < 3; ++k) {
        gr[3 + k] = c1 * s1A1[k] + c2 * s2A1[k]; gr[6 + k] = c1 * s1A2[k] + c2 * s2A2[k];
        gr[12 + k] = c1 * s1B1[k] + c2 * s2B1[k]; gr[15 + k] = c1 * s1B2[k] + c2 * s2B2[k];
    }
}

}  // namespace gf
