// gf_element_mfma2.hpp -- full-pass element kernel (p = 2, 3) with TWO waves per element, so that two waves share a SIMD.
//
// kl_element_mfma_kernel holds one wave per SIMD (254 VGPRs + 168 AGPRs: the 6 K + 9 dR/dc + 3 dR/dh accumulator tiles
// alone are 144 registers) and the matrix pipe is busy a third of the time: nothing runs under the pointwise phase, the row
// expansion or the block stores of the only resident wave.  Here the element's work is split between the two waves of its
// workgroup WITHOUT duplicating any of it:
//     wave 0: residual, K (6 tiles) and dR/dh (3 tiles)          -- expands the rows of G only
//     wave 1: dR/dc (9 tiles) and the body-force tiles           -- expands the rows of Hc = G + PzZ
// Both read the pointwise records of phase 1 from LDS (phase 1 itself: six lanes per Gauss point over both waves, one record
// column each).  Each wave needs at most 256 registers, so two of them -- of different elements -- are resident per SIMD and
// one's VALU / LDS / store phases sit under the other's MFMAs.  Same operand layout, DPP-fed T formation and element-block
// layout as kl_element_mfma_kernel (gf_element_mfma.hpp); the gather is unchanged.
// Reference path: the same integrals (GOLDFISH/nonmatching_opt.py:941-1015 via PENGoLINS assembly).
#pragma once
#include "gf_element_mfma.hpp"

namespace gf {

// one record column c = ic + 3 cc of shell_point_cols (the two-column routine of the one-wave kernel) per caller: six callers
// (ic = 0..2, cc = 0..1) fill the record; the caller with lead = true also writes the scalar part
template <bool REF, int CC>
GF_HD inline void shell_point_col(const double* z, const double* Z, double t, double E, double nu, int ic, const double* d, bool lead, double* im) {
    const double f3[3] = {1.0, 1.0, 2.0};
    double nt[3], n[3], Nt[3], N[3];
    cross3(z, z + 3, nt); const double j = sqrt(dot3(nt, nt)), ij = 1.0 / j;
    cross3(Z, Z + 3, Nt); const double Jn = sqrt(dot3(Nt, Nt)), iJn = 1.0 / Jn;
    for (int k = 0; k < 3; ++k) { n[k] = nt[k] * ij; N[k] = Nt[k] * iJn; }
    double C[6], dC[3][6], J;
    material(Z, Z + 3, E, nu, C, dC, J);
    double eps[3], kap[3];
    eps[0] = 0.5 * (dot3(z, z) - dot3(Z, Z));
    eps[1] = 0.5 * (dot3(z + 3, z + 3) - dot3(Z + 3, Z + 3));
    eps[2] = dot3(z, z + 3) - dot3(Z, Z + 3);
    for (int k = 0; k < 3; ++k) kap[k] = f3[k] * (dot3(Z + 6 + 3 * k, N) - dot3(z + 6 + 3 * k, n));
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
    double JZ[3];
    if (CC == 0) cross3(Z + 3, N, JZ); else cross3(N, Z, JZ);
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
    const double Sic[3] = {d[1] * (-v[2]) + d[2] * v[1], d[0] * v[2] + d[2] * (-v[0]), d[0] * (-v[1]) + d[1] * v[0]};
    const int c = ic + 3 * CC;
    double col[3], COL[3];
    if (CC == 0) { cross3(d, z + 3, col); cross3(d, Z + 3, COL); } else { cross3(z, d, col); cross3(Z, d, COL); }
    const double nc = dot3(n, col), NC = dot3(N, COL);
    double Dn[3], DN[3];
    for (int i = 0; i < 3; ++i) { Dn[i] = (col[i] - n[i] * nc) * ij; DN[i] = (COL[i] - N[i] * NC) * iJn; }
    double ez[3], eZ[3], bg[3], bG[3];
    if (CC == 0) { ez[0] = g1c; ez[1] = 0.0; ez[2] = g2c; eZ[0] = -G1c; eZ[1] = 0.0; eZ[2] = -G2c; }
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
    im[IM_JZJ + c] = dot3(d, JZ) / J;
    if constexpr (REF) {
        const double a0 = CC == 0 ? 2 * G1c : 0.0, a1 = CC == 0 ? 0.0 : 2 * G2c, a2 = CC == 0 ? G2c : G1c;
        double ce[3], cb2[3];
        symmv(C, eZ, ce); symmv(C, bG, cb2);
        for (int k = 0; k < 3; ++k) {
            im[IM_JDNV + 6 * k + c] = J * t * (dCe[0][k] * a0 + dCe[1][k] * a1 + dCe[2][k] * a2 + ce[k]);
            im[IM_JDMO + 6 * k + c] = J * t3 * (dCk[0][k] * a0 + dCk[1][k] * a1 + dCk[2][k] * a2 + cb2[k]);
        }
    }
    double QB[3];
    for (int a = 0; a < 3; ++a) QB[a] = Q[a][0] * col[0] + Q[a][1] * col[1] + Q[a][2] * col[2];
    for (int r = 0; r < 6; ++r) {
        double e[3] = {0, 0, 0}, Br[3];
        e[r % 3] = 1.0;
        if (r < 3) cross3(e, z + 3, Br); else cross3(z, e, Br);
        double h = dot3(Br, QB);
        if (CC == 1 && r < 3) h -= Sic[r];
        if (r <= c) im[IM_HMN + 6 * r - r * (r - 1) / 2 - r + c] = J * h;
    }
}

template <int P>
__global__ __launch_bounds__(128, 2) void kl_element_mfma2_kernel(DevModel M, int e_first, int flags, double* __restrict__ blk) {
    static_assert(P == 2 || P == 3, "one 16 x 16 tile: p <= 3");
    using Cfg = ElemCfg<P>;
    constexpr int P1 = P + 1, NB = P1 * P1, NG = NB, ND = 3 * NB, NGRP = (NG + 3) / 4;
    const int tid = threadIdx.x, role = tid >> 6, lane = tid & 63, x = lane & 15, kk = lane >> 4;
    const long long e = (long long)e_first + blockIdx.x;
    if (e >= M.nelem) return;
    const ElemDesc ed = M.edesc[e];
    const PatchDev& Pt = M.patches[ed.patch];
    // patch constants (E, nu, f[3], pd[3]: contiguous in PatchDev) staged in LDS: read from memory inside the Gauss-point loop they
    // are vector loads behind a vmcnt wait each (the compiler cannot move them across stores), held in registers they cost 16 VGPRs
    __shared__ double s_pc[8];
    if (threadIdx.x < 8) s_pc[threadIdx.x] = (&Pt.E)[threadIdx.x];
    const double* const pf = s_pc + 2; const double* const ppd = s_pc + 5;

    __shared__ __attribute__((aligned(16))) double s_g[4 * 3 * 16];
    double (*s_c)[3] = reinterpret_cast<double (*)[3]>(s_g);
    double (*s_d)[3] = reinterpret_cast<double (*)[3]>(s_g + 3 * NB);
    double* s_h = s_g + 6 * NB; double* s_w = s_g + 7 * NB;
    __shared__ double s_tu[P1 * 3 * P1], s_tv[P1 * 3 * P1], s_wg[2 * P1];
    __shared__ __attribute__((aligned(16))) double s_im[NG][IM_SIZE];

    // ---- phase 0: stage control-point data and 1-D tables
    if (tid < NB) {
        const long long g = ed.g0 + (tid % P1) + (long long)(tid / P1) * ed.nu;
        const double4 c4 = reinterpret_cast<const double4*>(M.cp4)[g];
        const double ux = M.u[3 * g], uy = M.u[3 * g + 1], uz = M.u[3 * g + 2];
        s_c[tid][0] = c4.x; s_c[tid][1] = c4.y; s_c[tid][2] = c4.z; s_w[tid] = c4.w;
        s_d[tid][0] = c4.x + ux; s_d[tid][1] = c4.y + uy; s_d[tid][2] = c4.z + uz;
        s_h[tid] = M.h[g];
    }
    if (tid >= 64 && tid - 64 < P1 * 3 * P1) { s_tu[tid - 64] = M.tab[ed.tabu + tid - 64]; s_tv[tid - 64] = M.tab[ed.tabv + tid - 64]; }
    if (tid >= 64 + 48 && tid - 112 < P1) { s_wg[tid - 112] = M.tab[ed.wu + tid - 112]; s_wg[P1 + tid - 112] = M.tab[ed.wv + tid - 112]; }
    __syncthreads();

#ifndef GF_MFMA2_PHASE1_SPLIT
    // ---- phase 1 on wave 0: three lanes per Gauss point, two record columns each (as in kl_element_mfma_kernel).  Spreading it over
    //      both waves (six lanes per Gauss point, -DGF_MFMA2_PHASE1_SPLIT) makes BOTH waves execute the scalar part of the closed
    //      forms: more wave instructions in total, and the SIMD's issue slots are what this kernel is short of.
    if (role == 0) {
        const int gp = x < NG ? x : NG - 1, ic = kk < 3 ? kk : 0, gu = gp % P1, gv = gp / P1;
        const bool act = kk < 3 && x < NG;
        double* im = s_im[gp];
        double W[6], th = 0.0;
        if (act) {
            double Ac[6], Ad[6];
            for (int k = 0; k < 6; ++k) { W[k] = 0.0; Ac[k] = 0.0; Ad[k] = 0.0; }
            double U[3][P1];
            for (int d = 0; d < 3; ++d) for (int j = 0; j < P1; ++j) U[d][j] = s_tu[(gu * 3 + d) * P1 + j];
#pragma unroll
            for (int jv = 0; jv < P1; ++jv) {
                const double v0 = s_tv[(gv * 3 + 0) * P1 + jv], v1 = s_tv[(gv * 3 + 1) * P1 + jv], v2 = s_tv[(gv * 3 + 2) * P1 + jv];
                double S[3][3], Sh = 0.0;
                for (int q = 0; q < 3; ++q) for (int d = 0; d < 3; ++d) S[q][d] = 0.0;
#pragma unroll
                for (int ju = 0; ju < P1; ++ju) {
                    const int a = ju + P1 * jv;
                    const double qv[3] = {s_c[a][ic], s_d[a][ic], s_w[a]};
                    for (int q = 0; q < 3; ++q) for (int d = 0; d < 3; ++d) S[q][d] += U[d][ju] * qv[q];
                    Sh += U[0][ju] * s_h[a];
                }
                th += v0 * Sh;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    double* A = q == 0 ? Ac : (q == 1 ? Ad : W);
                    A[0] += v0 * S[q][0]; A[1] += v0 * S[q][1]; A[2] += v1 * S[q][0];
                    A[3] += v0 * S[q][2]; A[4] += v2 * S[q][0]; A[5] += v1 * S[q][1];
                }
            }
            W[0] = 1.0 / W[0];
            double R[6];
            rationalize6(Ac, W, R);
            for (int mm = 0; mm < 5; ++mm) im[3 * mm + ic] = R[mm + 1];
            rationalize6(Ad, W, R);
            for (int mm = 0; mm < 5; ++mm) im[15 + 3 * mm + ic] = R[mm + 1];
        }
        wave_lds_sync();
        double z[15], Z[15];
        if (act) for (int k = 0; k < 15; ++k) { Z[k] = im[k]; z[k] = im[15 + k]; }
        wave_lds_sync();
        if (act) {
            const double dsel[3] = {ic == 0 ? 1.0 : 0.0, ic == 1 ? 1.0 : 0.0, ic == 2 ? 1.0 : 0.0};
            shell_point_cols<true>(z, Z, th, s_pc[0], s_pc[1], ic, dsel, kk == 0, im);
            if (kk == 0) {
                for (int k = 0; k < 6; ++k) im[IM_W + k] = W[k];
                im[IM_WQ] = s_wg[gu] * s_wg[P1 + gv];
            }
        }
    }
    __syncthreads();
#else
    // ---- phase 1: six lanes per Gauss point (gp = tid % 16, part = tid / 16 < 6 = ic + 3 cc): kinematics + pointwise closed forms.
    //      Lane (gp, ic, cc) sums component ic of the reference (cc = 0) or deformed (cc = 1) control points (sum factorisation,
    //      quotient rule), the six lanes exchange their components through the Gauss point's record, and each produces one
    //      record column c = ic + 3 cc.
    {
        const int part = tid >> 4, gp = (tid & 15) < NG ? (tid & 15) : NG - 1, ic = part % 3, cc = part < 3 ? 0 : 1, gu = gp % P1, gv = gp / P1;
        const bool act = part < 6 && (tid & 15) < NG;
        double* im = s_im[gp];
        double W[6], th = 0.0;
        if (act) {
            double Aq[6];
            for (int k = 0; k < 6; ++k) { W[k] = 0.0; Aq[k] = 0.0; }
            double U[3][P1];
            for (int d = 0; d < 3; ++d) for (int j = 0; j < P1; ++j) U[d][j] = s_tu[(gu * 3 + d) * P1 + j];
#pragma unroll
            for (int jv = 0; jv < P1; ++jv) {
                const double v0 = s_tv[(gv * 3 + 0) * P1 + jv], v1 = s_tv[(gv * 3 + 1) * P1 + jv], v2 = s_tv[(gv * 3 + 2) * P1 + jv];
                double S[2][3], Sh = 0.0;
                for (int q = 0; q < 2; ++q) for (int d = 0; d < 3; ++d) S[q][d] = 0.0;
#pragma unroll
                for (int ju = 0; ju < P1; ++ju) {
                    const int a = ju + P1 * jv;
                    const double qv[2] = {cc == 0 ? s_c[a][ic] : s_d[a][ic], s_w[a]};
                    for (int q = 0; q < 2; ++q) for (int d = 0; d < 3; ++d) S[q][d] += U[d][ju] * qv[q];
                    Sh += U[0][ju] * s_h[a];
                }
                th += v0 * Sh;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    double* A = q == 0 ? Aq : W;
                    A[0] += v0 * S[q][0]; A[1] += v0 * S[q][1]; A[2] += v1 * S[q][0];
                    A[3] += v0 * S[q][2]; A[4] += v2 * S[q][0]; A[5] += v1 * S[q][1];
                }
            }
            W[0] = 1.0 / W[0];
            double R[6];
            rationalize6(Aq, W, R);
            for (int mm = 0; mm < 5; ++mm) im[15 * cc + 3 * mm + ic] = R[mm + 1];
        }
        __syncthreads();
        double z[15], Z[15];
        if (act) for (int k = 0; k < 15; ++k) { Z[k] = im[k]; z[k] = im[15 + k]; }
        __syncthreads();                                   // all six lanes hold z, Z before the record overwrites the exchange slots
        if (act) {
            const double dsel[3] = {ic == 0 ? 1.0 : 0.0, ic == 1 ? 1.0 : 0.0, ic == 2 ? 1.0 : 0.0};
            if (cc == 0) shell_point_col<true, 0>(z, Z, th, s_pc[0], s_pc[1], ic, dsel, part == 0, im);
            else shell_point_col<true, 1>(z, Z, th, s_pc[0], s_pc[1], ic, dsel, false, im);
            if (part == 0) {
                for (int k = 0; k < 6; ++k) im[IM_W + k] = W[k];
                im[IM_WQ] = s_wg[gu] * s_wg[P1 + gv];
            }
        }
    }
    __syncthreads();

#endif

    // ---- lane constants of the row expansion (see kl_element_mfma_kernel)
    const bool tang = x < 6;
    const int r = x < 15 ? x : 14, mr = r / 3, ir = r - 3 * mr;
    const int kr = mr >= 2 ? mr - 2 : 0, rt = tang ? r : 0;
    const double mt = tang ? 1.0 : 0.0, m0 = (mr == 0) ? 1.0 : 0.0, m1 = (mr == 1) ? 1.0 : 0.0;
    const double f3c = tang ? 0.0 : ((kr == 2) ? 2.0 : 1.0);
    const double ck[3] = {(!tang && kr == 0) ? 1.0 : 0.0, (!tang && kr == 1) ? 1.0 : 0.0, (!tang && kr == 2) ? 1.0 : 0.0};
    const double dij[3] = {(tang && ir == 0) ? 1.0 : 0.0, (tang && ir == 1) ? 1.0 : 0.0, (tang && ir == 2) ? 1.0 : 0.0};
    const int oE2 = IM_G + (tang ? 3 * (1 - mr) + ir : 0);
    const int oJ0 = IM_JNV + (mr == 0 ? 0 : 2), oJ1 = IM_JNV + (mr == 1 ? 1 : 2);
    int oX[6];
    for (int s = 0; s < 6; ++s) oX[s] = tang ? IM_HMN + hmn_idx(r, s) : IM_DN + 6 * ir + s;

    const bool has_bf = (pf[0] != 0.0) || (pf[1] != 0.0) || (pf[2] != 0.0);
    const int xb = x < NB ? x : 0, ju = xb % P1, jv = xb / P1;
    const double bval = x < NB ? 1.0 : 0.0;
    double* const out = blk + (size_t)blockIdx.x * Cfg::BLK;
    constexpr int QI[6] = {0, 0, 0, 1, 1, 2}, QJ[6] = {0, 1, 2, 1, 2, 2};

    // basis function x at the Gauss point of this lane's group and the common part of the row expansion
    auto basis = [&](const double* im, int gu, int gv, double (&phi)[5], double& R0, double& n0) {
        const double u0 = s_tu[(gu * 3 + 0) * P1 + ju], u1 = s_tu[(gu * 3 + 1) * P1 + ju], u2 = s_tu[(gu * 3 + 2) * P1 + ju];
        const double v0 = s_tv[(gv * 3 + 0) * P1 + jv], v1 = s_tv[(gv * 3 + 1) * P1 + jv], v2 = s_tv[(gv * 3 + 2) * P1 + jv];
        const double Nb[6] = {u0 * v0, u1 * v0, u0 * v1, u2 * v0, u0 * v2, u1 * v1};
        double R[6];
        rationalize6(Nb, im + IM_W, R);
        for (int k = 0; k < 5; ++k) phi[k] = bval * R[k + 1];
        R0 = bval * R[0]; n0 = bval * Nb[0];
    };
    // row r of G (HC = false) or of Hc = G + PzZ (HC = true)
    auto expand = [&](const double* im, double (&row)[15], auto HC_) {
        constexpr bool HC = decltype(HC_)::value;
        const double gr = im[IM_G + rt], e0 = m0 * gr, e1 = m1 * gr, e2 = mt * im[oE2];
        const double fnr = f3c * im[IM_N + ir];
        const double b0 = mt * im[IM_BG + rt] + ck[0] * fnr, b1 = mt * im[IM_BG + 6 + rt] + ck[1] * fnr, b2 = mt * im[IM_BG + 12 + rt] + ck[2] * fnr;
        const double pzr = im[IM_PZ + r], xfac = mt + (1.0 - mt) * im[IM_JMOF + kr];
        const double jn[2] = {im[oJ0], im[oJ1]};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            double g = e0 * im[IM_CEZ + s] + e1 * im[IM_CEZ + 6 + s] + e2 * im[IM_CEZ + 12 + s]
                     + b0 * im[IM_CBG + s] + b1 * im[IM_CBG + 6 + s] + b2 * im[IM_CBG + 12 + s] - xfac * im[oX[s]] + dij[s % 3] * jn[s / 3];
            if constexpr (HC) g += pzr * im[IM_JZJ + s] + e0 * im[IM_JDNV + s] + e1 * im[IM_JDNV + 6 + s] + e2 * im[IM_JDNV + 12 + s]
                                 - (b0 * im[IM_JDMO + s] + b1 * im[IM_JDMO + 6 + s] + b2 * im[IM_JDMO + 12 + s]);
            row[s] = g;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double fc = (c == 2) ? 2.0 : 1.0;
            const double gam = fc * (b0 * im[IM_CT3 + sym3(0, c)] + b1 * im[IM_CT3 + sym3(1, c)] + b2 * im[IM_CT3 + sym3(2, c)]);
            const double alpha = mt * (fc * im[IM_CBG + 6 * c + rt]) + (1.0 - mt) * gam, beta = mt * im[IM_JMOF + c];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                double g = im[IM_N + jj] * alpha - beta * im[IM_DN + 6 * jj + rt];
                if constexpr (HC) g -= gam * im[IM_NB + jj];
                row[6 + 3 * c + jj] = g;
            }
        }
        dpp_source_fence(row);
    };

    if (role == 0) {
        // ================= wave 0: residual, K, dR/dh =================
        const bool doK = (flags & GF_ASM_K_BIT) != 0, doH = (flags & GF_ASM_H_BIT) != 0;
        gf_d4 accK[6], accH[3];
        for (int q = 0; q < 6; ++q) accK[q] = gf_d4{0, 0, 0, 0};
        for (int q = 0; q < 3; ++q) accH[q] = gf_d4{0, 0, 0, 0};
        double accR[3] = {0.0, 0.0, 0.0};
        for (int grp = 0; grp < NGRP; ++grp) {
            const int gp = 4 * grp + kk, gpc = gp < NG ? gp : NG - 1;
            const double* im = s_im[gpc];
            const double wq = gp < NG ? im[IM_WQ] : 0.0;
            double phi[5], R0, n0;
            basis(im, gpc % P1, gpc / P1, phi, R0, n0);
            double gR[15];
            for (int s = 0; s < 15; ++s) gR[s] = 0.0;
            if (doK) expand(im, gR, std::false_type{});
            {
                const double ls = has_bf ? load_scalar(im, ppd) : 0.0;
                for (int i = 0; i < 3; ++i) {
                    double rz = 0.0;
                    for (int m = 0; m < 5; ++m) rz += phi[m] * im[IM_PZ + 3 * m + i];
                    accR[i] += wq * (rz - ls * pf[i] * R0);
                }
            }
            double pb[5];
            for (int m = 0; m < 5; ++m) pb[m] = wq * phi[m];
            if (doH) {
                double nn = 0.0;
                for (int k = 0; k < 3; ++k) nn += phi[2 + k] * im[IM_JCK4 + k] * (k == 2 ? 2.0 : 1.0);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double g1i = im[IM_G + i], g2i = im[IM_G + 3 + i];
                    double rh = phi[0] * (im[IM_JCE] * g1i + im[IM_JCE + 2] * g2i) + phi[1] * (im[IM_JCE + 1] * g2i + im[IM_JCE + 2] * g1i);
                    for (int k = 0; k < 3; ++k) rh -= im[IM_JCK4 + k] * (phi[0] * im[IM_BG + 6 * k + i] + phi[1] * im[IM_BG + 6 * k + 3 + i]);
                    rh -= im[IM_N + i] * nn;
                    accH[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(wq * rh, n0, accH[i], 0, 0, 0);
                }
            }
            if (doK) {
                static_for<5>([&](auto m_) {
                    constexpr int m = decltype(m_)::value;
                    double t[6];
                    static_for<6>([&](auto q_) { constexpr int q = decltype(q_)::value; t[q] = row_dot<3 * m + QI[q], QJ[q]>(gR, pb); });
                    mfma_hazard_gap(t);
#pragma unroll
                    for (int q = 0; q < 6; ++q) accK[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[m], t[q], accK[q], 0, 0, 0);
                });
            }
        }
        // residual: sum the four Gauss-point groups (s_g: the staging data is dead since phase 1; only this wave uses it now)
        wave_lds_sync();
        for (int i = 0; i < 3; ++i) s_g[(kk * 16 + x) * 3 + i] = accR[i];
        wave_lds_sync();
        if (lane < ND && (flags & GF_ASM_R_BIT)) out[Cfg::OFF_R + lane] = s_g[lane] + s_g[48 + lane] + s_g[96 + lane] + s_g[144 + lane];
        const int b = x;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int a = kk + 4 * rr;
            if (a >= NB || b >= NB) continue;
            if (doK) {
#pragma unroll
                for (int ij = 0; ij < 6; ++ij) {
                    const int i = QI[ij], j = QJ[ij];
                    out[Cfg::OFF_K + (3 * a + i) * ND + 3 * b + j] = accK[ij][rr];
                    if (i < j) out[Cfg::OFF_K + (3 * b + j) * ND + 3 * a + i] = accK[ij][rr];
                }
            }
            if (doH) {
#pragma unroll
                for (int i = 0; i < 3; ++i) out[Cfg::OFF_H + (3 * a + i) * NB + b] = accH[i][rr];
            }
        }
    } else {
        // ================= wave 1: dR/dc (and the body-force tiles) =================
        const bool doC = (flags & GF_ASM_C_BIT) != 0;
        if (!doC) return;
        gf_d4 accC[9];
        for (int q = 0; q < 9; ++q) accC[q] = gf_d4{0, 0, 0, 0};
        gf_d4 accB[3] = {gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}};
        for (int grp = 0; grp < NGRP; ++grp) {
            const int gp = 4 * grp + kk, gpc = gp < NG ? gp : NG - 1;
            const double* im = s_im[gpc];
            const double wq = gp < NG ? im[IM_WQ] : 0.0;
            double phi[5], R0, n0;
            basis(im, gpc % P1, gpc / P1, phi, R0, n0);
            (void)n0;
            double hR[15];
            expand(im, hR, std::true_type{});
            double pb[5];
            for (int m = 0; m < 5; ++m) pb[m] = wq * phi[m];
            static_for<5>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                double t[9];
                static_for<9>([&](auto q_) { constexpr int q = decltype(q_)::value; t[q] = row_dot<3 * m + q / 3, q % 3>(hR, pb); });
                mfma_hazard_gap(t);
#pragma unroll
                for (int q = 0; q < 9; ++q) accC[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[m], t[q], accC[q], 0, 0, 0);
            });
            if (has_bf) {
                const LoadGeom lg = load_geom(im, ppd);
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const double jz = load_dz_dot(im, ppd, lg, f, pb[0], pb[1]);
                    accB[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(R0, jz, accB[f], 0, 0, 0);
                }
            }
        }
        if (has_bf) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int f = 0; f < 3; ++f) accC[3 * i + f] -= pf[i] * accB[f];
        }
        const int b = x;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int a = kk + 4 * rr;
            if (a >= NB || b >= NB) continue;
#pragma unroll
            for (int q = 0; q < 9; ++q) out[Cfg::OFF_C + (3 * a + q / 3) * ND + 3 * b + q % 3] = accC[q][rr];
        }
    }
}

}  // namespace gf
