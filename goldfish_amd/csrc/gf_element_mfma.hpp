// gf_element_mfma.hpp -- element kernel (p = 3, and p = 2 with a padded tile) whose a-b contraction runs on the FP64 matrix pipe.
//
// Why: the VALU kernel (kl_element_kernel) is bound by per-wave instruction issue (~900 instructions per
// Gauss point and wave for ~340 FP64 operations; the T tile makes a round trip through LDS).  With (p+1)^2 = 16
// basis functions the element matrices are exactly one 16x16 MFMA tile per (i,j) component:
//     K^{ij}[a][b] = sum_gp sum_m  phi_a[gp][m] * T^{ij}_b[gp][m],   T^{ij}_b[gp][m] = w_gp sum_m' G[(m,i),(m',j)] phi_b[gp][m']
// and v_mfma_f64_16x16x4 contracts 4 Gauss points at once (k = Gauss point of the lane's 16-lane group).
// Lane (x, kk) = (lane % 16, lane / 16) holds phi_x at Gauss point 4*grp + kk in registers: it supplies the A operand
// (row a = x) and computes the B operand T_b (column b = x) on the fly.  Lane x < 15 also expands row x of the pointwise
// Hessians G, Hc of its group's Gauss point and KEEPS it in registers; the FMAs that form T read those entries from the
// owning lane through DPP row_newbcast (v_fmac_f64_dpp) -- no T tile, no expanded Hessian in LDS, no cross-wave barrier
// (one wave per element), 300 MFMAs instead of 4800 FMA instructions per element.
// tools/ubench_mfma_loop.hip measures this inner loop at 68 cycles per (component, m) unit and checks the operand layout:
//     A[i][k]: lane = i + 16 k      B[k][j]: lane = j + 16 k      D[i][j]: lane = j + 16 (i % 4), register i / 4.
// Reference path: the same integrals as kl_element_kernel (GOLDFISH/nonmatching_opt.py:941-1015 via PENGoLINS assembly).
#pragma once
#include <type_traits>

namespace gf {

typedef double gf_d4 __attribute__((ext_vector_type(4)));

// t += (value of g held by lane LANE of this lane's 16-lane row) * p.   gfx950 has the DPP form of v_fmac_f64 (row_newbcast
// only); the compiler does not fold a DPP move into FP64 FMAs, hence the inline assembly.  The DPP source must not have been
// written by a VALU instruction in the two preceding slots (dpp_source_fence below).
template <int LANE> __device__ __forceinline__ void fmac_bcast(double& t, double g, double p) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(g), "v"(p), "n"(LANE));
}
// sum_m' (entry [3 m' + J] of the row held by lane LANE of this lane's 16-lane group) * p[m']
template <int LANE, int J> __device__ __forceinline__ double row_dot(const double (&g)[15], const double (&p)[5]) {
    double t = 0.0;
    fmac_bcast<LANE>(t, g[J], p[0]);
    fmac_bcast<LANE>(t, g[3 + J], p[1]);
    fmac_bcast<LANE>(t, g[6 + J], p[2]);
    fmac_bcast<LANE>(t, g[9 + J], p[3]);
    fmac_bcast<LANE>(t, g[12 + J], p[4]);
    return t;
}
// The row registers are written by VALU instructions and read through DPP by inline assembly the hazard recogniser cannot
// see: tying them to a 2-wait-state nop keeps every producer in front of it and every DPP read behind it.
__device__ __forceinline__ void dpp_source_fence(double (&g)[15]) {
    asm volatile("s_nop 1" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]),
                             "+v"(g[8]), "+v"(g[9]), "+v"(g[10]), "+v"(g[11]), "+v"(g[12]), "+v"(g[13]), "+v"(g[14]));
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}
// One wave per workgroup: its LDS operations execute in order, so cross-lane hand-over through LDS needs neither s_barrier
// nor the global-memory fence of __syncthreads().
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// VALU result -> MFMA operand needs two wait states; the FMAs above are opaque to the compiler's hazard recogniser
// (the operands are tied to the nop so that it stays between the last FMA and the first MFMA of a batch)
__device__ __forceinline__ void mfma_hazard_gap(double (&t)[6]) {
    asm volatile("s_nop 1" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]));
}
__device__ __forceinline__ void mfma_hazard_gap(double (&t)[9]) {
    asm volatile("s_nop 1" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]));
}

// P = 3: the tile is exactly full.  P = 2: 9 basis functions / 9 Gauss points use the same code with the lanes x >= 9 and the
// Gauss-point slots >= 9 of the third group padded by zeros (a third of the tile, still far fewer instructions than the VALU path).
template <int P, bool WITHC = true>        // WITHC = false: Newton pass (no dR/dCP): the Hc row and its MFMAs are compiled out
__global__ __launch_bounds__(64) void kl_element_mfma_kernel(DevModel M, int e_first, int flags, double* __restrict__ blk) {
    static_assert(P == 2 || P == 3, "one 16 x 16 tile: p <= 3");
    using Cfg = ElemCfg<P>;
    constexpr int P1 = P + 1, NB = P1 * P1, NG = NB, ND = 3 * NB, NGRP = (NG + 3) / 4;
    const int tid = threadIdx.x, x = tid & 15, kk = tid >> 4;
    const long long e = (long long)e_first + blockIdx.x;
    if (e >= M.nelem) return;
    const ElemDesc ed = M.edesc[e];
    const PatchDev& Pt = M.patches[ed.patch];
    // patch constants (E, nu, f[3], pd[3]: contiguous in PatchDev) staged in LDS: read from memory inside the Gauss-point loop they
    // are vector loads behind a vmcnt wait each (the compiler cannot move them across stores), held in registers they cost 16 VGPRs
    __shared__ double s_pc[8];
    if (threadIdx.x < 8) s_pc[threadIdx.x] = (&Pt.E)[threadIdx.x];
    const double* const pf = s_pc + 2; const double* const ppd = s_pc + 5;

    __shared__ __attribute__((aligned(16))) double s_g[4 * 3 * 16];  // control-point staging (phases 0-1), residual reduction at the end
    double (*s_c)[3] = reinterpret_cast<double (*)[3]>(s_g);
    double (*s_d)[3] = reinterpret_cast<double (*)[3]>(s_g + 3 * NB);
    double* s_h = s_g + 6 * NB; double* s_w = s_g + 7 * NB;
    __shared__ double s_tu[P1 * 3 * P1], s_tv[P1 * 3 * P1], s_wg[2 * P1];
    __shared__ __attribute__((aligned(16))) double s_im[NG][IM_SIZE];

    unsigned long long tstamp = 0; (void)tstamp;
#ifdef GF_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tstamp = clock64();
#endif
    // ---- phase 0: stage control-point data and 1-D tables (the loads are issued first; the lane constants of the row
    //      expansion further down do not depend on them)
    double4 c4 = {0, 0, 0, 0}; double ux = 0, uy = 0, uz = 0, hh = 0, ttu = 0, ttv = 0, twu = 0, twv = 0;
    if (tid < NB) {
        const long long g = ed.g0 + (tid % P1) + (long long)(tid / P1) * ed.nu;
        c4 = reinterpret_cast<const double4*>(M.cp4)[g];
        ux = M.u[3 * g]; uy = M.u[3 * g + 1]; uz = M.u[3 * g + 2];
        hh = M.h[g];
    }
    if (tid < P1 * 3 * P1) { ttu = M.tab[ed.tabu + tid]; ttv = M.tab[ed.tabv + tid]; }
    if (tid < P1) { twu = M.tab[ed.wu + tid]; twv = M.tab[ed.wv + tid]; }
    // ---- lane constants of the row expansion: lane x < 15 expands row r = x = 3 m_r + i_r of G = Pzz and Hc = Pzz + PzZ.
    //      Tangent rows (r < 6) and curvature rows share ONE code path: the closed forms have the same shape
    //          G[r][s]  = sum_k e_k(r) CEZ[k][s] + b_k(r) CBG[k][s] - X(r,s) + delta      (tangent columns s < 6)
    //          PzZ[r][s] = Pz[r] JZJ[s] + sum_k e_k(r) JDNV[k][s] - b_k(r) JDMO[k][s]
    //      with e_k = 0, b_k = f_k n_i delta_{k,k_r}, X = Jmo_k dn_i/dg_s on curvature rows (kl_point.hpp ez_entry/bz_entry),
    //      so the row type only selects lane-constant masks and offsets -- no divergent branches.
    const bool tang = x < 6;
    const int r = x < 15 ? x : 14, mr = r / 3, ir = r - 3 * mr;
    const int kr = mr >= 2 ? mr - 2 : 0, rt = tang ? r : 0;                   // curvature component of a curvature row; tangent row index (clamped)
    const double mt = tang ? 1.0 : 0.0, m0 = (mr == 0) ? 1.0 : 0.0, m1 = (mr == 1) ? 1.0 : 0.0;
    const double f3c = tang ? 0.0 : ((kr == 2) ? 2.0 : 1.0);
    const double ck[3] = {(!tang && kr == 0) ? 1.0 : 0.0, (!tang && kr == 1) ? 1.0 : 0.0, (!tang && kr == 2) ? 1.0 : 0.0};
    const double dij[3] = {(tang && ir == 0) ? 1.0 : 0.0, (tang && ir == 1) ? 1.0 : 0.0, (tang && ir == 2) ? 1.0 : 0.0};
    const int oE2 = IM_G + (tang ? 3 * (1 - mr) + ir : 0);
    const int oJ0 = IM_JNV + (mr == 0 ? 0 : 2), oJ1 = IM_JNV + (mr == 1 ? 1 : 2);
    int oX[6];
    for (int s = 0; s < 6; ++s) oX[s] = tang ? IM_HMN + hmn_idx(r, s) : IM_DN + 6 * ir + s;
    if (tid < NB) {
        s_c[tid][0] = c4.x; s_c[tid][1] = c4.y; s_c[tid][2] = c4.z; s_w[tid] = c4.w;
        s_d[tid][0] = c4.x + ux; s_d[tid][1] = c4.y + uy; s_d[tid][2] = c4.z + uz;
        s_h[tid] = hh;
    }
    if (tid < P1 * 3 * P1) { s_tu[tid] = ttu; s_tv[tid] = ttv; }
    if (tid < P1) { s_wg[tid] = twu; s_wg[P1 + tid] = twv; }
    wave_lds_sync();
    GF_STAMP(0, tstamp);

    // ---- phase 1: three lanes per Gauss point (gp = x, part ic = kk < 3): kinematics + pointwise closed forms --------
    // Lane (gp, ic) sums component ic of the reference and deformed control points (sum factorisation over the tensor-product
    // basis: per row jv of control points the three u-sums, then the six (du, dv) combinations; the rational derivatives
    // follow by the quotient rule, rationalize6 being linear in the B-spline values), the three lanes exchange their
    // components through the Gauss point's (not yet written) record, and each produces the record columns c = ic, 3 + ic.
    {
        const int gp = x < NG ? x : NG - 1, ic = kk < 3 ? kk : 0, gu = gp % P1, gv = gp / P1;
        const bool act = kk < 3 && x < NG;
        double* im = s_im[gp];
        double W[6], t = 0.0;
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
                t += v0 * Sh;
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
        wave_lds_sync();                                   // all three lanes hold z, Z before the record overwrites the exchange slots
        if (act) {
            const double dsel[3] = {ic == 0 ? 1.0 : 0.0, ic == 1 ? 1.0 : 0.0, ic == 2 ? 1.0 : 0.0};
            shell_point_cols<WITHC>(z, Z, t, s_pc[0], s_pc[1], ic, dsel, kk == 0, im);
            if (kk == 0) {
                for (int k = 0; k < 6; ++k) im[IM_W + k] = W[k];
                im[IM_WQ] = s_wg[gu] * s_wg[P1 + gv];
            }
        }
    }
    wave_lds_sync();
    GF_STAMP(1, tstamp);

    const bool doK = (flags & GF_ASM_K_BIT) != 0, doC = WITHC && (flags & GF_ASM_C_BIT) != 0, doH = (flags & GF_ASM_H_BIT) != 0;
    const bool has_bf = (pf[0] != 0.0) || (pf[1] != 0.0) || (pf[2] != 0.0);
    const int xb = x < NB ? x : 0, ju = xb % P1, jv = xb / P1;
    const double bval = x < NB ? 1.0 : 0.0;                  // lanes beyond the basis functions contribute zero rows / columns

    gf_d4 accK[6], accC[9], accH[3];
    for (int q = 0; q < 6; ++q) accK[q] = gf_d4{0, 0, 0, 0};
    for (int q = 0; q < 9; ++q) accC[q] = gf_d4{0, 0, 0, 0};
    for (int q = 0; q < 3; ++q) accH[q] = gf_d4{0, 0, 0, 0};
    double accR[3] = {0.0, 0.0, 0.0};
    gf_d4 accB[3] = {gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}};   // body force: sum_gp R_a (dJ/dZ . phi_b)_f, scaled by -f_i at the end

    GF_STAMP(2, tstamp);
    for (int grp = 0; grp < NGRP; ++grp) {
        const int gp = 4 * grp + kk, gpc = gp < NG ? gp : NG - 1, gu = gpc % P1, gv = gpc / P1;   // Gauss point of this lane's group
        const double* im = s_im[gpc];
        const double wq = gp < NG ? im[IM_WQ] : 0.0;        // padded Gauss-point slots contribute nothing
        // -- basis function x at this Gauss point (registers)
        double phi[5], R0, n0;
        {
            const double u0 = s_tu[(gu * 3 + 0) * P1 + ju], u1 = s_tu[(gu * 3 + 1) * P1 + ju], u2 = s_tu[(gu * 3 + 2) * P1 + ju];
            const double v0 = s_tv[(gv * 3 + 0) * P1 + jv], v1 = s_tv[(gv * 3 + 1) * P1 + jv], v2 = s_tv[(gv * 3 + 2) * P1 + jv];
            const double Nb[6] = {u0 * v0, u1 * v0, u0 * v1, u2 * v0, u0 * v2, u1 * v1};
            double R[6];
            rationalize6(Nb, im + IM_W, R);
            for (int k = 0; k < 5; ++k) phi[k] = bval * R[k + 1];
            R0 = bval * R[0]; n0 = bval * Nb[0];
        }
        GF_STAMP(3, tstamp);
        // -- row r of G and Hc at this Gauss point
        double gR[15], hR[15];                     // row r of G and Hc; entry (m', j) at [3 m' + j]
        for (int s = 0; s < 15; ++s) { gR[s] = 0.0; hR[s] = 0.0; }
        if (doK || doC) {
            const double gr = im[IM_G + rt], e0 = m0 * gr, e1 = m1 * gr, e2 = mt * im[oE2];
            const double fnr = f3c * im[IM_N + ir];
            const double b0 = mt * im[IM_BG + rt] + ck[0] * fnr, b1 = mt * im[IM_BG + 6 + rt] + ck[1] * fnr, b2 = mt * im[IM_BG + 12 + rt] + ck[2] * fnr;
            const double pzr = im[IM_PZ + r], xfac = mt + (1.0 - mt) * im[IM_JMOF + kr];
            const double jn[2] = {im[oJ0], im[oJ1]};
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const double g = e0 * im[IM_CEZ + s] + e1 * im[IM_CEZ + 6 + s] + e2 * im[IM_CEZ + 12 + s]
                               + b0 * im[IM_CBG + s] + b1 * im[IM_CBG + 6 + s] + b2 * im[IM_CBG + 12 + s] - xfac * im[oX[s]] + dij[s % 3] * jn[s / 3];
                gR[s] = g;
                if constexpr (WITHC) {
                    const double zz = pzr * im[IM_JZJ + s] + e0 * im[IM_JDNV + s] + e1 * im[IM_JDNV + 6 + s] + e2 * im[IM_JDNV + 12 + s]
                                    - (b0 * im[IM_JDMO + s] + b1 * im[IM_JDMO + 6 + s] + b2 * im[IM_JDMO + 12 + s]);
                    hR[s] = g + zz;
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {                                       // curvature columns (c, jj)
                const double fc = (c == 2) ? 2.0 : 1.0;
                const double gam = fc * (b0 * im[IM_CT3 + sym3(0, c)] + b1 * im[IM_CT3 + sym3(1, c)] + b2 * im[IM_CT3 + sym3(2, c)]);
                const double alpha = mt * (fc * im[IM_CBG + 6 * c + rt]) + (1.0 - mt) * gam, beta = mt * im[IM_JMOF + c];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    const double g = im[IM_N + jj] * alpha - beta * im[IM_DN + 6 * jj + rt];
                    gR[6 + 3 * c + jj] = g;
                    if constexpr (WITHC) hR[6 + 3 * c + jj] = g - gam * im[IM_NB + jj];
                }
            }
            GF_STAMP(4, tstamp);
            dpp_source_fence(gR);
            if constexpr (WITHC) dpp_source_fence(hR);
        }
        GF_STAMP(5, tstamp);
        // -- residual and dR/dh prefactors of basis function x at this Gauss point
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
        GF_STAMP(6, tstamp);
        // -- contraction: one MFMA per (component, m); the B operand T_b is formed from the expanded row on the fly
        // K component (i, j), m: T_b = w sum_m' G[(m,i),(m',j)] phi_b[m'] -- the five entries are gR[3 m' + j] of lane 3 m + i.
        // The B operands of all components of one m are formed as independent FMA chains before their MFMAs are issued.
        constexpr int QI[6] = {0, 0, 0, 1, 1, 2}, QJ[6] = {0, 1, 2, 1, 2, 2};
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
        if (doC) {
            static_for<5>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                double t[9];
                static_for<9>([&](auto q_) { constexpr int q = decltype(q_)::value; t[q] = row_dot<3 * m + q / 3, q % 3>(hR, pb); });
                mfma_hazard_gap(t);
#pragma unroll
                for (int q = 0; q < 9; ++q) accC[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[m], t[q], accC[q], 0, 0, 0);
            });
            if (has_bf) {                    // d(-f . u dA)/dc : -w f_i R_a (dJ/dZ . phi_b): one tile per f, the factor -f_i is applied once at the end
                const LoadGeom lg = load_geom(im, ppd);
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const double jz = load_dz_dot(im, ppd, lg, f, pb[0], pb[1]);
                    accB[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(R0, jz, accB[f], 0, 0, 0);
                }
            }
        }
    }
    if (has_bf && doC) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int f = 0; f < 3; ++f) accC[3 * i + f] -= pf[i] * accB[f];
    }

    // ---- residual: sum the four Gauss-point groups
#ifdef GF_STAMPS
    { const unsigned long long t1_ = clock64(); stamp_acc[7] += t1_ - tstamp; tstamp = t1_; }
    if ((blockIdx.x & 31) == 0 && tid == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_stamps[k], stamp_acc[k]);
#endif
    wave_lds_sync();
    for (int i = 0; i < 3; ++i) s_g[(kk * 16 + x) * 3 + i] = accR[i];
    wave_lds_sync();
    double* out = blk + (size_t)blockIdx.x * Cfg::BLK;
    if (tid < ND && (flags & GF_ASM_R_BIT)) out[Cfg::OFF_R + tid] = s_g[tid] + s_g[48 + tid] + s_g[96 + tid] + s_g[144 + tid];
    // ---- write the element block once: register rr of lane (x, kk) is entry (a, b) = (kk + 4 rr, x)
    constexpr int IJ_I[6] = {0, 0, 0, 1, 1, 2}, IJ_J[6] = {0, 1, 2, 1, 2, 2};
    const int b = x;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int a = kk + 4 * rr;
        if (a >= NB || b >= NB) continue;
        if (doK) {
#pragma unroll
            for (int ij = 0; ij < 6; ++ij) {
                const int i = IJ_I[ij], j = IJ_J[ij];
                out[Cfg::OFF_K + (3 * a + i) * ND + 3 * b + j] = accK[ij][rr];
                if (i < j) out[Cfg::OFF_K + (3 * b + j) * ND + 3 * a + i] = accK[ij][rr];
            }
        }
        if (doC) {
#pragma unroll
            for (int q = 0; q < 9; ++q) out[Cfg::OFF_C + (3 * a + q / 3) * ND + 3 * b + q % 3] = accC[q][rr];
        }
        if (doH) {
#pragma unroll
            for (int i = 0; i < 3; ++i) out[Cfg::OFF_H + (3 * a + i) * NB + b] = accH[i][rr];
        }
    }
}

}  // namespace gf
