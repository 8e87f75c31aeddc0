// gf_element_mfma4.hpp -- p = 4 element kernel on the FP64 matrix pipe (same scheme as gf_element_mfma.hpp).
//
// 25 basis functions = 2 x 2 MFMA tiles of 16 (9 columns / rows of the second tiles are padding), 25 Gauss points = 7 groups
// of 4 (3 padded slots with zero weight).  The 2 x 2 x 15 accumulator tiles (240 doubles per lane) do not fit the register
// file, so the element is processed in three passes whose accumulators each fit the 256 AGPRs: pass K (residual + K: the tile
// pairs (0, 0), (1, 1) with the components i <= j and (1, 0) with all nine; (0, 1) follows from the symmetry at the store:
// 21 tiles) and one pass per b tile for dR/dCP + dR/dh (2 x 9 + 2 x 3 tiles).  Round 2 ran two passes over the b tiles with
// K, dR/dCP and dR/dh together (36 tiles = 288 registers > 256: the compiler moved tiles in and out around the MFMAs).
// Phase 1 is one lane per Gauss point (25 x 3 lanes would not fit a wave).  One wave per element, 40.5 KB LDS -> four per CU.
#pragma once
#include "gf_gauss_loop.hpp"

namespace gf {

template <bool WITHC = true>               // WITHC = false: Newton pass (no dR/dCP)
__global__ __launch_bounds__(64) void kl_element_mfma4_kernel(DevModel M, int e_first, int flags, double* __restrict__ blk) {
    using Cfg = ElemCfg<4>;
    constexpr int P = 4, P1 = 5, NB = 25, NG = 25, ND = 75, NGRP = 7, NT1 = NB - 16;       // NT1: basis functions in the second tile
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

    __shared__ __attribute__((aligned(16))) double s_g[8 * NB];      // control-point staging (phases 0-1), then the residual reduction (2 x 2 x NB x 3 <= 8 NB... see below)
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
    // ---- phase 0
    if (tid < NB) {
        const long long g = ed.g0 + (tid % P1) + (long long)(tid / P1) * ed.nu;
        const double4 c4 = reinterpret_cast<const double4*>(M.cp4)[g];
        const double ux = M.u[3 * g], uy = M.u[3 * g + 1], uz = M.u[3 * g + 2];
        s_c[tid][0] = c4.x; s_c[tid][1] = c4.y; s_c[tid][2] = c4.z; s_w[tid] = c4.w;
        s_d[tid][0] = ux; s_d[tid][1] = uy; s_d[tid][2] = uz;            // displacement coefficients (kl_strains)
        s_h[tid] = M.h[g];
    }
    for (int k = tid; k < P1 * 3 * P1; k += 64) { s_tu[k] = M.tab[ed.tabu + k]; s_tv[k] = M.tab[ed.tabv + k]; }
    if (tid < P1) { s_wg[tid] = M.tab[ed.wu + tid]; s_wg[P1 + tid] = M.tab[ed.wv + tid]; }
    wave_lds_sync();
    GF_STAMP(0, tstamp);

    // ---- phase 1: one lane per Gauss point (sum-factorised control-point sums, quotient rule, pointwise record)
    if (tid < NG) {
        const int gu = tid % P1, gv = tid / P1;
        double Ac[3][6], Ad[3][6], W[6], t = 0.0;
        for (int k = 0; k < 6; ++k) { W[k] = 0.0; for (int i = 0; i < 3; ++i) { Ac[i][k] = 0.0; Ad[i][k] = 0.0; } }
        double U[3][P1];
        for (int d = 0; d < 3; ++d) for (int j = 0; j < P1; ++j) U[d][j] = s_tu[(gu * 3 + d) * P1 + j];
#pragma unroll
        for (int jv = 0; jv < P1; ++jv) {
            const double v0 = s_tv[(gv * 3 + 0) * P1 + jv], v1 = s_tv[(gv * 3 + 1) * P1 + jv], v2 = s_tv[(gv * 3 + 2) * P1 + jv];
            double S[7][3], Sh = 0.0;
            for (int q = 0; q < 7; ++q) for (int d = 0; d < 3; ++d) S[q][d] = 0.0;
#pragma unroll
            for (int ju = 0; ju < P1; ++ju) {
                const int a = ju + P1 * jv;
                const double qv[7] = {s_c[a][0], s_c[a][1], s_c[a][2], s_d[a][0], s_d[a][1], s_d[a][2], s_w[a]};
                for (int q = 0; q < 7; ++q) for (int d = 0; d < 3; ++d) S[q][d] += U[d][ju] * qv[q];
                Sh += U[0][ju] * s_h[a];
            }
            t += v0 * Sh;
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                double* A = q < 3 ? Ac[q] : (q < 6 ? Ad[q - 3] : W);
                A[0] += v0 * S[q][0]; A[1] += v0 * S[q][1]; A[2] += v1 * S[q][0];
                A[3] += v0 * S[q][2]; A[4] += v2 * S[q][0]; A[5] += v1 * S[q][1];
            }
        }
        W[0] = 1.0 / W[0];
        double z[15], Z[15], dz[15], R[6];
        for (int i = 0; i < 3; ++i) {
            rationalize6(Ac[i], W, R);
            for (int m = 0; m < 5; ++m) Z[3 * m + i] = R[m + 1];
            rationalize6(Ad[i], W, R);                       // s_d holds the displacement coefficients: dz = z - Z (kl_point.hpp: kl_strains)
            for (int m = 0; m < 5; ++m) { dz[3 * m + i] = R[m + 1]; z[3 * m + i] = Z[3 * m + i] + R[m + 1]; }
        }
        double* im = s_im[tid];
        shell_point(z, Z, dz, t, s_pc[0], s_pc[1], im);
        for (int k = 0; k < 6; ++k) im[IM_W + k] = W[k];
        im[IM_WQ] = s_wg[gu] * s_wg[P1 + gv];
    }
    wave_lds_sync();
    GF_STAMP(1, tstamp);

    const bool doK = (flags & GF_ASM_K_BIT) != 0, doC = WITHC && (flags & GF_ASM_C_BIT) != 0, doH = (flags & GF_ASM_H_BIT) != 0;
    const bool has_bf = (pf[0] != 0.0) || (pf[1] != 0.0) || (pf[2] != 0.0);
    // basis functions of this lane: tile 0 -> x, tile 1 -> 16 + x (padding for x >= NT1)
    const int bf[2] = {x, x < NT1 ? 16 + x : 0};
    const double bval[2] = {1.0, x < NT1 ? 1.0 : 0.0};
    constexpr int IJ_I[6] = {0, 0, 0, 1, 1, 2}, IJ_J[6] = {0, 1, 2, 1, 2, 2};
    constexpr int QI[6] = {0, 0, 0, 1, 1, 2}, QJ[6] = {0, 1, 2, 1, 2, 2};
    double* out = blk + (size_t)blockIdx.x * Cfg::BLK;
    const RowLane RLg(x);          // lane constants of the row expansion (round 2 re-derived them per group: with 288 accumulator registers a dozen of them were spilled)
    // basis functions of both tiles at the lane's Gauss point of a group (registers)
    auto basis = [&](const double* im, int gu, int gv, double (&phi)[2][5], double (&R0)[2], double (&n0)[2]) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int ju = bf[tl] % P1, jv = bf[tl] / P1;
            const double u0 = s_tu[(gu * 3 + 0) * P1 + ju], u1 = s_tu[(gu * 3 + 1) * P1 + ju], u2 = s_tu[(gu * 3 + 2) * P1 + ju];
            const double v0 = s_tv[(gv * 3 + 0) * P1 + jv], v1 = s_tv[(gv * 3 + 1) * P1 + jv], v2 = s_tv[(gv * 3 + 2) * P1 + jv];
            const double Nb[6] = {u0 * v0, u1 * v0, u0 * v1, u2 * v0, u0 * v2, u1 * v1};
            double R[6];
            rationalize6(Nb, im + IM_W, R);
            for (int k = 0; k < 5; ++k) phi[tl][k] = bval[tl] * R[k + 1];
            R0[tl] = bval[tl] * R[0]; n0[tl] = bval[tl] * Nb[0];
        }
    };

    // ================= pass K: residual + K.  The tile pairs (a tile, b tile) = (0, 0), (1, 1) with the six components i <= j and (1, 0) with all
    // nine -- K^(ij)[a][b] = K^(ji)[b][a] gives the pair (0, 1) at the store -- : 21 accumulator tiles (168 registers) instead of the 24 of two
    // b-tile passes, ONE row expansion of G per Gauss-point group, T_b of both b tiles formed once.
    {
        gf_d4 accK00[6], accK11[6], accK10[9];
        for (int q = 0; q < 6; ++q) { accK00[q] = gf_d4{0, 0, 0, 0}; accK11[q] = gf_d4{0, 0, 0, 0}; }
        for (int q = 0; q < 9; ++q) accK10[q] = gf_d4{0, 0, 0, 0};
        double accR[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        for (int grp = 0; grp < NGRP; ++grp) {
            const int gp = 4 * grp + kk, gpc = gp < NG ? gp : NG - 1;
            const int gu = gpc % P1, gv = gpc / P1;
            const double* im = s_im[gpc];
            const double wq = gp < NG ? im[IM_WQ] : 0.0;               // padded Gauss-point slots contribute nothing
            double phi[2][5], R0[2], n0[2];
            basis(im, gu, gv, phi, R0, n0);
            GF_STAMP(2, tstamp);
            double gR[15], hR[15];
            for (int s = 0; s < 15; ++s) { gR[s] = 0.0; hR[s] = 0.0; }
            if (doK) {
                RLg.template expand<false>(im, gR, hR); dpp_source_fence(gR);
            }
            GF_STAMP(3, tstamp);
            {
                const double ls = has_bf ? load_scalar(im, ppd) : 0.0;
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
                    for (int i = 0; i < 3; ++i) {
                        double rz = 0.0;
                        for (int m = 0; m < 5; ++m) rz += phi[ta][m] * im[IM_PZ + 3 * m + i];
                        accR[ta][i] += wq * (rz - ls * pf[i] * R0[ta]);
                    }
            }
            GF_STAMP(4, tstamp);
            if (doK) {
                double pb0[5], pb1[5];
                for (int m = 0; m < 5; ++m) { pb0[m] = wq * phi[0][m]; pb1[m] = wq * phi[1][m]; }
                static_for<5>([&](auto m_) {
                    constexpr int m = decltype(m_)::value;
                    double t0[9], t1[6];
                    static_for<9>([&](auto q_) { constexpr int q = decltype(q_)::value; t0[q] = row_dot<3 * m + q / 3, q % 3>(gR, pb0); });
                    static_for<6>([&](auto q_) { constexpr int q = decltype(q_)::value; t1[q] = row_dot<3 * m + QI[q], QJ[q]>(gR, pb1); });
                    mfma_hazard_gap(t0); mfma_hazard_gap(t1);
#pragma unroll
                    for (int q = 0; q < 6; ++q) accK00[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[0][m], t0[3 * QI[q] + QJ[q]], accK00[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 9; ++q) accK10[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[1][m], t0[q], accK10[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 6; ++q) accK11[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[1][m], t1[q], accK11[q], 0, 0, 0);
                });
            }
            GF_STAMP(5, tstamp);
        }
        // ---- K of the element block: register rr of lane (x, kk) is (a, b) = (16 ta + kk + 4 rr, 16 tb + x)
        if (doK) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int a0 = kk + 4 * rr, a1 = 16 + kk + 4 * rr, b0 = x, b1 = 16 + x;
#pragma unroll
                for (int ij = 0; ij < 6; ++ij) {
                    const int i = IJ_I[ij], j = IJ_J[ij];
                    out[Cfg::OFF_K + (3 * a0 + i) * ND + 3 * b0 + j] = accK00[ij][rr];
                    if (i < j) out[Cfg::OFF_K + (3 * b0 + j) * ND + 3 * a0 + i] = accK00[ij][rr];
                    if (a1 < NB && b1 < NB) {
                        out[Cfg::OFF_K + (3 * a1 + i) * ND + 3 * b1 + j] = accK11[ij][rr];
                        if (i < j) out[Cfg::OFF_K + (3 * b1 + j) * ND + 3 * a1 + i] = accK11[ij][rr];
                    }
                }
                if (a1 < NB) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        out[Cfg::OFF_K + (3 * a1 + q / 3) * ND + 3 * b0 + q % 3] = accK10[q][rr];
                        out[Cfg::OFF_K + (3 * b0 + q % 3) * ND + 3 * a1 + q / 3] = accK10[q][rr];          // the (0, 1) pair by symmetry
                    }
                }
            }
        }
        GF_STAMP(7, tstamp);
        // ---- residual: sum the four Gauss-point slots of a group (two steps through the staging area: 2 x 32 x 3 doubles)
        wave_lds_sync();
        if (kk >= 2) for (int ta = 0; ta < 2; ++ta) for (int i = 0; i < 3; ++i) s_g[(((kk - 2) * 2 + ta) * 16 + x) * 3 + i] = accR[ta][i];
        wave_lds_sync();
        if (kk < 2) for (int ta = 0; ta < 2; ++ta) for (int i = 0; i < 3; ++i) accR[ta][i] += s_g[((kk * 2 + ta) * 16 + x) * 3 + i];
        wave_lds_sync();
        if (kk == 1) for (int ta = 0; ta < 2; ++ta) for (int i = 0; i < 3; ++i) s_g[(ta * 16 + x) * 3 + i] = accR[ta][i];
        wave_lds_sync();
        if (kk == 0 && (flags & GF_ASM_R_BIT)) for (int ta = 0; ta < 2; ++ta) {
            const int a = 16 * ta + x;
            if (a < NB) for (int i = 0; i < 3; ++i) out[Cfg::OFF_R + 3 * a + i] = accR[ta][i] + s_g[(ta * 16 + x) * 3 + i];
        }
        wave_lds_sync();
    }

    // ================= passes C: dR/dCP and dR/dh, one pass per b tile (2 x 9 + 2 x 3 accumulator tiles = 192 registers): T_b of the pass's b tile
    // feeds both a tiles
    if (doC || doH) for (int tb = 0; tb < 2; ++tb) {
        gf_d4 accC[2][9], accH[2][3];
        for (int ta = 0; ta < 2; ++ta) {
            for (int q = 0; q < 9; ++q) accC[ta][q] = gf_d4{0, 0, 0, 0};
            for (int q = 0; q < 3; ++q) accH[ta][q] = gf_d4{0, 0, 0, 0};
        }
        for (int grp = 0; grp < NGRP; ++grp) {
            const int gp = 4 * grp + kk, gpc = gp < NG ? gp : NG - 1;
            const int gu = gpc % P1, gv = gpc / P1;
            const double* im = s_im[gpc];
            const double wq = gp < NG ? im[IM_WQ] : 0.0;
            double phi[2][5], R0[2], n0[2];
            basis(im, gu, gv, phi, R0, n0);
            GF_STAMP(2, tstamp);
            double gR[15], hR[15];
            for (int s = 0; s < 15; ++s) { gR[s] = 0.0; hR[s] = 0.0; }
            if constexpr (WITHC) if (doC) {
                RLg.template expand<true>(im, gR, hR); dpp_source_fence(hR);
            }
            GF_STAMP(3, tstamp);
            double pb[5];
            for (int m = 0; m < 5; ++m) pb[m] = wq * (tb == 0 ? phi[0][m] : phi[1][m]);
            const double n0b = tb == 0 ? n0[0] : n0[1];
            if (doH) {
#pragma unroll
                for (int ta = 0; ta < 2; ++ta) {
                    double nn = 0.0;
                    for (int k = 0; k < 3; ++k) nn += phi[ta][2 + k] * im[IM_JCK4 + k] * (k == 2 ? 2.0 : 1.0);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const double g1i = im[IM_G + i], g2i = im[IM_G + 3 + i];
                        double rh = phi[ta][0] * (im[IM_JCE] * g1i + im[IM_JCE + 2] * g2i) + phi[ta][1] * (im[IM_JCE + 1] * g2i + im[IM_JCE + 2] * g1i);
                        for (int k = 0; k < 3; ++k) rh -= im[IM_JCK4 + k] * (phi[ta][0] * im[IM_BG + 6 * k + i] + phi[ta][1] * im[IM_BG + 6 * k + 3 + i]);
                        rh -= im[IM_N + i] * nn;
                        accH[ta][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(wq * rh, n0b, accH[ta][i], 0, 0, 0);
                    }
                }
            }
            GF_STAMP(4, tstamp);
            if constexpr (WITHC) if (doC) {
                static_for<5>([&](auto m_) {
                    constexpr int m = decltype(m_)::value;
                    double t[9];
                    static_for<9>([&](auto q_) { constexpr int q = decltype(q_)::value; t[q] = row_dot<3 * m + q / 3, q % 3>(hR, pb); });
                    mfma_hazard_gap(t);
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        accC[0][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[0][m], t[q], accC[0][q], 0, 0, 0);
                        accC[1][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[1][m], t[q], accC[1][q], 0, 0, 0);
                    }
                });
                if (has_bf) {                    // d(-f . u dA)/dc : -w f_i R_a (dJ/dZ . phi_b)
                    const LoadGeom lg = load_geom(im, ppd);        // behind the uniform branch: a model without distributed loads does not pay for it
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        const double jz = load_dz_dot(im, ppd, lg, f, pb[0], pb[1]);
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            accC[0][3 * i + f] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pf[i] * R0[0], jz, accC[0][3 * i + f], 0, 0, 0);
                            accC[1][3 * i + f] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pf[i] * R0[1], jz, accC[1][3 * i + f], 0, 0, 0);
                        }
                    }
                }
            }
            GF_STAMP(6, tstamp);
        }
        // ---- the (a tiles, b tile tb) part of dR/dCP and dR/dh
        const int b = 16 * tb + x;
        if (b < NB) {
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int a = 16 * ta + kk + 4 * rr;
                    if (a >= NB) continue;
                    if constexpr (WITHC) if (doC) {
#pragma unroll
                        for (int q = 0; q < 9; ++q) out[Cfg::OFF_C + (3 * a + q / 3) * ND + 3 * b + q % 3] = accC[ta][q][rr];
                    }
                    if (doH) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) out[Cfg::OFF_H + (3 * a + i) * NB + b] = accH[ta][i][rr];
                    }
                }
        }
        GF_STAMP(7, tstamp);              // the block stores of this pass
    }
#ifdef GF_STAMPS
    if ((blockIdx.x & 31) == 0 && tid == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_stamps[k], stamp_acc[k]);
#endif
}

}  // namespace gf
