// gf_element_mfma.hpp -- element-BLOCK kernel (p = 3, and p = 2 with a padded tile): one knot-span element per workgroup = one wave,
// the element block (48 x 48 K, 48 x 48 dR/dCP, 48 x 16 dR/dh, 48 R) written once and summed by kl_gather1_kernel.  GF_ASSEMBLY=block
// selects it; the default for p = 2, 3 is the row-record path (gf_element_rec.hpp), against which this one is the cross-check of
// tests/test_gpu_parity.py.  Phase 1 and the Gauss-point group step are shared with that path (gf_gauss_loop.hpp).
// Reference path: the same integrals as kl_element_kernel (GOLDFISH/nonmatching_opt.py:941-1015 via PENGoLINS assembly).
#pragma once
#include "gf_gauss_loop.hpp"

namespace gf {

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
    const RowLane L(x);                                  // lane constants of the row expansion (independent of the loads above)
    if (tid < NB) {
        s_c[tid][0] = c4.x; s_c[tid][1] = c4.y; s_c[tid][2] = c4.z; s_w[tid] = c4.w;
        s_d[tid][0] = ux; s_d[tid][1] = uy; s_d[tid][2] = uz;            // displacement coefficients (kl_strains)
        s_h[tid] = hh;
    }
    if (tid < P1 * 3 * P1) { s_tu[tid] = ttu; s_tv[tid] = ttv; }
    if (tid < P1) { s_wg[tid] = twu; s_wg[P1 + tid] = twv; }
    wave_lds_sync();
    GF_STAMP(0, tstamp);

    // ---- phase 1: three lanes per Gauss point: kinematics + pointwise closed forms -> s_im
    point_phase<P, WITHC>(x, kk, s_tu, s_tv, s_c, s_d, s_w, s_h, s_pc, s_wg, s_wg + P1, s_im);
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

    for (int grp = 0; grp < NGRP; ++grp) {
        const int gp = 4 * grp + kk, gpc = gp < NG ? gp : NG - 1, gu = gpc % P1, gv = gpc / P1;   // Gauss point of this lane's group
        const double* im = s_im[gpc];
        const double wq = gp < NG ? im[IM_WQ] : 0.0;        // padded Gauss-point slots contribute nothing
        gauss_group<P, WITHC>(L, im, wq, s_tu, s_tv, gu, gv, ju, jv, bval, doK, doC, doH, has_bf, pf, ppd, accK, accC, accH, accB, accR, SfLane{} GF_GROUP_STAMP_ARGS);
    }
    GF_STAMP(4, tstamp);
    if (has_bf && doC) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int f = 0; f < 3; ++f) accC[3 * i + f] -= pf[i] * accB[f];
    }

    // ---- residual: sum the four Gauss-point groups
#ifdef GF_STAMPS
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
