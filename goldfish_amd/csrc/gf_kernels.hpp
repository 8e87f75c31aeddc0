// gf_kernels.hpp -- HIP kernels of the shell assembly + sensitivity hot path (gfx950).
//
//   kl_element_kernel   K1-K4 of SURVEY.md 2.2: one knot-span element per workgroup; basis and
//                       second derivatives at the Gauss points from 1-D tables staged in LDS;
//                       per-Gauss-point pointwise Hessians (kl_point.hpp) in LDS; contraction
//                       phi_a^T G phi_b in FP64 VALU registers; element blocks written once.
//   kl_gather_kernel    row-owner, atomic-free accumulation of element blocks into the static CSR
//                       value arrays (each nnz written exactly once), Dirichlet handling fused.
//   pen_*               K5-K7: penalty coupling, point kernel + deterministic owner gathers.
//   csr_apply*          K9: y += A x, y += A^T x on the block-CSR layout.
#pragma once
#include <hip/hip_runtime.h>
#include "gf_setup.hpp"
#include "kl_point.hpp"

namespace gf {

struct DevModel {
    const PatchDev* patches; const double* tab; const int* ints; const int* elem_patch; const int* cp_patch;
    const double* cp4; const double* u; const double* h; const unsigned char* zero;
    const unsigned short* nb_meta;   // per nb_c entry: box slot | Dirichlet flags of the column dofs | self (gf_setup.hpp)
    const CpDesc* cpdesc;            // per-control-point descriptors (gather)
    const ElemDesc* edesc;           // per-element descriptors (MFMA element kernel)
    const unsigned char* pen_row;    // 1: the control point owns penalty rows (pen_owner_kernel writes them before the gather adds the shell part)
    const long long* nb_ptr_s; const int* nb_s; const long long* nb_ptr_c; const int* nb_c;
    long long total_cp, nelem;
};

template <int P> struct ElemCfg {
    static constexpr int P1 = P + 1, NB = P1 * P1, NG = NB, ND = 3 * NB;
#ifndef GF_AG3
#define GF_AG3 2
#endif
    static constexpr int AG = (P == 2) ? 3 : (P == 3 ? GF_AG3 : 5);  // a's per lane
    static constexpr int NAG = (NB + AG - 1) / AG;
    static constexpr int NT = ((NAG * NB + 63) / 64) * 64;            // threads per element
    static constexpr int BLK = ND * ND + ND * ND + ND * NB + ND;      // doubles per element block (K, C[3], H, R)
    static constexpr int OFF_K = 0, OFF_C = ND * ND, OFF_H = 2 * ND * ND, OFF_R = 2 * ND * ND + ND * NB;
};

// tensor-product B-spline values/derivatives of local function a at Gauss point (gu, gv)
template <int P> __device__ __forceinline__ void bspline6(const double* tu, const double* tv, int gu, int gv, int a, double* Nb) {
    constexpr int P1 = P + 1;
    const int ju = a % P1, jv = a / P1;
    const double u0 = tu[(gu * 3 + 0) * P1 + ju], u1 = tu[(gu * 3 + 1) * P1 + ju], u2 = tu[(gu * 3 + 2) * P1 + ju];
    const double v0 = tv[(gv * 3 + 0) * P1 + jv], v1 = tv[(gv * 3 + 1) * P1 + jv], v2 = tv[(gv * 3 + 2) * P1 + jv];
    Nb[0] = u0 * v0; Nb[1] = u1 * v0; Nb[2] = u0 * v1; Nb[3] = u2 * v0; Nb[4] = u0 * v2; Nb[5] = u1 * v1;
}
// W[0] holds 1/W (the reciprocal is taken once per Gauss point by the caller), W[1..5] the derivatives of W
__device__ __forceinline__ void rationalize6(const double* Nb, const double* W, double* R) {
    const double iW = W[0];
    R[0] = Nb[0] * iW;
    R[1] = (Nb[1] - R[0] * W[1]) * iW; R[2] = (Nb[2] - R[0] * W[2]) * iW;
    R[3] = (Nb[3] - 2 * R[1] * W[1] - R[0] * W[3]) * iW;
    R[4] = (Nb[4] - 2 * R[2] * W[2] - R[0] * W[4]) * iW;
    R[5] = (Nb[5] - R[1] * W[2] - R[2] * W[1] - R[0] * W[5]) * iW;
}

#if defined(GF_STAMPS) || defined(GF_STAMPS_PEN)
__device__ unsigned long long g_stamps[8];
#endif
#ifdef GF_STAMPS
#define GF_STAMP(slot, t0) do { const unsigned long long t1_ = clock64(); stamp_acc[slot] += t1_ - (t0); (t0) = t1_; } while (0)
#else
#define GF_STAMP(slot, t0) do { } while (0)
#endif
template <int P>
__global__ __launch_bounds__(ElemCfg<P>::NT, (P >= 4 ? 1 : ElemCfg<P>::NT / 64)) void kl_element_kernel(DevModel M, int e_first, int flags, double* __restrict__ blk) {
    using Cfg = ElemCfg<P>;
    constexpr int P1 = Cfg::P1, NB = Cfg::NB, NG = Cfg::NG, ND = Cfg::ND, AG = Cfg::AG, NAG = Cfg::NAG, NT = Cfg::NT;
    constexpr int TS = 5 * 16 + 2;                 // T row per b: [m][16 q-slots] (+2 pad: conflict-free 16-B reads)
    const int tid = threadIdx.x;
    const long long e = (long long)e_first + blockIdx.x;
    if (e >= M.nelem) return;
    const PatchDev& Pt = M.patches[M.elem_patch[e]];
    const int le = int(e - Pt.elem_off), eu = le % Pt.nelu, ev = le / Pt.nelu;
    const int iu0 = M.ints[Pt.spu + eu] - P, iv0 = M.ints[Pt.spv + ev] - P;

    // LDS: the control-point staging buffers are only live in phases 0-1 and share storage with T (phase 2)
    constexpr int STAGE = NB * 8, TBUF = NB * TS, UNI = STAGE > TBUF ? STAGE : TBUF;
    __shared__ __attribute__((aligned(16))) double s_uni[UNI];
    double (*s_c)[3] = reinterpret_cast<double (*)[3]>(s_uni);
    double (*s_d)[3] = reinterpret_cast<double (*)[3]>(s_uni + 3 * NB);
    double* s_h = s_uni + 6 * NB; double* s_w = s_uni + 7 * NB;
    double* s_T = s_uni;
    __shared__ double s_tu[P1 * 3 * P1], s_tv[P1 * 3 * P1], s_wg[2 * P1];
    __shared__ double s_im[NG][IM_SIZE];
    __shared__ __attribute__((aligned(16))) double s_phi[NB][6];   // [a][0..4] = first/second derivatives, [5] = value
    __shared__ double s_n0[NB];
    // G = Pzz and Hc = Pzz + PzZ stored as [(m,i)][j][m' (pad 6)]: the 5 operands of one T output are
    // contiguous and 16-B aligned (ds_read_b128 instead of half-rate ds_read2_b64)
    __shared__ __attribute__((aligned(16))) double s_GH[2 * 270];
    double* s_G = s_GH; double* s_Hc = s_GH + 270;

    unsigned long long tstamp = 0; (void)tstamp;
#ifdef GF_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tstamp = clock64();
#endif
    // ---- phase 0: stage control-point data and 1-D tables ------------------------------------
    if (tid < NB) {
        const long long g = Pt.cp_off + (iu0 + tid % P1) + (long long)(iv0 + tid / P1) * Pt.nu;
        const double4 c4 = reinterpret_cast<const double4*>(M.cp4)[g];
        const double ux = M.u[3 * g], uy = M.u[3 * g + 1], uz = M.u[3 * g + 2];
        s_c[tid][0] = c4.x; s_c[tid][1] = c4.y; s_c[tid][2] = c4.z; s_w[tid] = c4.w;
        s_d[tid][0] = ux; s_d[tid][1] = uy; s_d[tid][2] = uz;            // displacement coefficients: the strains are evaluated from their derivatives (kl_strains)
        s_h[tid] = M.h[g];
    }
    for (int k = tid; k < P1 * 3 * P1; k += NT) { s_tu[k] = M.tab[Pt.tabu + eu * P1 * 3 * P1 + k]; s_tv[k] = M.tab[Pt.tabv + ev * P1 * 3 * P1 + k]; }
    if (tid < P1) { s_wg[tid] = M.tab[Pt.wu + eu * P1 + tid]; s_wg[P1 + tid] = M.tab[Pt.wv + ev * P1 + tid]; }
    __syncthreads();
    GF_STAMP(0, tstamp);

    // ---- phase 1: one lane per Gauss point: kinematics + pointwise closed forms ----------------
    if (tid < NG) {
        const int gu = tid % P1, gv = tid / P1;
        double W[6] = {0, 0, 0, 0, 0, 0}, Nb[6], R[6];
        for (int a = 0; a < NB; ++a) { bspline6<P>(s_tu, s_tv, gu, gv, a, Nb); for (int k = 0; k < 6; ++k) W[k] += Nb[k] * s_w[a]; }
        W[0] = 1.0 / W[0];
        double z[15], Z[15], dz[15], t = 0.0;
        for (int k = 0; k < 15; ++k) { dz[k] = 0.0; Z[k] = 0.0; }
        for (int a = 0; a < NB; ++a) {
            bspline6<P>(s_tu, s_tv, gu, gv, a, Nb); rationalize6(Nb, W, R);
            t += Nb[0] * s_h[a];
            for (int m = 0; m < 5; ++m) for (int i = 0; i < 3; ++i) { Z[3 * m + i] += R[m + 1] * s_c[a][i]; dz[3 * m + i] += R[m + 1] * s_d[a][i]; }
        }
        for (int k = 0; k < 15; ++k) z[k] = Z[k] + dz[k];
        double* im = s_im[tid];
        shell_point(z, Z, dz, t, Pt.E, Pt.nu_, im);
        for (int k = 0; k < 6; ++k) im[IM_W + k] = W[k];
        im[IM_WQ] = s_wg[gu] * s_wg[P1 + gv];
    }
    __syncthreads();
    GF_STAMP(1, tstamp);

    // ---- phase 2: per Gauss point: expand Hessians, T = G phi_b, contract with phi_a --------------
    const int b = tid % NB, ag = tid / NB;
    const bool lane_ok = ag < NAG;
    double accK[AG][6], accC[AG][9], accH[AG][3];
    for (int k = 0; k < AG; ++k) { for (int q = 0; q < 6; ++q) accK[k][q] = 0.0; for (int q = 0; q < 9; ++q) accC[k][q] = 0.0; for (int q = 0; q < 3; ++q) accH[k][q] = 0.0; }
    double accR = 0.0;                                    // tid < ND: residual entry (a, i) = (tid/3, tid%3)
    const bool has_bf = (Pt.f[0] != 0.0) || (Pt.f[1] != 0.0) || (Pt.f[2] != 0.0);
    constexpr int IJ_I[6] = {0, 0, 0, 1, 1, 2}, IJ_J[6] = {0, 1, 2, 1, 2, 2};

    // Loop-invariant per-thread descriptors (hoisted by hand: the index arithmetic of the expansion
    // slots and of the T rows otherwise re-executes for every Gauss point).
    // The 225 entries of G = Pzz and Hc = Pzz + PzZ are ordered by structural category
    // [A: tangent x tangent | B: curvature x tangent | B': transpose of B | C: curvature x curvature]
    // so that a wave executes at most two short code paths per pass.
    constexpr int NIT = (225 + NT - 1) / NT, LPB = NT / NB, NTO = (75 + LPB - 1) / LPB;
    int xcat[NIT], x0[NIT], x1[NIT], x2[NIT], x3[NIT], x4[NIT], x5[NIT], x6[NIT], x7[NIT]; double xf[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int slot = tid + it * NT;
        xcat[it] = 4; x0[it] = x1[it] = x2[it] = x3[it] = x4[it] = x5[it] = x6[it] = x7[it] = 0; xf[it] = 0.0;
        if (slot < 36) {
            const int r = slot / 6, s = slot - 6 * r, m = r / 3, i = r - 3 * m, mm = s / 3, j = s - 3 * mm;
            xcat[it] = 0; x0[it] = r; x1[it] = s; x2[it] = m; x3[it] = IM_G + 3 * (1 - m) + i; x4[it] = IM_JNV + (m == mm ? m : 2);
            x5[it] = IM_HMN + hmn_idx(r, s); x6[it] = (3 * r + j) * 6 + mm; xf[it] = (i == j) ? 1.0 : 0.0;
        } else if (slot < 90) {
            const int q = slot - 36, rr = q / 6, s = q - 6 * rr, k = rr / 3, i = rr - 3 * k, r = 6 + rr;
            xcat[it] = 1; x0[it] = r; x1[it] = s; x2[it] = IM_N + i; x3[it] = 6 * k + s; x4[it] = IM_JMOF + k; x5[it] = IM_DN + 6 * i + s;
            x6[it] = (3 * r + s % 3) * 6 + s / 3; x7[it] = (3 * s + i) * 6 + 2 + k; xf[it] = (k == 2 ? 2.0 : 1.0);
        } else if (slot < 144) {
            const int q = slot - 90, ss = q / 6, r = q - 6 * ss, kk = ss / 3, jj = ss - 3 * kk, s = 6 + ss;
            xcat[it] = 2; x0[it] = r; x1[it] = kk; x2[it] = IM_N + jj; x3[it] = 6 * kk + r; x4[it] = IM_JMOF + kk; x5[it] = IM_DN + 6 * jj + r;
            x6[it] = (3 * r + jj) * 6 + 2 + kk; (void)s; xf[it] = (kk == 2 ? 2.0 : 1.0);
        } else if (slot < 225) {
            const int q = slot - 144, rr = q / 9, ss = q - 9 * rr, k = rr / 3, i = rr - 3 * k, kk = ss / 3, jj = ss - 3 * kk;
            const int lo = k < kk ? k : kk, hi = k < kk ? kk : k;
            xcat[it] = 3; x0[it] = IM_N + i; x1[it] = IM_CT3 + lo * (5 - lo) / 2 + hi; x2[it] = jj; x6[it] = (3 * (6 + rr) + jj) * 6 + 2 + kk;
            xf[it] = (k == 2 ? 2.0 : 1.0) * (kk == 2 ? 2.0 : 1.0);
        }
    }
    // T outputs of this lane: o < 30 -> K (q = ij, from G), o >= 30 -> dR/dc (q = 6 + 3i + f, from Hc); stored at T[b][m*16 + q]
    int rowT[NTO], dstT[NTO];
#pragma unroll
    for (int t = 0; t < NTO; ++t) {
        const int o = ag + t * LPB, oo = o < 75 ? o : 0;
        if (oo < 30) { const int ij = oo / 5, m = oo - 5 * ij; rowT[t] = ((3 * m + IJ_I[ij]) * 3 + IJ_J[ij]) * 6; dstT[t] = m * 16 + ij; }
        else { const int q = (oo - 30) / 5, m = (oo - 30) - 5 * q; rowT[t] = 270 + ((3 * m + q / 3) * 3 + q % 3) * 6; dstT[t] = m * 16 + 6 + q; }
    }
    const bool doK = (flags & GF_ASM_K_BIT) != 0, doC = (flags & GF_ASM_C_BIT) != 0;
    const int pju = tid % P1, pjv = (tid / P1) % P1;              // basis function handled by tid < NB
    const int ra = tid / 3 < NB ? tid / 3 : 0, ri = tid % 3;      // residual entry handled by tid < ND
    double* myT = s_T + b * TS;
    // the two small prefactor jobs (residual rz, dR/dh rh) go to different waves when there are two
    constexpr bool SPLIT_R = (ND <= 64) && (NT >= 128);
    const bool do_rz = tid < ND, do_rh = SPLIT_R ? (tid >= 64 && tid - 64 < ND) : (tid < ND);
    const int hidx = SPLIT_R ? (tid >= 64 && tid - 64 < ND ? tid - 64 : 0) : (tid < ND ? tid : 0), ha = hidx / 3, hi = hidx % 3;
    GF_STAMP(2, tstamp);

    for (int gp = 0; gp < NG; ++gp) {
        const double* im = s_im[gp];
        const double wq = im[IM_WQ];
        // -- S1: basis at this Gauss point, expansion of G and Hc
        if (tid < NB) {
            const int gu = gp % P1, gv = gp / P1;
            const double u0 = s_tu[(gu * 3 + 0) * P1 + pju], u1 = s_tu[(gu * 3 + 1) * P1 + pju], u2 = s_tu[(gu * 3 + 2) * P1 + pju];
            const double v0 = s_tv[(gv * 3 + 0) * P1 + pjv], v1 = s_tv[(gv * 3 + 1) * P1 + pjv], v2 = s_tv[(gv * 3 + 2) * P1 + pjv];
            const double Nb[6] = {u0 * v0, u1 * v0, u0 * v1, u2 * v0, u0 * v2, u1 * v1};
            double R[6];
            rationalize6(Nb, im + IM_W, R);
            for (int k = 0; k < 5; ++k) s_phi[tid][k] = R[k + 1];
            s_phi[tid][5] = R[0];
            s_n0[tid] = Nb[0];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (xcat[it] == 0) {
                const int r = x0[it], s = x1[it];
                const double gr = im[IM_G + r];
                const double e0 = x2[it] == 0 ? gr : 0.0, e1 = x2[it] == 1 ? gr : 0.0, e2 = im[x3[it]];
                const double b0 = im[IM_BG + r], b1 = im[IM_BG + 6 + r], b2 = im[IM_BG + 12 + r];
                const double g = e0 * im[IM_CEZ + s] + e1 * im[IM_CEZ + 6 + s] + e2 * im[IM_CEZ + 12 + s]
                               + b0 * im[IM_CBG + s] + b1 * im[IM_CBG + 6 + s] + b2 * im[IM_CBG + 12 + s] - im[x5[it]] + xf[it] * im[x4[it]];
                const double zz = im[IM_PZ + r] * im[IM_JZJ + s]
                                + e0 * im[IM_JDNV + s] + e1 * im[IM_JDNV + 6 + s] + e2 * im[IM_JDNV + 12 + s]
                                - (b0 * im[IM_JDMO + s] + b1 * im[IM_JDMO + 6 + s] + b2 * im[IM_JDMO + 12 + s]);
                s_G[x6[it]] = g; s_Hc[x6[it]] = g + zz;
            } else if (xcat[it] == 1) {
                const int r = x0[it], s = x1[it];
                const double fn = xf[it] * im[x2[it]];
                const double g = fn * im[IM_CBG + x3[it]] - im[x4[it]] * im[x5[it]];
                const double zz = im[IM_PZ + r] * im[IM_JZJ + s] - fn * im[IM_JDMO + x3[it]];
                s_G[x6[it]] = g; s_G[x7[it]] = g; s_Hc[x6[it]] = g + zz;
            } else if (xcat[it] == 2) {
                const int r = x0[it], kk = x1[it], jj = x2[it] - IM_N;
                const double g = xf[it] * im[x2[it]] * im[IM_CBG + x3[it]] - im[x4[it]] * im[x5[it]];
                const int c1 = kk == 0 ? 1 : 2 + kk, c2 = kk == 0 ? 2 : 3 + kk;                         // sym3(1,kk), sym3(2,kk)
                const double zz = -(im[IM_BG + r] * im[IM_CT3 + kk] + im[IM_BG + 6 + r] * im[IM_CT3 + c1] + im[IM_BG + 12 + r] * im[IM_CT3 + c2]) * xf[it] * im[IM_NB + jj];
                s_Hc[x6[it]] = g + zz;
            } else if (xcat[it] == 3) {
                const double c = xf[it] * im[x0[it]] * im[x1[it]];
                const double nj = im[IM_N + x2[it]];
                s_G[x6[it]] = c * nj; s_Hc[x6[it]] = c * (nj - im[IM_NB + x2[it]]);
            }
        }
        GF_STAMP(3, tstamp);
        __syncthreads();
        GF_STAMP(4, tstamp);
        // -- S2: residual / dR/dh prefactors, T = G phi_b and Hc phi_b; operands of S3 are pulled into
        //    registers here so that S3 only reads T and s_rh and needs no barrier behind it
        if (do_rz) {
            double rz = 0.0;
            for (int m = 0; m < 5; ++m) rz += s_phi[ra][m] * im[IM_PZ + 3 * m + ri];
            accR += wq * (rz - (has_bf ? load_scalar(im, Pt.pd) : 0.0) * Pt.f[ri] * s_phi[ra][5]);
        }
        if (do_rh) {
            const double p1 = s_phi[ha][0], p2 = s_phi[ha][1];
            const double g1i = im[IM_G + hi], g2i = im[IM_G + 3 + hi];
            double rh = p1 * (im[IM_JCE] * g1i + im[IM_JCE + 2] * g2i) + p2 * (im[IM_JCE + 1] * g2i + im[IM_JCE + 2] * g1i);
            double nn = 0.0;
            for (int k = 0; k < 3; ++k) {
                rh -= im[IM_JCK4 + k] * (p1 * im[IM_BG + 6 * k + hi] + p2 * im[IM_BG + 6 * k + 3 + hi]);
                nn += s_phi[ha][2 + k] * im[IM_JCK4 + k] * (k == 2 ? 2.0 : 1.0);
            }
            rh -= im[IM_N + hi] * nn;
            s_T[(hidx / 5) * TS + (hidx % 5) * 16 + 15] = wq * rh;     // dR/dh prefactor parked in the unused q-slot 15 of T
        }
        double pb[5], pa[AG][5], pa0[AG];
        for (int m = 0; m < 5; ++m) pb[m] = wq * s_phi[b][m];
        for (int k = 0; k < AG; ++k) { const int a = ag * AG + k < NB ? ag * AG + k : NB - 1; pa0[k] = s_phi[a][5]; for (int m = 0; m < 5; ++m) pa[k][m] = s_phi[a][m]; }
        const double n0b = s_n0[b];
        double jz[3] = {0.0, 0.0, 0.0};
        if (has_bf) { const LoadGeom lg = load_geom(im, Pt.pd); for (int f = 0; f < 3; ++f) jz[f] = load_dz_dot(im, Pt.pd, lg, f, pb[0], pb[1]); }
#pragma unroll
        for (int t = 0; t < NTO; ++t) {
            const int o = ag + t * LPB;
            if (o < 75 && (o < 30 ? doK : doC)) {
                const double* gr = s_GH + rowT[t];                // rowT >= 270 addresses Hc
                const double2 g01 = *reinterpret_cast<const double2*>(gr), g23 = *reinterpret_cast<const double2*>(gr + 2);
                myT[dstT[t]] = g01.x * pb[0] + g01.y * pb[1] + g23.x * pb[2] + g23.y * pb[3] + gr[4] * pb[4];
            }
        }
        GF_STAMP(5, tstamp);
        __syncthreads();
        GF_STAMP(6, tstamp);
        // -- S3: contraction phi_a . T_b in registers
        if (lane_ok) {
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                double tq[16];
                const double2* src = reinterpret_cast<const double2*>(myT + m * 16);
#pragma unroll
                for (int q = 0; q < 8; ++q) { const double2 v = src[q]; tq[2 * q] = v.x; tq[2 * q + 1] = v.y; }
                for (int k = 0; k < AG; ++k) {
                    if (doK) for (int q = 0; q < 6; ++q) accK[k][q] += pa[k][m] * tq[q];
                    if (doC) for (int q = 0; q < 9; ++q) accC[k][q] += pa[k][m] * tq[6 + q];
                }
            }
            if (has_bf)            // d(-f . u dA)/dc : -w f_i R_a (dJ/dZ . phi_b)
                for (int k = 0; k < AG; ++k) for (int i = 0; i < 3; ++i) for (int f = 0; f < 3; ++f) accC[k][3 * i + f] -= Pt.f[i] * pa0[k] * jz[f];
            for (int k = 0; k < AG; ++k) { const int a = ag * AG + k < NB ? ag * AG + k : NB - 1; for (int i = 0; i < 3; ++i) { const int w = 3 * a + i; accH[k][i] += n0b * s_T[(w / 5) * TS + (w % 5) * 16 + 15]; } }
        }
        GF_STAMP(7, tstamp);
    }

#ifdef GF_STAMPS
    if ((threadIdx.x & 63) == 0 && (blockIdx.x & 31) == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_stamps[k], stamp_acc[k]);
#endif
    // ---- phase 3: write the element block once -------------------------------------------------------
    double* out = blk + (size_t)blockIdx.x * Cfg::BLK;
    if (tid < ND && (flags & GF_ASM_R_BIT)) out[Cfg::OFF_R + tid] = accR;
    if (lane_ok) for (int k = 0; k < AG; ++k) {
        const int a = ag * AG + k;
        if (a >= NB) break;
        if (flags & GF_ASM_K_BIT) for (int ij = 0; ij < 6; ++ij) {
            const int i = IJ_I[ij], j = IJ_J[ij];
            out[Cfg::OFF_K + (3 * a + i) * ND + 3 * b + j] = accK[k][ij];
            if (i < j) out[Cfg::OFF_K + (3 * b + j) * ND + 3 * a + i] = accK[k][ij];
        }
        // dR_(a,i)/dc_(b,f) = K_(a,i),(b,f) + Phi21 : accC already holds phi_a^T (Pzz + PzZ) phi_b
        if (flags & GF_ASM_C_BIT) for (int q = 0; q < 9; ++q) out[Cfg::OFF_C + (3 * a + q / 3) * ND + 3 * b + q % 3] = accC[k][q];
        if (flags & GF_ASM_H_BIT) for (int i = 0; i < 3; ++i) out[Cfg::OFF_H + (3 * a + i) * NB + b] = accH[k][i];
    }
}

// -------------------------------------------------------------------------------------------------
// Row-owner gather: one workgroup per control point a.  Every element containing a contributes the
// three dof rows (a, i) of its block: they are read as whole contiguous rows (coalesced) and summed,
// in fixed element order, into LDS accumulators laid out on a's neighbour box; each CSR entry of the
// rows is then written exactly once (Dirichlet handling fused).  No atomics, bitwise reproducible.
template <int P>
__global__ __launch_bounds__(256) void kl_gather_kernel(DevModel M, long long a_first, long long e_first, long long e_count, int flags,
                                                         const double* __restrict__ blk,
                                                         double* __restrict__ valK, double* __restrict__ valC0, double* __restrict__ valC1,
                                                         double* __restrict__ valC2, double* __restrict__ valH, double* __restrict__ R, int pen_add) {
    using Cfg = ElemCfg<P>;
    constexpr int P1 = Cfg::P1, NB = Cfg::NB, ND = Cfg::ND, WB = 2 * P + 1, NBOX = WB * WB;
    const long long a = a_first + blockIdx.x;
    if (a >= M.total_cp) return;
    const CpDesc& cd = M.cpdesc[a];                       // uniform (scalar) loads of one 96-byte record: no dependent index chain in front of the row loads
    const int ia = cd.ia, ja = cd.ja, eu0 = cd.eu0, ev0 = cd.ev0, i0 = cd.i0, j0 = cd.j0, j1 = cd.j1, wbox = cd.i1 - cd.i0 + 1;
    const long long ptr_c = M.nb_ptr_c[a], deg_c = M.nb_ptr_c[a + 1] - ptr_c, ptr_s = M.nb_ptr_s[a], deg_s = M.nb_ptr_s[a + 1] - ptr_s;
    const int tid = threadIdx.x;

    __shared__ double aK[3][NBOX][3], aC[3][3][NBOX], aH[3][NBOX], aR[3];
    // metadata of the row's entries: fetched now, used in the write phase (the loads overlap the element loop)
    constexpr int MAXMETA = 320;
    __shared__ unsigned short s_meta[MAXMETA];
    for (int k = tid; k < (int)deg_c && k < MAXMETA; k += 256) s_meta[k] = M.nb_meta[ptr_c + k];
    for (int k = tid; k < 9 * NBOX; k += 256) { (&aK[0][0][0])[k] = 0.0; (&aC[0][0][0])[k] = 0.0; }
    for (int k = tid; k < 3 * NBOX; k += 256) (&aH[0][0])[k] = 0.0;
    if (tid < 3) aR[tid] = 0.0;
    __syncthreads();
    // Wave w < 3 owns dof row i = w of K and dR/dCP, wave 3 owns dR/dh and R: an LDS slot is only ever
    // touched by one wave, whose LDS operations execute in program order, so the element loop needs
    // no barrier and the row loads of successive elements overlap.
    const int wave = tid >> 6, lane = tid & 63;
    // Elements are taken four at a time: the row loads of a group are all issued before the LDS adds
    // of the group, so up to 4 x 6 loads per lane are in flight (the adds keep the fixed element order).
    constexpr int NPASS = (ND + 63) / 64, NPH = (3 * NB + 63) / 64, UNR = 4;
    const int neu = cd.neu, nev = cd.nev, ne = neu * nev;
    for (int g0 = 0; g0 < ne; g0 += UNR) {
        const double* Bp[UNR]; int bu[UNR], bv[UNR], al[UNR]; bool ok[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int g = g0 + q, gg = g < ne ? g : 0, ku = gg % (neu > 0 ? neu : 1), kv = gg / (neu > 0 ? neu : 1);
            const long long e = cd.e00 + ku + (long long)kv * cd.nelu - e_first;
            ok[q] = g < ne && e >= 0 && e < e_count;
            Bp[q] = blk + (size_t)(ok[q] ? e : 0) * Cfg::BLK;
            bu[q] = cd.bu[ku]; bv[q] = cd.bv[kv]; al[q] = (ia - bu[q]) + (ja - bv[q]) * P1;
        }
        if (wave < 3) {
            const int i = wave;
            double vK[UNR][NPASS], vC[UNR][NPASS];
#pragma unroll
            for (int q = 0; q < UNR; ++q)
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int c = lane + 64 * ps;
                    const bool on = ok[q] && c < ND;
                    vK[q][ps] = (on && (flags & GF_ASM_K_BIT)) ? Bp[q][Cfg::OFF_K + (3 * al[q] + i) * ND + c] : 0.0;
                    vC[q][ps] = (on && (flags & GF_ASM_C_BIT)) ? Bp[q][Cfg::OFF_C + (3 * al[q] + i) * ND + c] : 0.0;
                }
#pragma unroll
            for (int q = 0; q < UNR; ++q)
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int c = lane + 64 * ps;
                    if (ok[q] && c < ND) {
                        const int bl = c / 3, j = c - 3 * bl, ks = (bu[q] + bl % P1 - i0) + (bv[q] + bl / P1 - j0) * wbox;
                        aK[i][ks][j] += vK[q][ps]; aC[j][i][ks] += vC[q][ps];
                    }
                }
        } else {
            double vH[UNR][NPH], vR[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
#pragma unroll
                for (int ps = 0; ps < NPH; ++ps) {
                    const int w = lane + 64 * ps;
                    vH[q][ps] = (ok[q] && w < 3 * NB && (flags & GF_ASM_H_BIT)) ? Bp[q][Cfg::OFF_H + (3 * al[q] + w / NB) * NB + w % NB] : 0.0;
                }
                vR[q] = (ok[q] && lane < 3 && (flags & GF_ASM_R_BIT)) ? Bp[q][Cfg::OFF_R + 3 * al[q] + lane] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                if (!ok[q]) continue;
#pragma unroll
                for (int ps = 0; ps < NPH; ++ps) {
                    const int w = lane + 64 * ps;
                    if (w < 3 * NB) { const int i = w / NB, bl = w - i * NB, ks = (bu[q] + bl % P1 - i0) + (bv[q] + bl / P1 - j0) * wbox; aH[i][ks] += vH[q][ps]; }
                }
                if (lane < 3) aR[lane] += vR[q];
            }
        }
    }
    __syncthreads();
    // the penalty rows of an interface control point were written before (pen_owner_kernel); the shell part is added to them
    const bool padd = pen_add && M.pen_row[a];
    // wave i < 3 writes dof row i of K and dR/dCP, one neighbour per lane: its box slot, the Dirichlet flags of its dofs and
    // the diagonal flag come from one 16-bit load (nb_meta) -- no dependent index loads, no divisions; wave 3 writes dR/dh,
    // whose neighbour list IS the box in slot order
    if (wave < 3) {
        const int i = wave;
        const bool zrow = M.zero[3 * a + i] != 0;
        if (flags & GF_ASM_K_BIT) {                        // one matrix entry per lane: consecutive lanes write consecutive doubles
            double* dst = valK + 9 * ptr_c + (long long)i * 3 * deg_c;
            for (int c = lane; c < 3 * (int)deg_c; c += 64) {
                const int k = c / 3, j = c - 3 * k;
                const unsigned meta = s_meta[k];
                const int ks = (meta & 127) == 127 ? -1 : int(meta & 127);
                double v = 0.0;
                if (zrow || (meta & (128u << j))) v = ((meta & 1024u) && i == j) ? 1.0 : 0.0;
                else { if (ks >= 0) v = aK[i][ks][j]; if (padd) v += dst[c]; }
                dst[c] = v;
            }
        }
        for (int k = lane; k < (int)deg_c; k += 64) {
            const unsigned meta = s_meta[k];
            const int ks = (meta & 127) == 127 ? -1 : int(meta & 127);
            if (flags & GF_ASM_C_BIT) {
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    double* dst = (f == 0 ? valC0 : (f == 1 ? valC1 : valC2)) + 3 * ptr_c + (long long)i * deg_c + k;
                    double v = 0.0;
                    if (!zrow) { if (ks >= 0) v = aC[f][i][ks]; if (padd) v += *dst; }
                    *dst = v;
                }
            }
        }
    } else if (flags & GF_ASM_H_BIT) {
        for (int k = lane; k < (int)deg_s; k += 64) {
#pragma unroll
            for (int i = 0; i < 3; ++i) valH[3 * ptr_s + (long long)i * deg_s + k] = aH[i][k];
        }
    }
    if ((flags & GF_ASM_R_BIT) && tid < 3) R[3 * a + tid] = aR[tid] + (padd ? R[3 * a + tid] : 0.0);
}

// Write phase of the one-wave gathers: the three dof rows of control point a from the box accumulators in LDS (aK [3][NBOX][3],
// aC [3 f][3 i][NBOX], aH [3][NBOX]); Dirichlet rows / columns, the coupling-only columns and the penalty rows written before
// (pen_owner_kernel) are handled here.
// Every neighbour list fits the LDS copy of its metadata (gf_create checks deg <= GATHER_MAXMETA).  A fallback to the global array
// (k < MAXMETA ? s_meta[k] : M.nb_meta[..]) compiles to a FLAT load: it counts in vmcnt, and vmcnt is in order -- every iteration of
// the write loops then waits for the previous iteration's store to complete (the write phase ran at 2 TB/s because of it).
constexpr int GATHER_MAXMETA = 320;
template <int NBOX, bool WITHC>
__device__ __forceinline__ void gather_write_rows(const DevModel& M, long long a, int lane, bool doK, bool doC, bool doH, bool padd, unsigned zmask,
                                                  long long ptr_c, long long deg_c, long long ptr_s, long long deg_s, const unsigned short* s_meta,
                                                  const double* aK, const double* aC, const double* aH,
                                                  double* __restrict__ valK, double* __restrict__ valC0, double* __restrict__ valC1, double* __restrict__ valC2, double* __restrict__ valH) {
    constexpr int MAXMETA = GATHER_MAXMETA;
    if (doK) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bool zrow = (zmask >> i) & 1u;
            double* dst = valK + 9 * ptr_c + (long long)i * 3 * deg_c;
            for (int c = lane; c < 3 * (int)deg_c; c += 64) {
                const int k = c / 3, j = c - 3 * k;
                const unsigned meta = s_meta[k];
                const int ks = (meta & 127) == 127 ? -1 : int(meta & 127);
                double v = 0.0;
                if (zrow || (meta & (128u << j))) v = ((meta & 1024u) && i == j) ? 1.0 : 0.0;
                else { if (ks >= 0) v = aK[(i * NBOX + ks) * 3 + j]; if (padd) v += dst[c]; }
                dst[c] = v;
            }
        }
    }
    if constexpr (WITHC) if (doC) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bool zrow = (zmask >> i) & 1u;
            for (int k = lane; k < (int)deg_c; k += 64) {
                const unsigned meta = s_meta[k];
                const int ks = (meta & 127) == 127 ? -1 : int(meta & 127);
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    double* dst = (f == 0 ? valC0 : (f == 1 ? valC1 : valC2)) + 3 * ptr_c + (long long)i * deg_c + k;
                    double v = 0.0;
                    if (!zrow) { if (ks >= 0) v = aC[(f * 3 + i) * NBOX + ks]; if (padd) v += *dst; }
                    *dst = v;
                }
            }
        }
    }
    if (doH) {
        for (int k = lane; k < (int)deg_s; k += 64) {
#pragma unroll
            for (int i = 0; i < 3; ++i) valH[3 * ptr_s + (long long)i * deg_s + k] = aH[i * NBOX + k];
        }
    }
}

// One-wave gather: ONE wave per control point takes the three dof rows itself (WITHC: 24 row loads in flight per group of
// four elements, otherwise 12).  The four-wave gather above is bound by the latency chain of a workgroup rather than by
// bandwidth when few bytes are asked for (Newton pass R + K: 5.7 ms for 42 % of the bytes of the full pass); with one wave
// per control point 2-3x as many control points are resident per CU: R + K pass 19.7 -> 16.9 ms, dR/dh-only 10.5 -> 7.1,
// dR/dCP-only 21.7 -> 19.1, full pass 26.4 -> 25.9 ms at C4.  Same fixed element order per accumulator slot: bitwise the
// same sums as the four-wave kernel (which stays for GF_GATHER1=0).
#ifndef GF_GATHER1_UNR
#define GF_GATHER1_UNR 4
#endif
// With at most 88 registers a wave of this kernel then fits next to a wave of the MFMA element kernel (424 of the 512 registers of
// a SIMD), which is what lets the gather of one chunk run under the element kernel of the next (gf_lib.hip, overlap).
template <int P, bool WITHC>
__global__ __launch_bounds__(64) void kl_gather1_kernel(DevModel M, long long a_first, long long e_first, long long e_count, int flags,
                                                        const double* __restrict__ blk, double* __restrict__ valK, double* __restrict__ valC0,
                                                        double* __restrict__ valC1, double* __restrict__ valC2, double* __restrict__ valH,
                                                        double* __restrict__ R, int pen_add) {
    using Cfg = ElemCfg<P>;
    constexpr int P1 = Cfg::P1, NB = Cfg::NB, ND = Cfg::ND, WB = 2 * P + 1, NBOX = WB * WB;
    const long long a = a_first + blockIdx.x;
    if (a >= M.total_cp) return;
    const CpDesc& cd = M.cpdesc[a];
    const int ia = cd.ia, ja = cd.ja, i0 = cd.i0, j0 = cd.j0, wbox = cd.i1 - cd.i0 + 1;
    const long long ptr_c = M.nb_ptr_c[a], deg_c = M.nb_ptr_c[a + 1] - ptr_c, ptr_s = M.nb_ptr_s[a], deg_s = M.nb_ptr_s[a + 1] - ptr_s;
    const int lane = threadIdx.x;
    const unsigned zmask = (M.zero[3 * a] ? 1u : 0u) | (M.zero[3 * a + 1] ? 2u : 0u) | (M.zero[3 * a + 2] ? 4u : 0u);   // requested now, used by the write phase
    const bool pen_row_a = M.pen_row[a] != 0;
    __shared__ double aK[3][NBOX][3], aC[WITHC ? 3 : 1][3][NBOX], aH[3][NBOX], aR[3];
    constexpr int MAXMETA = GATHER_MAXMETA;
    __shared__ unsigned short s_meta[MAXMETA];
    for (int k = lane; k < (int)deg_c && k < MAXMETA; k += 64) s_meta[k] = M.nb_meta[ptr_c + k];
    for (int k = lane; k < 9 * NBOX; k += 64) { (&aK[0][0][0])[k] = 0.0; if constexpr (WITHC) (&aC[0][0][0])[k] = 0.0; }
    for (int k = lane; k < 3 * NBOX; k += 64) (&aH[0][0])[k] = 0.0;
    if (lane < 3) aR[lane] = 0.0;
    __syncthreads();
    constexpr int NPASS = (ND + 63) / 64, NPH = (3 * NB + 63) / 64, UNR = GF_GATHER1_UNR;
    const int neu = cd.neu, nev = cd.nev, ne = neu * nev;
    const bool doC = WITHC && (flags & GF_ASM_C_BIT) != 0;
    const bool doK = (flags & GF_ASM_K_BIT) != 0, doH = (flags & GF_ASM_H_BIT) != 0, doR = (flags & GF_ASM_R_BIT) != 0;
    for (int g0 = 0; g0 < ne; g0 += UNR) {
        const double* Bp[UNR]; int bu[UNR], bv[UNR], al[UNR]; bool ok[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int g = g0 + q, gg = g < ne ? g : 0, ku = gg % (neu > 0 ? neu : 1), kv = gg / (neu > 0 ? neu : 1);
            const long long e = cd.e00 + ku + (long long)kv * cd.nelu - e_first;
            ok[q] = g < ne && e >= 0 && e < e_count;
            Bp[q] = blk + (size_t)(ok[q] ? e : 0) * Cfg::BLK;
            bu[q] = cd.bu[ku]; bv[q] = cd.bv[kv]; al[q] = (ia - bu[q]) + (ja - bv[q]) * P1;
        }
        double vK[UNR][3][NPASS], vC[WITHC ? UNR : 1][3][NPASS], vH[UNR][NPH], vR[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int c = lane + 64 * ps;
                    vK[q][i][ps] = (ok[q] && c < ND && doK) ? Bp[q][Cfg::OFF_K + (3 * al[q] + i) * ND + c] : 0.0;
                    if constexpr (WITHC) vC[q][i][ps] = (ok[q] && c < ND && doC) ? Bp[q][Cfg::OFF_C + (3 * al[q] + i) * ND + c] : 0.0;
                }
#pragma unroll
            for (int ps = 0; ps < NPH; ++ps) {
                const int w = lane + 64 * ps;
                vH[q][ps] = (ok[q] && w < 3 * NB && doH) ? Bp[q][Cfg::OFF_H + (3 * al[q] + w / NB) * NB + w % NB] : 0.0;
            }
            vR[q] = (ok[q] && lane < 3 && doR) ? Bp[q][Cfg::OFF_R + 3 * al[q] + lane] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            if (!ok[q]) continue;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int c = lane + 64 * ps;
                if (c < ND) {
                    const int bl = c / 3, j = c - 3 * bl, ks = (bu[q] + bl % P1 - i0) + (bv[q] + bl / P1 - j0) * wbox;
#pragma unroll
                    for (int i = 0; i < 3; ++i) { aK[i][ks][j] += vK[q][i][ps]; if constexpr (WITHC) aC[j][i][ks] += vC[q][i][ps]; }
                }
            }
#pragma unroll
            for (int ps = 0; ps < NPH; ++ps) {
                const int w = lane + 64 * ps;
                if (w < 3 * NB) { const int i = w / NB, bl = w - i * NB, ks = (bu[q] + bl % P1 - i0) + (bv[q] + bl / P1 - j0) * wbox; aH[i][ks] += vH[q][ps]; }
            }
            if (lane < 3) aR[lane] += vR[q];
        }
    }
    __syncthreads();
    const bool padd = pen_add && pen_row_a;
    gather_write_rows<NBOX, WITHC>(M, a, lane, doK, doC, doH, padd, zmask, ptr_c, deg_c, ptr_s, deg_s, s_meta, &aK[0][0][0], &aC[0][0][0], &aH[0][0],
                                   valK, valC0, valC1, valC2, valH);
    if (doR && lane < 3) R[3 * a + lane] = aR[lane] + (padd ? R[3 * a + lane] : 0.0);
}

// Residual-only gather (flags == R, e.g. DispImOpeartion.apply_nonlinear, line searches): one thread per owned control
// point sums the three residual entries of its <= (p+1)^2 element blocks in the same fixed order as kl_gather_kernel.
template <int P>
__global__ __launch_bounds__(256) void kl_rgather_kernel(DevModel M, long long a_first, long long a_end, long long e_first, long long e_count,
                                                          const double* __restrict__ blk, double* __restrict__ R, int pen_add, int blk_stride, int off_r) {
    constexpr int P1 = P + 1;                               // blk_stride / off_r: element-block layout (ElemCfg) or the residual-only blocks of the walking kernel
    const long long a = a_first + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= a_end) return;
    const PatchDev& Pt = M.patches[M.cp_patch[a]];
    const int la = int(a - Pt.cp_off), ia = la % Pt.nu, ja = la / Pt.nu;
    const int* spu = M.ints + Pt.spu; const int* spv = M.ints + Pt.spv; const int* c2u = M.ints + Pt.c2u; const int* c2v = M.ints + Pt.c2v;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int ev = c2v[2 * ja]; ev <= c2v[2 * ja + 1]; ++ev) for (int eu = c2u[2 * ia]; eu <= c2u[2 * ia + 1]; ++eu) {
        const long long e = Pt.elem_off + eu + (long long)ev * Pt.nelu - e_first;
        if (e < 0 || e >= e_count) continue;
        const int al = (ia - (spu[eu] - P)) + (ja - (spv[ev] - P)) * P1;
        const double* B = blk + (size_t)e * blk_stride + off_r + 3 * al;
        for (int i = 0; i < 3; ++i) acc[i] += B[i];
    }
    const bool padd = pen_add && M.pen_row[a];
    for (int i = 0; i < 3; ++i) R[3 * a + i] = acc[i] + (padd ? R[3 * a + i] : 0.0);
}

// ------------------------------------------------------------------------------------- functionals
// K10 of SURVEY.md 2.2: strain energy W = sum int Psi, volume V = sum int t dA and their gradients.
// One wave works on NE = 64 / (p+1)^2 elements at once (4 for p = 3): phase A one lane per (element, Gauss point) --
// sum-factorised kinematics and the pointwise function --, phase B one lane per (element, control point, component)
// contraction with the basis.  Per-element gradients go to a block [NB][11] (+ We, Ve), summed per control point by
// kl_fgather_kernel in fixed element order.
//   KIND 0  strain energy and volume (IntEnergyExOperation / VolumeExOperation): slots 0-2 dW/du, 3-5 the reference part
//           of dW/dc, 6-8 dV/dc, 9 dW/dh, 10 dV/dh; We = W_e, Ve = V_e
//   KIND 1  stress aggregation forms (MaxvMStressExOperation, operations/max_vmstress_exop.py:167-186): per element
//           sum_gp wq J g(sigma_vM), g = exp(rho (sigma - m_s)) [mode 0] or (sigma / m_s)^rho [mode 1], sigma_vM at the
//           station xi3 = sgn t/2 (shell_stress_point): slots 0-2 dI/du, 3-5 reference part of dI/dc, 9 dI/dh (6-8, 10
//           zero); We = I_e, Ve = max_gp sigma_vM
//   KIND 2  shape regularisation of the eVTOL demo (demos_om/shape_opt/eVTOL/int_energy_regu_exop.py:30-38):
//           sum_gp wq c_s |grad_s(P_f - P_f^0)|^2 J with the surface gradient on the CURRENT geometry,
//           |grad_s D|^2 J = |D_,1 G2 - D_,2 G1|^2 / J (D = homogeneous coordinate f minus its initial value);
//           slots 3-5 d/dc (0-2, 6-10 zero); We = value, Ve = 0
template <int P> struct FunCfg { static constexpr int NB = (P + 1) * (P + 1), STRIDE = NB * 11 + 2; };
struct StressCfg { int mode, measure; double rho, sgn; const double* m_list; int field; const double* cp0; };   // KIND 2: field, cp0 (initial homogeneous coordinate), m_list = coefficient per patch

template <int P, int KIND>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KIND == 2 ? 1 : 2))) void kl_pointfun_kernel(DevModel M, int e_first, int e_count, StressCfg S,
                                                                                                 double* __restrict__ blk, size_t blk_stride) {
    constexpr int P1 = P + 1, NB = P1 * P1, NG = NB, ND = 3 * NB, NE = 64 / NG, TS = P1 * 3 * P1;
    const int tid = threadIdx.x;
    const long long eb = (long long)blockIdx.x * NE;                   // first element of this wave (local to the chunk)
    __shared__ double s_c[NE][NB][3], s_d[NE][NB][3], s_h[NE][NB], s_w[NE][NB];
    __shared__ double s_tu[NE][TS], s_tv[NE][TS], s_wg[NE][2 * P1];
    __shared__ double s_fe[NE][NG][FE_SIZE + 8];                       // + W[6], wq, t (KIND 0) / sigma (KIND 1)
    __shared__ double s_c0[KIND == 2 ? NE : 1][NB];                    // KIND 2: initial coordinate of the regularised field
    for (int k = tid; k < NE * NB; k += 64) {
        const int el = k / NB, a = k - el * NB;
        if (eb + el >= e_count) continue;
        const ElemDesc& ed = M.edesc[e_first + eb + el];
        const long long g = ed.g0 + (a % P1) + (long long)(a / P1) * ed.nu;
        const double4 c4 = reinterpret_cast<const double4*>(M.cp4)[g];
        s_c[el][a][0] = c4.x; s_c[el][a][1] = c4.y; s_c[el][a][2] = c4.z; s_w[el][a] = c4.w;
        s_d[el][a][0] = M.u[3 * g]; s_d[el][a][1] = M.u[3 * g + 1]; s_d[el][a][2] = M.u[3 * g + 2];      // displacement coefficients (kl_strains)
        s_h[el][a] = M.h[g];
        if constexpr (KIND == 2) s_c0[el][a] = S.cp0[g];
    }
    for (int k = tid; k < NE * TS; k += 64) {
        const int el = k / TS, j = k - el * TS;
        if (eb + el >= e_count) continue;
        const ElemDesc& ed = M.edesc[e_first + eb + el];
        s_tu[el][j] = M.tab[ed.tabu + j]; s_tv[el][j] = M.tab[ed.tabv + j];
    }
    for (int k = tid; k < NE * 2 * P1; k += 64) {
        const int el = k / (2 * P1), j = k - el * 2 * P1;
        if (eb + el >= e_count) continue;
        const ElemDesc& ed = M.edesc[e_first + eb + el];
        s_wg[el][j] = j < P1 ? M.tab[ed.wu + j] : M.tab[ed.wv + j - P1];
    }
    __syncthreads();
    {
        const int el = tid / NG, gpi = tid - el * NG;
        if (el < NE && eb + el < e_count) {
            const int pid = M.edesc[e_first + eb + el].patch;
            const PatchDev& Pt = M.patches[pid];
            const int gu = gpi % P1, gv = gpi / P1;
            // sum factorisation over the tensor-product basis + quotient rule (as in the MFMA element kernels)
            double Ac[3][6], Ad[3][6], W[6], t = 0.0;
            for (int k = 0; k < 6; ++k) { W[k] = 0.0; for (int i = 0; i < 3; ++i) { Ac[i][k] = 0.0; Ad[i][k] = 0.0; } }
            double U[3][P1];
            for (int d = 0; d < 3; ++d) for (int j = 0; j < P1; ++j) U[d][j] = s_tu[el][(gu * 3 + d) * P1 + j];
            double D1 = 0.0, D2 = 0.0;                                  // KIND 2: non-rational derivatives of the coordinate difference
#pragma unroll
            for (int jv = 0; jv < P1; ++jv) {
                const double v0 = s_tv[el][(gv * 3 + 0) * P1 + jv], v1 = s_tv[el][(gv * 3 + 1) * P1 + jv], v2 = s_tv[el][(gv * 3 + 2) * P1 + jv];
                double Sm[7][3], Sh = 0.0, Sd0 = 0.0, Sd1 = 0.0;
                for (int q = 0; q < 7; ++q) for (int d = 0; d < 3; ++d) Sm[q][d] = 0.0;
#pragma unroll
                for (int ju = 0; ju < P1; ++ju) {
                    const int a = ju + P1 * jv;
                    const double qv[7] = {s_c[el][a][0], s_c[el][a][1], s_c[el][a][2], s_d[el][a][0], s_d[el][a][1], s_d[el][a][2], s_w[el][a]};
                    for (int q = 0; q < 7; ++q) for (int d = 0; d < 3; ++d) Sm[q][d] += U[d][ju] * qv[q];
                    Sh += U[0][ju] * s_h[el][a];
                    if constexpr (KIND == 2) {
                        const double dc = (S.field == 0 ? qv[0] : (S.field == 1 ? qv[1] : qv[2])) - s_c0[el][a];
                        Sd0 += U[0][ju] * dc; Sd1 += U[1][ju] * dc;
                    }
                }
                t += v0 * Sh;
                if constexpr (KIND == 2) { D1 += v0 * Sd1; D2 += v1 * Sd0; }
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    double* A = q < 3 ? Ac[q] : (q < 6 ? Ad[q - 3] : W);
                    A[0] += v0 * Sm[q][0]; A[1] += v0 * Sm[q][1]; A[2] += v1 * Sm[q][0];
                    A[3] += v0 * Sm[q][2]; A[4] += v2 * Sm[q][0]; A[5] += v1 * Sm[q][1];
                }
            }
            W[0] = 1.0 / W[0];
            double z[15], Z[15], dz[15], R[6];
            for (int i = 0; i < 3; ++i) {
                rationalize6(Ac[i], W, R);
                for (int m = 0; m < 5; ++m) Z[3 * m + i] = R[m + 1];
                rationalize6(Ad[i], W, R);
                for (int m = 0; m < 5; ++m) { dz[3 * m + i] = R[m + 1]; z[3 * m + i] = Z[3 * m + i] + R[m + 1]; }
            }
            double* fe = s_fe[el][gpi];
            if constexpr (KIND == 2) {
                // D_,alpha: non-rational derivatives of the homogeneous coordinate difference (spline.grad(cpFuncs[f]) of the reference)
                double wv[3], nt[3];
                for (int k = 0; k < 3; ++k) wv[k] = D1 * Z[3 + k] - D2 * Z[k];
                cross3(Z, Z + 3, nt);
                const double J = sqrt(dot3(nt, nt)), iJ = 1.0 / J, r = dot3(wv, wv) * iJ, cf = S.m_list[pid];
                double JZ[6], nn[3] = {nt[0] * iJ, nt[1] * iJ, nt[2] * iJ};
                cross3(Z + 3, nn, JZ); cross3(nn, Z, JZ + 3);
                for (int c = 0; c < FE_SIZE; ++c) fe[c] = 0.0;
                fe[FE_PSI] = cf * r;
                for (int k = 0; k < 3; ++k) {
                    fe[FE_PZR + k] = cf * (-2.0 * D2 * wv[k] * iJ - r * JZ[k] * iJ);
                    fe[FE_PZR + 3 + k] = cf * (2.0 * D1 * wv[k] * iJ - r * JZ[3 + k] * iJ);
                }
                fe[FE_PT] = cf * 2.0 * dot3(wv, Z + 3) * iJ;            // d/dD_,1  (slots reused: no thickness derivative here)
                fe[FE_J] = -cf * 2.0 * dot3(wv, Z) * iJ;                 // d/dD_,2
                fe[FE_SIZE + 7] = 0.0;
            } else if constexpr (KIND == 0) {
                shell_energy_point(z, Z, dz, t, Pt.E, Pt.nu_, fe);
                fe[FE_SIZE + 7] = t;
            } else {
                shell_stress_point(z, Z, dz, t, Pt.E, Pt.nu_, S.sgn, S.measure, fe);
                const double sig = fe[0], J = fe[1], ms = S.m_list[pid];
                double g, gp;                                           // g(sigma), g'(sigma)
                if (S.mode == 0) { g = exp(S.rho * (sig - ms)); gp = S.rho * g; }
                else { const double r = sig / ms; gp = sig > 0.0 ? S.rho * pow(r, S.rho - 1.0) / ms : 0.0; g = sig > 0.0 ? pow(r, S.rho) : 0.0; }
                const double Jgp = J * gp;
                fe[FE_PSI] = J * g; fe[FE_PT] *= Jgp;
                for (int c = 0; c < 15; ++c) { fe[FE_PZ + c] *= Jgp; fe[FE_PZR + c] = Jgp * fe[FE_PZR + c] + (c < 6 ? g * fe[FE_JZ + c] : 0.0); }
                fe[FE_SIZE + 7] = sig;
            }
            for (int k = 0; k < 6; ++k) fe[FE_SIZE + k] = W[k];
            fe[FE_SIZE + 6] = s_wg[el][gu] * s_wg[el][P1 + gv];
        }
    }
    __syncthreads();
    for (int w = tid; w < NE * ND; w += 64) {
        const int el = w / ND, r = w - el * ND;
        if (eb + el >= e_count) continue;
        const int a = r / 3, i = r - 3 * a;
        double du = 0.0, dc = 0.0, dv = 0.0, dwh = 0.0, dvh = 0.0;
        for (int gp = 0; gp < NG; ++gp) {
            const double* fe = s_fe[el][gp]; const double wq = fe[FE_SIZE + 6];
            double Nb[6], R[6];
            bspline6<P>(s_tu[el], s_tv[el], gp % P1, gp / P1, a, Nb); rationalize6(Nb, fe + FE_SIZE, R);
            double pz = 0.0, pZ = 0.0;
            for (int m = 0; m < 5; ++m) { pz += R[m + 1] * fe[FE_PZ + 3 * m + i]; pZ += R[m + 1] * fe[FE_PZR + 3 * m + i]; }
            du += wq * pz; dc += wq * pZ;
            if constexpr (KIND == 2) { if (i == S.field) dc += wq * (Nb[1] * fe[FE_PT] + Nb[2] * fe[FE_J]); }
            else dwh += wq * Nb[0] * fe[FE_PT];
            if constexpr (KIND == 0) {
                dv += wq * fe[FE_SIZE + 7] * (R[1] * fe[FE_JZ + i] + R[2] * fe[FE_JZ + 3 + i]);
                dvh += wq * Nb[0] * fe[FE_J];
            }
        }
        double* out = blk + (size_t)(eb + el) * blk_stride;
        out[a * 11 + i] = du; out[a * 11 + 3 + i] = dc; out[a * 11 + 6 + i] = dv;
        if (i == 0) { out[a * 11 + 9] = dwh; out[a * 11 + 10] = dvh; }
    }
    if (tid < NE && eb + tid < e_count) {
        double We = 0.0, Ve = 0.0;
        for (int gp = 0; gp < NG; ++gp) {
            const double* fe = s_fe[tid][gp]; const double wq = fe[FE_SIZE + 6];
            We += wq * fe[FE_PSI];
            if constexpr (KIND == 0) Ve += wq * fe[FE_SIZE + 7] * fe[FE_J]; else if constexpr (KIND == 1) Ve = fmax(Ve, fe[FE_SIZE + 7]);
        }
        double* out = blk + (size_t)(eb + tid) * blk_stride;
        out[NB * 11] = We; out[NB * 11 + 1] = Ve;
    }
}

// Compliance C = sum int f . u_hom dA (homogeneous displacement function, compliance_exop.py:24-26) and its
// gradients; same block layout as kl_pointfun_kernel: slots 0-2 dC/du, 6-8 dC/dc, We = C_e (3-5, 9, 10 zero).
template <int P>
__global__ __launch_bounds__(64) void kl_compliance_kernel(DevModel M, int e_first, const double* __restrict__ forces, double* __restrict__ blk, size_t blk_stride) {
    constexpr int P1 = P + 1, NB = P1 * P1, NG = NB, ND = 3 * NB;
    const int tid = threadIdx.x;
    const long long e = (long long)e_first + blockIdx.x;
    if (e >= M.nelem) return;
    const int pid = M.elem_patch[e];
    const PatchDev& Pt = M.patches[pid];
    const int le = int(e - Pt.elem_off), eu = le % Pt.nelu, ev = le / Pt.nelu;
    const int iu0 = M.ints[Pt.spu + eu] - P, iv0 = M.ints[Pt.spv + ev] - P;
    const double f0 = forces[3 * pid], f1 = forces[3 * pid + 1], f2 = forces[3 * pid + 2];
    __shared__ double s_c[NB][3], s_u[NB][3], s_w[NB];
    __shared__ double s_tu[P1 * 3 * P1], s_tv[P1 * 3 * P1], s_wg[2 * P1];
    __shared__ double s_g[NG][12];          // wq*J, wq*(f.u_hom), J1[3], J2[3], 1/W, W_1, W_2
    if (tid < NB) {
        const long long g = Pt.cp_off + (iu0 + tid % P1) + (long long)(iv0 + tid / P1) * Pt.nu;
        for (int k = 0; k < 3; ++k) { s_c[tid][k] = M.cp4[4 * g + k]; s_u[tid][k] = M.u[3 * g + k]; }
        s_w[tid] = M.cp4[4 * g + 3];
    }
    for (int k = tid; k < P1 * 3 * P1; k += 64) { s_tu[k] = M.tab[Pt.tabu + eu * P1 * 3 * P1 + k]; s_tv[k] = M.tab[Pt.tabv + ev * P1 * 3 * P1 + k]; }
    if (tid < P1) { s_wg[tid] = M.tab[Pt.wu + eu * P1 + tid]; s_wg[P1 + tid] = M.tab[Pt.wv + ev * P1 + tid]; }
    __syncthreads();
    if (tid < NG) {
        const int gu = tid % P1, gv = tid / P1;
        double W[6] = {0, 0, 0, 0, 0, 0}, Nb[6], R[6];
        for (int a = 0; a < NB; ++a) { bspline6<P>(s_tu, s_tv, gu, gv, a, Nb); for (int k = 0; k < 6; ++k) W[k] += Nb[k] * s_w[a]; }
        W[0] = 1.0 / W[0];
        double G1[3] = {0, 0, 0}, G2[3] = {0, 0, 0}, Uh[3] = {0, 0, 0};
        for (int a = 0; a < NB; ++a) {
            bspline6<P>(s_tu, s_tv, gu, gv, a, Nb); rationalize6(Nb, W, R);
            for (int k = 0; k < 3; ++k) { G1[k] += R[1] * s_c[a][k]; G2[k] += R[2] * s_c[a][k]; Uh[k] += Nb[0] * s_u[a][k]; }
        }
        double Nt[3]; cross3(G1, G2, Nt);
        const double J = sqrt(dot3(Nt, Nt)), wq = s_wg[gu] * s_wg[P1 + gv];
        const double Nn[3] = {Nt[0] / J, Nt[1] / J, Nt[2] / J};
        double J1[3], J2[3]; cross3(G2, Nn, J1); cross3(Nn, G1, J2);
        const double fu = f0 * Uh[0] + f1 * Uh[1] + f2 * Uh[2];
        s_g[tid][0] = wq * J; s_g[tid][1] = wq * fu;
        for (int k = 0; k < 3; ++k) { s_g[tid][2 + k] = J1[k]; s_g[tid][5 + k] = J2[k]; }
        s_g[tid][8] = W[0]; s_g[tid][9] = W[1]; s_g[tid][10] = W[2]; s_g[tid][11] = J * fu * wq;
    }
    __syncthreads();
    double* out = blk + (size_t)blockIdx.x * blk_stride;
    for (int w = tid; w < ND; w += 64) {
        const int a = w / 3, i = w - 3 * a;
        const double fi = i == 0 ? f0 : (i == 1 ? f1 : f2);
        double du = 0.0, dc = 0.0;
        for (int gp = 0; gp < NG; ++gp) {
            double Nb[6];
            bspline6<P>(s_tu, s_tv, gp % P1, gp / P1, a, Nb);
            const double iW = s_g[gp][8], R0 = Nb[0] * iW, R1 = (Nb[1] - R0 * s_g[gp][9]) * iW, R2 = (Nb[2] - R0 * s_g[gp][10]) * iW;
            du += s_g[gp][0] * fi * Nb[0];
            dc += s_g[gp][1] * (s_g[gp][2 + i] * R1 + s_g[gp][5 + i] * R2);
        }
        out[a * 11 + i] = du; out[a * 11 + 3 + i] = 0.0; out[a * 11 + 6 + i] = dc;
        if (i == 0) { out[a * 11 + 9] = 0.0; out[a * 11 + 10] = 0.0; }
    }
    if (tid == 0) { double Ce = 0.0; for (int gp = 0; gp < NG; ++gp) Ce += s_g[gp][11]; out[NB * 11] = Ce; out[NB * 11 + 1] = 0.0; }
}

// one thread per owned control point: ordered sum over its elements.  fun: [dWdu ndof | dWdcp 3*tcp | dWdh tcp | dVdcp 3*tcp | dVdh tcp]
template <int P>
__global__ __launch_bounds__(256) void kl_fgather_kernel(DevModel M, long long a_first, long long a_end, long long e_first, long long e_count, int apply_bcs,
                                                          const double* __restrict__ blk, size_t blk_stride, double* __restrict__ fun, double* __restrict__ We, double* __restrict__ Ve) {
    constexpr int P1 = P + 1;
    const long long a = a_first + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long eidx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (eidx < e_count) { We[e_first + eidx] = blk[(size_t)eidx * blk_stride + P1 * P1 * 11]; Ve[e_first + eidx] = blk[(size_t)eidx * blk_stride + P1 * P1 * 11 + 1]; }
    if (a >= a_end) return;
    const PatchDev& Pt = M.patches[M.cp_patch[a]];
    const int la = int(a - Pt.cp_off), ia = la % Pt.nu, ja = la / Pt.nu;
    const int* spu = M.ints + Pt.spu; const int* spv = M.ints + Pt.spv; const int* c2u = M.ints + Pt.c2u; const int* c2v = M.ints + Pt.c2v;
    double acc[11];
    for (int k = 0; k < 11; ++k) acc[k] = 0.0;
    for (int ev = c2v[2 * ja]; ev <= c2v[2 * ja + 1]; ++ev) for (int eu = c2u[2 * ia]; eu <= c2u[2 * ia + 1]; ++eu) {
        const long long e = Pt.elem_off + eu + (long long)ev * Pt.nelu - e_first;
        if (e < 0 || e >= e_count) continue;
        const int al = (ia - (spu[eu] - P)) + (ja - (spv[ev] - P)) * P1;
        const double* B = blk + (size_t)e * blk_stride + al * 11;
        for (int k = 0; k < 11; ++k) acc[k] += B[k];
    }
    const long long T = M.total_cp;
    for (int i = 0; i < 3; ++i) {
        fun[3 * a + i] = (apply_bcs && M.zero[3 * a + i]) ? 0.0 : acc[i];
        fun[3 * T + i * T + a] = acc[i] + acc[3 + i];
        fun[7 * T + i * T + a] = acc[6 + i];
    }
    fun[6 * T + a] = acc[9]; fun[10 * T + a] = acc[10];
}
// Per-segment sums (and maxima) of per-element / per-vertex values, in fixed order: one workgroup per segment (the elements of a
// patch, the vertices of an interface), strided partial sums per thread, fixed tree.  The few per-segment numbers are what crosses
// to the host (gf_functionals, gf_compliance, gf_stress_forms, gf_shape_regu) instead of nelem-long arrays.
__global__ __launch_bounds__(256) void seg_reduce_kernel(const long long* __restrict__ seg_off, const double* __restrict__ a, const double* __restrict__ b,
                                                         double* __restrict__ sum_a, double* __restrict__ sum_b, double* __restrict__ max_b) {
    __shared__ double sa[256], sb[256], sm[256];
    const long long i0 = seg_off[blockIdx.x], i1 = seg_off[blockIdx.x + 1];
    double xa = 0.0, xb = 0.0, xm = 0.0;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) { xa += a[i]; if (b) { xb += b[i]; xm = fmax(xm, b[i]); } }
    sa[threadIdx.x] = xa; sb[threadIdx.x] = xb; sm[threadIdx.x] = xm;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { sa[threadIdx.x] += sa[threadIdx.x + off]; sb[threadIdx.x] += sb[threadIdx.x + off]; sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + off]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sum_a[blockIdx.x] = sa[0]; if (sum_b) sum_b[blockIdx.x] = sb[0]; if (max_b) max_b[blockIdx.x] = sm[0]; }
}
__global__ void pen_energy_kernel(long long npts, const double* __restrict__ pbuf, double* __restrict__ en) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < npts) en[v] = pbuf[(size_t)v * PB_STRIDE + PB_EN];
}

// ------------------------------------------------------------------------------------------ penalty
struct DevPenalty {
    const int* pt_iface; const int* pt_base; const double* pt_nu; const double* pt_tau; const double* pt_wt;
    const int* if_patch; const double* if_alpha;
    const PenEntry* entries; const long long* ent_ptr; const int* row_cp;
    const unsigned short* slots;     // p <= 3: per visit and side the 16 window positions' indices in the row's neighbour list (pen_row16_kernel)
    long long npts, nrow_groups;
};

// one thread per mortar vertex: kinematics of both sides + pointwise gradient/Hessians -> pbuf
template <int P>
__global__ __launch_bounds__(64) void pen_point_kernel(DevModel M, DevPenalty Q, double* __restrict__ pbuf, int grad_only) {
    constexpr int P1 = P + 1, NB = P1 * P1;
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= Q.npts) return;
    const int itf = Q.pt_iface[v];
    double y[18], Y[12], dY[12];
    for (int k = 0; k < 18; ++k) y[k] = 0.0;
    for (int k = 0; k < 12; ++k) { Y[k] = 0.0; dY[k] = 0.0; }
    for (int sd = 0; sd < 2; ++sd) {
        const PatchDev& Pt = M.patches[Q.if_patch[2 * itf + sd]];
        const int iu0 = Q.pt_base[4 * v + 2 * sd], iv0 = Q.pt_base[4 * v + 2 * sd + 1];
        const double* nu = Q.pt_nu + ((size_t)v * 2 + sd) * 3 * NB;
        for (int a = 0; a < NB; ++a) {
            const long long g = Pt.cp_off + (iu0 + a % P1) + (long long)(iv0 + a / P1) * Pt.nu;
            const double r0 = nu[a], r1 = nu[NB + a], r2 = nu[2 * NB + a];
            for (int k = 0; k < 3; ++k) {
                const double c = M.cp4[4 * g + k], uu = M.u[3 * g + k];
                y[9 * sd + k] += r0 * uu; dY[6 * sd + k] += r1 * uu; dY[6 * sd + 3 + k] += r2 * uu;
                Y[6 * sd + k] += r1 * c; Y[6 * sd + 3 + k] += r2 * c;
            }
        }
        for (int k = 0; k < 6; ++k) y[9 * sd + 3 + k] = Y[6 * sd + k] + dY[6 * sd + k];      // deformed tangents = reference + displacement tangents
    }
    // grad_only: 0 = gradient + both Hessian blocks, 1 = gradient only, 2 = gradient + Hyy (Newton pass), 3 = gradient + HyC
    penalty_point(y, Y, dY, Q.pt_tau + 2 * v, Q.if_alpha[2 * itf], Q.if_alpha[2 * itf + 1], Q.pt_wt[v], pbuf + (size_t)v * PB_STRIDE, grad_only == 1,
                  grad_only == 2 ? 1 : (grad_only == 3 ? 2 : 3));
}

// Moving intersections (reference nonmatching_opt.py:1042-1341 dRIGAdxi_sub): derivative of the penalty
// residual rows of a mortar vertex along six directions -- 0..3: the vertex slides on side sd = dir / 2 in parametric
// direction d = dir % 2 (y, Y move with the second derivatives of the basis, and the row weights nu_a move), 4..5: the
// curve tangent tau_d changes (a neighbouring vertex of side A moved).  One thread per (vertex, direction): forward-mode
// pass of the vertex gradient in dual numbers, then the contraction with both sides' basis values:
//   out[(v*6 + dir)][sd'][a][i] = sum_m nu_m,a^(sd') d(grad)[9 sd' + 3 m + i] + [sd' == sd] sum_m d(nu_m,a)/d(xi_d) grad[9 sd + 3 m + i]
// REV = true: the blocks are not written -- each (vertex, direction) is contracted with lam over the rows this handle owns (control points below owned_cp,
// Dirichlet rows skipped: the reference zeroes them, nonmatching_opt.py:1057-1062) and only that scalar goes out: out[t] = sum_{side', a, i} block lam[dof]:
// the reverse-mode product (dR/dxi)^T lam without the blocks ever leaving the device (0.64 GB over PCIe at C4 for the 1.5 ms of this kernel).
template <int P, bool REV = false>
__global__ __launch_bounds__(64) void pen_dxi_kernel(DevModel M, DevPenalty Q, const double* __restrict__ pt_nu2, double* __restrict__ out, long long v_first, long long v_count,
                                                     const double* __restrict__ lam = nullptr, long long owned_cp = 0) {
    constexpr int P1 = P + 1, NB = P1 * P1;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // (vertex of the range, direction)
    if (t >= v_count * 6) return;
    const long long v = v_first + t / 6; const int dir = int(t % 6), sdd = dir >> 1, d = dir & 1;
    const int itf = Q.pt_iface[v];
    Dual y[18], Y[12], tau[2], gr[18];
    for (int k = 0; k < 18; ++k) y[k] = {0.0, 0.0};
    for (int k = 0; k < 12; ++k) Y[k] = {0.0, 0.0};
    for (int sd = 0; sd < 2; ++sd) {
        const PatchDev& Pt = M.patches[Q.if_patch[2 * itf + sd]];
        const int iu0 = Q.pt_base[4 * v + 2 * sd], iv0 = Q.pt_base[4 * v + 2 * sd + 1];
        const double* nu = Q.pt_nu + ((size_t)v * 2 + sd) * 3 * NB;
        const double* n2 = pt_nu2 + ((size_t)v * 2 + sd) * 3 * NB;
        const bool seed = dir < 4 && sd == sdd;
        for (int a = 0; a < NB; ++a) {
            const long long g = Pt.cp_off + (iu0 + a % P1) + (long long)(iv0 + a / P1) * Pt.nu;
            const double r0 = nu[a], r1 = nu[NB + a], r2 = nu[2 * NB + a];
            // d/dxi_d of (R, R_u, R_v): d = 0 -> (R_u, R_uu, R_uv), d = 1 -> (R_v, R_uv, R_vv)
            const double s0 = seed ? (d == 0 ? r1 : r2) : 0.0, s1 = seed ? (d == 0 ? n2[a] : n2[2 * NB + a]) : 0.0, s2 = seed ? (d == 0 ? n2[2 * NB + a] : n2[NB + a]) : 0.0;
            for (int k = 0; k < 3; ++k) {
                const double c = M.cp4[4 * g + k], uu = M.u[3 * g + k];
                y[9 * sd + k] = y[9 * sd + k] + Dual{r0 * uu, s0 * uu};
                y[9 * sd + 3 + k] = y[9 * sd + 3 + k] + Dual{r1 * (c + uu), s1 * (c + uu)};
                y[9 * sd + 6 + k] = y[9 * sd + 6 + k] + Dual{r2 * (c + uu), s2 * (c + uu)};
                Y[6 * sd + k] = Y[6 * sd + k] + Dual{r1 * c, s1 * c};
                Y[6 * sd + 3 + k] = Y[6 * sd + 3 + k] + Dual{r2 * c, s2 * c};
            }
        }
    }
    tau[0] = {Q.pt_tau[2 * v], dir == 4 ? 1.0 : 0.0}; tau[1] = {Q.pt_tau[2 * v + 1], dir == 5 ? 1.0 : 0.0};
    penalty_grad_t<Dual>(y, Y, tau, Q.if_alpha[2 * itf], Q.if_alpha[2 * itf + 1], Q.pt_wt[v], gr);
    double* o = REV ? nullptr : out + (size_t)t * (2 * NB * 3);
    double acc = 0.0;
    for (int sd = 0; sd < 2; ++sd) {
        const double* nu = Q.pt_nu + ((size_t)v * 2 + sd) * 3 * NB;
        const double* n2 = pt_nu2 + ((size_t)v * 2 + sd) * 3 * NB;
        const bool seed = dir < 4 && sd == sdd;
        const PatchDev& Pt = M.patches[Q.if_patch[2 * itf + sd]];
        const int iu0 = Q.pt_base[4 * v + 2 * sd], iv0 = Q.pt_base[4 * v + 2 * sd + 1];
        for (int a = 0; a < NB; ++a) {
            const double r0 = nu[a], r1 = nu[NB + a], r2 = nu[2 * NB + a];
            const double s0 = seed ? (d == 0 ? r1 : r2) : 0.0, s1 = seed ? (d == 0 ? n2[a] : n2[2 * NB + a]) : 0.0, s2 = seed ? (d == 0 ? n2[2 * NB + a] : n2[NB + a]) : 0.0;
            const long long g = Pt.cp_off + (iu0 + a % P1) + (long long)(iv0 + a / P1) * Pt.nu;
            for (int i = 0; i < 3; ++i) {
                const double b = r0 * gr[9 * sd + i].d + r1 * gr[9 * sd + 3 + i].d + r2 * gr[9 * sd + 6 + i].d
                               + s0 * gr[9 * sd + i].v + s1 * gr[9 * sd + 3 + i].v + s2 * gr[9 * sd + 6 + i].v;
                if constexpr (REV) { if (g < owned_cp && !M.zero[3 * g + i]) acc += b * lam[3 * g + i]; }
                else o[(sd * NB + a) * 3 + i] = b;
            }
        }
    }
    if constexpr (REV) out[t] = acc;
}

// Penalty rows of one owned control point a (one wave each): residual entries and the coupling blocks
// of K and dR/dCP.  Every lane OWNS up to PEN_SL neighbour slots k of a (k = lane + 64 sl) and keeps
// their 3x3 K and dR/dc blocks in registers.  The visits (mortar vertices whose support contains a) come
// from a host-built list that carries the vertex id, side, local index and both support windows, so the
// kernel has no dependent index loads; two visits are processed per iteration with all their loads issued
// together (the loop is latency bound).  Per visit the Hessian rows are contracted with nu_a once
// (w-vectors in LDS, four buffers: one barrier per pair); a lane whose slot's control point lies in the
// support window adds its blocks.  Fixed visit order, nothing shared: bitwise reproducible.
// The kernel WRITES the rows (the gather of these control points adds the shell part afterwards).
constexpr int PEN_MAXDEG = 64 * 5;
template <int P, int PEN_SL, bool WITHC = true, bool WITHK = true>     // WITHC = false (Newton pass): no dR/dCP blocks; WITHK = false (linearize after a Newton solve): no K blocks
__global__ __launch_bounds__(64) void pen_owner_kernel(DevModel M, DevPenalty Q, int flags, int maxdeg, const double* __restrict__ pbuf, double* __restrict__ R,
                                                        double* __restrict__ valK, double* __restrict__ valC0, double* __restrict__ valC1, double* __restrict__ valC2) {
    constexpr int P1 = P + 1, NB = P1 * P1;
    // workgroup w runs on XCD w % 8: give every XCD a contiguous range of row groups, so that the vertex records shared by
    // neighbouring control points are fetched into one L2 instead of eight (2.82 -> 2.73 ms at C4)
    const long long chunk = (Q.nrow_groups + 7) / 8;
    const long long gidx = (long long)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (gidx >= Q.nrow_groups) return;
    const int tid = threadIdx.x;
    const long long e0 = Q.ent_ptr[gidx], e1 = Q.ent_ptr[gidx + 1];
    const int a = Q.row_cp[gidx];
    const long long ptr_c = M.nb_ptr_c[a], deg_c = M.nb_ptr_c[a + 1] - ptr_c;
    const bool mats = (flags & (GF_ASM_K_BIT | GF_ASM_C_BIT)) != 0;
    // [buffer][i][side t][0..8 wK (m, j) | 9 pad | 10..15 wC (m', j)]: 128-byte groups read back as 16-byte pairs (-6 %).  Measured and
    // dropped: requesting the next pair's vertex data before the current pair is contracted (229 instead of 162 registers, two waves
    // per SIMD instead of three: 724 instead of 621 us on the 8 x 8-patch slice) -- the loop lives on occupancy.
    __shared__ __attribute__((aligned(16))) double s_w[4][3][2][16];
    // owned slots
    int sp[PEN_SL], si[PEN_SL], sj[PEN_SL];
    double kk[WITHK ? PEN_SL : 1][9], cc[WITHC ? PEN_SL : 1][9];
#pragma unroll
    for (int sl = 0; sl < PEN_SL; ++sl) {
        const int k = tid + 64 * sl;
        sp[sl] = -1; si[sl] = 0; sj[sl] = 0;
        if (mats && sl * 64 < maxdeg && k < deg_c) {
            const int bcp = M.nb_c[ptr_c + k], pb = M.cp_patch[bcp];
            const PatchDev& Pb = M.patches[pb];
            const int lb = int(bcp - Pb.cp_off);
            sp[sl] = pb; si[sl] = lb % Pb.nu; sj[sl] = lb / Pb.nu;
        }
        for (int q = 0; q < 9; ++q) { if constexpr (WITHK) kk[sl][q] = 0.0; if constexpr (WITHC) cc[sl][q] = 0.0; }
    }
    double racc = 0.0;                                  // tid < 3: residual entry (a, tid)
    const int iK = tid < 54 ? tid / 18 : 0, cK = tid < 54 ? tid - 18 * iK : 0, iC = tid < 36 ? tid / 12 : 0, cC = tid < 36 ? tid - 12 * iC : 0;
    int pair = 0;
    PenEntry En[2];                                     // visits of the next iteration, fetched one iteration ahead
    if (e0 < e1) { En[0] = Q.entries[e0]; En[1] = Q.entries[e0 + 1 < e1 ? e0 + 1 : e0]; }
    for (long long e = e0; e < e1; e += 2) {
        const int nu_ = (e + 1 < e1) ? 2 : 1;
        PenEntry E[2];
        E[0] = En[0]; E[1] = En[1];
        if (e + 2 < e1) { En[0] = Q.entries[e + 2]; En[1] = Q.entries[e + 3 < e1 ? e + 3 : e + 2]; }
        // -- all loads of the pair
        double n3[2][3], hk[2][3], hc[2][3], g3[2][3], bv[2][PEN_SL][3]; int tt[2][PEN_SL];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long v = E[u].v; const int s = E[u].sal >> 8, al = E[u].sal & 255;
            const double* na = Q.pt_nu + ((size_t)v * 2 + s) * 3 * NB; const double* pb = pbuf + (size_t)v * PB_STRIDE;
            n3[u][0] = na[al]; n3[u][1] = na[NB + al]; n3[u][2] = na[2 * NB + al];
            if (mats) {                                  // residual-only: the Hessian slots of the vertex record are not even computed
                if constexpr (WITHK) { const double* h = pb + PB_HYY + (9 * s + iK) * 18 + cK; hk[u][0] = h[0]; hk[u][1] = h[3 * 18]; hk[u][2] = h[6 * 18]; }
                else { hk[u][0] = hk[u][1] = hk[u][2] = 0.0; }
                if constexpr (WITHC) { const double* h = pb + PB_HYC + (9 * s + iC) * 12 + cC; hc[u][0] = h[0]; hc[u][1] = h[3 * 12]; hc[u][2] = h[6 * 12]; }
                else { hc[u][0] = hc[u][1] = hc[u][2] = 0.0; }
            } else { for (int q = 0; q < 3; ++q) { hk[u][q] = 0.0; hc[u][q] = 0.0; } }
            { const double* g = pb + PB_GRAD + 9 * s + (tid < 3 ? tid : 0); g3[u][0] = g[0]; g3[u][1] = g[3]; g3[u][2] = g[6]; }
#pragma unroll
            for (int sl = 0; sl < PEN_SL; ++sl) {
                // a self-interface (pA == pB) would need both sides per slot; not supported (rejected at create)
                const int t = sp[sl] < 0 ? -1 : (sp[sl] == E[u].pA ? 0 : (sp[sl] == E[u].pB ? 1 : -1));
                const int base = t == 1 ? E[u].baseB : E[u].baseA;
                const int di = si[sl] - (base & 0xffff), dj = sj[sl] - (base >> 16);
                const bool in = t >= 0 && di >= 0 && di <= P && dj >= 0 && dj <= P;
                tt[u][sl] = in ? t : -1;
                const double* nb = Q.pt_nu + ((size_t)v * 2 + (in ? t : 0)) * 3 * NB + (in ? di + dj * P1 : 0);
                if (mats) { bv[u][sl][0] = nb[0]; bv[u][sl][1] = nb[NB]; bv[u][sl][2] = nb[2 * NB]; } else { bv[u][sl][0] = bv[u][sl][1] = bv[u][sl][2] = 0.0; }
            }
        }
        // -- w-vectors of both visits, one barrier
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u < nu_) {
                double (*w)[2][16] = s_w[2 * pair + u];
                if constexpr (WITHK) { if (mats && tid < 54) w[iK][cK / 9][cK % 9] = n3[u][0] * hk[u][0] + n3[u][1] * hk[u][1] + n3[u][2] * hk[u][2]; }
                if constexpr (WITHC) { if (mats && tid < 36) w[iC][cC / 6][10 + cC % 6] = n3[u][0] * hc[u][0] + n3[u][1] * hc[u][1] + n3[u][2] * hc[u][2]; }
                if (tid < 3) racc += n3[u][0] * g3[u][0] + n3[u][1] * g3[u][1] + n3[u][2] * g3[u][2];
            }
        }
        __syncthreads();
        if (mats) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u >= nu_) continue;
                const double (*w)[2][16] = s_w[2 * pair + u];
#pragma unroll
                for (int sl = 0; sl < PEN_SL; ++sl) {
                    const int t = tt[u][sl];
                    if (t < 0) continue;
                    const double b0 = bv[u][sl][0], b1 = bv[u][sl][1], b2 = bv[u][sl][2];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        double wl[16];
                        const double2* wp = reinterpret_cast<const double2*>(w[i][t]);
#pragma unroll
                        for (int q = 0; q < 8; ++q) { if ((q < 5 && WITHK) || (q >= 5 && WITHC)) { const double2 v2 = wp[q]; wl[2 * q] = v2.x; wl[2 * q + 1] = v2.y; } else { wl[2 * q] = 0.0; wl[2 * q + 1] = 0.0; } }
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            if constexpr (WITHK) kk[sl][3 * i + j] += wl[j] * b0 + wl[3 + j] * b1 + wl[6 + j] * b2;
                            if constexpr (WITHC) cc[sl][3 * i + j] += wl[10 + j] * b1 + wl[13 + j] * b2;
                        }
                    }
                }
            }
        }
        pair ^= 1;            // the next pair writes the other two buffers: one barrier per pair suffices
    }
    if ((flags & GF_ASM_R_BIT) && tid < 3) R[3 * (long long)a + tid] = racc;          // the gather adds the shell part
    if (!mats) return;
#pragma unroll
    for (int sl = 0; sl < PEN_SL; ++sl) {
        if (sl * 64 >= maxdeg || sp[sl] < 0) continue;
        const int k = tid + 64 * sl;
        for (int i = 0; i < 3; ++i) {                    // every entry of the rows is written (Dirichlet rows/columns are overwritten by the gather)
            for (int j = 0; j < 3; ++j) {
                if constexpr (WITHK) { if (flags & GF_ASM_K_BIT) valK[9 * ptr_c + (long long)i * 3 * deg_c + 3 * k + j] = kk[sl][3 * i + j]; }
                if constexpr (WITHC) { if (flags & GF_ASM_C_BIT) { double* dst = j == 0 ? valC0 : (j == 1 ? valC1 : valC2); dst[3 * ptr_c + (long long)i * deg_c + k] = cc[sl][3 * i + j]; } }
            }
        }
    }
}

// point loads and Dirichlet rows of the residual
__global__ void residual_finish_kernel(long long ndof, const unsigned char* zero, long long npl, const long long* pl_dof, const double* pl_val, double* R) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npl) R[pl_dof[i]] -= pl_val[i];                 // dofs are distinct (duplicates pre-summed on the host)
}
__global__ void zero_rows_kernel(long long ndof, const unsigned char* zero, double* R) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ndof && zero[i]) R[i] = 0.0;
}

// ------------------------------------------------------------------------------------------ apply
// y += A x, one wave per control point a (its three dof rows share one pass over the neighbour list and
// the x gathers; every lane has up to nine independent, lane-contiguous value loads in flight).
// BW = 3: K (block columns 3*nb + j), BW = 1: dR/dCP, dR/dh (columns nb).  Fixed reduction order.
template <int BW>
__global__ __launch_bounds__(256) void csr_apply_kernel(long long ncp, const long long* __restrict__ nb_ptr, const int* __restrict__ nb,
                                                         const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
    const long long a = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (a >= ncp) return;
    const long long ptr = nb_ptr[a], deg = nb_ptr[a + 1] - ptr;
    const double* v = val + 3 * BW * ptr;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (long long k = lane; k < deg; k += 64) {
        const long long col = (long long)nb[ptr + k] * BW;
        double xv[BW];
#pragma unroll
        for (int j = 0; j < BW; ++j) xv[j] = x[col + j];
#pragma unroll
        for (int j = 0; j < BW; ++j) {
            s0 += v[BW * k + j] * xv[j];
            s1 += v[BW * deg + BW * k + j] * xv[j];
            s2 += v[2 * BW * deg + BW * k + j] * xv[j];
        }
    }
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    if (lane == 0) { y[3 * a] += s0; y[3 * a + 1] += s1; y[3 * a + 2] += s2; }
}
// y += A^T x for the ndof x total_cp matrices (dR/dCP_f, dR/dh) WITHOUT atomics: one wave per column (control point b) gathers
// its entries through the symmetric neighbour relation -- entry (row (a, i), column b) sits at position rev of a's list, where
// rev = position of b in a's list, precomputed -- and adds them in fixed order (lane partial sums over k, fixed shuffle tree).
// Rows with x = 0 are skipped (the ghost rows of a shard are not assembled).  Bitwise reproducible reverse-mode products.
template <int BW>
__global__ __launch_bounds__(256) void csr_apply_tdet_kernel(long long ncp, const long long* __restrict__ nb_ptr, const int* __restrict__ nb, const int* __restrict__ rev,
                                                              const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
    const long long b = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (b >= ncp) return;
    const long long ptr = nb_ptr[b], deg = nb_ptr[b + 1] - ptr;
    double s[BW];
#pragma unroll
    for (int j = 0; j < BW; ++j) s[j] = 0.0;
    for (long long k = lane; k < deg; k += 64) {
        const long long a = nb[ptr + k];
        const long long pa = nb_ptr[a], da = nb_ptr[a + 1] - pa;
        const double* v = val + 3 * BW * pa + BW * rev[ptr + k];         // BW = 1: dR/dCP_f, dR/dh; BW = 3: K (on a shard, where the local K is not symmetric)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double xv = x[3 * a + i];
            if (xv != 0.0) {
#pragma unroll
                for (int j = 0; j < BW; ++j) s[j] += v[(long long)i * BW * da + j] * xv;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < BW; ++j) {
        double t = s[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (lane == 0) y[BW * b + j] += t;
    }
}

}  // namespace gf
