// gf_element_strip.hpp -- p = 3 MFMA element kernel that accumulates along a strip of elements before writing.
//
// kl_element_mfma_kernel writes one 43 KB block per element and the gather reads every block back: 26 GB each way per C4
// step, 5.3x the algorithmic bytes.  Here one wave walks a strip of elements (fixed u-span eu, v-spans ev = 0 .. nelv-1 of a
// patch) and keeps the MFMA accumulators across elements: moving one control-point row up shifts the local index of both
// control points of a pair by one row, i.e. a -> a - 4 is the next accumulator register and b -> b - 4 is a DPP row shift by
// four lanes.  A pair (a, b) leaves the 4-row window when its lower row does; only then is it written, already summed over
// the (up to four) elements of the strip that contain it.  Output per strip: for every control-point row (v index, local u
// index, dof i) one record of STRIP_RS doubles [K: 28 neighbour slots x 3 | dR/dc: 28 x 3 | dR/dh: 28 | R], neighbour slot =
// (local u index of b) + 4 (row(b) - row(a) + 3).  2.3x fewer bytes written here and read by the gather, which sums the (at
// most four) strips containing a control point -- still in fixed order, without atomics.
//
// Status (measured, C4): gather 7.5 -> 4.9 ms and element-kernel traffic 26 -> 11.5 GB, but this kernel takes 22.6 ms against
// 15.1 ms for kl_element_mfma_kernel: the DPP shifts of the 72 accumulator doubles cost 16 % and the flush stores 15 % (in a
// persistent loop the next element's input loads queue behind them: vmcnt is in order).  Net 31.0 vs 26.4 ms per step, so the
// path is OFF by default (GF_STRIP=1 enables it; parity-tested).  Rotating the lane/register <-> control-point-row mapping
// instead of moving data was tried (four compile-time flush instances) and is slower still (register spills).  Next: stage the
// flush through LDS for wide stores issued after the next element's loads.
#pragma once
#include "gf_element_mfma.hpp"

namespace gf {

constexpr int STRIP_RS = 200;          // doubles per (row, dof) record: 84 K + 84 dR/dc + 28 dR/dh + 1 R, padded

// lane i <- lane i + 4 within each row of 16 lanes (zero fill): b -> b - 4
__device__ __forceinline__ double shl4(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)b, 0x104, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(b >> 32), 0x104, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ gf_d4 shift_row(gf_d4 v) { return gf_d4{shl4(v[1]), shl4(v[2]), shl4(v[3]), 0.0}; }

__global__ __launch_bounds__(64) void kl_element_strip_kernel(DevModel M, const StripDesc* __restrict__ strips, int s_first, int flags, double* __restrict__ scratch) {
    constexpr int P = 3, P1 = 4, NB = 16, NG = 16, ND = 48;
    const int tid = threadIdx.x, x = tid & 15, kk = tid >> 4;
    const StripDesc sd = strips[s_first + blockIdx.x];
    const PatchDev& Pt = M.patches[sd.patch];
    double* const sout = scratch + sd.out_off;

    __shared__ __attribute__((aligned(16))) double s_g[4 * ND];     // control-point staging (phases 0-1), residual reduction at a flush
    double (*s_c)[3] = reinterpret_cast<double (*)[3]>(s_g);
    double (*s_d)[3] = reinterpret_cast<double (*)[3]>(s_g + 3 * NB);
    double* s_h = s_g + 6 * NB; double* s_w = s_g + 7 * NB;
    __shared__ double s_tu[P1 * 3 * P1], s_tv[P1 * 3 * P1], s_wg[2 * P1];
    __shared__ __attribute__((aligned(16))) double s_im[NG][IM_SIZE];
    unsigned long long tstamp = 0; (void)tstamp;

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
    const bool doK = (flags & GF_ASM_K_BIT) != 0, doC = (flags & GF_ASM_C_BIT) != 0, doH = (flags & GF_ASM_H_BIT) != 0, doR = (flags & GF_ASM_R_BIT) != 0;
    const bool has_bf = (Pt.f[0] != 0.0) || (Pt.f[1] != 0.0) || (Pt.f[2] != 0.0);
    const int ju = x % P1, jv = x / P1;

    gf_d4 accK[6], accC[9], accH[3];
    for (int q = 0; q < 6; ++q) accK[q] = gf_d4{0, 0, 0, 0};
    for (int q = 0; q < 9; ++q) accC[q] = gf_d4{0, 0, 0, 0};
    for (int q = 0; q < 3; ++q) accH[q] = gf_d4{0, 0, 0, 0};
    double accR[3] = {0.0, 0.0, 0.0};                    // lane (x, kk): partial (over this lane's Gauss-point slot) residual of a = x

    // Write the pairs whose lower control-point row is the window's first row (global row `grow`), then move the window up.
    constexpr int IJ_I[6] = {0, 0, 0, 1, 1, 2}, IJ_J[6] = {0, 1, 2, 1, 2, 2};
    auto flush_shift = [&](int grow) {
        // residual of the leaving row: sum the four Gauss-point slots through LDS
        if (doR) {
            wave_lds_sync();
            if (x < 4) for (int i = 0; i < 3; ++i) s_g[(kk * 4 + x) * 3 + i] = accR[i];
            wave_lds_sync();
            if (tid < 12) { const int a = tid / 3, i = tid - 3 * a; sout[(size_t)((grow * 4 + a) * 3 + i) * STRIP_RS + 196] = s_g[tid] + s_g[12 + tid] + s_g[24 + tid] + s_g[36 + tid]; }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            // rr == 0: a = (kk, first row), every b of the window (offset x / 4 >= 0);  rr > 0: b in the first row (lanes x < 4), a in row rr
            if (rr > 0 && x >= 4) continue;
            if (grow + rr >= sd.nv) continue;                                     // rows past the patch (flushes after the last element)
            const int rowa = (grow + rr) * 4 + kk, rowb = (grow + (rr == 0 ? x / 4 : 0)) * 4 + x % 4;
            const int slotb = x % 4 + 4 * ((rr == 0 ? x / 4 : -rr) + 3), slota = kk + 4 * ((rr == 0 ? -(x / 4) : rr) + 3);
            const bool rowb_ok = rr > 0 || grow + x / 4 < sd.nv;
            double* const ra = sout + (size_t)rowa * 3 * STRIP_RS;
            double* const rb = sout + (size_t)rowb * 3 * STRIP_RS;
            if (doK) {
#pragma unroll
                for (int ij = 0; ij < 6; ++ij) {
                    const int i = IJ_I[ij], j = IJ_J[ij];
                    ra[i * STRIP_RS + 3 * slotb + j] = accK[ij][rr];
                    if (i < j && rowb_ok) rb[j * STRIP_RS + 3 * slota + i] = accK[ij][rr];
                }
            }
            if (doC) {
#pragma unroll
                for (int q = 0; q < 9; ++q) ra[(q / 3) * STRIP_RS + 84 + 3 * slotb + q % 3] = accC[q][rr];
            }
            if (doH) {
#pragma unroll
                for (int i = 0; i < 3; ++i) ra[i * STRIP_RS + 168 + slotb] = accH[i][rr];
            }
        }
        for (int q = 0; q < 6; ++q) accK[q] = shift_row(accK[q]);
        for (int q = 0; q < 9; ++q) accC[q] = shift_row(accC[q]);
        for (int q = 0; q < 3; ++q) accH[q] = shift_row(accH[q]);
        for (int i = 0; i < 3; ++i) accR[i] = shl4(accR[i]);
    };

    int grow = M.ints[Pt.spv] - P;                         // global control-point row of the window's first row
    for (int ev = 0; ev <= sd.nelv; ++ev) {
        // rows below the next element's window (after the last element: the remaining four rows) are complete
        const int iv0 = ev < sd.nelv ? M.ints[Pt.spv + ev] - P : grow + 4;
        while (grow < iv0) { flush_shift(grow); ++grow; }
        if (ev == sd.nelv) break;
        const long long e = (long long)sd.e_first + (long long)ev * Pt.nelu;
        const ElemDesc ed = M.edesc[e];
        // ---- phase 0
        double4 c4 = {0, 0, 0, 0}; double ux = 0, uy = 0, uz = 0, hh = 0, ttu = 0, ttv = 0, twu = 0, twv = 0;
        if (tid < NB) {
            const long long g = ed.g0 + (tid % P1) + (long long)(tid / P1) * ed.nu;
            c4 = reinterpret_cast<const double4*>(M.cp4)[g];
            ux = M.u[3 * g]; uy = M.u[3 * g + 1]; uz = M.u[3 * g + 2];
            hh = M.h[g];
        }
        if (tid < P1 * 3 * P1) { ttu = M.tab[ed.tabu + tid]; ttv = M.tab[ed.tabv + tid]; }
        if (tid < P1) { twu = M.tab[ed.wu + tid]; twv = M.tab[ed.wv + tid]; }
        wave_lds_sync();                                   // the previous element's (and flush's) LDS traffic is complete
        if (tid < NB) {
            s_c[tid][0] = c4.x; s_c[tid][1] = c4.y; s_c[tid][2] = c4.z; s_w[tid] = c4.w;
            s_d[tid][0] = c4.x + ux; s_d[tid][1] = c4.y + uy; s_d[tid][2] = c4.z + uz;
            s_h[tid] = hh;
        }
        if (tid < P1 * 3 * P1) { s_tu[tid] = ttu; s_tv[tid] = ttv; }
        if (tid < P1) { s_wg[tid] = twu; s_wg[P1 + tid] = twv; }
        wave_lds_sync();

        // ---- phase 1: three lanes per Gauss point (gp = x, part ic = kk < 3): kinematics + pointwise closed forms --------
        // Lane (gp, ic) sums component ic of the reference and deformed control points (sum factorisation over the tensor-product
        // basis: per row jv of control points the three u-sums, then the six (du, dv) combinations; the rational derivatives
        // follow by the quotient rule, rationalize6 being linear in the B-spline values), the three lanes exchange their
        // components through the Gauss point's (not yet written) record, and each produces the record columns c = ic, 3 + ic.
        {
            const int gp = x, ic = kk < 3 ? kk : 0, gu = gp % P1, gv = gp / P1;
            const bool act = kk < 3;
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
                shell_point_cols(z, Z, t, Pt.E, Pt.nu_, ic, dsel, kk == 0, im);
                if (kk == 0) {
                    for (int k = 0; k < 6; ++k) im[IM_W + k] = W[k];
                    im[IM_WQ] = s_wg[gu] * s_wg[P1 + gv];
                }
            }
        }
        wave_lds_sync();


        for (int grp = 0; grp < 4; ++grp) {
            const int gu = kk, gv = grp;                         // Gauss point of this lane's group: gp = gu + 4 gv
            const double* im = s_im[4 * grp + kk];
            const double wq = im[IM_WQ];
            // -- basis function x at this Gauss point (registers)
            double phi[5], R0, n0;
            {
                const double u0 = s_tu[(gu * 3 + 0) * P1 + ju], u1 = s_tu[(gu * 3 + 1) * P1 + ju], u2 = s_tu[(gu * 3 + 2) * P1 + ju];
                const double v0 = s_tv[(gv * 3 + 0) * P1 + jv], v1 = s_tv[(gv * 3 + 1) * P1 + jv], v2 = s_tv[(gv * 3 + 2) * P1 + jv];
                const double Nb[6] = {u0 * v0, u1 * v0, u0 * v1, u2 * v0, u0 * v2, u1 * v1};
                double R[6];
                rationalize6(Nb, im + IM_W, R);
                for (int k = 0; k < 5; ++k) phi[k] = R[k + 1];
                R0 = R[0]; n0 = Nb[0];
            }
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
                    const double zz = pzr * im[IM_JZJ + s] + e0 * im[IM_JDNV + s] + e1 * im[IM_JDNV + 6 + s] + e2 * im[IM_JDNV + 12 + s]
                                    - (b0 * im[IM_JDMO + s] + b1 * im[IM_JDMO + 6 + s] + b2 * im[IM_JDMO + 12 + s]);
                    gR[s] = g; hR[s] = g + zz;
                }
    #pragma unroll
                for (int c = 0; c < 3; ++c) {                                       // curvature columns (c, jj)
                    const double fc = (c == 2) ? 2.0 : 1.0;
                    const double gam = fc * (b0 * im[IM_CT3 + sym3(0, c)] + b1 * im[IM_CT3 + sym3(1, c)] + b2 * im[IM_CT3 + sym3(2, c)]);
                    const double alpha = mt * (fc * im[IM_CBG + 6 * c + rt]) + (1.0 - mt) * gam, beta = mt * im[IM_JMOF + c];
    #pragma unroll
                    for (int jj = 0; jj < 3; ++jj) {
                        const double g = im[IM_N + jj] * alpha - beta * im[IM_DN + 6 * jj + rt];
                        gR[6 + 3 * c + jj] = g; hR[6 + 3 * c + jj] = g - gam * im[IM_NB + jj];
                    }
                }
                    dpp_source_fence(gR); dpp_source_fence(hR);
            }
            // -- residual and dR/dh prefactors of basis function x at this Gauss point
            {
                const double ls = has_bf ? load_scalar(im, Pt.pd) : 0.0;
                for (int i = 0; i < 3; ++i) {
                    double rz = 0.0;
                    for (int m = 0; m < 5; ++m) rz += phi[m] * im[IM_PZ + 3 * m + i];
                    accR[i] += wq * (rz - ls * Pt.f[i] * R0);
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
                if (has_bf) {                    // d(-f . u dA)/dc : -w f_i R_a (dJ/dZ . phi_b)
                    const LoadGeom lg = load_geom(im, Pt.pd);
    #pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        const double jz = load_dz_dot(im, Pt.pd, lg, f, pb[0], pb[1]);
    #pragma unroll
                        for (int i = 0; i < 3; ++i) accC[3 * i + f] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pt.f[i] * R0, jz, accC[3 * i + f], 0, 0, 0);
                    }
                }
            }
        }

    }
}

}  // namespace gf
