// gf_element_rec4.hpp -- p = 4: the MFMA element kernel as a WALK over a strip of elements that keeps its accumulators across elements and
// stores every control-point pair once per work item ("row records"), like gf_element_rec.hpp does for p = 2, 3, and the gather of those records.
//
// Why: kl_element_mfma4_kernel writes one 105 KB block per element (2 x 75^2 + 75 x 25 + 75 doubles) and kl_gather_kernel<4> reads every block back:
// 41.7 + 50.9 GB of HBM-side traffic per pass over one GPU's share of C5 for 4.94 GB of algorithmic bytes (profiles/traffic_p4_c5share.json, round 4
// correction).  Walking along v, a pair of control points stays in ONE accumulator register while the window passes its rows, so it is stored once per
// strip it lies in (5 times instead of 25): 37.8 KB per element, nothing read back, no launch order.
//
// Tiles.  25 basis functions per element = a 5 x 5 window of control points.  Operand lane x of tile t holds the basis function
//     tile 0: u index x / 5 (0..2), v slot x % 5, x < 15        tile 1: u index 3 + x / 5 (3..4), v slot x % 5, x < 10
// with slot = (control-point row) mod 5: moving the window changes which entry of the 1-D v table a lane evaluates (jv = (slot - first row) mod 5), never
// where a sum sits.  The 2 x 2 x 15 accumulator tiles do not fit the register file (as in gf_element_mfma4.hpp), and accumulators that must survive from
// element to element cannot be time-multiplexed inside one walk, so the three passes of the block kernel become THREE WALKS (three launches), each with its
// own phase 1:   PASS 0: residual + K ((0,0), (1,1): i <= j; (1,0): all nine; the transposed entries are written from the symmetry K^(ij)[a][b] = K^(ji)[b][a])
//                PASS 1 / 2: dR/dCP + dR/dh for the b tile PASS - 1 (T_b feeds both a tiles).
//
// Record of row rho of a work item (Rec4Cfg<NC>::SZ doubles; NC = 21 values per ORDERED pair in the full layout, 9 for the Newton pass): the pairs (A, B) whose
// lower row is rho, in three PARTS, each written by exactly one of the walks (so no cache line is written by two kernels: with the components of all walks
// interleaved the dR/dCP walks read-modify-wrote each other's lines, 8.5 GB per launch for 5.2 GB of stores):
//     part K   (PASS 0):  9 components K^(ij) at 3 i + j,                                    B columns ub = 0..4
//     part C0  (PASS 1): 12 components dR/dCP^(if) at 3 i + f, dR/dh^(i) at 9 + i,           B columns ub = 0..2  (b tile 0)
//     part C1  (PASS 2): the same 12 components,                                             B columns ub = 3..4  (b tile 1)
// and inside a part (NCp components, Wp columns)
//     area 1  [ua][c][d][ub']        A = (iu0 + ua, rho),      B = (iu0 + ub, rho + d), d = 0..4
//     area 2  [ua][d - 1][c][ub']    A = (iu0 + ua, rho + d),  B = (iu0 + ub, rho),     d = 1..4
// -- every ordered pair with all its components, so the gather of a control point a reads contiguous runs ([ua = its u index]) and needs no mirrored reads.
// Record index = (item - first item of the chunk) * rec_rows + (rho - first row of the item).
// Reference path: the same integrals as kl_element_mfma4_kernel (GOLDFISH/nonmatching_opt.py:941-1015 via PENGoLINS' assembly).
#pragma once
#include "gf_gauss_loop.hpp"

namespace gf {

// part p = 0 (K), 1 (C0), 2 (C1): components, columns, first column, offset in the row record, size of its area 1
struct Rec4Part { int nc, w, ub0, off, a1; };
__host__ __device__ constexpr Rec4Part rec4_part(int p) {
    return p == 0 ? Rec4Part{9, 5, 0, 0, 5 * 9 * 5 * 5} : (p == 1 ? Rec4Part{12, 3, 0, 2025, 5 * 12 * 5 * 3} : Rec4Part{12, 2, 3, 2025 + 1620, 5 * 12 * 5 * 2});
}
template <int NC> struct Rec4Cfg { static constexpr int NPART = NC == 21 ? 3 : 1, SZ = NC == 21 ? 4725 : 2025; };
static_assert(rec4_part(0).a1 + 5 * 4 * 9 * 5 == 2025 && rec4_part(1).a1 + 5 * 4 * 12 * 3 == 1620 && rec4_part(2).off + rec4_part(2).a1 + 5 * 4 * 12 * 2 == 4725, "record parts");
struct Rec4Out { double* rec; double* rblk; int rec_rows; long long e_first; };     // rblk: 75 residual doubles per element of the chunk (element e at e - e_first)

__device__ __forceinline__ int mod5(int v) { int r = v % 5; return r < 0 ? r + 5 : r; }

template <int PASS, int NC>            // PASS 0: R + K; 1, 2: dR/dCP + dR/dh of b tile PASS - 1.  NC: values per pair in the record (21: full layout, 9: K only)
__global__ __launch_bounds__(64) void kl_element_rec4_kernel(DevModel M, const WalkItem* __restrict__ items, int flags, Rec4Out O) {
    static_assert(PASS == 0 || NC == 21, "the dR/dCP / dR/dh walks write the full record layout");
    using RC = Rec4Cfg<NC>;
    constexpr int P = 4, P1 = 5, NB = 25, NG = 25, ND = 75, NGRP = 7, TS = P1 * 3 * P1;
    const int tid = threadIdx.x;
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni64 = [&](long long v) { return (long long)(((unsigned long long)(unsigned)uni((int)((unsigned long long)v >> 32)) << 32) | (unsigned)uni((int)(unsigned long long)v)); };
    WalkItem it = items[blockIdx.x];
    it.patch = uni(it.patch); it.eu = uni(it.eu); it.ev0 = uni(it.ev0); it.nel = uni(it.nel); it.iu0 = uni(it.iu0);
    const PatchDev& Pt = M.patches[it.patch];
    const int p_nu = uni(Pt.nu), p_nelu = uni(Pt.nelu), p_tabu = uni(Pt.tabu), p_tabv = uni(Pt.tabv), p_wu = uni(Pt.wu), p_wv = uni(Pt.wv), p_spv = uni(Pt.spv);
    const long long p_cp_off = uni64(Pt.cp_off), p_elem_off = uni64(Pt.elem_off);
    // row advance from every element of the item to the next one (after the last element every row leaves: P1); one byte each -- with an int table the
    // kernel's LDS passes 40 KB and only three waves fit a CU
    __shared__ unsigned char s_dv[REC_MAX_NEL + 1];
    for (int k = threadIdx.x; k < it.nel && k <= REC_MAX_NEL; k += 64)
        s_dv[k] = (unsigned char)(k + 1 < it.nel ? M.ints[p_spv + it.ev0 + k + 1] - M.ints[p_spv + it.ev0 + k] : P1);
    const int iv_first = uni(M.ints[p_spv + it.ev0] - P);
    __shared__ double s_pc[8];
    if (threadIdx.x < 8) s_pc[threadIdx.x] = (&Pt.E)[threadIdx.x];
    const double* const pf = s_pc + 2; const double* const ppd = s_pc + 5;

    __shared__ __attribute__((aligned(16))) double s_g[8 * NB];      // control points of the element (phases 0-1), then the residual reduction
    double (*s_c)[3] = reinterpret_cast<double (*)[3]>(s_g);
    double (*s_d)[3] = reinterpret_cast<double (*)[3]>(s_g + 3 * NB);
    double* s_h = s_g + 6 * NB; double* s_w = s_g + 7 * NB;
    __shared__ double s_tu[TS], s_tv[TS], s_wg[2 * P1];
    __shared__ __attribute__((aligned(16))) double s_im[NG][IM_SIZE];
    wave_lds_sync();

    const bool doK = PASS == 0 && (flags & GF_ASM_K_BIT) != 0, doC = PASS != 0 && (flags & GF_ASM_C_BIT) != 0, doH = PASS != 0 && (flags & GF_ASM_H_BIT) != 0;
    const bool has_bf = (pf[0] != 0.0) || (pf[1] != 0.0) || (pf[2] != 0.0);
    // ---- inputs of one element: 64 bytes per control point (c_xy | c_zw | u_xy | u_z, h: lane task = 4 * local index + quarter, 100 tasks in two rounds),
    //      the v table and the v weights.  Requested in front of the flush stores (vmcnt is in order: a load behind the stores would wait for all of them),
    //      parked in LDS behind them.
    struct Fetch { double2 cp[2]; double tv[2], wv; };
    // (lane constants of fetch / park / flush are re-derived from an opaque copy of the lane id where they are used: kept alive across the element loop they
    //  are spilled to scratch, and every reload is an exposed memory round trip for a wave that runs alone on its SIMD)
    auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
    auto fetch = [&](int ev, int iv0f) {
        Fetch F;
        const int tid = opaque((int)threadIdx.x);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int task = tid + 64 * r, pa_cp = task >> 2, pa_q = task & 3;
            F.cp[r] = double2{0.0, 0.0};
            if (pa_cp < NB) {
                const long long g = p_cp_off + (it.iu0 + pa_cp % P1) + (long long)(iv0f + pa_cp / P1) * p_nu;
                if (pa_q < 2) F.cp[r] = reinterpret_cast<const double2*>(M.cp4 + 4 * g)[pa_q];
                else if (pa_q == 2) { F.cp[r].x = M.u[3 * g]; F.cp[r].y = M.u[3 * g + 1]; }
                else { F.cp[r].x = M.u[3 * g + 2]; F.cp[r].y = M.h[g]; }
            }
            const int k = tid + 64 * r;
            F.tv[r] = k < TS ? M.tab[p_tabv + ev * TS + k] : 0.0;
        }
        F.wv = tid < P1 ? M.tab[p_wv + ev * P1 + tid] : 0.0;
        return F;
    };
    auto park = [&](const Fetch& F) {
        const int tid = opaque((int)threadIdx.x);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int task = tid + 64 * r, a = task >> 2, pa_q = task & 3;
            if (a < NB) {
                if (pa_q == 0) { s_c[a][0] = F.cp[r].x; s_c[a][1] = F.cp[r].y; }
                else if (pa_q == 1) { s_c[a][2] = F.cp[r].x; s_w[a] = F.cp[r].y; }
                else if (pa_q == 2) { s_d[a][0] = F.cp[r].x; s_d[a][1] = F.cp[r].y; }     // displacement coefficients (the strains are evaluated from their derivatives: kl_strains)
                else { s_d[a][2] = F.cp[r].x; s_h[a] = F.cp[r].y; }
            }
            const int k = tid + 64 * r;
            if (k < TS) s_tv[k] = F.tv[r];
        }
        if (tid < P1) s_wg[P1 + tid] = F.wv;
        wave_lds_sync();
    };
    {   // prologue: u table and weights of the strip, inputs of the first element
        for (int k = tid; k < TS; k += 64) s_tu[k] = M.tab[p_tabu + it.eu * TS + k];
        if (tid < P1) s_wg[tid] = M.tab[p_wu + it.eu * P1 + tid];
        const Fetch F = fetch(it.ev0, iv_first);
        park(F);
    }
    const __amdgpu_buffer_rsrc_t rR = buf_rsrc(O.rec + (size_t)blockIdx.x * O.rec_rows * RC::SZ, (unsigned)(O.rec_rows * RC::SZ * 8));

    // accumulators of the pass (they live across the elements of the item)
    constexpr int NK00 = PASS == 0 ? 6 : 1, NK10 = PASS == 0 ? 9 : 1, NCC = PASS != 0 ? 9 : 1, NCH = PASS != 0 ? 3 : 1;
    gf_d4 accK00[NK00], accK11[NK00], accK10[NK10], accC[2][NCC], accH[2][NCH];
    for (int q = 0; q < NK00; ++q) { accK00[q] = gf_d4{0, 0, 0, 0}; accK11[q] = gf_d4{0, 0, 0, 0}; }
    for (int q = 0; q < NK10; ++q) accK10[q] = gf_d4{0, 0, 0, 0};
    for (int ta = 0; ta < 2; ++ta) {
        for (int q = 0; q < NCC; ++q) accC[ta][q] = gf_d4{0, 0, 0, 0};
        for (int q = 0; q < NCH; ++q) accH[ta][q] = gf_d4{0, 0, 0, 0};
    }
    constexpr int QI[6] = {0, 0, 0, 1, 1, 2}, QJ[6] = {0, 1, 2, 1, 2, 2};

    unsigned long long tstamp = 0; (void)tstamp;
#ifdef GF_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tstamp = clock64();
#endif
    int iv0 = iv_first;
    for (int t = 0; t < it.nel; ++t) {
        const int ev = it.ev0 + t;
        const long long e = p_elem_off + it.eu + (long long)ev * p_nelu;
        wave_lds_sync();
        const int iv0n = iv0 + uni((int)s_dv[t]);
        const bool more = t + 1 < it.nel;
        // ---- phase 1: one lane per Gauss point (sum-factorised control-point sums, quotient rule, pointwise record)
        if (tid < NG) {
            const int gu = tid % P1, gv = tid / P1;
            double Ac[3][6], Ad[3][6], W[6], th = 0.0;
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
                th += v0 * Sh;
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
            shell_point<PASS != 0>(z, Z, dz, th, s_pc[0], s_pc[1], im);      // the K walk needs no reference-configuration derivatives
            for (int k = 0; k < 6; ++k) im[IM_W + k] = W[k];
            im[IM_WQ] = s_wg[gu] * s_wg[P1 + gv];
        }
        wave_lds_sync();
        GF_STAMP(0, tstamp);

        // ---- roles of this lane, derived HERE from an opaque copy of the lane id: nothing lane-constant is alive during phase 1, whose working set fills
        //      the arch VGPRs while the accumulators of the walk sit in the AGPRs (kept across the element loop these ~50 registers went to scratch).
        //      B operand (column) of tile t: lane x; A operand rows held in register rr of a D tile: x_a = kk + 4 rr
        const int tidl = opaque((int)threadIdx.x), x = tidl & 15, kk = tidl >> 4;
        const int slot_b = x % 5, jq = x / 5;
        const bool bvalid[2] = {x < 15, x < 10};
        const int ub[2] = {jq < 3 ? jq : 0, jq < 2 ? 3 + jq : 3};
        const double bval[2] = {bvalid[0] ? 1.0 : 0.0, bvalid[1] ? 1.0 : 0.0};
        const RowLane RLg(x);
        // this lane's basis functions in the current element: tile t -> (u index ub[t], v index = (slot - first row) mod 5)
        const int jvb = mod5(slot_b - iv0);
        auto basis = [&](const double* im, int gu, int gv, double (&phi)[2][5], double (&R0)[2], double (&n0)[2]) {
            const double v0 = s_tv[(gv * 3 + 0) * P1 + jvb], v1 = s_tv[(gv * 3 + 1) * P1 + jvb], v2 = s_tv[(gv * 3 + 2) * P1 + jvb];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int ju = ub[tl];
                const double u0 = s_tu[(gu * 3 + 0) * P1 + ju], u1 = s_tu[(gu * 3 + 1) * P1 + ju], u2 = s_tu[(gu * 3 + 2) * P1 + ju];
                const double Nb[6] = {u0 * v0, u1 * v0, u0 * v1, u2 * v0, u0 * v2, u1 * v1};
                double R[6];
                rationalize6(Nb, im + IM_W, R);
                for (int k = 0; k < 5; ++k) phi[tl][k] = bval[tl] * R[k + 1];
                R0[tl] = bval[tl] * R[0]; n0[tl] = bval[tl] * Nb[0];
            }
        };

        double accR[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
        if constexpr (PASS == 0) {
            for (int grp = 0; grp < NGRP; ++grp) {
                const int gp = 4 * grp + kk, gpc = gp < NG ? gp : NG - 1;
                const int gu = gpc % P1, gv = gpc / P1;
                const double* im = s_im[gpc];
                const double wq = gp < NG ? im[IM_WQ] : 0.0;               // padded Gauss-point slots contribute nothing
                double phi[2][5], R0[2], n0[2];
                basis(im, gu, gv, phi, R0, n0);
                GF_STAMP(1, tstamp);
                double gR[15], hR[15];
                for (int s = 0; s < 15; ++s) { gR[s] = 0.0; hR[s] = 0.0; }
                if (doK) { RLg.template expand<false>(im, gR, hR); dpp_source_fence(gR); }
                GF_STAMP(2, tstamp);
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
                GF_STAMP(3, tstamp);
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
                GF_STAMP(4, tstamp);
            }
        } else {
            constexpr int tb = PASS - 1;
            // body force: sum_gp R_a (dJ/dZ . phi_b)_f per a tile, scaled by -f_i behind the group loop (3 + 3 MFMAs per group instead of 9 + 9 into the nine
            // (i, f) tiles: f_i is a constant of the patch) -- as the p <= 3 kernels do (gauss_group: accB)
            gf_d4 accB[2][3] = {{gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}}, {gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}}};
            for (int grp = 0; grp < NGRP; ++grp) {
                const int gp = 4 * grp + kk, gpc = gp < NG ? gp : NG - 1;
                const int gu = gpc % P1, gv = gpc / P1;
                const double* im = s_im[gpc];
                const double wq = gp < NG ? im[IM_WQ] : 0.0;
                double phi[2][5], R0[2], n0[2];
                basis(im, gu, gv, phi, R0, n0);
                GF_STAMP(1, tstamp);
                double gR[15], hR[15];
                for (int s = 0; s < 15; ++s) { gR[s] = 0.0; hR[s] = 0.0; }
                if (doC) { RLg.template expand<true>(im, gR, hR); dpp_source_fence(hR); }
                GF_STAMP(2, tstamp);
                double pb[5];
                for (int m = 0; m < 5; ++m) pb[m] = wq * phi[tb][m];
                const double n0b = n0[tb];
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
                GF_STAMP(3, tstamp);
                if (doC) {
                    static_for<5>([&](auto m_) {
                        constexpr int m = decltype(m_)::value;
                        double tq[9];
                        static_for<9>([&](auto q_) { constexpr int q = decltype(q_)::value; tq[q] = row_dot<3 * m + q / 3, q % 3>(hR, pb); });
                        mfma_hazard_gap(tq);
#pragma unroll
                        for (int q = 0; q < 9; ++q) {
                            accC[0][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[0][m], tq[q], accC[0][q], 0, 0, 0);
                            accC[1][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[1][m], tq[q], accC[1][q], 0, 0, 0);
                        }
                    });
                    if (has_bf) {                    // d(-f . u dA)/dc : -w f_i R_a (dJ/dZ . phi_b)
                        const LoadGeom lg = load_geom(im, ppd);
#pragma unroll
                        for (int f = 0; f < 3; ++f) {
                            const double jz = load_dz_dot(im, ppd, lg, f, pb[0], pb[1]);
                            accB[0][f] = __builtin_amdgcn_mfma_f64_16x16x4f64(R0[0], jz, accB[0][f], 0, 0, 0);
                            accB[1][f] = __builtin_amdgcn_mfma_f64_16x16x4f64(R0[1], jz, accB[1][f], 0, 0, 0);
                        }
                    }
                }
                GF_STAMP(4, tstamp);
            }
            if (doC && has_bf) {
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int f = 0; f < 3; ++f) accC[ta][3 * i + f] -= pf[i] * accB[ta][f];
            }
        }

        // ---- residual of this element (PASS 0): sum the four Gauss-point slots of a group through the staging area, entry (local a, i) of the element
        wave_lds_sync();
        if constexpr (PASS == 0) {
            if (kk >= 2) for (int ta = 0; ta < 2; ++ta) for (int i = 0; i < 3; ++i) s_g[(((kk - 2) * 2 + ta) * 16 + x) * 3 + i] = accR[ta][i];
            wave_lds_sync();
            if (kk < 2) for (int ta = 0; ta < 2; ++ta) for (int i = 0; i < 3; ++i) accR[ta][i] += s_g[((kk * 2 + ta) * 16 + x) * 3 + i];
            wave_lds_sync();
            if (kk == 1) for (int ta = 0; ta < 2; ++ta) for (int i = 0; i < 3; ++i) s_g[(ta * 16 + x) * 3 + i] = accR[ta][i];
            wave_lds_sync();
            if (kk == 0 && (flags & GF_ASM_R_BIT)) for (int ta = 0; ta < 2; ++ta) {
                if (bvalid[ta]) {
                    const int a = ub[ta] + P1 * jvb;
                    for (int i = 0; i < 3; ++i) O.rblk[(size_t)(e - O.e_first) * ND + 3 * a + i] = accR[ta][i] + s_g[(ta * 16 + x) * 3 + i];
                }
            }
            wave_lds_sync();
        }

        // ---- the next element's inputs are requested in front of the flush stores
        Fetch Fn;
        if (more) Fn = fetch(ev + 1, iv0n);
        GF_STAMP(5, tstamp);

        // ---- flush: the pairs whose lower row leaves the window (rows < iv0n) are complete for this item
        // (the A-side roles of the four registers of a D tile are re-derived here from an opaque copy of the lane id: kept in registers across the Gauss-point
        //  loop they are two dozen values the register allocator has to spill)
        const int tid2 = opaque((int)threadIdx.x), kk2 = tid2 >> 4, x2 = tid2 & 15;
        const int rowb = iv0 + mod5(x2 % 5 - iv0);
        const bool bvalid2[2] = {x2 < 15, x2 < 10};
        const int ub2[2] = {x2 / 5 < 3 ? x2 / 5 : 0, x2 / 5 < 2 ? 3 + x2 / 5 : 3};
        int rowa[4], ua[2][4]; bool avalid[2][4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int xa = kk2 + 4 * rr, q = xa / 5;
            rowa[rr] = iv0 + mod5(xa % 5 - iv0);
            avalid[0][rr] = xa < 15; avalid[1][rr] = xa < 10;
            ua[0][rr] = q < 3 ? q : 0; ua[1][rr] = q < 2 ? 3 + q : 3;
        }
        // byte offsets of component 0 of the ordered pair (A = (uA, rA), B = (uB, rB)) and its component stride
        constexpr Rec4Part PT = rec4_part(PASS);           // the part of the row records this walk writes
        auto pair_off = [&](int uA, int rA, int uB, int rB, unsigned& cs) {        // uB: column inside the part (ub - PT.ub0)
            const int rho = rA < rB ? rA : rB, d = rA < rB ? rB - rA : rA - rB;
            const bool a1 = rA <= rB;
            cs = 8u * (unsigned)(a1 ? 5 * PT.w : PT.w);
            return 8u * (unsigned)((rho - iv_first) * RC::SZ + PT.off + (a1 ? (uA * PT.nc * 5 + d) * PT.w + uB : PT.a1 + (uA * 4 + d - 1) * PT.nc * PT.w + uB));
        };
        auto flush_tile = [&](auto tag, int ta, int tb2, gf_d4* acc, int ncomp) {
            constexpr int KIND = decltype(tag)::value;       // 0: K diagonal quadrant (6 comps i <= j, mirrored); 1: K (1,0) quadrant (9 comps, mirrored); 2: dR/dCP (9); 3: dR/dh (3)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int rA = rowa[rr], lo = rA < rowb ? rA : rowb;
                const bool ok = avalid[ta][rr] && bvalid2[tb2] && lo < iv0n;
                // no branch around the stores: a pair that is not flushed gets an offset beyond the buffer's num_records and the hardware's range check drops
                // the store.  Straight-line code lets the compiler COUNT the stores behind the fetch loads (s_waitcnt vmcnt(n) instead of vmcnt(0): the park
                // below then does not wait for the stores to drain), and the flush issues without exec-mask bookkeeping.
                constexpr unsigned OOB = 0xF0000000u;
                unsigned cs, cst;
                unsigned o = pair_off(ua[ta][rr], rA, ub2[tb2] - PT.ub0, rowb, cs);
                o = ok ? o : OOB;
                if constexpr (KIND <= 1) {
                    unsigned ot = pair_off(ub2[tb2], rowb, ua[ta][rr], rA, cst);
                    ot = ok ? ot : OOB;
                    if constexpr (KIND == 0) {
#pragma unroll
                        for (int q = 0; q < 6; ++q) {
                            buf_st(rR, o + (3 * QI[q] + QJ[q]) * cs, acc[q][rr]);
                            if (QI[q] != QJ[q]) buf_st(rR, ot + (3 * QJ[q] + QI[q]) * cst, acc[q][rr]);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 9; ++q) {
                            buf_st(rR, o + q * cs, acc[q][rr]);
                            buf_st(rR, ot + (3 * (q % 3) + q / 3) * cst, acc[q][rr]);
                        }
                    }
                } else if constexpr (KIND == 2) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) buf_st(rR, o + q * cs, acc[q][rr]);
                } else {
#pragma unroll
                    for (int q = 0; q < 3; ++q) buf_st(rR, o + (9 + q) * cs, acc[q][rr]);
                }
#pragma unroll
                for (int q = 0; q < (KIND == 0 ? 6 : (KIND == 3 ? 3 : 9)); ++q) acc[q][rr] = ok ? 0.0 : acc[q][rr];
                (void)ncomp;
            }
        };
        if constexpr (PASS == 0) {
            if (doK) {
                flush_tile(std::integral_constant<int, 0>{}, 0, 0, accK00, 6);
                flush_tile(std::integral_constant<int, 0>{}, 1, 1, accK11, 6);
                flush_tile(std::integral_constant<int, 1>{}, 1, 0, accK10, 9);
            }
        } else {
            constexpr int tb = PASS - 1;
            if (doC) { flush_tile(std::integral_constant<int, 2>{}, 0, tb, accC[0], 9); flush_tile(std::integral_constant<int, 2>{}, 1, tb, accC[1], 9); }
            if (doH) { flush_tile(std::integral_constant<int, 3>{}, 0, tb, accH[0], 3); flush_tile(std::integral_constant<int, 3>{}, 1, tb, accH[1], 3); }
        }
        GF_STAMP(6, tstamp);
        // ---- park the next element's inputs
        wave_lds_sync();
        if (more) park(Fn);
        iv0 = iv0n;
        GF_STAMP(7, tstamp);
    }
#ifdef GF_STAMPS
    if ((blockIdx.x & 7) == 0 && tid == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_stamps[k], stamp_acc[k]);
#endif
}

// Record gather, p = 4: ONE wave per control point a = (ia, ja) sums, item by item (strips ascending, segments ascending: a fixed order), what the row
// records hold for its three dof rows, then writes the rows (gather_write_rows: Dirichlet entries, coupling-only columns, penalty rows).  From a work item
// (strip iu0 .. iu0 + 4) with ua = ia - iu0:
//   G1  record ja,     area 1 [ua][c][d][ub'] of every part:      the pairs (a, b = (iu0 + ub, ja + d)), d = 0..4: 225 + 180 + 120 doubles in three runs
//   G3  record ja - d, area 2 [ua][d - 1][c][ub'] of every part:  the pairs (a, b = (iu0 + ub, ja - d)), d = 1..4: 45 + 36 + 24 doubles per d
// Every ordered pair is stored with all its components (the element kernel writes the transposed K entries), so there are no mirrored reads.  A pair is
// present in an item only if one of the item's elements holds both rows (RecCp4::info, bit 16 + (jb - ja + 4)); everything else in a record row is never
// written and never read.
template <int NC>
__global__ __launch_bounds__(64) void kl_gather_rec4_kernel(DevModel M, long long a_first, long long a_end, int flags, const double* __restrict__ rec, long long row_base,
                                                            const RecCp4* __restrict__ reccp, double* __restrict__ valK, double* __restrict__ valC0,
                                                            double* __restrict__ valC1, double* __restrict__ valC2, double* __restrict__ valH, int pen_add) {
    using RC = Rec4Cfg<NC>;
    constexpr int WB = 9, NBOX = WB * WB, SZ = RC::SZ;
    constexpr bool WITHC = NC == 21;
    // workgroup w runs on XCD w % 8: every XCD takes a contiguous range of control points (the record lines shared by neighbours meet in one L2)
    const long long chunk = (a_end - a_first + 7) / 8;
    const long long a = a_first + (long long)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if ((long long)(blockIdx.x >> 3) >= chunk || a >= a_end) return;
    const CpDesc& cd = M.cpdesc[a];
    const int ja = cd.ja, j0 = cd.j0, wbox = cd.i1 - cd.i0 + 1;
    const long long ptr_c = M.nb_ptr_c[a], deg_c = M.nb_ptr_c[a + 1] - ptr_c, ptr_s = M.nb_ptr_s[a], deg_s = M.nb_ptr_s[a + 1] - ptr_s;
    const int lane = threadIdx.x;
    __shared__ double acc[21 * NBOX];                   // aK [3][NBOX][3] | aH [3][NBOX] | aC [3 f][3 i][NBOX]
    __shared__ unsigned short s_meta[GATHER_MAXMETA];
    // the work items of a: (row, info) per item, parked in LDS by the first lanes and read back uniformly per item (held in registers the 22-dword
    // descriptor went to VGPRs + scratch with dynamic indexing: the kernel is short of SGPRs)
    __shared__ int s_it[2 * 10 + 2];
    if (lane < 22) s_it[lane] = reinterpret_cast<const int*>(reccp + a)[lane < 20 ? 2 + lane : lane - 20];      // [0..19]: it[k].row, it[k].info; [20]: nit, [21]: flags
    const bool pen_row_a = M.pen_row[a] != 0;
    for (int k = lane; k < 21 * NBOX; k += 64) acc[k] = 0.0;
    const bool doC = WITHC && (flags & GF_ASM_C_BIT) != 0, doK = (flags & GF_ASM_K_BIT) != 0, doH = WITHC && (flags & GF_ASM_H_BIT) != 0;
    // accumulator address of component c at box slot 0 and its stride in the slot (3 for K, 1 otherwise)
    auto comp_base = [&](int c) { return c < 9 ? (c / 3) * NBOX * 3 + c % 3 : (c < 18 ? 12 * NBOX + (((c - 9) % 3) * 3 + (c - 9) / 3) * NBOX : 9 * NBOX + (c - 18) * NBOX); };
    // ---- per-lane task table (independent of the item).  Task t = lane + 64 ps runs through G1 (area 1) of the parts, then G3 (area 2) of the parts:
    //      tk = accumulator address for iu0 = i0 | stride3 << 15 | presence bit << 16 | valid << 20 | (stride of the value in ua) << 21; to = offset of the value
    //      relative to the row record of ja (+ ua * stride)
    constexpr int NPART = RC::NPART;
    constexpr int NTASK = NC == 21 ? 525 + 420 : 225 + 180, NV = (NTASK + 63) / 64;
    unsigned tk[NV]; int to[NV];
#pragma unroll
    for (int ps = 0; ps < NV; ++ps) {
        int e = lane + 64 * ps, c = 0, drow = 0, ubx = 0, off = 0, mul = 0;
        bool found = false;
#pragma unroll
        for (int pp = 0; pp < NPART; ++pp) {               // G1: area 1 of part pp, [c][d][ub']
            constexpr Rec4Part Q0 = rec4_part(0), Q1 = rec4_part(1), Q2 = rec4_part(2);
            const Rec4Part Q = pp == 0 ? Q0 : (pp == 1 ? Q1 : Q2);
            const int n = Q.nc * 5 * Q.w;
            if (!found && e < n) { found = true; c = (pp == 0 ? 0 : 9) + e / (5 * Q.w); drow = (e % (5 * Q.w)) / Q.w; ubx = Q.ub0 + e % Q.w; off = Q.off + e; mul = n; }
            if (!found) e -= n;
        }
#pragma unroll
        for (int pp = 0; pp < NPART; ++pp) {               // G3: area 2 of part pp, [d - 1][c][ub']
            constexpr Rec4Part Q0 = rec4_part(0), Q1 = rec4_part(1), Q2 = rec4_part(2);
            const Rec4Part Q = pp == 0 ? Q0 : (pp == 1 ? Q1 : Q2);
            const int n = 4 * Q.nc * Q.w;
            if (!found && e < n) {
                found = true;
                const int d = 1 + e / (Q.nc * Q.w), r = e % (Q.nc * Q.w);
                c = (pp == 0 ? 0 : 9) + r / Q.w; drow = -d; ubx = Q.ub0 + r % Q.w; off = -d * SZ + Q.off + Q.a1 + (d - 1) * Q.nc * Q.w + r; mul = n;
            }
            if (!found) e -= n;
        }
        const bool en = c < 9 ? doK : (c < 18 ? doC : doH);
        const bool ok = found && en;
        tk[ps] = (unsigned)((comp_base(c) + (ubx + (ja + drow - j0) * wbox) * (c < 9 ? 3 : 1)) & 0x7fff) | (c < 9 ? 1u << 15 : 0u) | ((unsigned)(4 + drow) << 16) | (ok ? 1u << 20 : 0u) | ((unsigned)mul << 21);
        to[ps] = off;
    }
    struct Item { double v[NV]; int du; unsigned pm; };
    auto load_item = [&](int n) {
        Item I;
        const int row = __builtin_amdgcn_readfirstlane(s_it[2 * n]); const unsigned info = (unsigned)__builtin_amdgcn_readfirstlane(s_it[2 * n + 1]);
        I.du = int(info & 255u); I.pm = info >> 16;
        const int uaa = int((info >> 8) & 255u);
        const double* Rja = rec + (size_t)((long long)row - row_base) * SZ;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int mul = int(tk[v] >> 21);
            const bool ok = ((tk[v] >> 20) & 1) && ((I.pm >> ((tk[v] >> 16) & 15)) & 1);
            // unconditional load (an absent pair reads the head of the row record; its value is never added): see kl_gather_rec_kernel
            I.v[v] = Rja[ok ? to[v] + uaa * mul : 0];
        }
        return I;
    };
    auto add_item = [&](const Item& I) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bool ok = ((tk[v] >> 20) & 1) && ((I.pm >> ((tk[v] >> 16) & 15)) & 1);
            if (ok) (void)__hip_atomic_fetch_add(&acc[(tk[v] & 0x7fff) + I.du * ((tk[v] >> 15) & 1 ? 3 : 1)], I.v[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    wave_lds_sync();
    const int nit = __builtin_amdgcn_readfirstlane(s_it[20]), zflags = __builtin_amdgcn_readfirstlane(s_it[21]);
    constexpr int NBT = 2;                              // work items whose loads are in flight together
    Item B[NBT];
#pragma unroll
    for (int q = 0; q < NBT; ++q) if (q < nit) B[q] = load_item(q);
    constexpr int NM = (GATHER_MAXMETA + 63) / 64;
    unsigned short mt[NM];
#pragma unroll
    for (int q = 0; q < NM; ++q) { const int k = lane + 64 * q; mt[q] = k < (int)deg_c ? M.nb_meta[ptr_c + k] : (unsigned short)0; }
    for (int n0 = 0; n0 < nit; n0 += NBT) {
#pragma unroll
        for (int q = 0; q < NBT; ++q) {
            if (n0 + q < nit) add_item(B[q]);
            if (n0 + NBT + q < nit) B[q] = load_item(n0 + NBT + q);
        }
    }
#pragma unroll
    for (int q = 0; q < NM; ++q) { const int k = lane + 64 * q; if (k < GATHER_MAXMETA) s_meta[k] = mt[q]; }
    wave_lds_sync();
    const bool padd = pen_add && pen_row_a;
    gather_write_rows<NBOX, WITHC>(M, a, lane, doK, doC, doH, padd, (unsigned)zflags, ptr_c, deg_c, ptr_s, deg_s, s_meta, acc, acc + 12 * NBOX, acc + 9 * NBOX,
                                   valK, valC0, valC1, valC2, valH);
}

}  // namespace gf
