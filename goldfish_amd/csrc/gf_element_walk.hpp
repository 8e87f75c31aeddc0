// gf_element_walk.hpp -- MFMA element kernel (p = 2, 3) that walks a strip of elements and accumulates straight into the CSR
// value arrays: no element blocks, no gather.
//
// kl_element_mfma_kernel writes one 43 KB block per element and the gather reads every block back (26 GB each way per C4
// step, 5.3x the algorithmic bytes; the gather is a quarter of the step).  Here one wave walks the elements ev = ev0 .. of a
// strip (fixed u-span eu) and KEEPS the MFMA accumulators across elements.  A control-point pair (A, B) lives in the
// accumulator slot addressed by the rows' indices modulo 4:
//     operand lane x  <->  basis function (u index x / 4, v slot x % 4),  v slot s holds the control-point row with row % 4 = s,
//     D[a][b]: lane (x, kk), register rr  <->  A = (iu0 + rr, row of slot kk),  B = (iu0 + x / 4, row of slot x % 4),
// so moving to the next element changes WHICH basis function a lane evaluates (a different entry of the 1-D v table), never
// where a pair's partial sum sits: no shifts, no moves.  When the window leaves a row, the pairs that have it as their lower
// row (28 of 64 lanes, all registers) are complete for this strip: their sums are added to the CSR entries and the slots are
// zeroed for the row that enters.  A pair receives contributions from up to p + 1 strips and (when a strip is cut into
// segments) two segments: the work items are launched in classes (eu mod (p + 1), segment parity); items of one class share
// no pair, classes run in ascending order, the first class touching an entry stores, later ones add to what is there --
// a fixed summation order, no atomics, bitwise reproducible.  vmcnt counts loads and stores together, in order: a load issued
// behind the flush stores would wait for every one of them, so the kernel never issues one there -- the next element's inputs
// are fetched before the group loop into a few registers and parked in LDS (rows in a ring of 8: the window of the next element
// never collides with the rows the flush still needs), and all reads of a flush precede its first store.  Dirichlet rows / columns are overwritten at every flush; the
// penalty rows are added afterwards (pen_owner_kernel<.., ADD = true>).  The residual still goes through a 3 (p+1)^2-double
// block per element and kl_rgather_kernel.
// Traffic per C4 step: ~11 GB of partial sums written + ~6 GB read back, instead of 26 GB written + 29 GB read + 6 GB written.
// Reference path: the same integrals as kl_element_mfma_kernel (GOLDFISH/nonmatching_opt.py:941-1015 via PENGoLINS assembly).
#pragma once
#include "gf_element_mfma.hpp"

namespace gf {

struct WalkOut { double* valK; double* valC0; double* valC1; double* valC2; double* valH; double* rblk; const WalkPatch* wpatch; };

// The value arrays are addressed through buffer resources that span ONE patch (base in SGPRs, 32-bit byte offset per lane):
// half the address registers and arithmetic of flat 64-bit pointers, and an offset beyond the range reads as zero -- a pair
// that starts in this class, or a Dirichlet entry, "reads" its old value from there: no branch, no zero page.
typedef unsigned gf_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t walk_rsrc(double* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); }
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, unsigned off, double v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(gf_u2, v), r, (int)off, 0, 0); }
constexpr unsigned WALK_OOR = 0xFFFFE000u;     // byte offset beyond every patch range (HostModel::build_walk keeps the ranges below 2^32 - 4096)

template <int P, bool WITHC = true>
__global__ __launch_bounds__(64) void kl_element_walk_kernel(DevModel M, const WalkItem* __restrict__ items, int item_first, int flags,
                                                              const RowDesc* __restrict__ rowdesc, WalkOut O) {
    static_assert(P == 2 || P == 3, "one 16 x 16 tile: p <= 3");
    constexpr int P1 = P + 1, NB = P1 * P1, NG = NB, ND = 3 * NB, NGRP = (NG + 3) / 4, TS = P1 * 3 * P1;
    const int tid = threadIdx.x, x = tid & 15, kk = tid >> 4;
    const WalkItem it = items[item_first + blockIdx.x];
    const PatchDev& Pt = M.patches[it.patch];

    __shared__ __attribute__((aligned(16))) double s_g[4 * 3 * 16];  // control-point staging (phases 0-1), residual reduction at the end
    double (*s_c)[3] = reinterpret_cast<double (*)[3]>(s_g);
    double (*s_d)[3] = reinterpret_cast<double (*)[3]>(s_g + 3 * NB);
    double* s_h = s_g + 6 * NB; double* s_w = s_g + 7 * NB;
    __shared__ double s_tu[TS], s_tv[2][TS], s_wgu[P1], s_wgv[2][P1];            // v tables double buffered (the next element's are parked while this one's are in use)
    __shared__ __attribute__((aligned(16))) double s_im[NG][IM_SIZE];
    __shared__ __attribute__((aligned(16))) double s_raw[8][4][8];               // control points of the windows, ring over the row index: c_x, c_y, c_z, w, u_x, u_y, u_z, h
    __shared__ __attribute__((aligned(16))) int s_rd[8][4][16];                  // their row descriptors

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

    const bool doK = (flags & GF_ASM_K_BIT) != 0, doC = WITHC && (flags & GF_ASM_C_BIT) != 0, doH = (flags & GF_ASM_H_BIT) != 0;
    const bool has_bf = (Pt.f[0] != 0.0) || (Pt.f[1] != 0.0) || (Pt.f[2] != 0.0);
    const int jub = x >> 2, sb = x & 3;                   // this lane's basis function: u index, v slot
    const int jubc = jub < P1 ? jub : 0;

    gf_d4 accK[6], accC[9], accH[3];
    for (int q = 0; q < 6; ++q) accK[q] = gf_d4{0, 0, 0, 0};
    for (int q = 0; q < 9; ++q) accC[q] = gf_d4{0, 0, 0, 0};
    for (int q = 0; q < 3; ++q) accH[q] = gf_d4{0, 0, 0, 0};

    // ---- input fetch of one element: 16 bytes per lane of the window's control points (lane = 4 * local index + quarter:
    //      c_xy | c_zw | u_xy | u_z, h) and of their row descriptors, the v table and v weights; parked in LDS by park()
    const int pa_cp = tid >> 2, pa_q = tid & 3, pa_ju = pa_cp % P1, pa_jv = pa_cp / P1;
    struct Fetch { double2 cp; int4 rd; double tv, wv; };
    auto fetch = [&](const ElemDesc& ed) {
        Fetch F; F.cp = double2{0.0, 0.0}; F.rd = int4{0, 0, 0, 0}; F.tv = 0.0; F.wv = 0.0;
        if (pa_cp < NB) {
            const long long g = ed.g0 + pa_ju + (long long)pa_jv * ed.nu;
            if (pa_q < 2) F.cp = reinterpret_cast<const double2*>(M.cp4 + 4 * g)[pa_q];
            else if (pa_q == 2) { F.cp.x = M.u[3 * g]; F.cp.y = M.u[3 * g + 1]; }
            else { F.cp.x = M.u[3 * g + 2]; F.cp.y = M.h[g]; }
            F.rd = reinterpret_cast<const int4*>(rowdesc + g)[pa_q];
        }
        if (tid < TS) F.tv = M.tab[ed.tabv + tid];
        if (tid < P1) F.wv = M.tab[ed.wv + tid];
        return F;
    };
    auto park = [&](const Fetch& F, int iv0f, int buf) {
        if (pa_cp < NB) {
            const int rs = (iv0f + pa_jv) & 7;
            *reinterpret_cast<double2*>(&s_raw[rs][pa_ju][2 * pa_q]) = F.cp;
            *reinterpret_cast<int4*>(&s_rd[rs][pa_ju][4 * pa_q]) = F.rd;
        }
        if (tid < TS) s_tv[buf][tid] = F.tv;
        if (tid < P1) s_wgv[buf][tid] = F.wv;
    };
    {   // prologue: u table and weights of the strip, inputs of the first element
        const long long e0 = Pt.elem_off + it.eu + (long long)it.ev0 * Pt.nelu;
        const ElemDesc ed = M.edesc[e0];
        if (tid < TS) s_tu[tid] = M.tab[ed.tabu + tid];
        if (tid < P1) s_wgu[tid] = M.tab[ed.wu + tid];
        const Fetch F = fetch(ed);
        park(F, M.ints[Pt.spv + it.ev0] - P, 0);
    }

    const int ev_end = it.ev0 + it.nel;
    const bool seg_even = (it.seg & 1) == 0;
    const WalkPatch wp = O.wpatch[it.patch];
    const __amdgpu_buffer_rsrc_t rK = walk_rsrc(O.valK + wp.kbase, wp.kbytes), rC0 = walk_rsrc(O.valC0 + wp.cbase, wp.cbytes),
                                 rC1 = walk_rsrc(O.valC1 + wp.cbase, wp.cbytes), rC2 = walk_rsrc(O.valC2 + wp.cbase, wp.cbytes), rH = walk_rsrc(O.valH + wp.hbase, wp.hbytes);
    // ---- flush of the pairs whose lower row leaves the window (first row iv0f, next element's first row iv0nf): they are
    //      complete for this item.  PART 0: K and dR/dh, at the end of the element; PART 1: dR/dCP, behind phase 1 of the NEXT
    //      element (its inputs are in LDS already, so no load is issued in between and the stores of part 0 have drained) --
    //      two halves so that all reads of a half (every one precedes its first store) fit the register file.
    auto flush = [&](auto PART_, int iv0f, int iv0nf) {
        constexpr int PART = decltype(PART_)::value;
#ifndef GF_WALK_HPART
#define GF_WALK_HPART 0                    // dR/dh rides with K (measured: fewer spills than with the dR/dCP half)
#endif
        const bool fK = PART == 0 && doK, fH = PART == GF_WALK_HPART && doH, fC = PART == 1 && doC;
        const int rowa = iv0f + ((kk - iv0f) & 3), rowb = iv0f + ((sb - iv0f) & 3);       // control-point rows of this lane's slots
        const bool live = (rowa - iv0f) < P1 && (rowb - iv0f) < P1 && jub < P1;
        // any Dirichlet dof among the window's control points (wave-uniform): only then the flush carries the constraint logic
        const bool anybc = __builtin_amdgcn_ballot_w64(((rowb - iv0f) < P1 && jub < P1) ? s_rd[rowb & 7][jubc][8] != 0 : false) != 0;
        if (live && (rowa < iv0nf || rowb < iv0nf) && (fK || fC || fH)) {
            const int* dB = s_rd[rowb & 7][jub];
            const int Bu = it.iu0 + jub;
            const int offKB = dB[0], degB = dB[1], i0B = dB[5], j0B = dB[6], wbB = dB[7], zB = dB[8], louB = dB[9], hiuB = dB[10], lovB = dB[11], hivB = dB[12];
            // first touch along the walk direction: a pair seen by two segments is started by the even one
            const int lovA = s_rd[rowa & 7][0][11], hivA = s_rd[rowa & 7][0][12];
            const int t0 = lovA > lovB ? lovA : lovB, t1 = hivA < hivB ? hivA : hivB;
            const bool vfirst = (t0 >= it.ev0 && t1 < ev_end) || seg_even;
            constexpr int IJ_I[6] = {0, 0, 0, 1, 1, 2}, IJ_J[6] = {0, 1, 2, 1, 2, 2};
            unsigned kA[P1], kB[P1], cA[P1], hA[P1], sK[P1], sC[P1], sH[P1], zA[P1], gate[P1];
#pragma unroll
            for (int rr = 0; rr < P1; ++rr) {
                const int* dA = s_rd[rowa & 7][rr];
                const int Au = it.iu0 + rr;
                const int offKA = dA[0], dgA = dA[1], offCA = dA[2], offHA = dA[3], dgsA = dA[4], i0A = dA[5], j0A = dA[6], wbA = dA[7], louA = dA[9], hiuA = dA[10];
                zA[rr] = dA[8];
                const int s0 = louA > louB ? louA : louB, s1 = hiuA < hiuB ? hiuA : hiuB;
                const bool first = vfirst && ((it.eu % P1 == 0) || (it.eu == s0 && s0 / P1 == s1 / P1));
                gate[rr] = first ? WALK_OOR : 0u;                              // offset | gate: out of range, reads 0
                const int slotAB = (Bu - i0A) + (rowb - j0A) * wbA, slotBA = (Au - i0B) + (rowa - j0B) * wbB;
                kA[rr] = 8u * (offKA + 3 * slotAB); kB[rr] = 8u * (offKB + 3 * slotBA); cA[rr] = 8u * (offCA + slotAB); hA[rr] = 8u * (offHA + slotAB);
                sK[rr] = 24u * dgA; sC[rr] = 8u * dgA; sH[rr] = 8u * dgsA;
            }
            const unsigned sKB = 24u * degB;
            // -- pass 1: every read of this half (what the earlier classes left) before its first store -- vmcnt is in order, a
            //    load behind a store would wait for it.  Unconditional loads: nothing for the compiler to predicate.
            double tK[P1][6], tC[P1][9], tH[P1][3];
#pragma unroll
            for (int rr = 0; rr < P1; ++rr) {
                if (fK) {
#pragma unroll
                    for (int ij = 0; ij < 6; ++ij) {
                        const int i = IJ_I[ij], j = IJ_J[ij];
                        unsigned off = (kA[rr] + i * sK[rr] + 8 * j) | gate[rr];
                        if (anybc) off = (((zA[rr] >> i) & 1) || ((zB >> j) & 1)) ? WALK_OOR : off;
                        tK[rr][ij] = buf_ld(rK, off);
                    }
                }
                if constexpr (WITHC) if (fC) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        const int i = q / 3, f = q % 3;
                        unsigned off = (cA[rr] + i * sC[rr]) | gate[rr];
                        if (anybc) off = ((zA[rr] >> i) & 1) ? WALK_OOR : off;
                        tC[rr][q] = buf_ld(f == 0 ? rC0 : (f == 1 ? rC1 : rC2), off);
                    }
                }
                if (fH) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) tH[rr][i] = buf_ld(rH, (hA[rr] + i * sH[rr]) | gate[rr]);
                }
            }
            // -- pass 2: sums and stores, then the slots start from zero for the rows that enter
#pragma unroll
            for (int rr = 0; rr < P1; ++rr) {
                const bool diag = (rr == jub) && (kk == sb);
                if (fK) {
#pragma unroll
                    for (int ij = 0; ij < 6; ++ij) {
                        const int i = IJ_I[ij], j = IJ_J[ij];
                        double v = tK[rr][ij] + accK[ij][rr];
                        if (anybc) { if (((zA[rr] >> i) & 1) || ((zB >> j) & 1)) v = (diag && i == j) ? 1.0 : 0.0; }
                        buf_st(rK, kA[rr] + i * sK[rr] + 8 * j, v);
                        if (i < j) buf_st(rK, kB[rr] + j * sKB + 8 * i, v);          // K is symmetric: entry (B, j), (A, i)
                    }
                }
                if constexpr (WITHC) if (fC) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        double v = tC[rr][q] + accC[q][rr];
                        if (anybc) { if ((zA[rr] >> (q / 3)) & 1) v = 0.0; }
                        buf_st(q % 3 == 0 ? rC0 : (q % 3 == 1 ? rC1 : rC2), cA[rr] + (q / 3) * sC[rr], v);
                    }
                }
                if (fH) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) buf_st(rH, hA[rr] + i * sH[rr], tH[rr][i] + accH[i][rr]);
                }
            }
            if constexpr (PART == 0) for (int q = 0; q < 6; ++q) accK[q] = gf_d4{0, 0, 0, 0};
            if constexpr (PART == 1) for (int q = 0; q < 9; ++q) accC[q] = gf_d4{0, 0, 0, 0};
            if constexpr (PART == GF_WALK_HPART) for (int q = 0; q < 3; ++q) accH[q] = gf_d4{0, 0, 0, 0};
        }
    };

    int prev_iv0 = 0, prev_iv0n = 0;
    unsigned long long tstamp = 0; (void)tstamp;
#ifdef GF_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tstamp = clock64();
#endif
    for (int t = 0; t < it.nel; ++t) {
        const int ev = it.ev0 + t, buf = t & 1;
        const long long e = Pt.elem_off + it.eu + (long long)ev * Pt.nelu;
        const int iv0 = M.ints[Pt.spv + ev] - P;
        const bool more = t + 1 < it.nel;
        const int iv0n = more ? M.ints[Pt.spv + ev + 1] - P : iv0 + 4;      // after the last element of the item every row leaves
        // ---- phase 0: this element's control points from the ring
        wave_lds_sync();
        if (tid < NB) {
            const double* rw = s_raw[(iv0 + tid / P1) & 7][tid % P1];
            s_c[tid][0] = rw[0]; s_c[tid][1] = rw[1]; s_c[tid][2] = rw[2]; s_w[tid] = rw[3];
            s_d[tid][0] = rw[0] + rw[4]; s_d[tid][1] = rw[1] + rw[5]; s_d[tid][2] = rw[2] + rw[6];
            s_h[tid] = rw[7];
        }
        wave_lds_sync();
        const double* const tv = s_tv[buf];
        GF_STAMP(0, tstamp);

        // ---- phase 1: three lanes per Gauss point (gp = x, part ic = kk < 3): kinematics + pointwise closed forms
        {
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
                    const double v0 = tv[(gv * 3 + 0) * P1 + jv], v1 = tv[(gv * 3 + 1) * P1 + jv], v2 = tv[(gv * 3 + 2) * P1 + jv];
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
            wave_lds_sync();                                   // all three lanes hold z, Z before the record overwrites the exchange slots
            if (act) {
                const double dsel[3] = {ic == 0 ? 1.0 : 0.0, ic == 1 ? 1.0 : 0.0, ic == 2 ? 1.0 : 0.0};
                shell_point_cols<WITHC>(z, Z, th, Pt.E, Pt.nu_, ic, dsel, kk == 0, im);
                if (kk == 0) {
                    for (int k = 0; k < 6; ++k) im[IM_W + k] = W[k];
                    im[IM_WQ] = s_wgu[gu] * s_wgv[buf][gv];
                }
            }
        }
        wave_lds_sync();

        GF_STAMP(1, tstamp);
        // ---- second half of the previous element's flush (dR/dCP): the stores of its first half have drained meanwhile
        if constexpr (WITHC) { if (t > 0) flush(std::integral_constant<int, 1>{}, prev_iv0, prev_iv0n); }

        // ---- the next element's inputs are requested now (they land during the group loop) and parked in LDS behind it:
        //      no load is ever issued behind the flush stores
        GF_STAMP(2, tstamp);
        Fetch Fn; Fn.cp = double2{0.0, 0.0}; Fn.rd = int4{0, 0, 0, 0}; Fn.tv = 0.0; Fn.wv = 0.0;
        if (more) Fn = fetch(M.edesc[e + Pt.nelu]);
        GF_STAMP(3, tstamp);

        // this lane's basis function in the current element: u index jub, v index = (slot - first row) mod 4
        const int jvb = (sb - iv0) & 3;
        const bool bok = jub < P1 && jvb < P1;
        const int jvc = jvb < P1 ? jvb : 0;
        const double bval = bok ? 1.0 : 0.0;                 // lanes beyond the basis functions contribute zero rows / columns
        double accR[3] = {0.0, 0.0, 0.0};
        gf_d4 accB[3] = {gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}};   // body force: sum_gp R_a (dJ/dZ . phi_b)_f, scaled by -f_i behind the loop

        for (int grp = 0; grp < NGRP; ++grp) {
            const int gp = 4 * grp + kk, gpc = gp < NG ? gp : NG - 1, gu = gpc % P1, gv = gpc / P1;   // Gauss point of this lane's group
            const double* im = s_im[gpc];
            const double wq = gp < NG ? im[IM_WQ] : 0.0;        // padded Gauss-point slots contribute nothing
            double phi[5], R0, n0;
            {
                const double u0 = s_tu[(gu * 3 + 0) * P1 + jubc], u1 = s_tu[(gu * 3 + 1) * P1 + jubc], u2 = s_tu[(gu * 3 + 2) * P1 + jubc];
                const double v0 = tv[(gv * 3 + 0) * P1 + jvc], v1 = tv[(gv * 3 + 1) * P1 + jvc], v2 = tv[(gv * 3 + 2) * P1 + jvc];
                const double Nb[6] = {u0 * v0, u1 * v0, u0 * v1, u2 * v0, u0 * v2, u1 * v1};
                double R[6];
                rationalize6(Nb, im + IM_W, R);
                for (int k = 0; k < 5; ++k) phi[k] = bval * R[k + 1];
                R0 = bval * R[0]; n0 = bval * Nb[0];
            }
            // -- row r of G and Hc at this Gauss point
            double gR[15], hR[15];
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
                dpp_source_fence(gR);
                if constexpr (WITHC) dpp_source_fence(hR);
            }
            // -- residual and dR/dh prefactors of this lane's basis function at this Gauss point
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
            constexpr int QI[6] = {0, 0, 0, 1, 1, 2}, QJ[6] = {0, 1, 2, 1, 2, 2};
            if (doK) {
                static_for<5>([&](auto m_) {
                    constexpr int m = decltype(m_)::value;
                    double tq[6];
                    static_for<6>([&](auto q_) { constexpr int q = decltype(q_)::value; tq[q] = row_dot<3 * m + QI[q], QJ[q]>(gR, pb); });
                    mfma_hazard_gap(tq);
#pragma unroll
                    for (int q = 0; q < 6; ++q) accK[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[m], tq[q], accK[q], 0, 0, 0);
                });
            }
            if (doC) {
                static_for<5>([&](auto m_) {
                    constexpr int m = decltype(m_)::value;
                    double tq[9];
                    static_for<9>([&](auto q_) { constexpr int q = decltype(q_)::value; tq[q] = row_dot<3 * m + q / 3, q % 3>(hR, pb); });
                    mfma_hazard_gap(tq);
#pragma unroll
                    for (int q = 0; q < 9; ++q) accC[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[m], tq[q], accC[q], 0, 0, 0);
                });
                if (has_bf) {                    // d(-f . u dA)/dc : -w f_i R_a (dJ/dZ . phi_b): one tile per f, the factor -f_i is applied behind the loop
                    const LoadGeom lg = load_geom(im, Pt.pd);
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        const double jz = load_dz_dot(im, Pt.pd, lg, f, pb[0], pb[1]);
                        accB[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(R0, jz, accB[f], 0, 0, 0);
                    }
                }
            }
        }
        if (has_bf && doC) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int f = 0; f < 3; ++f) accC[3 * i + f] -= Pt.f[i] * accB[f];
        }
        GF_STAMP(4, tstamp);

        // ---- park the next element's inputs; residual of this element: sum the four Gauss-point groups (entry (local a, i)
        //      comes from lane 4 ju + ((iv0 + jv) & 3))
        wave_lds_sync();
        if (more) park(Fn, iv0n, buf ^ 1);
        for (int i = 0; i < 3; ++i) s_g[(kk * 16 + x) * 3 + i] = accR[i];
        wave_lds_sync();
        if (tid < ND && (flags & GF_ASM_R_BIT)) {
            const int a = tid / 3, i = tid - 3 * a, xs = 4 * (a % P1) + ((iv0 + a / P1) & 3), w = 3 * xs + i;
            O.rblk[(size_t)e * ND + tid] = s_g[w] + s_g[48 + w] + s_g[96 + w] + s_g[144 + w];
        }

        GF_STAMP(5, tstamp);
        // ---- flush, first half (K, dR/dh); the dR/dCP half follows behind phase 1 of the next element
        flush(std::integral_constant<int, 0>{}, iv0, iv0n);
        prev_iv0 = iv0; prev_iv0n = iv0n;
        GF_STAMP(6, tstamp);
    }
#ifdef GF_STAMPS
    if ((blockIdx.x & 7) == 0 && tid == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_stamps[k], stamp_acc[k]);
#endif
    if constexpr (WITHC) flush(std::integral_constant<int, 1>{}, prev_iv0, prev_iv0n);      // (with WITHC = false there is no second half)
}

}  // namespace gf
