// gf_element_rec.hpp -- MFMA element kernel (p = 2, 3) that walks a strip of elements, keeps the accumulators across elements and
// STORES every control-point pair once per work item, when its lower row leaves the window: "row records" instead of element blocks.
//
// kl_element_mfma_kernel writes one 43 KB block per element (18 tiles of 16 x 16 plus the mirrored K tiles) and the gather reads every
// block back: a pair of control points is stored up to (p+1)^2 times.  Walking along v, a pair of control points lives in the
// accumulator slot given by its rows' indices modulo 4 (operand lane x <-> basis function (u index x / 4, v slot x % 4)): moving on
// changes which entry of the 1-D v table a lane evaluates, never where a sum sits, so the (p+1) elements of a strip that contribute
// to a pair are summed in registers and a pair is stored once per strip it lies in ((p+1) times instead of (p+1)^2): 16 KB per
// element instead of 43 KB.  Nothing is read back and nothing depends on launch order: the flush is a burst of plain stores (a
// variant that added the sums straight into the CSR arrays -- read-modify-write, launch classes -- was measured slower in round 2 and
// is gone), and kl_gather_rec_kernel sums the <= p + 1 strips (x <= 2 segments) of every pair in a fixed order: bitwise reproducible.
//
// Record of row rho of a work item (RecCfg<WITHC>::SZ = 112 NT doubles, NT tiles): the pairs (A, B) whose lower row is rho,
//     area 1  [rr][tile q][c], c < 16:        A = (iu0 + rr, rho),       B = (iu0 + c / 4, the row >= rho with row % 4 = c % 4)
//     area 2  [rr][rk][tile q][jub], rk < 3:  A = (iu0 + rr, rho + 1 + rk),  B = (iu0 + jub, rho)
// so that what the gather of one control point needs from a record is contiguous (a control point reads [rr = its u index] of both
// areas).  Tiles: K (i <= j) 0..5, dR/dh 6..8, dR/dCP (i, f) 9..17.  Record index = item * rec_rows + (rho - first row of the item).
// Reference path: the same integrals as kl_element_mfma_kernel (GOLDFISH/nonmatching_opt.py:941-1015 via PENGoLINS assembly).
#pragma once
#include "gf_gauss_loop.hpp"

namespace gf {

constexpr int REC_MAX_NEL = 255;      // elements per work item (HostModel::build_rec cuts longer strips)
template <bool WITHC> struct RecCfg { static constexpr int NT = WITHC ? 18 : 9, A2 = 64 * NT, SZ = 112 * NT, QK = 0, QH = 6, QC = 9; };

struct RecOut { double* rec; double* rblk; int rec_rows; };

#ifndef GF_SUMFACT_BUILD
#define GF_SUMFACT_BUILD 1                    // p = 3: row-side sum factorisation of the contraction (gf_gauss_loop.hpp: SfLane); 0: the 16 x 16 x 4 products of rounds 2 - 4
#endif
// LDS of one walking wave (one struct, so that the two instances of the walk below -- polynomial / rational patch -- share it)
template <int P> struct RecShared {
    static constexpr int P1 = P + 1, NB = P1 * P1, TS = P1 * 3 * P1;
    int iv[REC_MAX_NEL + 1];                            // first control-point row of every element of the item (+ one behind)
    double pc[8];                                       // patch constants E, nu, f[3], pd[3]
    double tu[TS], tv[2][TS], wgu[P1], wgv[2][P1];      // v tables double buffered (the next element's are parked while this one's are in use)
    __attribute__((aligned(16))) double g[4 * 3 * 16];  // control-point staging (phases 0-1), residual reduction at the end
    __attribute__((aligned(16))) double im[NB][IM_SIZE];
    __attribute__((aligned(16))) double raw[8][4][8];   // control points of the windows, ring over the row index: c_x, c_y, c_z, w, u_x, u_y, u_z, h
};

// SF: 0 the accumulator register is the u index of the row function (lane / 16 = slot of its row), 1 / 2 row-side sum factorisation on a polynomial / rational
// patch: the register is the SLOT of the row, lane / 16 the u index (gf_gauss_loop.hpp)
template <int P, bool WITHC, bool ALLF, int SF>
__device__ __forceinline__ void rec_walk(const DevModel& M, const WalkItem* __restrict__ items, int item, int flags, const RecOut& O, RecShared<P>& S) {
    static_assert(P == 2 || P == 3, "one 16 x 16 tile: p <= 3");
    using RC = RecCfg<WITHC>;
    constexpr int P1 = P + 1, NB = P1 * P1, NG = NB, ND = 3 * NB, NGRP = (NG + 3) / 4, TS = P1 * 3 * P1;
    const int tid = threadIdx.x, x = tid & 15, kk = tid >> 4;
    // The work item and the patch fields the element loop needs are wave-uniform but arrive through vector loads (the compiler does not prove
    // the addresses uniform); read inside the loop each of them is a load behind s_waitcnt vmcnt(0) -- and vmcnt also covers the 72 record
    // stores of the previous element.  They are read ONCE here into scalar registers, the items' span indices go to LDS.
    auto uni = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
    auto uni64 = [&](long long x) { return (long long)(((unsigned long long)(unsigned)uni((int)((unsigned long long)x >> 32)) << 32) | (unsigned)uni((int)(unsigned long long)x)); };
    WalkItem it = items[item];
    it.patch = uni(it.patch); it.eu = uni(it.eu); it.ev0 = uni(it.ev0); it.nel = uni(it.nel); it.iu0 = uni(it.iu0);
    const PatchDev& Pt = M.patches[it.patch];
    const int p_nu = uni(Pt.nu), p_nelu = uni(Pt.nelu), p_tabu = uni(Pt.tabu), p_tabv = uni(Pt.tabv), p_wu = uni(Pt.wu), p_wv = uni(Pt.wv), p_spv = uni(Pt.spv);
    const long long p_cp_off = uni64(Pt.cp_off), p_elem_off = uni64(Pt.elem_off);
    int* const s_iv = S.iv;
    for (int k = threadIdx.x; k <= it.nel && k <= REC_MAX_NEL; k += 64) s_iv[k] = M.ints[p_spv + it.ev0 + (k < it.nel ? k : it.nel - 1)] - P + (k < it.nel ? 0 : 4);
    // patch constants (E, nu, f[3], pd[3]: contiguous in PatchDev) staged in LDS: read from memory inside the Gauss-point loop they
    // are vector loads behind a vmcnt wait each (the compiler cannot move them across stores), held in registers they cost 16 VGPRs
    double* const s_pc = S.pc;
    if (threadIdx.x < 8) s_pc[threadIdx.x] = (&Pt.E)[threadIdx.x];
    const double* const pf = s_pc + 2; const double* const ppd = s_pc + 5;
    wave_lds_sync();

    double* const s_g = S.g;
    double (*s_c)[3] = reinterpret_cast<double (*)[3]>(s_g);
    double (*s_d)[3] = reinterpret_cast<double (*)[3]>(s_g + 3 * NB);
    double* s_h = s_g + 6 * NB; double* s_w = s_g + 7 * NB;
    double* const s_tu = S.tu; double (*s_tv)[TS] = S.tv; double* const s_wgu = S.wgu; double (*s_wgv)[P1] = S.wgv;
    double (*s_im)[IM_SIZE] = S.im;
    double (*s_raw)[4][8] = S.raw;

    const RowLane L(x);                                  // lane constants of the row expansion

    const bool doK = (flags & GF_ASM_K_BIT) != 0, doC = WITHC && (flags & GF_ASM_C_BIT) != 0, doH = (flags & GF_ASM_H_BIT) != 0;
    const bool has_bf = (pf[0] != 0.0) || (pf[1] != 0.0) || (pf[2] != 0.0);
    const int jub = x >> 2, sb = x & 3;                   // this lane's basis function: u index, v slot
    const int jubc = jub < P1 ? jub : 0;

    gf_d4 accK[6], accC[9], accH[3];                       // SF = 0: the walking accumulators (MFMA destinations)
    for (int q = 0; q < 6; ++q) accK[q] = gf_d4{0, 0, 0, 0};
    for (int q = 0; q < 9; ++q) accC[q] = gf_d4{0, 0, 0, 0};
    for (int q = 0; q < 3; ++q) accH[q] = gf_d4{0, 0, 0, 0};
    // SF != 0, passes with dR/dCP: the walking accumulators are FMA destinations (arch VGPRs) but 72 doubles of them next to phase 1 and the group step do not fit --
    // they are parked in AGPRs (AccReg) and every component's update reads, adds and writes back its four values (16 moves per component and group)
    // (the six K tiles stay in arch VGPRs; GF_SF_PARK_K=1 parks them too: 16 more moves per K component and group, 48 registers more for everything else)
#ifndef GF_SF_PARK_K
#define GF_SF_PARK_K 0
#endif
    constexpr bool AGP = SF != 0 && WITHC, AGPK = AGP && GF_SF_PARK_K != 0;
    AccReg aK[6][4], aC[9][4];
    if constexpr (AGP) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            if constexpr (AGPK) for (int q = 0; q < 6; ++q) acc_init(aK[q][sl]);
            for (int q = 0; q < 9; ++q) acc_init(aC[q][sl]);
        }
    }

    // ---- input fetch of one element: 16 bytes per lane of the window's control points (lane = 4 * local index + quarter:
    //      c_xy | c_zw | u_xy | u_z, h), the v table and v weights; requested behind phase 1, parked in LDS behind the group loop
    const int pa_cp = tid >> 2, pa_q = tid & 3, pa_ju = pa_cp % P1, pa_jv = pa_cp / P1;
    struct Fetch { double2 cp; double tv, wv; };
    // (everything is addressed from the patch record and the element row: no per-element descriptor, no dependent load)
    auto fetch = [&](int ev, int iv0f) {
        Fetch F; F.cp = double2{0.0, 0.0}; F.tv = 0.0; F.wv = 0.0;
        if (pa_cp < NB) {
            const long long g = p_cp_off + (it.iu0 + pa_ju) + (long long)(iv0f + pa_jv) * p_nu;
            if (pa_q < 2) F.cp = reinterpret_cast<const double2*>(M.cp4 + 4 * g)[pa_q];
            else if (pa_q == 2) { F.cp.x = M.u[3 * g]; F.cp.y = M.u[3 * g + 1]; }
            else { F.cp.x = M.u[3 * g + 2]; F.cp.y = M.h[g]; }
        }
        if (tid < TS) F.tv = M.tab[p_tabv + ev * TS + tid];
        if (tid < P1) F.wv = M.tab[p_wv + ev * P1 + tid];
        return F;
    };
    auto park = [&](const Fetch& F, int iv0f, int buf) {
        if (pa_cp < NB) *reinterpret_cast<double2*>(&s_raw[(iv0f + pa_jv) & 7][pa_ju][2 * pa_q]) = F.cp;
        if (tid < TS) s_tv[buf][tid] = F.tv;
        if (tid < P1) s_wgv[buf][tid] = F.wv;
    };
    const int iv_first = s_iv[0];
    {   // prologue: u table and weights of the strip, inputs of the first element
        if (tid < TS) s_tu[tid] = M.tab[p_tabu + it.eu * TS + tid];
        if (tid < P1) s_wgu[tid] = M.tab[p_wu + it.eu * P1 + tid];
        const Fetch F = fetch(it.ev0, iv_first);
        park(F, iv_first, 0);
    }
    SfLane sfl; sfl.au[0] = sfl.au[1] = sfl.au[2] = 0.0; sfl.rot = 0;
    if constexpr (SF != 0) {                             // the lane's row-side u index a1 = x & 3 at its Gauss point g1 = kk: constants of the strip
        wave_lds_sync();
        for (int k1 = 0; k1 < 3; ++k1) sfl.au[k1] = s_tu[(kk * 3 + k1) * P1 + (x & 3)];
    }

    // ---- flush: the pairs whose lower row leaves the window (first row iv0f, next element's first row iv0nf) are complete for
    //      this item; every lane holds pairs of exactly one lower row rho and stores its registers into that row's record.
    const __amdgpu_buffer_rsrc_t rR = buf_rsrc(O.rec + (size_t)item * O.rec_rows * RC::SZ, (unsigned)(O.rec_rows * RC::SZ * 8));
    // (KCH: which tiles -- the SF instances flush K and dR/dCP with their own addressing and the three dR/dh tiles, which every instance accumulates with the
    //  16 x 16 x 4 product, with the SF = 0 addressing)
    auto flush16 = [&](int iv0f, int iv0nf, bool fK, bool fC, bool fH) {
        const int rowa = iv0f + ((kk - iv0f) & 3), rowb = iv0f + ((sb - iv0f) & 3);       // control-point rows of this lane's slots
        const bool live = (rowa - iv0f) < P1 && (rowb - iv0f) < P1 && jub < P1;
        const int rho = rowa < rowb ? rowa : rowb;
        if (live && rho < iv0nf) {
            const int s = rho & 3;
            const bool own = kk == s;                                 // A's row is the lower one: area 1, else area 2 (A's row = rho + 1 + rk)
            const int rk = ((kk - s) & 3) - 1;
            const unsigned sr = 8u * (own ? RC::NT * 16 : 3 * RC::NT * 4), sq = 8u * (own ? 16 : 4);
            unsigned off = 8u * (unsigned)((rho - iv_first) * RC::SZ + (own ? x : RC::A2 + rk * RC::NT * 4 + jub));
#pragma unroll
            for (int rr = 0; rr < P1; ++rr, off += sr) {
                if (fK) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) buf_st(rR, off + (RC::QK + q) * sq, accK[q][rr]);
                }
                if (fH) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) buf_st(rR, off + (RC::QH + q) * sq, accH[q][rr]);
                }
                if constexpr (WITHC) if (fC) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) buf_st(rR, off + (RC::QC + q) * sq, accC[q][rr]);
                }
            }
            if (fK) for (int q = 0; q < 6; ++q) accK[q] = gf_d4{0, 0, 0, 0};
            if (fC) for (int q = 0; q < 9; ++q) accC[q] = gf_d4{0, 0, 0, 0};
            if (fH) for (int q = 0; q < 3; ++q) accH[q] = gf_d4{0, 0, 0, 0};
        }
    };
    auto flush = [&](int iv0f, int iv0nf) {
        if constexpr (SF != 0) {
            // register sl holds the pairs whose row function lies in control-point row rowa (the same for every lane), lane / 16 is its u index
            const int rowb = iv0f + ((sb - iv0f) & 3);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int rowa = iv0f + ((sl - iv0f) & 3);
                const int rho = rowa < rowb ? rowa : rowb;
                if (rho < iv0nf) {
                    const bool own = rowa <= rowb;                        // A's row is the lower one: area 1, else area 2 (A's row = rho + 1 + rk)
                    const int rk = rowa - rho - 1;
                    const unsigned sq = 8u * (own ? 16 : 4);
                    const unsigned off = 8u * (unsigned)((rho - iv_first) * RC::SZ + (own ? x : RC::A2 + rk * RC::NT * 4 + jub) + kk * (own ? RC::NT * 16 : 3 * RC::NT * 4));
                    if (doK) {
#pragma unroll
                        for (int q = 0; q < 6; ++q) { if constexpr (AGPK) { buf_st(rR, off + (RC::QK + q) * sq, acc_get(aK[q][sl])); acc_zero(aK[q][sl]); } else { buf_st(rR, off + (RC::QK + q) * sq, accK[q][sl]); accK[q][sl] = 0.0; } }
                    }
                    if constexpr (WITHC) if (doC) {
#pragma unroll
                        for (int q = 0; q < 9; ++q) { if constexpr (AGP) { buf_st(rR, off + (RC::QC + q) * sq, acc_get(aC[q][sl])); acc_zero(aC[q][sl]); } else { buf_st(rR, off + (RC::QC + q) * sq, accC[q][sl]); accC[q][sl] = 0.0; } }
                    }
                }
            }
            if (doH) flush16(iv0f, iv0nf, false, false, true);
        } else flush16(iv0f, iv0nf, doK, doC, doH);
    };

#if defined(GF_STAMPS) && defined(GF_STAMPS_FINE)
#define GF_OUTER_STAMP(slot) GF_STAMP((slot) < 2 ? (slot) : 7, tstamp)
#else
#define GF_OUTER_STAMP(slot) GF_STAMP(slot, tstamp)
#endif
    unsigned long long tstamp = 0; (void)tstamp;
#ifdef GF_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    tstamp = clock64();
#endif
    for (int t = 0; t < it.nel; ++t) {
        const int ev = it.ev0 + t, buf = t & 1;
        const long long e = p_elem_off + it.eu + (long long)ev * p_nelu;
        const int iv0 = s_iv[t];
        const bool more = t + 1 < it.nel;
        const int iv0n = s_iv[t + 1];                                       // after the last element of the item every row leaves (iv0 + 4)
        // ---- phase 0: this element's control points from the ring
        wave_lds_sync();
        if (tid < NB) {
            const double* rw = s_raw[(iv0 + tid / P1) & 7][tid % P1];
            s_c[tid][0] = rw[0]; s_c[tid][1] = rw[1]; s_c[tid][2] = rw[2]; s_w[tid] = rw[3];
            s_d[tid][0] = rw[4]; s_d[tid][1] = rw[5]; s_d[tid][2] = rw[6];      // displacement coefficients (kl_strains)
            s_h[tid] = rw[7];
        }
        wave_lds_sync();
        const double* const tv = s_tv[buf];
        GF_OUTER_STAMP(0);

        // ---- phase 1: three lanes per Gauss point: kinematics + pointwise closed forms -> s_im
        point_phase<P, WITHC>(x, kk, s_tu, tv, s_c, s_d, s_w, s_h, s_pc, s_wgu, s_wgv[buf], s_im);

        GF_OUTER_STAMP(1);
        // ---- the next element's inputs are requested now (they land during the group loop) and parked in LDS behind it:
        //      vmcnt counts loads and stores in order, so a load right behind the flush stores would wait for all of them
        GF_OUTER_STAMP(2);
        Fetch Fn; Fn.cp = double2{0.0, 0.0}; Fn.tv = 0.0; Fn.wv = 0.0;
        if (more) Fn = fetch(ev + 1, iv0n);
        GF_OUTER_STAMP(3);

        // this lane's basis function in the current element: u index jub, v index = (slot - first row) mod 4
        const int jvb = (sb - iv0) & 3;
        const bool bok = jub < P1 && jvb < P1;
        const int jvc = jvb < P1 ? jvb : 0;
        const double bval = bok ? 1.0 : 0.0;                 // lanes beyond the basis functions contribute zero rows / columns
        sfl.rot = iv0 & 3;
        double accR[3] = {0.0, 0.0, 0.0};
        gf_d4 accB[3] = {gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}, gf_d4{0, 0, 0, 0}};   // body force: sum_gp R_a (dJ/dZ . phi_b)_f, scaled by -f_i behind the loop
        for (int grp = 0; grp < NGRP; ++grp) {
            const int gp = 4 * grp + kk, gpc = gp < NG ? gp : NG - 1, gu = gpc % P1, gv = gpc / P1;   // Gauss point of this lane's group
            const double* im = s_im[gpc];
            const double wq = gp < NG ? im[IM_WQ] : 0.0;        // padded Gauss-point slots contribute nothing
            if constexpr (AGPK) gauss_group<P, WITHC, ALLF, SF>(L, im, wq, s_tu, tv, gu, gv, jubc, jvc, bval, doK, doC, doH, has_bf, pf, ppd, aK, aC, accH, accB, accR, sfl GF_GROUP_STAMP_ARGS);
            else if constexpr (AGP) gauss_group<P, WITHC, ALLF, SF>(L, im, wq, s_tu, tv, gu, gv, jubc, jvc, bval, doK, doC, doH, has_bf, pf, ppd, accK, aC, accH, accB, accR, sfl GF_GROUP_STAMP_ARGS);
            else gauss_group<P, WITHC, ALLF, SF>(L, im, wq, s_tu, tv, gu, gv, jubc, jvc, bval, doK, doC, doH, has_bf, pf, ppd, accK, accC, accH, accB, accR, sfl GF_GROUP_STAMP_ARGS);
        }
        if (SF == 0 && has_bf && doC) {          // (the SF instances carry the body-force term inside their dR/dCP components)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int f = 0; f < 3; ++f) accC[3 * i + f] -= pf[i] * accB[f];
        }
        GF_OUTER_STAMP(4);

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

        GF_OUTER_STAMP(5);
        // ---- store the pairs whose lower row leaves the window
        flush(iv0, iv0n);
        GF_OUTER_STAMP(6);
    }
#ifdef GF_STAMPS
    if ((blockIdx.x & 7) == 0 && tid == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_stamps[k], stamp_acc[k]);
#endif
}

// SF = 1 / 2: the work items order[0 .. grid) -- the items of the polynomial / of the rational patches (HostModel::rec_order); one kernel per kind, so that each
// instance has the register file to itself.  order = nullptr: item = workgroup.
template <int P, bool WITHC = true, bool ALLF = false, int SF = 0>      // ALLF: flags = R + K + dR/dCP + dR/dh known at compile time (gauss_group)
__global__ __launch_bounds__(64) void kl_element_rec_kernel(DevModel M, const WalkItem* __restrict__ items, const int* __restrict__ order, int flags, RecOut O) {
    __shared__ RecShared<P> S;
    const int item = order ? __builtin_amdgcn_readfirstlane(order[blockIdx.x]) : (int)blockIdx.x;
    rec_walk<P, WITHC, ALLF, SF>(M, items, item, flags, O, S);
}

// Record gather: ONE wave per control point a = (ia, ja) sums, strip by strip and segment by segment (a fixed order), what the row
// records hold for its three dof rows, then writes the rows (gather_write_rows: Dirichlet entries, coupling-only columns, penalty
// rows).  From a work item (strip iu0 .. iu0 + p, segment g) with rr_a = ia - iu0:
//   G1  record ja,      area 1 [rr_a][q][c]:            a as A, neighbours b = (iu0 + c / 4, ja + ((c - ja) & 3)): all tiles, contiguous
//   G3  record ja-1-rk, area 2 [rr_a][rk][q][jub]:      a as A, b = (iu0 + jub, ja - 1 - rk): all tiles, contiguous per rk
//   G2  record ja,      area 2 [rr][rk][q'][rr_a]:      a as B, b = (iu0 + rr, ja + 1 + rk): K^(i > j) = tile (j, i) of the pair (b, a)
//   G4  record ja-d,    area 1 [rr][q'][4 rr_a + ja%4]: a as B, b = (iu0 + rr, ja - d)
// A pair is present in an item only if one of the item's elements holds both rows; everything else in a record row is never
// written and never read.
template <int P, bool WITHC>
__global__ __launch_bounds__(64) void kl_gather_rec_kernel(DevModel M, long long a_first, long long a_end, const int* __restrict__ cp_list, int flags, const double* __restrict__ rec, int rec_rows,
                                                           const RecCp* __restrict__ reccp, double* __restrict__ valK, double* __restrict__ valC0,
                                                           double* __restrict__ valC1, double* __restrict__ valC2, double* __restrict__ valH, int pen_add) {
    using RC = RecCfg<WITHC>;
    constexpr int P1 = P + 1, WB = 2 * P + 1, NBOX = WB * WB, NT = RC::NT, SZ = RC::SZ, A2 = RC::A2;
    // workgroup w runs on XCD w % 8: every XCD takes a contiguous range of control points, so that the record lines shared by
    // neighbouring control points are fetched into one L2
    // cp_list: the control points to gather (ascending), entries a_first .. a_end - 1 of it; nullptr: the control points a_first .. a_end - 1
    const long long chunk = (a_end - a_first + 7) / 8;
    const long long idx = a_first + (long long)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if ((long long)(blockIdx.x >> 3) >= chunk || idx >= a_end) return;
    const long long a = cp_list ? (long long)cp_list[idx] : idx;
    const CpDesc& cd = M.cpdesc[a];
    const int ia = cd.ia, ja = cd.ja, i0 = cd.i0, j0 = cd.j0, wbox = cd.i1 - cd.i0 + 1;
    const long long ptr_c = M.nb_ptr_c[a], deg_c = M.nb_ptr_c[a + 1] - ptr_c, ptr_s = M.nb_ptr_s[a], deg_s = M.nb_ptr_s[a + 1] - ptr_s;
    const int lane = threadIdx.x;
    __shared__ double acc[21 * NBOX];                   // aK [3][NBOX][3] | aH [3][NBOX] | aC [3 f][3 i][NBOX]
    __shared__ unsigned short s_meta[GATHER_MAXMETA];
    const RecCp rc = reccp[a];                          // the work items of a (wave-uniform: scalar loads)
    const bool pen_row_a = M.pen_row[a] != 0;           // requested now, used by the write phase
    // One wave per workgroup: LDS operations execute in order, no barrier anywhere; nothing below waits for memory before the
    // record loads are issued (the neighbour metadata of the write phase is requested behind them).
    for (int k = lane; k < 21 * NBOX; k += 64) acc[k] = 0.0;
    // accumulator address of tile q: base + box slot * stride;  mirrored entry (i > j) of the tiles (0,1), (0,2), (1,2)
    auto tile_base = [&](int q) {
        const int qi = q < 3 ? 0 : (q < 5 ? 1 : 2), qj = q < 3 ? q : (q < 5 ? q - 2 : 2);
        return q < 6 ? qi * NBOX * 3 + qj : (q < 9 ? 9 * NBOX + (q - 6) * NBOX : 12 * NBOX + (((q - 9) % 3) * 3 + (q - 9) / 3) * NBOX);
    };
    auto mirror_base = [&](int m) { const int qi = m == 2 ? 1 : 0, qj = m == 0 ? 1 : 2; return qj * NBOX * 3 + qi; };
    const bool doC = WITHC && (flags & GF_ASM_C_BIT) != 0, doK = (flags & GF_ASM_K_BIT) != 0, doH = (flags & GF_ASM_H_BIT) != 0;
#ifdef GF_REC_NOITEMS
    const int nit = 0;
#else
    const int nit = rc.nit;
#endif
    constexpr int N1 = NT * 16, NP1 = (N1 + 63) / 64, N3 = 3 * NT * 4, NP3 = (N3 + 63) / 64, NV = NP1 + NP3 + 2;
    // ---- per-lane task table (independent of strip and segment): accumulator address for iu0 = i0, its stride in iu0 (1 or 3), the
    //      row offset d whose presence decides, the offset in the record (relative to the row record of ja) and its stride in rr_a.
    //      task v < NP1: G1, < NP1 + NP3: G3, then G2, G4.  tk = address | stride3 << 15 | d << 16 | valid << 20
    int tk[NV], to[NV];
#pragma unroll
    for (int ps = 0; ps < NP1; ++ps) {
        const int t = lane + 64 * ps, q = (t >> 4) < NT ? (t >> 4) : 0, c = t & 15, jub = c >> 2, dv = (c - ja) & 3;
        const bool en = q < 6 ? doK : (q < 9 ? doH : doC);
        const bool ok = t < N1 && en && jub < P1 && dv < P1;
        tk[ps] = ((tile_base(q) + (jub + (ja + dv - j0) * wbox) * (q < 6 ? 3 : 1)) & 0x7fff) | (q < 6 ? 1 << 15 : 0) | ((3 + dv) << 16) | (ok ? 1 << 20 : 0);
        to[ps] = t;                                                   // + rr_a * NT * 16
    }
#pragma unroll
    for (int ps = 0; ps < NP3; ++ps) {
        const int t = lane + 64 * ps, rk = t / (NT * 4) < 3 ? t / (NT * 4) : 0, r = t - (t / (NT * 4)) * NT * 4, q = r >> 2, jub = r & 3;
        const bool en = q < 6 ? doK : (q < 9 ? doH : doC);
        const bool ok = t < N3 && en && jub < P1 && rk + 1 < P1;
        tk[NP1 + ps] = ((tile_base(q) + (jub + (ja - 1 - rk - j0) * wbox) * (q < 6 ? 3 : 1)) & 0x7fff) | (q < 6 ? 1 << 15 : 0) | ((2 - rk) << 16) | (ok ? 1 << 20 : 0);
        to[NP1 + ps] = (-1 - rk) * SZ + A2 + rk * NT * 4 + r;          // + rr_a * 3 * NT * 4
    }
    {   // G2: 3 tiles x 3 rk x 4 rr (a as B, rows above)
        const int m = lane / 12 < 3 ? lane / 12 : 0, r = lane - 12 * (lane / 12), rk = r >> 2, rr = r & 3, qm = m == 2 ? 4 : m + 1;
        const bool ok = lane < 36 && doK && rr < P1 && rk + 1 < P1;
        tk[NP1 + NP3] = ((mirror_base(m) + (rr + (ja + 1 + rk - j0) * wbox) * 3) & 0x7fff) | (1 << 15) | ((4 + rk) << 16) | (ok ? 1 << 20 : 0);
        to[NP1 + NP3] = A2 + ((rr * 3 + rk) * NT + qm) * 4;           // + rr_a
    }
    {   // G4: 3 tiles x 4 d x 4 rr (a as B, own row and rows below)
        const int m = (lane >> 4) < 3 ? (lane >> 4) : 0, r = lane & 15, d = r >> 2, rr = r & 3, qm = m == 2 ? 4 : m + 1;
        const bool ok = lane < 48 && doK && rr < P1 && d < P1;
        tk[NP1 + NP3 + 1] = ((mirror_base(m) + (rr + (ja - d - j0) * wbox) * 3) & 0x7fff) | (1 << 15) | ((3 - d) << 16) | (ok ? 1 << 20 : 0);
        to[NP1 + NP3 + 1] = -d * SZ + (rr * NT + qm) * 16 + (ja & 3); // + 4 * rr_a
    }
    // ---- the work items that hold pairs of a: strips k (ascending), segments g (ascending); the loads of item n + 1 are in flight
    //      while item n is summed (one wave: its LDS operations execute in order; the targets of one instruction are distinct)
    struct Item { double v[NV]; int iu0; unsigned pm; };
    auto load_item = [&](int n) {
        Item I;
        int row = rc.it[0].row; unsigned info = rc.it[0].info;
#pragma unroll
        for (int q = 1; q < 8; ++q) if (q == n) { row = rc.it[q].row; info = rc.it[q].info; }       // register-resident table: no dynamic indexing
        I.iu0 = int(info & 255u); I.pm = info >> 16;
        const int rra = int((info >> 8) & 255u);
        const double* Rja = rec + (size_t)row * SZ;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int mul = v < NP1 ? NT * 16 : (v < NP1 + NP3 ? 3 * NT * 4 : (v == NP1 + NP3 ? 1 : 4));
            const bool ok = ((tk[v] >> 20) & 1) && ((I.pm >> ((tk[v] >> 16) & 7)) & 1);
            // unconditional load (an absent pair reads the head of the row record; its value is never added): a predicated load is a branch, and
            // behind branches the waitcnt insertion stops counting -- every ds_add_f64 then waited for ALL loads in flight (vmcnt(0)),
            // the next item's included
            I.v[v] = Rja[ok ? to[v] + rra * mul : 0];
        }
        return I;
    };
    auto add_item = [&](const Item& I) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bool ok = ((tk[v] >> 20) & 1) && ((I.pm >> ((tk[v] >> 16) & 7)) & 1);
            // ds_add_f64: one LDS instruction, no returned value to wait for (distinct targets within an instruction, program order between them)
            if (ok) (void)__hip_atomic_fetch_add(&acc[(tk[v] & 0x7fff) + I.iu0 * ((tk[v] >> 15) & 1 ? 3 : 1)], I.v[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
#ifndef GF_REC_BATCH
#define GF_REC_BATCH 2
#endif
    constexpr int NBT = GF_REC_BATCH;                   // work items whose loads are in flight together
    Item B[NBT];
#pragma unroll
    for (int q = 0; q < NBT; ++q) if (q < nit) B[q] = load_item(q);
    // neighbour metadata of the write phase: requested behind the first record loads, parked in LDS before the rows are written
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
#ifdef GF_REC_NOWRITE
    if (acc[lane] == 123.456)
#endif
    gather_write_rows<NBOX, WITHC>(M, a, lane, doK, doC, doH, padd, (unsigned)rc.flags, ptr_c, deg_c, ptr_s, deg_s, s_meta, acc, acc + 12 * NBOX, acc + 9 * NBOX,
                                   valK, valC0, valC1, valC2, valH);
}

}  // namespace gf
