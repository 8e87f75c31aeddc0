// gf_penalty_row16.hpp -- penalty rows of one owned control point (p = 2, 3), one 16-lane row of the wave per visit.
//
// pen_owner_kernel gives every lane <= PEN_SL neighbour slots of a and lets it add its 3 x 3 blocks whenever its slot's control point
// lies in a visit's support window: 32 of the >= 128 slot-lanes work per visit, the w vectors come back from LDS 45 times per slot
// and visit.  Here the lanes ARE the window: a 16-lane row takes one visit (four visits in flight per wave), lane c is the control
// point at window position c, first of side A, then of side B (the side is a compile-time constant of the pass).  The 90 w values of
// the visit (vertex Hessian rows contracted with nu_a) are computed once per row, six per lane, and reach the block update through
// v_fmac_f64_dpp row_newbcast from registers -- no LDS read, no barrier; the 18 block entries of (a, b) are added to the row's
// accumulators in LDS by ds_add_f64 at the slot of b in a's neighbour list (host-built per visit: DevPenalty::slots).  The next four
// visits' data is requested before the current four are contracted.  ~65 instead of ~190 instructions per visit.
// Summation order per entry: every 16-lane row walks its quarter of the visit list in order; where several rows hit one address in
// the same instruction the LDS unit serialises the lanes in a fixed order -- run-to-run reproducible (tests compare bitwise).
// Where the time goes (clock64 stamps, 8 x 8-patch slice, 24 visits per row on average): prologue 5.6 k, visit loop 31 k (5.2 k per batch of
// four visits, with two waves per SIMD), epilogue 6.1 k cycles per wave.  Measured without effect on that: run-length accumulation in
// registers (ds_add_f64 only when a lane's slot changes), a loop without the `cur = nxt` copy at its back edge (the compiler waits for the
// freshly requested registers there: s_waitcnt vmcnt(1) + 30 moves) with unconditional loads masked at use (exact vmcnt counts, the next
// batch really in flight), fewer masks: 426-441 us each -- neither the LDS unit nor the load latency nor the VALU count is the limit;
// the 37 vector loads per lane and batch (8-byte, four vertex records per instruction) through the texture addresser are the suspect
// (a record layout with the three Hessian rows of a w value contiguous would need 26; dropping 9 of the 37 loads in a timing experiment
// gave 384 instead of 426 us, so that layout is worth about 7 %).
// Measured and dropped earlier: a second batch of data in flight (513 / 444 vs 487 / 426 us on the 8 x 8-patch slice before / after the bank fix); one
// launch per class of neighbour counts (<= 64, <= 88, rest: more waves per CU for the narrow rows, but three tails: 463 vs 426 us).
// Round 4: the one-matrix instances (Newton pass: K only; linearize after a solve: dR/dc only) allocate only their own accumulators -- half the LDS, twice
// the waves per CU -- without effect on their time (K-only instance at C4: 1.26 -> 1.24 ms: the kernel is not short of waves).  The full pass as TWO such
// launches (K rows, then dR/dc rows) is slower than the fused instance (1.27 + 1.03 = 2.29 vs 1.80 ms at C4: the visit and vertex loads are issued twice,
// and those loads are what the kernel waits for).
// The kernel WRITES the rows (the gather adds the shell part), like pen_owner_kernel<.., ADD = false>.
// Reference path: nonmatching_opt.py:745-752, 789-801, 861-887 (penalty residual and its blocks of dR/du, dR/dCP).
#pragma once
#include "gf_gauss_loop.hpp"

namespace gf {

// p = 4 (round 5): the 5 x 5 window takes a 32-lane row = two DPP rows of 16; both halves compute the visit's 90 w values (six per lane position c % 16), so that
// every broadcast stays inside its DPP row; two visits in flight per wave instead of four.  (pen_owner_kernel was the p = 4 path until round 5: 3.4 ms on one
// GPU's share of C5 for < 1 % of the quadrature points.)
template <int P, bool WITHC, bool WITHK>
__global__ __launch_bounds__(64) void pen_row16_kernel(DevModel M, DevPenalty Q, int flags, const double* __restrict__ pbuf, double* __restrict__ R,
                                                        double* __restrict__ valK, double* __restrict__ valC0, double* __restrict__ valC1, double* __restrict__ valC2) {
    static_assert(P >= 2 && P <= 4, "the support window must fit a 32-lane row");
    constexpr int P1 = P + 1, NB = P1 * P1, LW = NB <= 16 ? 16 : 32, NV = 64 / LW;      // lanes per visit, visits in flight per wave
    const long long chunk = (Q.nrow_groups + 7) / 8;                           // XCD-contiguous ranges of row groups (see pen_owner_kernel)
    const long long gidx = (long long)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if ((long long)(blockIdx.x >> 3) >= chunk || gidx >= Q.nrow_groups) return;
    const int lane = threadIdx.x, g = lane / LW, c = lane % LW, c16 = lane & 15;      // c16: position in the DPP row (which of the 90 w values this lane computes)
    const long long e0 = Q.ent_ptr[gidx], e1 = Q.ent_ptr[gidx + 1];
    // row g takes the g-th quarter of the visit list: the four visits of a batch then lie ~n / 4 vertices apart and mostly hit
    // different slots (neighbouring vertices share their windows: the same addresses in one ds_add_f64)
    const long long nq = (e1 - e0 + NV - 1) / NV, eq0 = e0 + g * nq, eq1 = eq0 + nq < e1 ? eq0 + nq : e1;
    const int a = Q.row_cp[gidx];
    const long long ptr_c = M.nb_ptr_c[a], deg_c = M.nb_ptr_c[a + 1] - ptr_c;
    const bool mats = (flags & (GF_ASM_K_BIT | GF_ASM_C_BIT)) != 0;
    // accumulators [deg_c][9] for the K blocks, then [deg_c][9] for the dR/dc blocks (entry (i, j) at 3 i + j).  A stride of 9 doubles
    // = 18 banks walks all 32 banks in 16 slots: the ds_add_f64 of a window (slots s .. s + 3, s + 7 .. in a box 7 wide) meet two-way
    // at most (with 18 doubles per slot, 36 banks, every eighth slot collided: five-way; the kernel is LDS bound)
    extern __shared__ double s_acc[];
    double* const s_accK = s_acc; double* const s_accC = (WITHK && WITHC) ? s_acc + 9 * (size_t)deg_c : s_acc;      // a one-matrix instance has only its own accumulators (half the LDS: twice the waves per CU)
    __shared__ double s_r[NV][4];
    if (mats) for (int k = lane; k < (int)deg_c * ((WITHK && WITHC) ? 18 : 9); k += 64) s_acc[k] = 0.0;

    // the six w values of lane c: index c + 16 q of [wK (i, col) 54 | wC (i, col) 36]; offset of their Hessian row 0 in the vertex record
    // for side 0 and the stride between rows (the rows 9 s + 3 m + i, m = 0..2, are contracted with nu_a's value / d1 / d2)
    int woff[6], wstr[6]; bool wok[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int idx = c16 + 16 * q;
        if (idx < 54) { woff[q] = PB_HYY + (idx / 18) * 18 + idx % 18; wstr[q] = 18; wok[q] = WITHK && mats; }
        else if (idx < 90) { const int j = idx - 54; woff[q] = PB_HYC + (j / 12) * 12 + j % 12; wstr[q] = 12; wok[q] = WITHC && mats; }
        else { woff[q] = 0; wstr[q] = 0; wok[q] = false; }
    }
    // Software pipeline, three stages deep: the visit record (vertex id, side, local index of a) of batch n + 2, the vertex data of
    // batch n + 1 (its addresses need that record), the contraction of batch n (a second batch of data in flight: slower, 513 vs 487 us).  Two waves per SIMD fit the LDS accumulators, so the
    // two dependent round trips per batch would otherwise be the kernel's time.
    struct Head { int v, sal; long long ee; bool ok; };
    auto load_head = [&](long long eb) {
        Head H; const long long e = eq0 + eb;
        H.ok = e < eq1; H.ee = H.ok ? e : e0;
        H.v = Q.entries[H.ee].v; H.sal = Q.entries[H.ee].sal;
        return H;
    };
    struct Batch { double n3[3], h[6][3], bv[2][3], g3[3]; int slot[2]; bool ok; };
    auto load = [&](const Head& H) {
        Batch B;
        B.ok = H.ok;
        const long long v = H.v; const int s = H.sal >> 8, al = H.sal & 255;
        const double* na = Q.pt_nu + ((size_t)v * 2 + s) * 3 * NB; const double* pb = pbuf + (size_t)v * PB_STRIDE;
#pragma unroll
        for (int m = 0; m < 3; ++m) B.n3[m] = B.ok ? na[m * NB + al] : 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const double* h = pb + woff[q] + 9 * s * wstr[q];
#pragma unroll
            for (int m = 0; m < 3; ++m) B.h[q][m] = wok[q] ? h[3 * m * wstr[q]] : 0.0;
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) B.g3[m] = c < 3 ? pb[PB_GRAD + 9 * s + 3 * m + c] : 0.0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const double* nb = Q.pt_nu + ((size_t)v * 2 + t) * 3 * NB + (c < NB ? c : 0);
#pragma unroll
            for (int m = 0; m < 3; ++m) B.bv[t][m] = (c < NB && B.ok && mats) ? nb[m * NB] : 0.0;
            B.slot[t] = (c < NB && B.ok && mats) ? (int)Q.slots[((size_t)H.ee * 2 + t) * LW + c] : 0xFFFF;
        }
        return B;
    };
    double racc = 0.0;
    auto contract = [&](const Batch& B) {
        double wr[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) wr[q] = B.n3[0] * B.h[q][0] + B.n3[1] * B.h[q][1] + B.n3[2] * B.h[q][2];
        racc += B.n3[0] * B.g3[0] + B.n3[1] * B.g3[1] + B.n3[2] * B.g3[2];      // lanes c < 3: residual entry (a, c) of this row's visit
        asm volatile("s_nop 1" : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]), "+v"(wr[4]), "+v"(wr[5]));   // VALU result -> DPP source: two wait states
        if (!mats) return;
        static_for<2>([&](auto t_) {
            constexpr int T = decltype(t_)::value;
            double kk[9], cc[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) { kk[q] = 0.0; cc[q] = 0.0; }
            static_for<9>([&](auto ij_) {
                constexpr int IJ = decltype(ij_)::value, I = IJ / 3, J = IJ % 3;
                if constexpr (WITHK) {
                    static_for<3>([&](auto m_) {
                        constexpr int MM = decltype(m_)::value, IDX = I * 18 + 9 * T + 3 * MM + J;
                        fmac_bcast<IDX % 16>(kk[IJ], wr[IDX / 16], B.bv[T][MM]);
                    });
                }
                if constexpr (WITHC) {
                    static_for<2>([&](auto m_) {
                        constexpr int MM = decltype(m_)::value, IDX = 54 + I * 12 + 6 * T + 3 * MM + J;
                        fmac_bcast<IDX % 16>(cc[IJ], wr[IDX / 16], B.bv[T][MM + 1]);
                    });
                }
            });
            if (B.slot[T] != 0xFFFF) {
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    if constexpr (WITHK) (void)__hip_atomic_fetch_add(s_accK + B.slot[T] * 9 + q, kk[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if constexpr (WITHC) (void)__hip_atomic_fetch_add(s_accC + B.slot[T] * 9 + q, cc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        });
    };
#ifndef GF_PEN16_DEPTH
#define GF_PEN16_DEPTH 1
#endif
    if (e0 < e1) {
#if GF_PEN16_DEPTH == 1
        Head hd = load_head(0);
        Batch cur = load(hd);
        hd = load_head(1);
        for (long long eb = 0; eb < nq; ++eb) {
            Batch nxt = cur;
            if (eb + 1 < nq) { nxt = load(hd); hd = load_head(eb + 2); }
            contract(cur);
            cur = nxt;
        }
#else
        Head hd = load_head(0);
        Batch cur = load(hd);
        hd = load_head(1);
        Batch n1 = cur;
        if (1 < nq) n1 = load(hd);
        hd = load_head(2);
        for (long long eb = 0; eb < nq; ++eb) {
            Batch n2 = n1;
            if (eb + 2 < nq) { n2 = load(hd); hd = load_head(eb + 3); }
            contract(cur);
            cur = n1; n1 = n2;
        }
#endif
    }
    // ---- residual: the four rows' sums in fixed order; blocks: every entry of the rows is written (the gather adds the shell part)
    if (c < 3) s_r[g][c] = racc;
    wave_lds_sync();
    if ((flags & GF_ASM_R_BIT) && lane < 3) {
        if constexpr (NV == 4) R[3 * (long long)a + lane] = ((s_r[0][lane] + s_r[1][lane]) + s_r[2][lane]) + s_r[3][lane];
        else R[3 * (long long)a + lane] = s_r[0][lane] + s_r[1][lane];
    }
    if (!mats) return;
    for (int k = lane; k < (int)deg_c; k += 64) {
        const double* srcK = s_accK + k * 9; const double* srcC = s_accC + k * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if constexpr (WITHK) { if (flags & GF_ASM_K_BIT) valK[9 * ptr_c + (long long)i * 3 * deg_c + 3 * k + j] = srcK[3 * i + j]; }
                if constexpr (WITHC) { if (flags & GF_ASM_C_BIT) { double* dst = j == 0 ? valC0 : (j == 1 ? valC1 : valC2); dst[3 * ptr_c + (long long)i * deg_c + k] = srcC[3 * i + j]; } }
            }
    }
}

}  // namespace gf
