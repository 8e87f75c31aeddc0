// gf_gauss_loop.hpp -- the part every MFMA element kernel of degree p = 2, 3 shares (kl_element_rec_kernel: default path;
// kl_element_mfma_kernel: one block per element, the cross-check path): the per-Gauss-point kinematics + closed forms
// (point_phase) and the Gauss-point group step -- basis function of the lane, row expansion of the pointwise Hessians, formation
// of the B operands through DPP and the MFMA contraction (gauss_group).  A fix to the formulation lands here once.
//
// Why the contraction runs on the FP64 matrix pipe: with (p+1)^2 = 16 basis functions the element matrices are exactly one
// 16 x 16 MFMA tile per (i, j) component,
//     K^{ij}[a][b] = sum_gp sum_m  phi_a[gp][m] * T^{ij}_b[gp][m],   T^{ij}_b[gp][m] = w_gp sum_m' G[(m,i),(m',j)] phi_b[gp][m']
// and v_mfma_f64_16x16x4 contracts 4 Gauss points at once (k = Gauss point of the lane's 16-lane group).  Lane (x, kk) =
// (lane % 16, lane / 16) holds phi of its basis function at Gauss point 4 grp + kk in registers: it supplies the A operand and
// computes the B operand T_b on the fly.  Lane x < 15 also expands row x of the pointwise Hessians G, Hc of its group's Gauss
// point and KEEPS it in registers; the FMAs that form T read those entries from the owning lane through DPP row_newbcast
// (v_fmac_f64_dpp) -- no T tile, no expanded Hessian in LDS, no cross-wave barrier (one wave per workgroup), 300 MFMAs instead
// of 4800 FMA instructions per element.  tools/ubench_mfma_loop.hip measures this loop at 68 cycles per (component, m) unit and
// checks the operand layout:  A[i][k]: lane = i + 16 k    B[k][j]: lane = j + 16 k    D[i][j]: lane = j + 16 (i % 4), register i / 4.
// Reference path: GOLDFISH/nonmatching_opt.py:941-1015 (RIGA, dRIGAduIGA, dRIGAdCPIGA, dRIGAdh_th) via PENGoLINS' assembly.
#pragma once
#include <type_traits>

namespace gf {

typedef double gf_d4 __attribute__((ext_vector_type(4)));
typedef unsigned gf_u2 __attribute__((ext_vector_type(2)));

// t += (value of g held by lane LANE of this lane's 16-lane row) * p.   gfx950 has the DPP form of v_fmac_f64 (row_newbcast
// only); the compiler does not fold a DPP move into FP64 FMAs, hence the inline assembly.  The DPP source must not have been
// written by a VALU instruction in the two preceding slots (dpp_source_fence below; tools/check_dpp_hazard.py checks the code object).
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
// the same for a CURVATURE row (m >= 2) of G / Hc: its curvature columns are rank one in the components -- G[(2 + k, i), (2 + c, j)] = n_i n_j f_k f_c (J t^3 C)_kc,
// Hc: n_i (n_j - N_j) ... (kl_point.hpp:310, 327) -- so that part of the dot product is one product of two per-lane factors (t0) and only the two tangent
// columns come from the row's lane: a multiply and two DPP FMAs instead of an initialising move and five
template <int LANE, int J> __device__ __forceinline__ double row_dot2(double t0, const double (&g)[15], const double (&p)[5]) {
    double t = t0;
    fmac_bcast<LANE>(t, g[J], p[0]);
    fmac_bcast<LANE>(t, g[3 + J], p[1]);
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
// Buffer resources (base in SGPRs, 32-bit byte offset per lane): half the address registers and arithmetic of flat 64-bit pointers.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(double* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, unsigned off, double v) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(gf_u2, v), r, (int)off, 0, 0); }

// ---- phase 1: three lanes per Gauss point (gp = x, part ic = kk < 3): kinematics + pointwise closed forms ------------------------
// Lane (gp, ic) sums component ic of the reference and deformed control points (sum factorisation over the tensor-product basis:
// per row jv of control points the three u-sums, then the six (du, dv) combinations; the rational derivatives follow by the
// quotient rule, rationalize6 being linear in the B-spline values), the three lanes exchange their components through the Gauss
// point's (not yet written) record, and each produces the record columns c = ic, 3 + ic (kl_point.hpp: shell_point_cols).
// s_c / s_d: reference homogeneous control points / displacement coefficients of the element, s_w weights, s_h thickness; tu / tv: 1-D tables
// [gp][3][p+1]; wgu / wgv: Gauss weights x span length / 2; pc: E, nu of the patch.  Leaves s_im[gp] complete (one wave: in order).
template <int P, bool WITHC>
__device__ __forceinline__ void point_phase(int x, int kk, const double* tu, const double* tv, const double (*s_c)[3], const double (*s_d)[3],
                                            const double* s_w, const double* s_h, const double* pc, const double* wgu, const double* wgv,
                                            double (*s_im)[IM_SIZE]) {
    constexpr int P1 = P + 1, NG = P1 * P1;
    const int gp = x < NG ? x : NG - 1, ic = kk < 3 ? kk : 0, gu = gp % P1, gv = gp / P1;
    const bool act = kk < 3 && x < NG;
    double* im = s_im[gp];
    double W[6], th = 0.0;
    if (act) {
        double Ac[6], Ad[6];
        for (int k = 0; k < 6; ++k) { W[k] = 0.0; Ac[k] = 0.0; Ad[k] = 0.0; }
        double U[3][P1];
        for (int d = 0; d < 3; ++d) for (int j = 0; j < P1; ++j) U[d][j] = tu[(gu * 3 + d) * P1 + j];
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
    double z[15], Z[15], dz[15];                        // s_d holds the displacement coefficients: Ad sums to dz = z - Z (kl_point.hpp: kl_strains)
    if (act) for (int k = 0; k < 15; ++k) { Z[k] = im[k]; dz[k] = im[15 + k]; z[k] = Z[k] + dz[k]; }
    wave_lds_sync();                                   // all three lanes hold z, Z before the record overwrites the exchange slots
    if (act) {
        const double dsel[3] = {ic == 0 ? 1.0 : 0.0, ic == 1 ? 1.0 : 0.0, ic == 2 ? 1.0 : 0.0};
        shell_point_cols<WITHC>(z, Z, dz, th, pc[0], pc[1], ic, dsel, kk == 0, im);
        if (kk == 0) {
            for (int k = 0; k < 6; ++k) im[IM_W + k] = W[k];
            im[IM_WQ] = wgu[gu] * wgv[gv];
        }
    }
    wave_lds_sync();
}

// ---- lane constants of the row expansion: lane x < 15 expands row r = x = 3 m_r + i_r of G = Pzz and Hc = Pzz + PzZ.
//      Tangent rows (r < 6) and curvature rows share ONE code path: the closed forms have the same shape
//          G[r][s]  = sum_k e_k(r) CEZ[k][s] + b_k(r) CBG[k][s] - X(r,s) + delta      (tangent columns s < 6)
//          PzZ[r][s] = Pz[r] JZJ[s] + sum_k e_k(r) JDNV[k][s] - b_k(r) JDMO[k][s]
//      with e_k = 0, b_k = f_k n_i delta_{k,k_r}, X = Jmo_k dn_i/dg_s on curvature rows (kl_point.hpp ez_entry/bz_entry),
//      so the row type only selects lane-constant masks and offsets -- no divergent branches.
struct RowLane {
    int r, ir, kr, rt, oE2, oJ0, oJ1, oX[6];
    double mt, m0, m1, f3c, ck[3], dij[3];
    __device__ __forceinline__ RowLane() {}
    __device__ __forceinline__ explicit RowLane(int x) { init(x); }
    __device__ __forceinline__ void init(int x) {
        const bool tang = x < 6;
        r = x < 15 ? x : 14; const int mr = r / 3; ir = r - 3 * mr;
        kr = mr >= 2 ? mr - 2 : 0; rt = tang ? r : 0;                     // curvature component of a curvature row; tangent row index (clamped)
        mt = tang ? 1.0 : 0.0; m0 = (mr == 0) ? 1.0 : 0.0; m1 = (mr == 1) ? 1.0 : 0.0;
        f3c = tang ? 0.0 : ((kr == 2) ? 2.0 : 1.0);
        for (int k = 0; k < 3; ++k) { ck[k] = (!tang && kr == k) ? 1.0 : 0.0; dij[k] = (tang && ir == k) ? 1.0 : 0.0; }
        oE2 = IM_G + (tang ? 3 * (1 - mr) + ir : 0);
        oJ0 = IM_JNV + (mr == 0 ? 0 : 2); oJ1 = IM_JNV + (mr == 1 ? 1 : 2);
        for (int s = 0; s < 6; ++s) oX[s] = tang ? IM_HMN + hmn_idx(r, s) : IM_DN + 6 * ir + s;
    }
    // the lane-dependent entries of the record the expansion starts from: requested TOGETHER (and together with the caller's other loads) in front of a
    // scheduling barrier -- one wave per SIMD has nobody to hide an LDS round trip behind, and left to itself the compiler issues these reads one by one,
    // each followed by s_waitcnt lgkmcnt(0) and its first use (eight exposed round trips per Gauss-point group)
    struct Pre { double gr, e2r, nir, bg0, bg1, bg2, pzr, jmof, jn0, jn1, ox[6]; };
    __device__ __forceinline__ Pre load(const double* im) const {
        Pre q;
        q.gr = im[IM_G + rt]; q.e2r = im[oE2]; q.nir = im[IM_N + ir];
        q.bg0 = im[IM_BG + rt]; q.bg1 = im[IM_BG + 6 + rt]; q.bg2 = im[IM_BG + 12 + rt];
        q.pzr = im[IM_PZ + r]; q.jmof = im[IM_JMOF + kr]; q.jn0 = im[oJ0]; q.jn1 = im[oJ1];
#pragma unroll
        for (int s = 0; s < 6; ++s) q.ox[s] = im[oX[s]];
        return q;
    }
    // row r of G (gR) and of Hc (hR, WITHC) at the Gauss point with record im; entry (m', j) at [3 m' + j]
    template <bool WITHC> __device__ __forceinline__ void expand(const double* im, double (&gR)[15], double (&hR)[15]) const {
        const Pre q = load(im);
        expand<WITHC>(im, q, gR, hR);
    }
    template <bool WITHC> __device__ __forceinline__ void expand(const double* im, const Pre& q, double (&gR)[15], double (&hR)[15]) const {
        const double gr = q.gr, e0 = m0 * gr, e1 = m1 * gr, e2 = mt * q.e2r;
        const double fnr = f3c * q.nir;
        const double b0 = mt * q.bg0 + ck[0] * fnr, b1 = mt * q.bg1 + ck[1] * fnr, b2 = mt * q.bg2 + ck[2] * fnr;
        const double pzr = q.pzr, xfac = mt + (1.0 - mt) * q.jmof;
        const double jn[2] = {q.jn0, q.jn1};
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const double g = e0 * im[IM_CEZ + s] + e1 * im[IM_CEZ + 6 + s] + e2 * im[IM_CEZ + 12 + s]
                           + b0 * im[IM_CBG + s] + b1 * im[IM_CBG + 6 + s] + b2 * im[IM_CBG + 12 + s] - xfac * q.ox[s] + dij[s % 3] * jn[s / 3];
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
    }
    // the same expansion in two steps, for the full-pass instance of gauss_group: the row of G first (what the K batches read), the row of Hc = G + PzZ
    // later (from the row of G and ten scalars), so that the second step can be issued between the K MFMAs
    struct Mid { double e0, e1, e2, b0, b1, b2, pzr, gam[3]; };
    __device__ __forceinline__ void expand_g(const double* im, const Pre& q, double (&gR)[15], Mid& M) const {
        const double gr = q.gr;
        M.e0 = m0 * gr; M.e1 = m1 * gr; M.e2 = mt * q.e2r;
        const double fnr = f3c * q.nir;
        M.b0 = mt * q.bg0 + ck[0] * fnr; M.b1 = mt * q.bg1 + ck[1] * fnr; M.b2 = mt * q.bg2 + ck[2] * fnr;
        M.pzr = q.pzr;
        const double xfac = mt + (1.0 - mt) * q.jmof;
        const double jn[2] = {q.jn0, q.jn1};
#pragma unroll
        for (int s = 0; s < 6; ++s)
            gR[s] = M.e0 * im[IM_CEZ + s] + M.e1 * im[IM_CEZ + 6 + s] + M.e2 * im[IM_CEZ + 12 + s]
                  + M.b0 * im[IM_CBG + s] + M.b1 * im[IM_CBG + 6 + s] + M.b2 * im[IM_CBG + 12 + s] - xfac * q.ox[s] + dij[s % 3] * jn[s / 3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double fc = (c == 2) ? 2.0 : 1.0;
            M.gam[c] = fc * (M.b0 * im[IM_CT3 + sym3(0, c)] + M.b1 * im[IM_CT3 + sym3(1, c)] + M.b2 * im[IM_CT3 + sym3(2, c)]);
            const double alpha = mt * (fc * im[IM_CBG + 6 * c + rt]) + (1.0 - mt) * M.gam[c], beta = mt * im[IM_JMOF + c];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) gR[6 + 3 * c + jj] = im[IM_N + jj] * alpha - beta * im[IM_DN + 6 * jj + rt];
        }
    }
    __device__ __forceinline__ void expand_h(const double* im, const Mid& M, const double (&gR)[15], double (&hR)[15]) const {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const double zz = M.pzr * im[IM_JZJ + s] + M.e0 * im[IM_JDNV + s] + M.e1 * im[IM_JDNV + 6 + s] + M.e2 * im[IM_JDNV + 12 + s]
                            - (M.b0 * im[IM_JDMO + s] + M.b1 * im[IM_JDMO + 6 + s] + M.b2 * im[IM_JDMO + 12 + s]);
            hR[s] = gR[s] + zz;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) hR[6 + 3 * c + jj] = gR[6 + 3 * c + jj] - M.gam[c] * im[IM_NB + jj];
    }
};

// ---- one group of 4 Gauss points: the lane's Gauss point is 4 grp + kk (record im, weight wq = 0 on a padded slot), its basis
//      function has the 1-D indices (ju, jv) in the element (bval = 0: the lane holds no basis function: zero row / column).
//      Adds to the MFMA accumulators K (6 tiles, i <= j), dR/dCP (9), dR/dh (3), the body-force tiles B (3) and the residual sums.
// diagnostic build (-DGF_STAMPS -DGF_STAMPS_FINE, tools/stamps.py): the sections of the group step in stamp slots 2 .. 6
#if defined(GF_STAMPS) && defined(GF_STAMPS_FINE)
#define GF_GROUP_STAMP_PARAMS , unsigned long long* stamp_acc, unsigned long long& tstamp
#define GF_GROUP_STAMP_ARGS , stamp_acc, tstamp
#define GF_GROUP_STAMP(slot) GF_STAMP(slot, tstamp)
#else
#define GF_GROUP_STAMP_PARAMS
#define GF_GROUP_STAMP_ARGS
#define GF_GROUP_STAMP(slot) do { } while (0)
#endif
// ALLF: the full pass (R + K + dR/dCP + dR/dh) as a compile-time fact -- the flags then fold away and the whole group step is ONE basic block, which lets the
// scheduler move the plain FP64 work (residual / dR/dh prefactors, issued behind the first K batch) in between the MFMAs, into the unused part of their issue slots
template <int P, bool WITHC, bool ALLF = false>
__device__ __forceinline__ void gauss_group(const RowLane& L, const double* im, double wq, const double* tu, const double* tv, int gu, int gv,
                                            int ju, int jv, double bval, bool doK_, bool doC_, bool doH_, bool has_bf, const double* pf, const double* ppd,
                                            gf_d4 (&accK)[6], gf_d4 (&accC)[9], gf_d4 (&accH)[3], gf_d4 (&accB)[3], double (&accR)[3] GF_GROUP_STAMP_PARAMS) {
    constexpr int P1 = P + 1;
    const bool doK = ALLF || doK_, doC = (ALLF && WITHC) || doC_, doH = ALLF || doH_;
    // -- every load the basis function and the row expansion start from, in one batch (RowLane::load)
    const double u0 = tu[(gu * 3 + 0) * P1 + ju], u1 = tu[(gu * 3 + 1) * P1 + ju], u2 = tu[(gu * 3 + 2) * P1 + ju];
    const double v0 = tv[(gv * 3 + 0) * P1 + jv], v1 = tv[(gv * 3 + 1) * P1 + jv], v2 = tv[(gv * 3 + 2) * P1 + jv];
    const double Wl[6] = {im[IM_W], im[IM_W + 1], im[IM_W + 2], im[IM_W + 3], im[IM_W + 4], im[IM_W + 5]};
    RowLane::Pre pre;
    if (doK || doC) pre = L.load(im);
    __builtin_amdgcn_sched_barrier(0);
    // -- basis function of the lane at this Gauss point (registers)
    double phi[5], R0, n0;
    {
        const double Nb[6] = {u0 * v0, u1 * v0, u0 * v1, u2 * v0, u0 * v2, u1 * v1};
        double R[6];
        rationalize6(Nb, Wl, R);
        for (int k = 0; k < 5; ++k) phi[k] = bval * R[k + 1];
        R0 = bval * R[0]; n0 = bval * Nb[0];
    }
    GF_GROUP_STAMP(2);
    // -- row r of G and Hc at this Gauss point; entry (m', j) at [3 m' + j]
    double gR[15], hR[15];
    for (int s = 0; s < 15; ++s) { gR[s] = 0.0; hR[s] = 0.0; }
    RowLane::Mid mid;
    if constexpr (ALLF) {
        L.expand_g(im, pre, gR, mid);
        dpp_source_fence(gR);
    } else if (doK || doC) {
        L.template expand<WITHC>(im, pre, gR, hR);
        dpp_source_fence(gR);
        if constexpr (WITHC) dpp_source_fence(hR);
    }
    GF_GROUP_STAMP(3);
    // -- residual and dR/dh prefactors of the lane's basis function at this Gauss point (ALLF: issued behind the first K batch, see above)
    double pb[5];
    for (int m = 0; m < 5; ++m) pb[m] = wq * phi[m];
    auto prefactors = [&]() {
    {
        // ALLF: no branch on the load (pf = 0 without one: the product vanishes)
        const double ls = ALLF ? load_scalar(im, ppd) : (has_bf ? load_scalar(im, ppd) : 0.0);
        for (int i = 0; i < 3; ++i) {
            double rz = 0.0;
            for (int m = 0; m < 5; ++m) rz += phi[m] * im[IM_PZ + 3 * m + i];
            accR[i] += wq * (rz - ls * pf[i] * R0);
        }
    }
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
    };
    if constexpr (!ALLF) prefactors();
    // -- contraction: one MFMA per (component, m); the B operand T_b is formed from the expanded row on the fly.
    // K component (i, j), m: T_b = w sum_m' G[(m,i),(m',j)] phi_b[m'] -- the five entries are gR[3 m' + j] of lane 3 m + i.
    // The B operands of all components of one m are formed as independent FMA chains before their MFMAs are issued (a single
    // chain -> MFMA -> chain sequence serialises on the shared FP64 pipe).
    constexpr int QI[6] = {0, 0, 0, 1, 1, 2}, QJ[6] = {0, 1, 2, 1, 2, 2};
    // curvature rows: psi_k = f_k sum_c f_c (J t^3 C)_kc w phi_b,2+c (f = 1, 1, 2) and the products of the normals, per lane
    double psi[3], nn[6], nnb[9];
    if (doK || doC) {
        const double ct[6] = {im[IM_CT3], im[IM_CT3 + 1], im[IM_CT3 + 2], im[IM_CT3 + 3], im[IM_CT3 + 4], im[IM_CT3 + 5]};
        const double nv[3] = {im[IM_N], im[IM_N + 1], im[IM_N + 2]};
        const double p4 = 2.0 * pb[4];
#pragma unroll
        for (int k = 0; k < 3; ++k) psi[k] = (k == 2 ? 2.0 : 1.0) * (ct[sym3(k, 0)] * pb[2] + ct[sym3(k, 1)] * pb[3] + ct[sym3(k, 2)] * p4);
#pragma unroll
        for (int q = 0; q < 6; ++q) nn[q] = nv[QI[q]] * nv[QJ[q]];
        if constexpr (WITHC) {
            const double dn[3] = {nv[0] - im[IM_NB], nv[1] - im[IM_NB + 1], nv[2] - im[IM_NB + 2]};
#pragma unroll
            for (int q = 0; q < 9; ++q) nnb[q] = nv[q / 3] * dn[q % 3];
        }
    }
    GF_GROUP_STAMP(4);
    if (doK) {
        static_for<5>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            double tq[6];
            static_for<6>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                if constexpr (m < 2) tq[q] = row_dot<3 * m + QI[q], QJ[q]>(gR, pb);
                else tq[q] = row_dot2<3 * m + QI[q], QJ[q]>(nn[q] * psi[m - 2], gR, pb);
            });
            mfma_hazard_gap(tq);
#pragma unroll
            for (int q = 0; q < 6; ++q) accK[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[m], tq[q], accK[q], 0, 0, 0);
            if constexpr (ALLF && m == 0) prefactors();
            if constexpr (ALLF && WITHC && m == 1) { L.expand_h(im, mid, gR, hR); dpp_source_fence(hR); }
        });
    }
    GF_GROUP_STAMP(5);
    if (doC) {
        static_for<5>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            double tq[9];
            static_for<9>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                if constexpr (m < 2) tq[q] = row_dot<3 * m + q / 3, q % 3>(hR, pb);
                else tq[q] = row_dot2<3 * m + q / 3, q % 3>(nnb[q] * psi[m - 2], hR, pb);
            });
            mfma_hazard_gap(tq);
#pragma unroll
            for (int q = 0; q < 9; ++q) accC[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[m], tq[q], accC[q], 0, 0, 0);
        });
        if (has_bf) {                    // d(-f . u dA)/dc : -w f_i R_a (dJ/dZ . phi_b): one tile per f, the factor -f_i is applied behind the loop
            const LoadGeom lg = load_geom(im, ppd);
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const double jz = load_dz_dot(im, ppd, lg, f, pb[0], pb[1]);
                accB[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(R0, jz, accB[f], 0, 0, 0);
            }
        }
    }
    GF_GROUP_STAMP(6);
}

}  // namespace gf
