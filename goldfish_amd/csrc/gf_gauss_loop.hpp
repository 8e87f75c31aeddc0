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
// Round 5, p = 3 walking kernel (template argument SF of gauss_group): the ROW side of that product sum-factorised -- per u index on v_mfma_f64_4x4x4, the v
// direction by twelve FMAs per component (see SfLane below); the column side (T formation) is the one described here in every instance.
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
// The same dot products with their START tied to four values d (inline assembly that names them as inputs and does not read them): the chain cannot be scheduled
// in front of whatever produces d.  The SF contraction uses it to put the T formation of component q + 1 between the 4 x 4 x 4 products of component q and the FMAs
// that read their results -- the products' result latency is then covered by useful work instead of wait states (mfma4_result_gap).
struct Dep4 { double a, b, c, d; };
template <int LANE, int J> __device__ __forceinline__ double row_dot_dep(const double (&g)[15], const double (&p)[5], const Dep4& D) {
    double t;
    asm("v_mov_b64 %0, 0" : "=v"(t) : "v"(D.a), "v"(D.b), "v"(D.c), "v"(D.d));
    fmac_bcast<LANE>(t, g[J], p[0]);
    fmac_bcast<LANE>(t, g[3 + J], p[1]);
    fmac_bcast<LANE>(t, g[6 + J], p[2]);
    fmac_bcast<LANE>(t, g[9 + J], p[3]);
    fmac_bcast<LANE>(t, g[12 + J], p[4]);
    return t;
}
template <int LANE, int J> __device__ __forceinline__ double row_dot2_dep(double a, double b, const double (&g)[15], const double (&p)[5], const Dep4& D) {
    double t;
    asm("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b), "v"(D.a), "v"(D.b), "v"(D.c), "v"(D.d));
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
__device__ __forceinline__ void mfma_hazard_gap(double (&t)[5]) {
    asm volatile("s_nop 1" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]));
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
#ifdef GF_STAMPS_SFDBG       // diagnostic of the SF contraction: slots 2 / 3 / 4 = T formation / products / accumulation of the dR/dCP components, the front part of the group in slot 7
#define GF_GROUP_STAMP(slot) GF_STAMP(((slot) >= 2 && (slot) <= 4) ? 7 : (slot), tstamp)
#define GF_SFDBG_STAMP(slot) GF_STAMP(slot, tstamp)
#else
#define GF_GROUP_STAMP(slot) GF_STAMP(slot, tstamp)
#define GF_SFDBG_STAMP(slot) do { } while (0)
#endif
#else
#define GF_GROUP_STAMP_PARAMS
#define GF_GROUP_STAMP_ARGS
#define GF_GROUP_STAMP(slot) do { } while (0)
#define GF_SFDBG_STAMP(slot) do { } while (0)
#endif
// ---- row-side sum factorisation (SF; round 5, p = 3 walking kernel).  A group of the bicubic element is the four Gauss points g1 = kk of ONE v index g2 = grp,
//      and the row-side basis function factorises, phi_a[m](g1, g2) = sum_k2 F_{a1}[m][k2](g1, g2) B^(k2)_{a2}(g2)  (F: the part of rationalize6 that multiplies
//      the k2-th v derivative; B: the v table).  So the contraction over a group needs the 16 x 16 x 4 product  phi_a[m] x T_b[m]  only for the FOUR u indices a1:
//          X[k2][a1][b]    = sum_g1 sum_m F_{a1}[m][k2] T_b[m]     v_mfma_f64_4x4x4: A operand F at lane (a1 = lane & 3, g1 = lane >> 4), B operand T_b of the lane
//                                                                  (b = lane & 15, g1 = lane >> 4), result at lane b + 16 a1 -- 16.5 cycles instead of 64
//          K[(a1,a2)][b]  += sum_k2 B^(k2)_{a2}(g2) X[k2][a1][b]   12 FMAs with wave-uniform factors (the accumulator register is the SLOT of the row a2, gf_element_rec.hpp)
//      A polynomial patch (all weights equal, SF = 1) has one k2 per m (5 MFMAs per component), a rational one (SF = 2) nine (m, k2) pairs:
//      82 / 148 + 48 cycles on the FP64 pipe per component and group instead of 320.  The T formation, the expansions and the residual are unchanged.
// a value every lane holds, moved to scalar registers (an FMA takes one scalar operand: the twelve factors of a group cost no vector registers)
__device__ __forceinline__ double uniform_double(double v) {
    const gf_u2 w = __builtin_bit_cast(gf_u2, v);
    return __builtin_bit_cast(double, gf_u2{(unsigned)__builtin_amdgcn_readfirstlane((int)w.x), (unsigned)__builtin_amdgcn_readfirstlane((int)w.y)});
}
// A double parked in two accumulation registers: the asm statements that define and read it constrain both halves to AGPRs, so the value never occupies
// arch VGPRs between its uses (the walking accumulators of the SF instances: touched once per element).
struct AccReg { int lo, hi; };
__device__ __forceinline__ void acc_init(AccReg& a) { asm("v_accvgpr_write_b32 %0, 0" : "=a"(a.lo)); asm("v_accvgpr_write_b32 %0, 0" : "=a"(a.hi)); }
__device__ __forceinline__ void acc_zero(AccReg& a) { asm("v_accvgpr_write_b32 %0, 0" : "+a"(a.lo)); asm("v_accvgpr_write_b32 %0, 0" : "+a"(a.hi)); }
__device__ __forceinline__ double acc_get(const AccReg& a) {
    int lo, hi;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(a.lo)); asm("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(a.hi));
    return __builtin_bit_cast(double, gf_u2{(unsigned)lo, (unsigned)hi});
}
__device__ __forceinline__ void acc_put(AccReg& a, double v) {
    const gf_u2 w = __builtin_bit_cast(gf_u2, v);
    // "+a": the new value takes the register of the old one (a fresh "=a" value is copied back into the loop-carried register: one v_accvgpr_mov each)
    asm("v_accvgpr_write_b32 %0, %1" : "+a"(a.lo) : "v"((int)w.x)); asm("v_accvgpr_write_b32 %0, %1" : "+a"(a.hi) : "v"((int)w.y));
}
struct SfLane { double au[3]; int rot; };            // A^(k1)_{a1 = lane & 3}(g1 = lane >> 4) of the strip's u table; first control-point row of the element mod 4
template <int SF> __host__ __device__ constexpr bool sf_nz(int m, int k2) {
    return SF == 1 ? k2 == ((m == 0 || m == 2) ? 0 : (m == 3 ? 2 : 1)) : (k2 == 0 || (k2 == 1 && (m == 1 || m == 3 || m == 4)) || (k2 == 2 && m == 3));
}
// v_mfma_f64_4x4x4 with EVERY operand in arch VGPRs.  The builtin lets the compiler choose, and in a function that may use AGPRs it puts the results there: two
// v_accvgpr_read per product before the FMAs can use them, and the AGPRs are where the walking accumulators of the SF instances live (AccReg).  As inline
// assembly the instruction is opaque to the hazard recogniser, so the two software interlocks are spelled out: operands written by VALU need two wait states
// (mfma_hazard_gap, as for the 16 x 16 x 4 form), and a result needs the instruction's four passes + write-back before a VALU reads it (mfma4_result_gap: ten
// wait states against the six LLVM's table lists for the 4 x 4 x 4 DGEMM form).  Dependent products (the same X) are kept two products apart.
// (volatile: the products and the gaps keep the order they are written in -- the spacing of dependent products is a property of the source, not of the scheduler's mood)
__device__ __forceinline__ double mfma4_first(double a, double b) { double d; asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ void mfma4_acc(double& x, double a, double b) { asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma4_result_gap(double& a, double& b, double& c) { asm volatile("s_nop 7\n\ts_nop 1" : "+v"(a), "+v"(b), "+v"(c)); }
__device__ __forceinline__ void mfma4_result_gap(double& a, double& b, double& c, double& d) { asm volatile("s_nop 7\n\ts_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void mfma4_result_gap(double& x) { asm volatile("s_nop 7\n\ts_nop 1" : "+v"(x)); }
__device__ __forceinline__ void mfma_hazard_gap(double& a, double& b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }
// the operand gap of component q + 1 that also holds back the readers of component q's products (Dep4 above): everything that reads X comes behind it
__device__ __forceinline__ void mfma_pipeline_gap(double (&t)[5], double (&x)[4]) {
    asm volatile("s_nop 1" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
}

// ALLF: the full pass (R + K + dR/dCP + dR/dh) as a compile-time fact -- the flags then fold away and the whole group step is ONE basic block, which lets the
// scheduler move the plain FP64 work (residual / dR/dh prefactors, issued behind the first K batch) in between the MFMAs, into the unused part of their issue slots
// SF: acc[sl] += b0[sl] x0 + b1[sl] x1 + b2[sl] x2 for the four row slots -- on accumulators in arch VGPRs (gf_d4) or parked in AGPRs (AccReg[4]: read, FMAs, write back)
__device__ __forceinline__ void sf_accumulate(gf_d4& acc, const double (&bs)[3][4], double x0, double x1, double x2) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) acc[sl] = __builtin_fma(bs[2][sl], x2, __builtin_fma(bs[1][sl], x1, __builtin_fma(bs[0][sl], x0, acc[sl])));
}
__device__ __forceinline__ void sf_accumulate(AccReg (&acc)[4], const double (&bs)[3][4], double x0, double x1, double x2) {
    // the four slots side by side (four independent chains of three FMAs): slot by slot, every instruction would wait for the one in front of it
    double v[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) v[sl] = acc_get(acc[sl]);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) v[sl] = __builtin_fma(bs[0][sl], x0, v[sl]);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) v[sl] = __builtin_fma(bs[1][sl], x1, v[sl]);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) v[sl] = __builtin_fma(bs[2][sl], x2, v[sl]);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) acc_put(acc[sl], v[sl]);
}
__device__ __forceinline__ void sf_accumulate1(gf_d4& acc, const double (&b)[4], double x) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) acc[sl] = __builtin_fma(b[sl], x, acc[sl]);
}
__device__ __forceinline__ void sf_accumulate1(AccReg (&acc)[4], const double (&b)[4], double x) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) acc_put(acc[sl], __builtin_fma(b[sl], x, acc_get(acc[sl])));
}

// accumulators: gf_d4 arrays (SF = 0: MFMA destinations, register = u index of the row function; SF != 0: register = row slot), or AccReg[.][4] (SF != 0, parked in AGPRs)
template <int P, bool WITHC, bool ALLF = false, int SF = 0, class AK, class AC>
__device__ __forceinline__ void gauss_group(const RowLane& L, const double* im, double wq, const double* tu, const double* tv, int gu, int gv,
                                            int ju, int jv, double bval, bool doK_, bool doC_, bool doH_, bool has_bf, const double* pf, const double* ppd,
                                            AK& accK, AC& accC, gf_d4 (&accH)[3], gf_d4 (&accB)[3], double (&accR)[3], const SfLane& sf GF_GROUP_STAMP_PARAMS) {
    constexpr int P1 = P + 1;
    static_assert(SF == 0 || P == 3, "row-side sum factorisation: the groups of a bicubic element are its four v indices");
    const bool doK = ALLF || doK_, doC = (ALLF && WITHC) || doC_, doH = ALLF || doH_;
    // -- every load the basis function and the row expansion start from, in one batch (RowLane::load)
    const double u0 = tu[(gu * 3 + 0) * P1 + ju], u1 = tu[(gu * 3 + 1) * P1 + ju], u2 = tu[(gu * 3 + 2) * P1 + ju];
    const double v0 = tv[(gv * 3 + 0) * P1 + jv], v1 = tv[(gv * 3 + 1) * P1 + jv], v2 = tv[(gv * 3 + 2) * P1 + jv];
    const double Wl[6] = {im[IM_W], im[IM_W + 1], im[IM_W + 2], im[IM_W + 3], im[IM_W + 4], im[IM_W + 5]};
    RowLane::Pre pre;
    if (doK || doC) pre = L.load(im);
    double bs[3][4];                                     // SF: B^(k2)_{a2}(g2) of the row slots s (a2 = (s - first row) mod 4): the same for every lane
    if constexpr (SF != 0) {
#pragma unroll
        for (int k2 = 0; k2 < 3; ++k2)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) bs[k2][sl] = uniform_double(tv[(gv * 3 + k2) * P1 + ((sl - sf.rot) & 3)]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // SF: the row-side factors F[m][k2] of the lane's (a1, g1) at this group's g2 and the value kind Fv (k2 = 0 only)
    double F[5][3], Fv = 0.0;
    if constexpr (SF != 0) {
        for (int m = 0; m < 5; ++m) for (int k2 = 0; k2 < 3; ++k2) F[m][k2] = 0.0;
        const double iW = Wl[0];
        if constexpr (SF == 1) {
            const double f0 = iW * sf.au[0], f1 = iW * sf.au[1], f2 = iW * sf.au[2];
            F[0][0] = f1; F[1][1] = f0; F[2][0] = f2; F[3][2] = f0; F[4][1] = f1; Fv = f0;
        } else {                                         // rationalize6 applied to the products with v0 = 1 / v1 = 1 / v2 = 1 alone
            const double r0 = sf.au[0] * iW, r1 = (sf.au[1] - r0 * Wl[1]) * iW, r2 = -r0 * Wl[2] * iW;
            F[0][0] = r1; F[1][0] = r2; F[2][0] = (sf.au[2] - 2 * r1 * Wl[1] - r0 * Wl[3]) * iW;
            F[3][0] = (-2 * r2 * Wl[2] - r0 * Wl[4]) * iW; F[4][0] = (-r1 * Wl[2] - r2 * Wl[1] - r0 * Wl[5]) * iW; Fv = r0;
            F[1][1] = r0; F[3][1] = -2 * r0 * Wl[2] * iW; F[4][1] = (sf.au[1] - r0 * Wl[1]) * iW;
            F[3][2] = r0;
        }
    }
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
    // dR/dh: a rank-one tile per component -- ONE 16 x 16 x 4 product in every instance (the SF instances would need three 4 x 4 x 4 products, their operands, two
    // interlocks and the parked accumulators' moves: measured 2.0 k against 0.85 k cycles per group).  Its accumulators keep the layout of that product
    // (lane / 16 = slot of the row function's row, register = its u index): the walking kernel flushes the three dR/dh tiles with the SF = 0 addressing.
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
    double psi[3], nn[6], nnb[9], nv[3] = {0.0, 0.0, 0.0}, dn[3] = {0.0, 0.0, 0.0};
    if (doK || doC) {
        const double ct[6] = {im[IM_CT3], im[IM_CT3 + 1], im[IM_CT3 + 2], im[IM_CT3 + 3], im[IM_CT3 + 4], im[IM_CT3 + 5]};
        for (int k = 0; k < 3; ++k) nv[k] = im[IM_N + k];
        const double p4 = 2.0 * pb[4];
#pragma unroll
        for (int k = 0; k < 3; ++k) psi[k] = (k == 2 ? 2.0 : 1.0) * (ct[sym3(k, 0)] * pb[2] + ct[sym3(k, 1)] * pb[3] + ct[sym3(k, 2)] * p4);
        if constexpr (WITHC) for (int k = 0; k < 3; ++k) dn[k] = nv[k] - im[IM_NB + k];
        if constexpr (SF == 0) {
#pragma unroll
            for (int q = 0; q < 6; ++q) nn[q] = nv[QI[q]] * nv[QJ[q]];
            if constexpr (WITHC) {
#pragma unroll
                for (int q = 0; q < 9; ++q) nnb[q] = nv[q / 3] * dn[q % 3];
            }
        }
    }
    GF_GROUP_STAMP(4);
    if constexpr (SF != 0) {
        // Row-side sum factorisation (see SfLane): COMPONENT by component -- the five B operands of a component, its five (nine) 4 x 4 x 4 products into X[k2],
        // the twelve FMAs into the accumulators.  The accumulators are FMA destinations here (arch VGPRs, 144 of the 256), so what a component keeps live is
        // kept small: X is three registers, the products of the normals are formed per component.
        // order of the products: consecutive ones write different X (a dependent pair would wait for the first one's four passes)
        // order of the products: two products into the same partial sum are at least two products apart (a dependent product reads its accumulator while the
        // previous one is still in its four passes; nothing interlocks inline assembly).  The rational instance splits its five k2 = 0 products over X[0] and X[3].
        constexpr int NPR = SF == 1 ? 5 : 9, NX = SF == 1 ? 3 : 4;
        constexpr int PM[9] = {0, 1, 3, SF == 1 ? 2 : 3, SF == 1 ? 4 : 1, 3, 4, 2, 4}, PK[9] = {0, 1, 2, 0, SF == 1 ? 1 : 0, 1, 0, 0, 1}, PX[9] = {0, 1, 2, SF == 1 ? 0 : 3, SF == 1 ? 1 : 0, 1, 3, 0, 1};
        // (m, k2) -> X:  SF = 1: (0,0)->0 (1,1)->1 (3,2)->2 (2,0)->0 (4,1)->1;   SF = 2: (0,0)->0 (1,1)->1 (3,2)->2 (3,0)->3 (1,0)->0 (3,1)->1 (4,0)->3 (2,0)->0 (4,1)->1
        // tb: the B operand of the body-force term of a dR/dCP component (its row side is the VALUE kind: Fv, k2 = 0) -- one more product, into X[3]
        auto products = [&](const double (&tq)[5], double (&X)[4], auto withb_, double tb) {
            constexpr bool WITHB = decltype(withb_)::value;
            if constexpr (NX == 3 && !WITHB) X[3] = 0.0;
            if constexpr (NX == 3 && WITHB) X[3] = mfma4_first(Fv, tb);
            static_for<NPR>([&](auto n_) {
                constexpr int n = decltype(n_)::value, m = PM[n], k2 = PK[n], xi = PX[n];
                static_assert(sf_nz<SF>(m, k2) && (xi == k2 || (xi == 3 && k2 == 0)), "product list of the row-side factors");
                if constexpr (n < NX) X[xi] = mfma4_first(F[m][k2], tq[m]);      // the list starts with one product per partial sum
                else mfma4_acc(X[xi], F[m][k2], tq[m]);
            });
            if constexpr (NX == 4 && WITHB) mfma4_acc(X[3], Fv, tb);          // (the previous product into X[3] lies three products back)
        };
        auto settle = [&](double (&X)[4], auto& acc, auto withb_) {
            constexpr bool WITHB = decltype(withb_)::value;
            if constexpr (NX == 4 || WITHB) X[0] += X[3];
            sf_accumulate(acc, bs, X[0], X[1], X[2]);
        };
        constexpr std::false_type NOB{}; constexpr std::true_type WB{};
        // software pipeline over the components of a stage: T formation of q (tied behind the products of q - 1), ONE gap, the FMAs of q - 1, the products of q
        if (doK) {
            double Xp[4] = {0.0, 0.0, 0.0, 0.0};
            static_for<6>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                const double nq = nv[QI[q]] * nv[QJ[q]];
                const Dep4 D = {Xp[0], Xp[1], Xp[2], Xp[3]};
                double tq[5];
                static_for<5>([&](auto m_) {
                    constexpr int m = decltype(m_)::value;
                    if constexpr (q == 0) {
                        if constexpr (m < 2) tq[m] = row_dot<3 * m + QI[q], QJ[q]>(gR, pb);
                        else tq[m] = row_dot2<3 * m + QI[q], QJ[q]>(nq * psi[m - 2], gR, pb);
                    } else {
                        if constexpr (m < 2) tq[m] = row_dot_dep<3 * m + QI[q], QJ[q]>(gR, pb, D);
                        else tq[m] = row_dot2_dep<3 * m + QI[q], QJ[q]>(nq, psi[m - 2], gR, pb, D);
                    }
                });
                if constexpr (q == 0) mfma_hazard_gap(tq);
                else { mfma_pipeline_gap(tq, Xp); settle(Xp, accK[q - 1], NOB); }
                products(tq, Xp, NOB, 0.0);
                if constexpr (ALLF && q == 0) prefactors();
                if constexpr (ALLF && WITHC && q == 2) { L.expand_h(im, mid, gR, hR); dpp_source_fence(hR); }
            });
            mfma4_result_gap(Xp[0], Xp[1], Xp[2], Xp[3]);
            settle(Xp, accK[5], NOB);
        }
        GF_GROUP_STAMP(5);
        if (doC) {
            // body force, d(-f . u dA)/dc: -w f_i R_a (dJ/dZ . phi_b)_f with R_a = Fv B^(0)_{a2} -- one more product per (i, f) component (no tiles of its own)
            double jzf[3];
            const double pfu[3] = {uniform_double(pf[0]), uniform_double(pf[1]), uniform_double(pf[2])};
            {
                LoadGeom lg = {1.0, 0.0, 0.0};
                if (has_bf) lg = load_geom(im, ppd);
#pragma unroll
                for (int f = 0; f < 3; ++f) jzf[f] = has_bf ? -load_dz_dot(im, ppd, lg, f, pb[0], pb[1]) : 0.0;
            }
            double Xp[4] = {0.0, 0.0, 0.0, 0.0};
            static_for<9>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                const double nq = nv[q / 3] * dn[q % 3];
                const Dep4 D = {Xp[0], Xp[1], Xp[2], Xp[3]};
                double tq[5];
                static_for<5>([&](auto m_) {
                    constexpr int m = decltype(m_)::value;
                    if constexpr (q == 0) {
                        if constexpr (m < 2) tq[m] = row_dot<3 * m + q / 3, q % 3>(hR, pb);
                        else tq[m] = row_dot2<3 * m + q / 3, q % 3>(nq * psi[m - 2], hR, pb);
                    } else {
                        if constexpr (m < 2) tq[m] = row_dot_dep<3 * m + q / 3, q % 3>(hR, pb, D);
                        else tq[m] = row_dot2_dep<3 * m + q / 3, q % 3>(nq, psi[m - 2], hR, pb, D);
                    }
                });
                double tbq = pfu[q / 3] * jzf[q % 3];
                if constexpr (q == 0) asm volatile("s_nop 1" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]), "+v"(tq[4]), "+v"(tbq));
                else { asm volatile("s_nop 1" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]), "+v"(tq[4]), "+v"(tbq), "+v"(Xp[0]), "+v"(Xp[1]), "+v"(Xp[2]), "+v"(Xp[3])); settle(Xp, accC[q - 1], WB); }
                products(tq, Xp, WB, tbq);
            });
            mfma4_result_gap(Xp[0], Xp[1], Xp[2], Xp[3]);
            settle(Xp, accC[8], WB);
        }
        GF_GROUP_STAMP(6);
    } else {
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
}

}  // namespace gf
