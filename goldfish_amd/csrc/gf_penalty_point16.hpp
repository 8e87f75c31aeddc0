// gf_penalty_point16.hpp -- vertex records of the penalty coupling (p = 2, 3), 16 lanes per mortar vertex.
//
// pen_point_kernel gives a vertex to one thread: the two 12 x 12 Hessians of the rotation measures live in its registers (512 VGPRs +
// 2.7 KB of scratch, one wave per SIMD) and it writes its 4.5 KB record alone (64 lanes, 64 different lines per store).  Here a
// 16-lane row takes the vertex (four vertices per wave):
//   A  kinematics: lane c sums control point c of side A, then of side B, into its 30 partial sums of y (u, g1, g2 per side) and
//      Y (G1, G2 per side); the row adds them up through LDS;
//   B  the row computes the SMALL per-vertex quantities of penalty_point by role (lanes 0-5: one tangent-slot column each of the normal /
//      tangent derivatives, gradients g1, g2 and cross-product tables; lanes 6-9: the cores Q B, v of the four Hessians of M . n; lanes
//      10-15: the columns of the reference configuration) into a table of 315 doubles in LDS;
//   C1 the 12 x 12 tangent block W = c0 ar (g1 g1^T + e1 H1 + g2 g2^T + e2 H2) block by block (AA, AB = BA^T, BB: the same closed form for
//      every lane of a pass, one 3-term dot product per Hessian term) into LDS;
//   C2 every entry of the record is a lookup in W plus rank-one terms: lane c writes the entries c, c + 16, ... -- consecutive lanes on
//      consecutive addresses.
// Same formulas as penalty_point (kl_point.hpp: s_terms, hess_M_dot_n, hess_M_dot_t), entry by entry; same record layout.
// Default for p = 2, 3 (GF_PEN_POINT16=0: pen_point_kernel).  8 x 8-patch slice: 97 us against pen_point_kernel's 155 us; the first version,
// with the current configuration's table on one lane (~250 registers, one wave per SIMD), took 196 us.
#pragma once
#include "gf_gauss_loop.hpp"

namespace gf {

enum : int { PT_BCA = 0, PT_BCB = 18, PT_QB = 36, PT_V = 108, PT_QT = 120, PT_DNA = 129, PT_DNB = 147, PT_DT = 165, PT_SBA = 183, PT_SAB = 201, PT_STB = 219,
             PT_G1 = 237, PT_G2 = 249, PT_GR1 = 261, PT_GR2 = 273, PT_GL = 285, PT_SC = 303, PT_SIZE = 320 };
// scalars at PT_SC: 0 e1, 1 e2, 2 c0, 3 c0 * ar, 4 c0 * ad, 5 dt, 6 tau0, 7 tau1, 8..10 At, 11 energy

// core of the Hessian of M . n(g1, g2): Bc[c] = d(g1 x g2)/d(.)_c, QB[c][a] = sum_b Q[a][b] Bc[c][b], v (hess_M_dot_n)
__device__ inline void hess_core(const double* g1, const double* g2, const double* n, double j, const double* Mv, double* QB, double* v) {
    const double Mn = dot3(Mv, n), ij2 = 1.0 / (j * j);
    double Q[3][3];
    for (int a = 0; a < 3; ++a) {
        v[a] = (Mv[a] - Mn * n[a]) / j;
        for (int b = 0; b < 3; ++b) Q[a][b] = -(Mv[a] * n[b] + n[a] * Mv[b] + Mn * ((a == b ? 1.0 : 0.0) - 3.0 * n[a] * n[b])) * ij2;
    }
    for (int c = 0; c < 6; ++c) {
        double e[3] = {0, 0, 0}, Bc[3]; e[c % 3] = 1.0;
        if (c < 3) cross3(e, g2, Bc); else cross3(g1, e, Bc);
        for (int a = 0; a < 3; ++a) QB[3 * c + a] = Q[a][0] * Bc[0] + Q[a][1] * Bc[1] + Q[a][2] * Bc[2];
    }
}
__device__ inline void store_dn(double* T, const double Dn[3][6]) { for (int i = 0; i < 3; ++i) for (int c = 0; c < 6; ++c) T[6 * i + c] = Dn[i][c]; }

// phase B: the per-vertex table, spread over the 16 lanes of the vertex's row.  role 0..5: column c = role of the current configuration
// (d/d(g1, g2)_c of the normals and the tangent, everything that is indexed by one tangent slot of a side); role 6..9: the cores of the
// four Hessians of M . n; role 10..15: column role - 10 of the reference configuration (values and gradients only).  Every lane
// recomputes the normals and the tangent it needs (a few dozen flops) instead of holding whole derivative arrays.
__device__ __forceinline__ void unit_normal(const double* g1, const double* g2, double* n, double& j) {
    double t[3]; cross3(g1, g2, t); j = sqrt(dot3(t, t));
    for (int k = 0; k < 3; ++k) n[k] = t[k] / j;
}
// column c of Dn = (I - n n^T) / j * B and the B column itself
__device__ __forceinline__ void dn_column(const double* g1, const double* g2, const double* n, double j, int c, double* bc, double* dn) {
    double e[3] = {0, 0, 0}; e[c % 3] = 1.0;
    if (c < 3) cross3(e, g2, bc); else cross3(g1, e, bc);
    const double nc = dot3(n, bc);
    for (int i = 0; i < 3; ++i) dn[i] = (bc[i] - n[i] * nc) / j;
}
// one tangent-slot column of a configuration (gA = (g1, g2) of side A, gB of side B): the entries c and 6 + c of the gradients of
// s1 = nA . nB and s2 = at . (nA x nB); with T != nullptr also the column's part of the tables and (c == 0) the scalars
__device__ inline void config_column(const double* gA, const double* gB, const double* tau, int c, double* T, double& g1a, double& g1b, double& g2a, double& g2b,
                                     double& s1, double& s2, double& L, double* at) {
    double nA[3], nB[3], jA, jB, tt[3];
    unit_normal(gA, gA + 3, nA, jA); unit_normal(gB, gB + 3, nB, jB);
    for (int k = 0; k < 3; ++k) tt[k] = tau[0] * gA[k] + tau[1] * gA[3 + k];
    L = sqrt(dot3(tt, tt));
    for (int k = 0; k < 3; ++k) at[k] = tt[k] / L;
    double cAB[3], cBt[3], ctA[3];
    cross3(nA, nB, cAB); cross3(nB, at, cBt); cross3(at, nA, ctA);
    s1 = dot3(nA, nB); s2 = dot3(at, cAB);
    double bcA[3], bcB[3], dnA[3], dnB[3], dt[3];
    dn_column(gA, gA + 3, nA, jA, c, bcA, dnA);
    dn_column(gB, gB + 3, nB, jB, c, bcB, dnB);
    for (int i = 0; i < 3; ++i) dt[i] = tau[c / 3] * ((i == c % 3 ? 1.0 : 0.0) - at[i] * at[c % 3]) / L;
    g1a = dot3(dnA, nB); g1b = dot3(dnB, nA);
    g2a = dot3(dt, cAB) + dot3(dnA, cBt); g2b = dot3(dnB, ctA);
    if (!T) return;
    for (int k = 0; k < 3; ++k) { T[PT_BCA + 3 * c + k] = bcA[k]; T[PT_BCB + 3 * c + k] = bcB[k]; }
    double t0[3], t1[3], t2[3];
    cross3(nB, dnA, t0); cross3(nA, dnB, t1); cross3(at, dnB, t2);
    for (int i = 0; i < 3; ++i) {
        T[PT_DNA + 6 * i + c] = dnA[i]; T[PT_DNB + 6 * i + c] = dnB[i]; T[PT_DT + 6 * i + c] = dt[i];
        T[PT_SBA + 6 * i + c] = t0[i]; T[PT_SAB + 6 * i + c] = t1[i]; T[PT_STB + 6 * i + c] = t2[i];
    }
    if (c == 0) {   // hess_M_dot_t with M = cAB: Ht[r][c] = tau[r / 3] tau[c / 3] Qt[r % 3][c % 3]
        const double Ma = dot3(cAB, at), iL2 = 1.0 / (L * L);
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
            T[PT_QT + 3 * a + b] = -(cAB[a] * at[b] + at[a] * cAB[b] + Ma * ((a == b ? 1.0 : 0.0) - 3.0 * at[a] * at[b])) * iL2;
    }
}
__device__ inline void penalty_tables(int role, const double* y, const double* Y, const double* tau, double ad, double ar, double dt, double* T) {
    const double* gA = y + 3; const double* gB = y + 12;
    if (role < 6) {
        double g1a, g1b, g2a, g2b, s1, s2, L, at[3];
        config_column(gA, gB, tau, role, T, g1a, g1b, g2a, g2b, s1, s2, L, at);
        T[PT_G1 + role] = g1a; T[PT_G1 + 6 + role] = g1b; T[PT_G2 + role] = g2a; T[PT_G2 + 6 + role] = g2b;
        if (role == 0) { T[PT_SC] = s1; T[PT_SC + 1] = s2; T[PT_SC + 5] = dt; T[PT_SC + 6] = tau[0]; T[PT_SC + 7] = tau[1]; }   // s1, s2: turned into e1, e2 by the combine step
    } else if (role < 10) {                                           // the four Hessians of M . n: (gA, nB), (gB, nA), (gA, nB x at), (gB, at x nA)
        const int hq = role - 6;
        double nA[3], nB[3], jA, jB;
        unit_normal(gA, gA + 3, nA, jA); unit_normal(gB, gB + 3, nB, jB);
        double Mv[3];
        if (hq == 0) for (int k = 0; k < 3; ++k) Mv[k] = nB[k];
        else if (hq == 1) for (int k = 0; k < 3; ++k) Mv[k] = nA[k];
        else {
            double tt[3], at[3];
            for (int k = 0; k < 3; ++k) tt[k] = tau[0] * gA[k] + tau[1] * gA[3 + k];
            const double L = sqrt(dot3(tt, tt));
            for (int k = 0; k < 3; ++k) at[k] = tt[k] / L;
            if (hq == 2) cross3(nB, at, Mv); else cross3(at, nA, Mv);
        }
        const bool sideA = hq == 0 || hq == 2;
        hess_core(sideA ? gA : gB, (sideA ? gA : gB) + 3, sideA ? nA : nB, sideA ? jA : jB, Mv, T + PT_QB + 18 * hq, T + PT_V + 3 * hq);
    } else {                                                          // reference configuration: values and gradients only
        const int c = role - 10;
        double g1a, g1b, g2a, g2b, S1, S2, Lr, At[3];
        config_column(Y, Y + 6, tau, c, nullptr, g1a, g1b, g2a, g2b, S1, S2, Lr, At);
        T[PT_GR1 + c] = g1a; T[PT_GR1 + 6 + c] = g1b; T[PT_GR2 + c] = g2a; T[PT_GR2 + 6 + c] = g2b;
        if (c == 0) {
            T[PT_SC + 8] = At[0]; T[PT_SC + 9] = At[1]; T[PT_SC + 10] = At[2];
            T[PT_SC + 2] = dt * Lr;                                   // c0
            T[PT_SC + 3] = S1; T[PT_SC + 4] = S2;                      // (replaced by c0 ar, c0 ad in the combine step)
        }
    }
}
// combine step (one lane, behind the six): e1, e2, the scaled constants, the gradient of the energy, the energy
__device__ inline void penalty_combine(const double* y, double ad, double ar, double* T) {
    double* sc = T + PT_SC;
    // e1 = s1 - S1, e2 = s2 - S2 evaluated from the displacement tangents (y + 30: dY; y + 18: Y) without the cancellation (kl_point.hpp: pen_rot_measures)
    double e1, e2;
    pen_rot_measures(y + 18, y + 30, sc + 6, e1, e2);
    const double c0 = sc[2];
    const double d[3] = {y[0] - y[9], y[1] - y[10], y[2] - y[11]};
    const int tan[12] = {3, 4, 5, 6, 7, 8, 12, 13, 14, 15, 16, 17};
    for (int k = 0; k < 18; ++k) T[PT_GL + k] = 0.0;
    for (int k = 0; k < 3; ++k) { T[PT_GL + k] = c0 * ad * d[k]; T[PT_GL + 9 + k] = -c0 * ad * d[k]; }
    for (int k = 0; k < 12; ++k) T[PT_GL + tan[k]] = c0 * ar * (e1 * T[PT_G1 + k] + e2 * T[PT_G2 + k]);
    sc[0] = e1; sc[1] = e2; sc[3] = c0 * ar; sc[4] = c0 * ad;
    sc[11] = c0 * (0.5 * ad * dot3(d, d) + 0.5 * ar * (e1 * e1 + e2 * e2));
}

// Hessian of M . n, entry (r, c): Bc[r] . QB[c] -+ skew(v) on the (g1, g2) / (g2, g1) blocks
__device__ __forceinline__ double hess_entry(const double* Bc, const double* QB, const double* v, int r, int c) {
    double h = Bc[3 * r] * QB[3 * c] + Bc[3 * r + 1] * QB[3 * c + 1] + Bc[3 * r + 2] * QB[3 * c + 2];
    const int a = r % 3, b = c % 3;
    if ((r < 3) != (c < 3) && a != b) {
        const double s = ((b - a + 3) % 3 == 1) ? -v[3 - a - b] : v[3 - a - b];          // skew(v)[a][b]
        // r < 3 <= c: H[a][3 + b] -= S[a][b];  r >= 3 > c: H[3 + a][b] += S[a][b]
        h += (r < 3) ? -s : s;
    }
    return h;
}
__device__ __forceinline__ double col_dot(const double* X, int r, const double* Yc, int c) { return X[r] * Yc[c] + X[6 + r] * Yc[6 + c] + X[12 + r] * Yc[12 + c]; }
// H1[tr][tc], H2[tr][tc] of the 12 tangent slots (A: 0..5, B: 6..11)
__device__ __forceinline__ void hpair(const double* T, int tr, int tc, double& h1, double& h2) {
    if (tr < 6 && tc < 6) {
        h1 = hess_entry(T + PT_BCA, T + PT_QB, T + PT_V, tr, tc);
        const double ht = T[PT_SC + 6 + tr / 3] * T[PT_SC + 6 + tc / 3] * T[PT_QT + 3 * (tr % 3) + tc % 3];
        const double X = -col_dot(T + PT_DT, tr, T + PT_SBA, tc), Xt = -col_dot(T + PT_DT, tc, T + PT_SBA, tr);
        h2 = ht + hess_entry(T + PT_BCA, T + PT_QB + 36, T + PT_V + 6, tr, tc) + X + Xt;
    } else if (tr >= 6 && tc >= 6) {
        h1 = hess_entry(T + PT_BCB, T + PT_QB + 18, T + PT_V + 3, tr - 6, tc - 6);
        h2 = hess_entry(T + PT_BCB, T + PT_QB + 54, T + PT_V + 9, tr - 6, tc - 6);
    } else {
        const int r = tr < 6 ? tr : tc, c = (tr < 6 ? tc : tr) - 6;                        // (A slot, B slot): the blocks are each other's transposes
        h1 = col_dot(T + PT_DNA, r, T + PT_DNB, c);
        h2 = col_dot(T + PT_DT, r, T + PT_SAB, c) - col_dot(T + PT_DNA, r, T + PT_STB, c);
    }
}
__device__ __forceinline__ int tslot_of(int r) { return r < 3 ? -1 : (r < 9 ? r - 3 : (r < 12 ? -1 : r - 6)); }
// Hyy[r][c] from the tangent block W (12 x 12) and the displacement block
__device__ __forceinline__ double hyy_lookup(const double* T, const double* W, int r, int c) {
    const int tr = tslot_of(r), tc = tslot_of(c);
    if (tr < 0 && tc < 0) return (r % 9 == c % 9) ? ((r == c) ? T[PT_SC + 4] : -T[PT_SC + 4]) : 0.0;      // displacement block: +- c0 alpha_d I
    if (tr < 0 || tc < 0) return 0.0;
    return W[12 * tr + tc];
}

template <int P>
__global__ __launch_bounds__(64) void pen_point16_kernel(DevModel M, DevPenalty Q, double* __restrict__ pbuf, int grad_only) {
    static_assert(P == 2 || P == 3, "the support window must fit a 16-lane row");
    constexpr int P1 = P + 1, NB = P1 * P1;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const long long v0 = (long long)blockIdx.x * 4 + g;
    const bool vok = v0 < Q.npts;
    const long long v = vok ? v0 : Q.npts - 1;
    // LDS: the partial sums of phase A and the tables of phases B / C share the space (the sums are consumed before the tables are written)
    __shared__ __attribute__((aligned(16))) double s_buf[4][PT_SIZE + 144 > 480 ? PT_SIZE + 144 : 480];
    __shared__ __attribute__((aligned(16))) double s_y[4][44];            // y (18) | Y (12) | dY (12: the displacement tangents)
    const int itf = Q.pt_iface[v];
    // ---- A: kinematics
    {
        double part[30];
#pragma unroll
        for (int k = 0; k < 30; ++k) part[k] = 0.0;
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            const PatchDev& Pt = M.patches[Q.if_patch[2 * itf + sd]];
            const int iu0 = Q.pt_base[4 * v + 2 * sd], iv0 = Q.pt_base[4 * v + 2 * sd + 1];
            if (c < NB) {
                const double* nu = Q.pt_nu + ((size_t)v * 2 + sd) * 3 * NB;
                const long long gcp = Pt.cp_off + (iu0 + c % P1) + (long long)(iv0 + c / P1) * Pt.nu;
                const double r0 = nu[c], r1 = nu[NB + c], r2 = nu[2 * NB + c];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double cc = M.cp4[4 * gcp + k], uu = M.u[3 * gcp + k];
                    part[9 * sd + k] = r0 * uu; part[9 * sd + 3 + k] = r1 * uu; part[9 * sd + 6 + k] = r2 * uu;       // displacement tangents for now
                    part[18 + 6 * sd + k] = r1 * cc; part[18 + 6 * sd + 3 + k] = r2 * cc;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 30; ++k) s_buf[g][16 * k + c] = part[k];
    }
    wave_lds_sync();
    for (int k = c; k < 30; k += 16) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += s_buf[g][16 * k + q];               // fixed order
        s_y[g][k] = s;
    }
    wave_lds_sync();
    if (c < 12) {                                                           // deformed tangents = reference + displacement tangents; the latter are kept (dY)
        const int ys = c < 6 ? 3 + c : 6 + c;
        const double dy = s_y[g][ys];
        s_y[g][30 + c] = dy; s_y[g][ys] = s_y[g][18 + c] + dy;
    }
    wave_lds_sync();
    // ---- B: the small quantities (all 16 lanes of the row, by role), then the combine step
    double* T = s_buf[g]; double* W = s_buf[g] + PT_SIZE;
    const double ad = Q.if_alpha[2 * itf], ar = Q.if_alpha[2 * itf + 1];
    penalty_tables(c, s_y[g], s_y[g] + 18, Q.pt_tau + 2 * v, ad, ar, Q.pt_wt[v], T);
    wave_lds_sync();
    if (c == 0) penalty_combine(s_y[g], ad, ar, T);
    wave_lds_sync();
    // ---- C: the record.  grad_only: 0 = gradient + both Hessian blocks, 1 = gradient only, 2 = gradient + Hyy, 3 = gradient + HyC
    double* out = pbuf + (size_t)v * PB_STRIDE;
    if (vok) {
        for (int o = c; o < 18; o += 16) out[PB_GRAD + o] = T[PT_GL + o];
        if (c == 0) out[PB_EN] = T[PT_SC + 11];
    }
    if (grad_only == 1) return;
    // C1: W = c0 ar (g1 g1^T + e1 H1 + g2 g2^T + e2 H2), block by block (the same closed form for every lane of a pass)
    {
        const double c0ar = T[PT_SC + 3], e1 = T[PT_SC], e2 = T[PT_SC + 1];
        for (int idx = c; idx < 36; idx += 16) {                              // AA
            const int r = idx / 6, cc = idx - 6 * r;
            const double h1 = hess_entry(T + PT_BCA, T + PT_QB, T + PT_V, r, cc);
            const double ht = T[PT_SC + 6 + r / 3] * T[PT_SC + 6 + cc / 3] * T[PT_QT + 3 * (r % 3) + cc % 3];
            const double h2 = ht + hess_entry(T + PT_BCA, T + PT_QB + 36, T + PT_V + 6, r, cc) - col_dot(T + PT_DT, r, T + PT_SBA, cc) - col_dot(T + PT_DT, cc, T + PT_SBA, r);
            W[12 * r + cc] = c0ar * (T[PT_G1 + r] * T[PT_G1 + cc] + e1 * h1 + T[PT_G2 + r] * T[PT_G2 + cc] + e2 * h2);
        }
        for (int idx = c; idx < 36; idx += 16) {                              // BB
            const int r = idx / 6, cc = idx - 6 * r;
            const double h1 = hess_entry(T + PT_BCB, T + PT_QB + 18, T + PT_V + 3, r, cc), h2 = hess_entry(T + PT_BCB, T + PT_QB + 54, T + PT_V + 9, r, cc);
            W[12 * (6 + r) + 6 + cc] = c0ar * (T[PT_G1 + 6 + r] * T[PT_G1 + 6 + cc] + e1 * h1 + T[PT_G2 + 6 + r] * T[PT_G2 + 6 + cc] + e2 * h2);
        }
        for (int idx = c; idx < 36; idx += 16) {                              // AB and its transpose
            const int r = idx / 6, cc = idx - 6 * r;
            const double h1 = col_dot(T + PT_DNA, r, T + PT_DNB, cc), h2 = col_dot(T + PT_DT, r, T + PT_SAB, cc) - col_dot(T + PT_DNA, r, T + PT_STB, cc);
            const double w = c0ar * (T[PT_G1 + r] * T[PT_G1 + 6 + cc] + e1 * h1 + T[PT_G2 + r] * T[PT_G2 + 6 + cc] + e2 * h2);
            W[12 * r + 6 + cc] = w; W[12 * (6 + cc) + r] = w;
        }
    }
    wave_lds_sync();
    if (!vok) return;
    // C2: the entries of the record
    if (grad_only != 3) for (int o = c; o < 324; o += 16) out[PB_HYY + o] = hyy_lookup(T, W, o / 18, o % 18);
    if (grad_only != 2) {
        for (int o = c; o < 216; o += 16) {
            const int r = o / 12, cy = o - 12 * r, tr = tslot_of(r);
            const double c0Y = cy < 6 ? T[PT_SC + 5] * T[PT_SC + 6 + cy / 3] * T[PT_SC + 8 + cy % 3] : 0.0;
            const int tcol = cy < 6 ? 3 + cy : 6 + cy;                         // tangent slot cy of Y = column tan[cy] of y
            double val = hyy_lookup(T, W, r, tcol) + T[PT_GL + r] * c0Y / T[PT_SC + 2];
            if (tr >= 0) val -= T[PT_SC + 3] * (T[PT_G1 + tr] * T[PT_GR1 + cy] + T[PT_G2 + tr] * T[PT_GR2 + cy]);
            out[PB_HYC + o] = val;
        }
    }
}

}  // namespace gf
