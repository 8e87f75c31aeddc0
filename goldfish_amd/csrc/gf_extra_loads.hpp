// gf_extra_loads.hpp -- the two load models of the reference's demos that are not a dead load per unit area:
//   * follower pressure   dWext = p sqrt(det a / det A) a2 . z dA = p (x_,1 x x_,2) . z dxi   on the DEFORMED configuration
//     (demos_om/shape_opt/tube/tube_shape_opt_wint.py:303-324): residual, load stiffness (NOT symmetric element by element: the
//     element kernels' six-tile K cannot hold it) and dR/dCP -- both derivatives are the same expression, the deformed tangents
//     see c + U;
//   * dead edge traction  dWext = f . z |dX/dt| dt on a patch edge (``inner(f * bdry, z) * spline.ds``,
//     demos_om/thickness_opt/plate/plate_const_th_opt_wint.py:235-250): residual and, through the length measure, dR/dCP.
// One wave per owned control point a that carries such a load ("owner computes rows": no atomics, fixed order, bitwise reproducible),
// launched BEHIND the gather: the contributions are added to the rows of a that the gather has written (Dirichlet entries stay).
//     R_(a,i)            -= sum_gp w p R_a (g1 x g2)_i
//     K_(a,i),(b,j)      -= eps_ijk v_b,k ,   v_b = sum_gp w p R_a (phi_b,1 g2 - phi_b,2 g1)        (= the dR/dCP_j entry)
//     R_(a,i)            -= sum_egp w f_i R_a |X_t| ;   dR_(a,i)/dc_(b,k) -= sum_egp w f_i R_a (X_t,k / |X_t|) R_b,t
// The work is a few per cent of an element pass (no second derivatives, no material law): plain FP64 VALU code.
#pragma once

namespace gf {

template <int P>
__global__ __launch_bounds__(64) void kl_extra_loads_kernel(DevModel M, const int* __restrict__ cps, int ncps, int flags, double* __restrict__ R,
                                                            double* __restrict__ valK, double* __restrict__ valC0, double* __restrict__ valC1, double* __restrict__ valC2) {
    constexpr int P1 = P + 1, NB = P1 * P1, NG = NB, WB = 2 * P + 1, NBOX = WB * WB, TS = P1 * 3 * P1;
    if ((int)blockIdx.x >= ncps) return;
    const long long a = cps[blockIdx.x];
    const int lane = threadIdx.x;
    const CpDesc& cd = M.cpdesc[a];
    const PatchDev& Pt = M.patches[cd.patch];
    const int ia = cd.ia, ja = cd.ja, i0 = cd.i0, j0 = cd.j0, wbox = cd.i1 - cd.i0 + 1;
    const double press = Pt.press;
    const bool doR = (flags & GF_ASM_R_BIT) != 0, doK = (flags & GF_ASM_K_BIT) != 0, doC = (flags & GF_ASM_C_BIT) != 0;
    __shared__ double accV[NBOX][3];          // pressure: v_b of the box slots
    __shared__ double accEf[NBOX][9];         // edge tractions: blocks (i, k) = f_i q_b,k, q_b,k = -sum w R_a (X_t,k / |X_t|) R_b,t, summed over the loaded edges
    __shared__ double s_gp[NG][12];           // per Gauss point: g1[3], g2[3], 1/W, W1/W, W2/W, w p, -, -
    __shared__ double s_cp[NB][4];            // deformed homogeneous control points + weight of the element
    __shared__ double s_r[64][3];
    for (int k = lane; k < NBOX * 3; k += 64) (&accV[0][0])[k] = 0.0;
    for (int k = lane; k < NBOX * 9; k += 64) (&accEf[0][0])[k] = 0.0;
    double racc[3] = {0.0, 0.0, 0.0};         // residual of a: lanes hold partial sums, reduced at the end
    __syncthreads();

    // ---- follower pressure: the elements of a's support
    if (press != 0.0 && (doR || doK || doC)) {
        for (int kv = 0; kv < cd.nev; ++kv) for (int ku = 0; ku < cd.neu; ++ku) {
            const int eu = cd.eu0 + ku, ev = cd.ev0 + kv, bu = cd.bu[ku], bv = cd.bv[kv];
            const double* tu = M.tab + Pt.tabu + (size_t)eu * TS; const double* tv = M.tab + Pt.tabv + (size_t)ev * TS;
            if (lane < NB) {
                const long long g = Pt.cp_off + (bu + lane % P1) + (long long)(bv + lane / P1) * Pt.nu;
                const double4 c4 = reinterpret_cast<const double4*>(M.cp4)[g];
                s_cp[lane][0] = c4.x + M.u[3 * g]; s_cp[lane][1] = c4.y + M.u[3 * g + 1]; s_cp[lane][2] = c4.z + M.u[3 * g + 2]; s_cp[lane][3] = c4.w;
            }
            __syncthreads();
            const int la_u = ia - bu, la_v = ja - bv;                       // a's local indices in this element
            if (lane < NG) {                                                  // phase A: lane = Gauss point: deformed tangents (quotient rule), residual term
                const int gu = lane % P1, gv = lane / P1;
                double A0[3] = {0, 0, 0}, A1[3] = {0, 0, 0}, A2[3] = {0, 0, 0}, W0 = 0, W1 = 0, W2 = 0;
                for (int jv = 0; jv < P1; ++jv) for (int ju = 0; ju < P1; ++ju) {
                    const double u0 = tu[(gu * 3 + 0) * P1 + ju], u1 = tu[(gu * 3 + 1) * P1 + ju], v0 = tv[(gv * 3 + 0) * P1 + jv], v1 = tv[(gv * 3 + 1) * P1 + jv];
                    const double n0 = u0 * v0, n1 = u1 * v0, n2 = u0 * v1; const double* c = s_cp[ju + P1 * jv];
                    for (int k = 0; k < 3; ++k) { A0[k] += n0 * c[k]; A1[k] += n1 * c[k]; A2[k] += n2 * c[k]; }
                    W0 += n0 * c[3]; W1 += n1 * c[3]; W2 += n2 * c[3];
                }
                const double iW = 1.0 / W0, w1 = W1 * iW, w2 = W2 * iW;
                double* sg = s_gp[lane];
                for (int k = 0; k < 3; ++k) { const double x = A0[k] * iW; sg[k] = (A1[k] - x * W1) * iW; sg[3 + k] = (A2[k] - x * W2) * iW; }
                const double wp = M.tab[Pt.wu + eu * P1 + gu] * M.tab[Pt.wv + ev * P1 + gv] * press;
                sg[6] = iW; sg[7] = w1; sg[8] = w2; sg[9] = wp;
                const double Ra = tu[(gu * 3 + 0) * P1 + la_u] * tv[(gv * 3 + 0) * P1 + la_v] * iW;
                const double* g1 = sg; const double* g2 = sg + 3;
                racc[0] -= wp * Ra * (g1[1] * g2[2] - g1[2] * g2[1]);
                racc[1] -= wp * Ra * (g1[2] * g2[0] - g1[0] * g2[2]);
                racc[2] -= wp * Ra * (g1[0] * g2[1] - g1[1] * g2[0]);
            }
            __syncthreads();
            if (lane < NB && (doK || doC)) {                                  // phase B: lane = basis function b of the element
                const int ju = lane % P1, jv = lane / P1;
                double v[3] = {0, 0, 0};
                for (int gp = 0; gp < NG; ++gp) {
                    const int gu = gp % P1, gv = gp / P1; const double* sg = s_gp[gp];
                    const double iW = sg[6], u0 = tu[(gu * 3 + 0) * P1 + ju], u1 = tu[(gu * 3 + 1) * P1 + ju], v0 = tv[(gv * 3 + 0) * P1 + jv], v1 = tv[(gv * 3 + 1) * P1 + jv];
                    const double Rb = u0 * v0 * iW, p1 = u1 * v0 * iW - Rb * sg[7], p2 = u0 * v1 * iW - Rb * sg[8];
                    const double Ra = tu[(gu * 3 + 0) * P1 + la_u] * tv[(gv * 3 + 0) * P1 + la_v] * iW, f = sg[9] * Ra;
                    for (int k = 0; k < 3; ++k) v[k] += f * (p1 * sg[3 + k] - p2 * sg[k]);
                }
                const int slot = (bu + ju - i0) + (bv + jv - j0) * wbox;     // one lane per slot: plain adds
                for (int k = 0; k < 3; ++k) accV[slot][k] += v[k];
            }
            __syncthreads();
        }
    }

    // ---- dead edge tractions: a lies on the edge xi_d = side iff its index in direction d is the first / last one
    for (int e = 0; e < 4; ++e) {
        const double f0 = Pt.et[3 * e], f1 = Pt.et[3 * e + 1], f2 = Pt.et[3 * e + 2];
        if (f0 == 0.0 && f1 == 0.0 && f2 == 0.0) continue;
        const int d = e >> 1, side = e & 1, nd_ = d ? Pt.nv : Pt.nu;
        if ((d ? ja : ia) != (side ? nd_ - 1 : 0)) continue;
        // along the edge (direction t = 1 - d): the elements of a's support, their p + 1 edge control points, p + 1 Gauss points each
        const int net = d ? cd.neu : cd.nev;                                  // d = 1: the edge runs along u
        const int at = d ? ia : ja, t0 = d ? i0 : j0;                         // a's index / the box origin along the edge
        const int wt = d ? wbox : (cd.j1 - cd.j0 + 1);
        if (lane < wt) {                                                      // lane = control point b of the box row along the edge
            const int bt = t0 + lane;
            double q[3] = {0, 0, 0}, rs = 0.0;
            for (int ke = 0; ke < net; ++ke) {
                const int et = (d ? cd.eu0 : cd.ev0) + ke, bfirst = d ? cd.bu[ke] : cd.bv[ke];
                const double* tt = M.tab + (d ? Pt.tabu : Pt.tabv) + (size_t)et * TS;
                const int lb = bt - bfirst, lat = at - bfirst;
                for (int g = 0; g < P1; ++g) {
                    double A0[3] = {0, 0, 0}, A1[3] = {0, 0, 0}, W0 = 0, W1 = 0;
                    for (int j = 0; j < P1; ++j) {
                        const long long gg = Pt.cp_off + (d ? (long long)(bfirst + j) + (long long)ja * Pt.nu : (long long)ia + (long long)(bfirst + j) * Pt.nu);
                        const double4 c4 = reinterpret_cast<const double4*>(M.cp4)[gg];
                        const double n0 = tt[(g * 3 + 0) * P1 + j], n1 = tt[(g * 3 + 1) * P1 + j];
                        A0[0] += n0 * c4.x; A0[1] += n0 * c4.y; A0[2] += n0 * c4.z; W0 += n0 * c4.w;
                        A1[0] += n1 * c4.x; A1[1] += n1 * c4.y; A1[2] += n1 * c4.z; W1 += n1 * c4.w;
                    }
                    const double iW = 1.0 / W0;
                    double Xt[3];
                    for (int k = 0; k < 3; ++k) Xt[k] = (A1[k] - A0[k] * iW * W1) * iW;
                    const double len = sqrt(Xt[0] * Xt[0] + Xt[1] * Xt[1] + Xt[2] * Xt[2]);
                    const double w = M.tab[(d ? Pt.wu : Pt.wv) + et * P1 + g];
                    const double Ra = tt[(g * 3 + 0) * P1 + lat] * iW;
                    if (lane == 0) rs += w * Ra * len;
                    if (lb >= 0 && lb < P1) {
                        const double Rb = tt[(g * 3 + 0) * P1 + lb] * iW, Rbt = tt[(g * 3 + 1) * P1 + lb] * iW - Rb * W1 * iW;
                        for (int k = 0; k < 3; ++k) q[k] -= w * Ra * (Xt[k] / len) * Rbt;
                    }
                }
            }
            if (lane == 0) { racc[0] -= f0 * rs; racc[1] -= f1 * rs; racc[2] -= f2 * rs; }
            const int slot = d ? (lane + (ja - j0) * wbox) : ((ia - i0) + lane * wbox);
            const double ff[3] = {f0, f1, f2};
            for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) accEf[slot][3 * i + k] += ff[i] * q[k];
        }
        __syncthreads();
    }

    // ---- residual of a: sum the lanes' partial sums in lane order
    for (int i = 0; i < 3; ++i) s_r[lane][i] = racc[i];
    __syncthreads();
    if (lane < 3 && doR) {
        double t = 0.0;
        for (int l = 0; l < 64; ++l) t += s_r[l][lane];
        R[3 * a + lane] += t;                                                 // Dirichlet rows are zeroed afterwards (zero_rows_kernel)
    }
    if (!(doK || doC)) return;
    // ---- add to the rows of a (written by the gather): Dirichlet rows / columns keep their 0 / 1 entries
    const long long ptr_c = M.nb_ptr_c[a], deg_c = M.nb_ptr_c[a + 1] - ptr_c;
    const unsigned zmask = (M.zero[3 * a] ? 1u : 0u) | (M.zero[3 * a + 1] ? 2u : 0u) | (M.zero[3 * a + 2] ? 4u : 0u);
    for (int k = lane; k < (int)deg_c; k += 64) {
        const unsigned meta = M.nb_meta[ptr_c + k];
        if ((meta & 127u) == 127u) continue;                                  // coupling-only column: no shell-level load term
        const int slot = int(meta & 127u);
        const double v[3] = {accV[slot][0], accV[slot][1], accV[slot][2]};
        // block (i, j) = -eps_ijk v_k:  (0,1) -v2  (0,2) +v1  (1,0) +v2  (1,2) -v0  (2,0) -v1  (2,1) +v0
        const double blk[9] = {0.0, -v[2], v[1], v[2], 0.0, -v[0], -v[1], v[0], 0.0};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if ((zmask >> i) & 1u) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (doK && !(meta & (128u << j))) valK[9 * ptr_c + (long long)i * 3 * deg_c + 3 * k + j] += blk[3 * i + j];
                if (doC) { double* dst = (j == 0 ? valC0 : (j == 1 ? valC1 : valC2)) + 3 * ptr_c + (long long)i * deg_c + k; *dst += blk[3 * i + j] + accEf[slot][3 * i + j]; }
            }
        }
    }
}

}  // namespace gf
