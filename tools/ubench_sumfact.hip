// Microbenchmark + layout proof (tools/, not product): the SUM-FACTORISED contraction of one p = 3 element on v_mfma_f64_4x4x4.
//
// Today (gf_gauss_loop.hpp): K^(ij)_ab = sum_gp phi_a[m] G^(ij)[m][m'] phi_b[m'] as a (16 x 16) x (k = 16 Gauss points x 5 m) product per component:
// T_b = G phi_b by 100 v_fmac_f64_dpp + 20 v_mfma_f64_16x16x4 per component (26.9 k FMA), 15 components (6 K + 9 dR/dCP) per element.
// With the rational quotient folded into the pointwise matrix (phi_a[m] = sum_k Q[m][k] psi_a[k], psi_a[k] = A^(d1)_{a1}(g1) B^(d2)_{a2}(g2) pure tensor
// products, k = (d1, d2), d1 + d2 <= 2: six kinds; Gt = w Q^T G Q is 6 x 6 per Gauss point and component) the sum over the 4 x 4 Gauss points factorises:
//   S1  U[g1,g2][k][b1][d2'] = sum_{d1'} Gt[g1,g2][k][(d1',d2')] A^(d1')_{b1}(g1)                        36 FMA per lane          (VALU)
//   S2  X[d2][d2'][g2][b1][a1] = sum_{g1, d1} U[g1,g2][(d1,d2)][b1][d2'] A^(d1)_{a1}(g1)                 18 v_mfma_f64_4x4x4      (k = g1, one per kind and d2')
//   S3  Y[d2][b2][g2][b1][a1]  = sum_{d2'} B^(d2')_{b2}(g2) X[d2][d2'][g2][b1][a1]                       36 FMA per lane          (VALU, no data movement)
//   S4  K[a2][b2][b1][a1]      = sum_{g2, d2} B^(d2)_{a2}(g2) Y[d2][b2][g2][b1][a1]                      12 v_mfma_f64_4x4x4      (k = g2, one per d2 and b2)
// = 12.3 k FMA per component (2.2 x fewer), and with the lane assignment below NO value moves between lanes from S1 to S4:
//   lane l = (4 b1 + g2) + 16 g1 in S1 / the A operand of S2 (one Gauss point and one b1 per lane: the lane needs the 36 Gt entries of ITS Gauss point only),
//   S2: D = mfma(A = U, B = Atab[d1][a1][g1] at lane (4 * + a1) + 16 g1)  ->  X at lane (4 b1 + a1) + 16 g2,
//   S3: in place (every lane holds all d2' of its X), S4: D = mfma(A = Btab[d2][a2][g2] at lane (4 * + a2) + 16 g2, B = Y) -> K at lane (4 b1 + a1) + 16 a2.
// The kernel below computes one component exactly this way from random tables / Gt and is CHECKED against the plain six-fold sum on the host; then it times
// an element = 15 components, with Gt read from LDS (36 ds_read_b64 per lane and component) and a stand-in for the production of Gt (the 36 entries of a
// Gauss point by its four lanes, ~40 FMA + 9 LDS writes per lane and component: the real thing = expansion of G from the compact record + Q^T G Q).
// Output: cycles per element for (a) S1-S4 alone, (b) with the Gt stand-in; compare with the 43.1 k cycles of today's Gauss-point loop per p = 3 element
// (profiles/r04_p3_rec_element_stamps.txt: group loop) of 55.5 k for the whole element.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

constexpr int NK = 6;                                   // kinds (d1, d2): 0 (0,0) 1 (1,0) 2 (2,0) 3 (0,1) 4 (1,1) 5 (0,2)
__host__ __device__ constexpr int kd1(int k) { return k < 3 ? k : (k < 5 ? k - 3 : 0); }
__host__ __device__ constexpr int kd2(int k) { return k < 3 ? 0 : (k < 5 ? 1 : 2); }
__host__ __device__ constexpr int kid(int d1, int d2) { return d2 == 0 ? d1 : (d2 == 1 ? 3 + d1 : 5); }

// tables: At[d][a][g], Bt[d][a][g] (3 x 4 x 4); Gt[comp][g1][g2][k][k'] in LDS as [comp? no: one component at a time][gp = 4 g1 + g2][36]
// MODE 0: correctness (Gt of every component copied from global memory); 1: timing, Gt resident in LDS (the same 576 values for every component: no global
// access inside the loop); 2: timing with the stand-in for the production of Gt
template <int NCOMP, int MODE>
__global__ __launch_bounds__(64) void sumfact_kernel(const double* __restrict__ At, const double* __restrict__ Bt, const double* __restrict__ Gt_g, double* __restrict__ Kout,
                                                     int nelem, long long* __restrict__ cyc) {
    const int l = threadIdx.x, q = l & 15, hi = l >> 4;
    __shared__ double s_gt[16 * 36];                    // Gt of the current component: [gp][k][k']
    __shared__ double s_src[16 * 48];                   // stand-in source data of the Gt production
    // lane roles: S1 / S2-A: b1 = q >> 2, g2 = q & 3, g1 = hi;  S2-B: a1 = q & 3, g1 = hi;  S3 / S4-B: b1 = q >> 2, a1 = q & 3, g2 = hi;  S4-A: a2 = q & 3, g2 = hi
    const int s1_b1 = q >> 2, s1_g2 = q & 3, s1_g1 = hi, gp = 4 * s1_g1 + s1_g2;
    double a_b1[3], a_a1[3], b_g2[3][4], b_a2[3];       // lane-constant table values
    for (int d = 0; d < 3; ++d) {
        a_b1[d] = At[(d * 4 + s1_b1) * 4 + s1_g1];      // A^(d)_{b1}(g1): S1
        a_a1[d] = At[(d * 4 + (q & 3)) * 4 + hi];       // A^(d)_{a1}(g1): B operand of S2
        b_a2[d] = Bt[(d * 4 + (q & 3)) * 4 + hi];       // B^(d)_{a2}(g2): A operand of S4
        for (int b2 = 0; b2 < 4; ++b2) b_g2[d][b2] = Bt[(d * 4 + b2) * 4 + hi];       // B^(d2')_{b2}(g2): S3
    }
    for (int k = l; k < 16 * 48; k += 64) s_src[k] = 1.0 + 1e-3 * k;
    double acc[NCOMP][4];
    for (int c = 0; c < NCOMP; ++c) for (int b2 = 0; b2 < 4; ++b2) acc[c][b2] = 0.0;
    __syncthreads();
    const long long t0 = clock64();
    for (int e = 0; e < nelem; ++e) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
            // ---- Gt of component c into LDS
            if constexpr (MODE == 2) {
                // the four lanes of a Gauss point produce its 36 entries, 9 each: 5 FMA on 5 values read from LDS per entry (stand-in for the expansion of G from the
                // compact record + Q^T G Q: 45 FMA, 45 LDS reads, 9 LDS writes per lane and component)
                const int part = s1_b1;
                const double* s = s_src + gp * 48 + 9 * part;
                const double ce = 1e-9 * (e + c);
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const double v = s[j] * s[(j + 1) % 9 + 0] + s[(j + 2) % 9] * s[(j + 4) % 9] + s[(j + 5) % 9] * ce + 0.5 * s[(j + 7) % 9];
                    s_gt[gp * 36 + 9 * part + j] = v * 1e-3;
                }
                __syncthreads();
            } else if constexpr (MODE == 0) {
                for (int k = l; k < 576; k += 64) s_gt[k] = Gt_g[(size_t)c * 576 + k];
                __syncthreads();
            } else {
                if (e == 0 && c == 0) for (int k = l; k < 576; k += 64) s_gt[k] = Gt_g[k];
                if (l == ((e + c) & 63)) s_gt[l] += 1e-12;          // one entry changes per component: nothing of S1 - S4 is loop invariant
                __syncthreads();
            }
            // ---- S1 + S2: X[d2][d2'] (9 registers) at lane (4 b1 + a1) + 16 g2
            const double* g = s_gt + gp * 36;
            double X[3][3];
#pragma unroll
            for (int d2 = 0; d2 < 3; ++d2) for (int e2 = 0; e2 < 3; ++e2) X[d2][e2] = 0.0;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
#pragma unroll
                for (int e2 = 0; e2 < 3; ++e2) {         // d2'
                    double u = 0.0;
#pragma unroll
                    for (int e1 = 0; e1 + e2 <= 2; ++e1) u += g[k * 6 + kid(e1, e2)] * a_b1[e1];
                    X[kd2(k)][e2] = __builtin_amdgcn_mfma_f64_4x4x4f64(u, a_a1[kd1(k)], X[kd2(k)][e2], 0, 0, 0);
                }
            }
            // ---- S3 + S4
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) {
#pragma unroll
                for (int d2 = 0; d2 < 3; ++d2) {
                    const double y = b_g2[0][b2] * X[d2][0] + b_g2[1][b2] * X[d2][1] + b_g2[2][b2] * X[d2][2];
                    acc[c][b2] = __builtin_amdgcn_mfma_f64_4x4x4f64(b_a2[d2], y, acc[c][b2], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    const long long t1 = clock64();
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    // K[a2][b2][b1][a1] at lane (4 b1 + a1) + 16 a2, register b2
    for (int c = 0; c < NCOMP; ++c) for (int b2 = 0; b2 < 4; ++b2) Kout[((size_t)blockIdx.x * NCOMP + c) * 256 + 64 * b2 + l] = acc[c][b2];
}

int main() {
    srand(7);
    auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
    std::vector<double> At(48), Bt(48), Gt(15 * 576);
    for (auto& v : At) v = rnd();
    for (auto& v : Bt) v = rnd();
    for (auto& v : Gt) v = rnd();
    double *dA, *dB, *dG, *dK; long long* dC;
    const int NWG = 1024;
    hipMalloc(&dA, 48 * 8); hipMalloc(&dB, 48 * 8); hipMalloc(&dG, Gt.size() * 8); hipMalloc(&dK, (size_t)2 * NWG * 15 * 256 * 8); hipMalloc(&dC, 8);
    hipMemcpy(dA, At.data(), 48 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 48 * 8, hipMemcpyHostToDevice); hipMemcpy(dG, Gt.data(), Gt.size() * 8, hipMemcpyHostToDevice);
    // ---- correctness: one element, 15 components, against the six-fold sum
    hipLaunchKernelGGL((sumfact_kernel<15, 0>), dim3(1), dim3(64), 0, 0, dA, dB, dG, dK, 1, dC);
    std::vector<double> K(15 * 256);
    hipMemcpy(K.data(), dK, K.size() * 8, hipMemcpyDeviceToHost);
    double err = 0, mx = 0;
    for (int c = 0; c < 15; ++c)
        for (int a1 = 0; a1 < 4; ++a1) for (int a2 = 0; a2 < 4; ++a2) for (int b1 = 0; b1 < 4; ++b1) for (int b2 = 0; b2 < 4; ++b2) {
            double s = 0;
            for (int g1 = 0; g1 < 4; ++g1) for (int g2 = 0; g2 < 4; ++g2) for (int k = 0; k < 6; ++k) for (int kp = 0; kp < 6; ++kp)
                s += At[(kd1(k) * 4 + a1) * 4 + g1] * Bt[(kd2(k) * 4 + a2) * 4 + g2] * Gt[c * 576 + (4 * g1 + g2) * 36 + 6 * k + kp] * At[(kd1(kp) * 4 + b1) * 4 + g1] * Bt[(kd2(kp) * 4 + b2) * 4 + g2];
            const double got = K[c * 256 + 64 * b2 + (4 * b1 + a1) + 16 * a2];
            err = fmax(err, fabs(got - s)); mx = fmax(mx, fabs(s));
        }
    printf("layout check: max |K_sumfact - K_direct| = %.3e (max |K| %.3e) over 15 components x 256 pairs\n", err, mx);
    // ---- timing: one wave per SIMD (1024 workgroups of one wave on 256 CUs), 200 elements per wave
    const int NE = 200;
    for (int variant = 0; variant < 4; ++variant) {
        const int nwg = variant < 2 ? NWG : 2 * NWG;          // variants 2, 3: two waves per SIMD (226 / 2xx VGPRs, 10 KB of LDS per wave: they fit)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (variant % 2 == 0) hipLaunchKernelGGL((sumfact_kernel<15, 1>), dim3(nwg), dim3(64), 0, 0, dA, dB, dG, dK, NE, dC);
            else hipLaunchKernelGGL((sumfact_kernel<15, 2>), dim3(nwg), dim3(64), 0, 0, dA, dB, dG, dK, NE, dC);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        long long cyc = 0; hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost);
        printf("%s, %d wave(s) per SIMD: %.0f cycles per element (s_memtime, wave 0), %.3f ms for %d x %d elements = %.2f us per element and SIMD; C4 (589 824 elements): %.2f ms\n",
               variant % 2 == 0 ? "S1-S4, Gt resident in LDS" : "S1-S4 + stand-in for the production of Gt (4 lanes per Gauss point, LDS)", nwg / NWG, (double)cyc / NE, ms, nwg, NE,
               1e3 * ms / NE / (nwg / NWG), ms / NE * 589824.0 / nwg);
    }
    printf("today (profiles/r04_p3_rec_element_stamps.txt): 43.1 k cycles per element in the Gauss-point group loop (T formation + MFMAs + expansions + prefactors), 55.5 k per element in all\n");
    return 0;
}
