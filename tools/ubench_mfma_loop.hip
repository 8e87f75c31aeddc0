// Microbenchmark (tools/, not product): the inner loop an MFMA-based element kernel would run.
// Per (q, m): read one 5-wide row of G for this lane's Gauss point from LDS, form t = G_row . phi_b (5 FMA),
// then acc[q] += phi_a[m] (x) t as one v_mfma_f64_16x16x4 (k = 4 Gauss points).  Also checks the operand layout.
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NQ = 15, NM = 5, ROW = 6, GSZ = NQ * NM * ROW;      // doubles of G per Gauss point
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const double* __restrict__ gin, const double* __restrict__ phin, double* out, int iters) {
    __shared__ __attribute__((aligned(16))) double sG[4 * GSZ];
    for (int i = threadIdx.x; i < 4 * GSZ; i += 64 * WAVES) sG[i] = gin[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, x = lane & 15, kk = lane >> 4;
    double phi[NM];
    for (int m = 0; m < NM; ++m) phi[m] = phin[(kk * 16 + x) * NM + m];
    d4 acc[NQ];
    for (int q = 0; q < NQ; ++q) acc[q] = d4{0, 0, 0, 0};
    const double* g = sG + kk * GSZ;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const double2* r = reinterpret_cast<const double2*>(g + (q * NM + m) * ROW);
                const double2 g01 = r[0], g23 = r[1]; const double g4 = g[(q * NM + m) * ROW + 4];
                const double t = g01.x * phi[0] + g01.y * phi[1] + g23.x * phi[2] + g23.y * phi[3] + g4 * phi[4];
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi[m], t, acc[q], 0, 0, 0);
            }
    }
    double* o = out + ((size_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * NQ * 256;
    for (int q = 0; q < NQ; ++q) for (int r = 0; r < 4; ++r) o[q * 256 + (4 * kk + r) * 16 + x] = acc[q][r];
}
int main() {
    std::vector<double> G(4 * GSZ), P(64 * NM);
    srand(1);
    for (auto& v : G) v = rand() / (double)RAND_MAX - 0.5;
    for (auto& v : P) v = rand() / (double)RAND_MAX - 0.5;
    double *dG, *dP, *dO; const int maxblocks = 4096;
    hipMalloc(&dG, G.size() * 8); hipMalloc(&dP, P.size() * 8); hipMalloc(&dO, (size_t)maxblocks * 2 * NQ * 256 * 8);
    hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice);
    // layout check: acc_q[a][b] = sum_kk sum_m phi[kk][a][m] * (sum_m' G[kk][q][m][m'] phi[kk][b][m'])
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dG, dP, dO, 1); hipDeviceSynchronize();
    std::vector<double> O(NQ * 256); hipMemcpy(O.data(), dO, O.size() * 8, hipMemcpyDeviceToHost);
    double err = 0, err2 = 0;
    for (int q = 0; q < NQ; ++q) for (int a = 0; a < 16; ++a) for (int b = 0; b < 16; ++b) {
        double s = 0;
        for (int kk = 0; kk < 4; ++kk) for (int m = 0; m < NM; ++m) {
            double t = 0; for (int mp = 0; mp < NM; ++mp) t += G[kk * GSZ + (q * NM + m) * ROW + mp] * P[(kk * 16 + b) * NM + mp];
            s += P[(kk * 16 + a) * NM + m] * t;
        }
        err = fmax(err, fabs(s - O[q * 256 + a * 16 + b]));
        const int kk = a & 3, r = a >> 2;                      // alternative: D row = (lane/16) + 4*r
        err2 = fmax(err2, fabs(s - O[q * 256 + (4 * kk + r) * 16 + b]));
    }
    printf("layout check: max abs err %.3e (row = 4*(lane/16)+r)   %.3e (row = lane/16 + 4*r)\n", err, err2);
    const int iters = 2000;
    for (int waves = 1; waves <= 2; ++waves) for (int blocks : {256, 512, 1024, 2048}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() { if (waves == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, dG, dP, dO, iters); else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(128), 0, 0, dG, dP, dO, iters); };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double units = (double)blocks * waves * iters * NQ * NM;       // (q,m) units = MFMAs
        printf("waves/block %d blocks %4d: %.2f ms  %.1f TF (mfma+T flops)  %.1f ns per unit per CU-SIMD-slot\n", waves, blocks, ms,
               units * (2048 + 640) / (ms * 1e-3) / 1e12, ms * 1e6 / (units / 1024.0));
    }
    return 0;
}
