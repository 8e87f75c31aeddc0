// Check + timing (tools/, not product): the two diagonal-tile kernels of goldfish_amd/csrc/gf_solver.hip (row-per-thread, and round 5's blocked form) against a host
// L D L^T in long double on random symmetric tiles (positive definite and indefinite with a dominant diagonal), and their time per dependent launch.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/test_diag.hip -o /tmp/test_diag && /tmp/test_diag
#include "../goldfish_amd/csrc/gf_solver.hip"
#include <random>
namespace {
__global__ __launch_bounds__(256) void diag_rows_kernel(double* band, double* linv, double* dval, const long long* rowoff, double* stat) { diag_body_rows(band, linv, dval, rowoff, (int)blockIdx.x, stat); }
__global__ __launch_bounds__(256) void diag_blocked_kernel(double* band, double* linv, double* dval, const long long* rowoff, double* stat) { GF_TILE_SMEM; diag_body_blocked(band, linv, dval, rowoff, (int)blockIdx.x, stat, smem); }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int NT = 512;
    std::mt19937_64 rng(7); std::normal_distribution<double> N01;
    std::vector<double> A((size_t)NT * NB2), Lref((size_t)NT * NB2), Mref((size_t)NT * NB2), dref((size_t)NT * NB);
    for (int t = 0; t < NT; ++t) {
        double* a = &A[(size_t)t * NB2];
        std::vector<long double> w(NB2);
        for (int i = 0; i < NB; ++i) for (int j = 0; j <= i; ++j) { const double v = N01(rng); w[i * NB + j] = v; w[j * NB + i] = v; }
        for (int i = 0; i < NB; ++i) w[i * NB + i] = (t % 2 && i % 3 == 0 ? -1.0 : 1.0) * (40.0 + std::fabs(N01(rng)) * 10.0) * (t % 5 == 0 ? 1e6 : 1.0);
        for (int i = 0; i < NB; ++i) for (int j = 0; j < NB; ++j) a[i * NB + j] = j <= i ? (double)w[i * NB + j] : 1e300 * (i % 2 ? 1 : -1);       // what lies above the diagonal must not matter
        std::vector<long double> L(NB2, 0.0L), d(NB);
        for (int j = 0; j < NB; ++j) {
            long double s = w[j * NB + j]; for (int m = 0; m < j; ++m) s -= L[j * NB + m] * L[j * NB + m] * d[m];
            d[j] = s; L[j * NB + j] = 1.0L;
            for (int i = j + 1; i < NB; ++i) { long double v = w[i * NB + j]; for (int m = 0; m < j; ++m) v -= L[i * NB + m] * L[j * NB + m] * d[m]; L[i * NB + j] = v / s; }
        }
        std::vector<long double> M(NB2, 0.0L);
        for (int c = 0; c < NB; ++c) for (int r = 0; r < NB; ++r) { long double v = r == c ? 1.0L : 0.0L; for (int m = 0; m < r; ++m) v -= L[r * NB + m] * M[m * NB + c]; M[r * NB + c] = v; }
        for (int i = 0; i < NB; ++i) { dref[(size_t)t * NB + i] = (double)d[i]; for (int j = 0; j < NB; ++j) { Lref[(size_t)t * NB2 + i * NB + j] = j < i ? (double)L[i * NB + j] : (j == i ? (double)d[i] : 0.0); Mref[(size_t)t * NB2 + i * NB + j] = (double)M[i * NB + j]; } }
    }
    std::vector<long long> rowoff(NT); for (int t = 0; t < NT; ++t) rowoff[t] = t;
    double *dA, *dM, *dd, *dst; long long* dro;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dM, A.size() * 8)); CK(hipMalloc(&dd, dref.size() * 8)); CK(hipMalloc(&dst, (size_t)NT * 16)); CK(hipMalloc(&dro, NT * 8));
    CK(hipMemcpy(dro, rowoff.data(), NT * 8, hipMemcpyHostToDevice));
    int bad = 0;
    for (int which = 0; which < 2; ++which) {
        CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(dM, 0xff, A.size() * 8));
        if (which == 0) hipLaunchKernelGGL(diag_rows_kernel, dim3(NT), dim3(256), 0, 0, dA, dM, dd, dro, dst);
        else hipLaunchKernelGGL(diag_blocked_kernel, dim3(NT), dim3(256), 0, 0, dA, dM, dd, dro, dst);
        CK(hipDeviceSynchronize());
        std::vector<double> L(A.size()), M(A.size()), d(dref.size()), st((size_t)NT * 2);
        CK(hipMemcpy(L.data(), dA, A.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(M.data(), dM, A.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(d.data(), dd, d.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
        double eL = 0, eM = 0, eD = 0, eS = 0;
        for (int t = 0; t < NT; ++t) {
            double sL = 0, sM = 0, mn = 1e300, mx = 0;
            for (int q = 0; q < NB2; ++q) { sL = std::max(sL, std::fabs(Lref[(size_t)t * NB2 + q])); sM = std::max(sM, std::fabs(Mref[(size_t)t * NB2 + q])); }
            for (int q = 0; q < NB2; ++q) {
                const double a = L[(size_t)t * NB2 + q], b = M[(size_t)t * NB2 + q];
                eL = std::max(eL, std::isfinite(a) ? std::fabs(a - Lref[(size_t)t * NB2 + q]) / sL : 1e300); eM = std::max(eM, std::isfinite(b) ? std::fabs(b - Mref[(size_t)t * NB2 + q]) / sM : 1e300);
            }
            for (int i = 0; i < NB; ++i) { const double r = dref[(size_t)t * NB + i]; eD = std::max(eD, std::fabs(d[(size_t)t * NB + i] - r) / std::fabs(r)); mn = std::min(mn, std::fabs(r)); mx = std::max(mx, std::fabs(r)); }
            eS = std::max(eS, std::max(std::fabs(st[2 * t] - mn) / mn, std::fabs(st[2 * t + 1] - mx) / mx));
        }
        printf("%s: tile (d, L) max error %.2e, inverse of L %.2e, dval %.2e, pivot extremes %.2e (relative to the largest entry; %d tiles)\n", which ? "blocked" : "rows   ", eL, eM, eD, eS, NT);
        if (!(eL < 1e-12 && eM < 1e-12 && eD < 1e-12 && eS < 1e-12)) bad = 1;
    }
    // time per DEPENDENT launch (one tile, stream order) and per tile when many run side by side
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
        float ms1 = 0, msN = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int q = 0; q < 200; ++q) { if (which == 0) hipLaunchKernelGGL(diag_rows_kernel, dim3(1), dim3(256), 0, 0, dA, dM, dd, dro, dst); else hipLaunchKernelGGL(diag_blocked_kernel, dim3(1), dim3(256), 0, 0, dA, dM, dd, dro, dst); }
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms1, e0, e1));
            CK(hipEventRecord(e0, 0));
            for (int q = 0; q < 20; ++q) { if (which == 0) hipLaunchKernelGGL(diag_rows_kernel, dim3(NT), dim3(256), 0, 0, dA, dM, dd, dro, dst); else hipLaunchKernelGGL(diag_blocked_kernel, dim3(NT), dim3(256), 0, 0, dA, dM, dd, dro, dst); }
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&msN, e0, e1));
        }
        printf("%s: %.1f us per dependent one-tile launch (200 in a row, launch gap included), %.1f us per launch of %d tiles\n", which ? "blocked" : "rows   ", 1e3 * ms1 / 200, 1e3 * msN / 20, NT);
    }
    printf(bad ? "FAILED\n" : "ok\n");
    return bad;
}
