// Layout check (tools/, not product): v_mfma_f64_4x4x4 (4 blocks) against the hypothesis that it is the block diagonal of the 16 x 16 x 4 form:
//   A[I][k] at lane I + 16 k, B[k][J] at lane J + 16 k, D[i][J] (row I = 4 (J / 4) + i of column J) at lane J + 16 i, one register.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
__global__ void k(const double* a, const double* b, double* d) {
    d[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, 0, 0, 0);
}
int main() {
    double ha[64], hb[64], hd[64];
    srand(3);
    for (int l = 0; l < 64; ++l) { ha[l] = rand() / (double)RAND_MAX - 0.5; hb[l] = rand() / (double)RAND_MAX - 0.5; }
    double *da, *db, *dd; hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd); hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, e3 = 0;
    for (int J = 0; J < 16; ++J) for (int i = 0; i < 4; ++i) {
        double s1 = 0, s2 = 0, s3 = 0;
        for (int kk = 0; kk < 4; ++kk) {
            s1 += ha[4 * (J / 4) + i + 16 * kk] * hb[J + 16 * kk];                        // hypothesis of the header
            s2 += ha[4 * (J / 4) + kk + 16 * i] * hb[J + 16 * kk];                        // A with (i, k) swapped
            s3 += ha[16 * (J / 4) + 4 * kk + i] * hb[16 * (J / 4) + 4 * kk + (J % 4)];    // blocks = 16-lane rows
        }
        e1 = fmax(e1, fabs(s1 - hd[J + 16 * i])); e2 = fmax(e2, fabs(s2 - hd[J + 16 * i])); e3 = fmax(e3, fabs(s3 - hd[16 * (J / 4) + 4 * i + (J % 4)]));
    }
    printf("max abs error: block diagonal of 16x16x4 %.3e | A transposed %.3e | blocks = 16-lane rows %.3e\n", e1, e2, e3);
    return 0;
}
