// Microbenchmark: do v_mfma_f64_16x16x4_f64 and v_fma_f64 overlap on gfx950?  (tools/, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(256) void k(double* out, int iters) {
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    double v[16];
    for (int q = 0; q < 16; ++q) v[q] = q * 0.5 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = __builtin_fma(v[q], b, a);
        }
    }
    double s = acc0[0] + acc1[1] + acc2[2] + acc3[3];
    for (int q = 0; q < 16; ++q) s += v[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> float run(double* d, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double* d; hipMalloc(&d, sizeof(double) * 256 * 4096);
    const int iters = 20000;
    for (int blocks : {256, 512}) {
        float m1 = run<1>(d, blocks, iters), m2 = run<2>(d, blocks, iters), m3 = run<3>(d, blocks, iters);
        // per iteration per wave: 4 MFMA (4*2048 flop) ; 64 FMA instr (64*64*2 flop)
        double waves = blocks * 4.0;
        printf("blocks %d: mfma-only %.2f ms (%.1f TF), valu-only %.2f ms (%.1f TF), both %.2f ms (sum %.2f)\n", blocks, m1,
               waves * iters * 4 * 2048 / (m1 * 1e-3) / 1e12, m2, waves * iters * 64.0 * 128 / (m2 * 1e-3) / 1e12, m3, m1 + m2);
    }
    return 0;
}
