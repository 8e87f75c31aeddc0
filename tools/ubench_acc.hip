// Microbenchmark (tools/, not product): issue interval of the FP64 MFMAs with the accumulators in AGPRs vs arch VGPRs (inline asm pins the register class)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(64) void k(double* out, int iters, long long* cyc) {
    const long long c0 = clock64(), w0 = wall_clock64();
    constexpr int NT = 12;
    d4 acc[NT]; double acs[4 * NT];
    for (int q = 0; q < NT; ++q) acc[q] = d4{0, 0, 0, 0};
    for (int q = 0; q < 4 * NT; ++q) acs[q] = 0.0;
    double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            if (MODE == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[q]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int q = 0; q < 4 * NT; ++q) {
            if (MODE == 2) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(acs[q]) : "v"(a), "v"(b));
            if (MODE == 3) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acs[q]) : "v"(a), "v"(b));
        }
    }
    if (MODE >= 4) {           // the element kernels' pattern: tiles of four 4 x 4 x 4 products, A operand a4[s] per product, B operand t[q] per tile
        double a4[4] = {a, a + 1.0, a + 2.0, a + 3.0}, t[NT];
        for (int q = 0; q < NT; ++q) t[q] = b + q;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (MODE == 4) acc[q][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[s], t[q], acc[q][s], 0, 0, 0);
                    if (MODE == 5) acc[q][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[0], t[q], acc[q][s], 0, 0, 0);
                    if (MODE == 6) acc[q][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[s], t[0], acc[q][s], 0, 0, 0);
                }
        }
    }
    double s = 0;
    for (int q = 0; q < NT; ++q) s += acc[q][0] + acc[q][3] + acc[q][1] + acc[q][2];
    for (int q = 0; q < 4 * NT; ++q) s += acs[q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = clock64() - c0; cyc[1] = wall_clock64() - w0; }
}
template <int MODE> void run(double* d, const char* what, int per_iter, double flop) {
    const int iters = 4000, blocks = 1024;
    long long* dc; hipMalloc(&dc, 16); long long hc[2];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, iters, dc); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, iters, dc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    printf("%-52s %6.1f cycles per instruction  (%5.1f TFLOP/s, one wave per SIMD)\n", what, (double)hc[0] / iters / per_iter, (double)blocks * per_iter * flop * iters / (ms * 1e-3) / 1e12);
}
int main() {
    double* d; hipMalloc(&d, 8 * 64 * 1024);
    run<0>(d, "v_mfma_f64_16x16x4, accumulators in AGPRs", 12, 2048.0);
    run<1>(d, "v_mfma_f64_16x16x4, accumulators in arch VGPRs", 12, 2048.0);
    run<2>(d, "v_mfma_f64_4x4x4, accumulators in AGPRs", 48, 512.0);
    run<3>(d, "v_mfma_f64_4x4x4, accumulators in arch VGPRs", 48, 512.0);
    run<4>(d, "v_mfma_f64_4x4x4 tiles: A per product, B per tile", 48, 512.0);
    run<5>(d, "v_mfma_f64_4x4x4 tiles: one A, B per tile", 48, 512.0);
    run<6>(d, "v_mfma_f64_4x4x4 tiles: A per product, one B", 48, 512.0);
    return 0;
}
