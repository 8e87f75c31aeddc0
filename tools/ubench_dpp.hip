// Microbenchmark (tools/, not product): issue cost of v_fmac_f64_dpp (row_newbcast) vs plain v_fmac_f64, 16 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
template <int MODE> __global__ __launch_bounds__(64) void k(double* out, int iters) {
    double acc[16], g = threadIdx.x * 1e-3 + 1.0, p = 1.0 + threadIdx.x * 1e-6;
    for (int q = 0; q < 16; ++q) acc[q] = q;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[q]) : "v"(g), "v"(p));
                else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[q]) : "v"(g), "v"(p));
            }
    }
    double s = 0; for (int q = 0; q < 16; ++q) s += acc[q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
    double* d; hipMalloc(&d, 8 * 64 * 4096);
    const int iters = 20000;
    for (int blocks : {1024, 2048}) for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, iters); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, iters); };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 64;      // instructions per wave
        printf("blocks %d (%d wave/SIMD) %s: %.2f ms  %.2f ns per instruction per wave-slot\n", blocks, blocks / 1024, mode ? "v_fmac_f64_dpp" : "v_fmac_f64    ", ms, ms * 1e6 / n / (blocks / 1024));
    }
    return 0;
}
