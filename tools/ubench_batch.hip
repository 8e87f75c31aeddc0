// Microbenchmark (tools/, not product): one (component, m) batch of the element kernels' contraction -- NT B operands, each formed by a chain of five
// v_fmac_f64_dpp (row_newbcast) behind a zero initialisation, then NT v_mfma_f64_16x16x4 on NT accumulators -- in the orders
//   0: MFMAs only            1: DPP chains only          2: chains, then MFMAs (what gauss_group does)
//   3: software pipelined: the chains of batch i + 1 interleaved between the MFMAs of batch i (one chain behind every MFMA)
// Answers whether FP64 VALU work issued between the MFMAs hides under them on gfx950 (it does not if (3) costs what (2) does).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define DPP(t, g, p, L) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(g), "v"(p))
template <int NT, int MODE> __global__ __launch_bounds__(64) void k(double* out, int iters, long long* cyc) {
    const long long c0 = clock64(), w0 = wall_clock64();
    d4 acc[NT];
    for (int q = 0; q < NT; ++q) acc[q] = d4{0, 0, 0, 0};
    double g[5], p[5], phi = 1.0 + threadIdx.x * 1e-6;
    for (int m = 0; m < 5; ++m) { g[m] = threadIdx.x * 1e-3 + m; p[m] = 1.0 + m * 1e-3; }
    double t[NT], tn[NT];
    for (int q = 0; q < NT; ++q) { t[q] = q; tn[q] = q; }
    auto chain = [&](double& tt) { tt = 0.0; DPP(tt, g[0], p[0], 1); DPP(tt, g[1], p[1], 2); DPP(tt, g[2], p[2], 3); DPP(tt, g[3], p[3], 4); DPP(tt, g[4], p[4], 5); };
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi, t[q], acc[q], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < NT; ++q) chain(t[q]);
        } else if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < NT; ++q) chain(t[q]);
            asm volatile("s_nop 1");
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi, t[q], acc[q], 0, 0, 0);
        } else if (MODE == 4) {          // chains, then the tiles as four v_mfma_f64_4x4x4 each, A operand rotated through its 16-lane row
#pragma unroll
            for (int q = 0; q < NT; ++q) chain(t[q]);
            double a4[4] = {phi, phi, phi, phi};
            {
                const int lo = __double2loint(phi), hi = __double2hiint(phi);
                a4[1] = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x12c, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, 0x12c, 0xf, 0xf, false));
                a4[2] = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false));
                a4[3] = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x124, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xf, 0xf, false));
            }
            asm volatile("s_nop 1");
#pragma unroll
            for (int q = 0; q < NT; ++q)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[q][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[s], t[q], acc[q][s], 0, 0, 0);
            phi += 1e-9;
        } else {
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(phi, t[q], acc[q], 0, 0, 0);
                asm volatile("" : "+v"(acc[q][0]));       // keeps the MFMA in front of the chain in program order
                chain(tn[q]);
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) { const double s = t[q]; t[q] = tn[q]; tn[q] = s; }
        }
    }
    double s = 0;
    for (int q = 0; q < NT; ++q) s += acc[q][0] + acc[q][3] + t[q] + tn[q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = clock64() - c0; cyc[1] = wall_clock64() - w0; }
}
// v_mfma_f64_4x4x4 (four 4 x 4 x 4 blocks, 512 flop): back-to-back issue on independent accumulators
template <int NT> __global__ __launch_bounds__(64) void k4(double* out, int iters, long long* cyc) {
    const long long c0 = clock64(), w0 = wall_clock64();
    double acc[NT], a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    for (int q = 0; q < NT; ++q) acc[q] = q;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[q], 0, 0, 0);
    }
    double s = 0; for (int q = 0; q < NT; ++q) s += acc[q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = clock64() - c0; cyc[1] = wall_clock64() - w0; }
}
template <int NT> void run4(double* d, int blocks) {
    const int iters = 8000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    long long* dc; hipMalloc(&dc, 16); long long hc[2];
    hipLaunchKernelGGL((k4<NT>), dim3(blocks), dim3(64), 0, 0, d, iters, dc); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((k4<NT>), dim3(blocks), dim3(64), 0, 0, d, iters, dc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    printf("v_mfma_f64_4x4x4  blocks %4d: %6.1f cycles per instruction at %.2f GHz  (%5.1f TFLOP/s)\n", blocks, (double)hc[0] / iters / NT, (double)hc[0] / ((double)hc[1] * 10.0),
           (double)blocks * NT * 512.0 * iters / (ms * 1e-3) / 1e12);
    hipFree(dc);
}
template <int NT, int MODE> void run(double* d, const char* what, int blocks = 1024, int iters = 4000) {      // 1024 blocks: one wave per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    long long* dc; hipMalloc(&dc, 16); long long hc[2];
    hipLaunchKernelGGL((k<NT, MODE>), dim3(blocks), dim3(64), 0, 0, d, iters, dc); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((k<NT, MODE>), dim3(blocks), dim3(64), 0, 0, d, iters, dc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)hc[0] / ((double)hc[1] * 10.0);       // wall_clock64: 100 MHz
    printf("NT %2d  blocks %4d  %-44s %8.1f ns per batch = %6.1f cycles per MFMA slot at the measured %.2f GHz  (%5.1f TFLOP/s of MFMA work)\n", NT, blocks, what, ms * 1e6 / iters,
           (double)hc[0] / iters / NT, ghz, MODE == 1 ? 0.0 : (double)blocks * NT * 2048.0 * iters / (ms * 1e-3) / 1e12);
    hipFree(dc);
}
int main() {
    double* d; hipMalloc(&d, 8 * 64 * 8192);
    run<6, 0>(d, "MFMAs only"); run<6, 1>(d, "DPP chains only"); run<6, 2>(d, "chains, then MFMAs"); run<6, 3>(d, "chains interleaved with the previous MFMAs");
    run<6, 4>(d, "chains, then tiles of four 4x4x4"); run<9, 4>(d, "chains, then tiles of four 4x4x4"); run<15, 4>(d, "chains, then tiles of four 4x4x4");
    run<9, 0>(d, "MFMAs only"); run<9, 1>(d, "DPP chains only"); run<9, 2>(d, "chains, then MFMAs"); run<9, 3>(d, "chains interleaved with the previous MFMAs");
    run<15, 0>(d, "MFMAs only"); run<15, 1>(d, "DPP chains only"); run<15, 2>(d, "chains, then MFMAs"); run<15, 3>(d, "chains interleaved with the previous MFMAs");
    // is the matrix pipe's rate a per-wave or a per-SIMD limit, and what clock does the chip hold under it?
    run<15, 0>(d, "MFMAs only", 256); run<15, 0>(d, "MFMAs only", 512); run<15, 0>(d, "MFMAs only", 2048); run<15, 0>(d, "MFMAs only", 4096); run<15, 0>(d, "MFMAs only", 8192); run<15, 0>(d, "MFMAs only", 1024, 40000);
    run<15, 2>(d, "chains, then MFMAs", 256); run<15, 2>(d, "chains, then MFMAs", 2048);
    run4<16>(d, 1024); run4<16>(d, 2048); run4<16>(d, 4096);
    return 0;
}
