// Microbenchmark (tools/, not product): where the time of the solver's wide trailing update goes.  Includes the solver's translation unit and runs
//   V0  update_wide_kernel as the library launches it (one large front: nrow block rows behind a group of w block columns)
//   V1  the same loop without the global operand loads inside it (the first column's tiles parked again and again)
//   V2  V1 without parking and without barriers (MFMAs + LDS operand reads only)
//   V3  V2 without the LDS reads (MFMAs on register operands only: the issue rate of four waves x two workgroups per CU)
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/ubench_update.hip -o tools/ubench_update
#include "../goldfish_amd/csrc/gf_solver.hip"
#include <vector>

namespace {
template <int V>
__global__ __launch_bounds__(256) void ub_kernel(double* __restrict__ band, const double* __restrict__ wbuf, long long wstride, const long long* __restrict__ rowoff, int k0, int w, int nrow) {
    int gi, gj; tri_index((int)blockIdx.x, gi, gj);
    if (gi >= nrow) return;
    __shared__ __attribute__((aligned(16))) double sA[NB * LS], sB[NB * LS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = k0 + w + gi, j = k0 + w + gj;
    double* C = band + (size_t)(rowoff[i] + (gi - gj)) * NB2;
    double2 ra[8], rb[8];
    fetch_tile(wbuf + (size_t)(i - (k0 + 1)) * NB2, ra, tid);
    fetch_tile(band + (size_t)(rowoff[j] + (j - k0)) * NB2, rb, tid);
    d4 acc[4];
    for (int nj = 0; nj < 4; ++nj) for (int rg = 0; rg < 4; ++rg) acc[nj][rg] = C[(16 * wave + 4 * rg + (lane >> 4)) * NB + 16 * nj + (lane & 15)];
    if (V >= 2) { park_tile(ra, sA, tid); park_tile(rb, sB, tid); __syncthreads(); }
    for (int c = 0; c < w; ++c) {
        if (V == 1) { park_tile(ra, sA, tid); park_tile(rb, sB, tid); __syncthreads(); }
        if (V <= 2) tile_abt(sA, sB, acc, wave, lane, -1.0);
        else {
            double a = ra[0].x, b = rb[0].x;
#pragma unroll 4
            for (int k = 0; k < NB; k += 4) {
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) acc[nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[nj], 0, 0, 0);
                asm volatile("" : "+v"(a), "+v"(b));
            }
        }
        if (V == 1) __syncthreads();
    }
    for (int nj = 0; nj < 4; ++nj) for (int rg = 0; rg < 4; ++rg) C[(16 * wave + 4 * rg + (lane >> 4)) * NB + 16 * nj + (lane & 15)] = acc[nj][rg];
}
}

namespace {
template <int DMA>
__global__ __launch_bounds__(256) void wide_kernel(double* __restrict__ band, const double* __restrict__ wbuf, long long wstride, const long long* __restrict__ rowoff, int k0, int w, int nrow) {
    GF_TILE_SMEM;
    int gi, gj; tri_index((int)blockIdx.x, gi, gj);
    if (gi >= nrow) return;
    if (DMA) update_wide_tile_dma<HK>(band, wbuf, wstride, rowoff, k0, w, gi, gj, smem);
    else update_wide_tile(band, wbuf, wstride, rowoff, k0, w, gi, gj, smem);
}
// the DMA form with parts of HKT k and only its own LDS (4 x 64 x HKT doubles: HKT = 16 -> 32 KB, four workgroups per CU; 64 -> 128 KB, one)
template <int HKT>
__global__ __launch_bounds__(256) void wide_dma_kernel(double* __restrict__ band, const double* __restrict__ wbuf, long long wstride, const long long* __restrict__ rowoff, int k0, int w, int nrow) {
    __shared__ __attribute__((aligned(16))) double smem[4 * NB * HKT];
    int gi, gj; tri_index((int)blockIdx.x, gi, gj);
    if (gi >= nrow) return;
    update_wide_tile_dma<HKT>(band, wbuf, wstride, rowoff, k0, w, gi, gj, smem);
}
}
namespace {
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void wide_macro_kernel(double* __restrict__ band, const double* __restrict__ wbuf, long long wstride, const long long* __restrict__ rowoff, int k0, int w, int nrow) {
    __shared__ __attribute__((aligned(16))) double smem[8 * NB * 16];
    int mi, mj; tri_index((int)blockIdx.x, mi, mj);
    update_wide_macro_dma<16>(band, wbuf, wstride, rowoff, k0, w, mi, mj, nrow, smem);
}
}
namespace {
__global__ void fill_kernel(double* p, size_t n, unsigned seed) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = (t + 1) * 0x9E3779B97F4A7C15ull + seed; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[t] = (double)(x & 0xFFFFF) / 524288.0 - 1.0;
    }
}
}
int main(int argc, char** argv) {
    const int nrow = argc > 1 ? atoi(argv[1]) : 150, w = argc > 2 ? atoi(argv[2]) : 8, reps = 10;
    const int nblk = nrow + w;
    std::vector<long long> tri(nblk + 1);
    for (int I = 0; I <= nblk; ++I) tri[I] = (long long)I * (I + 1) / 2;
    const size_t ntiles = (size_t)tri[nblk] + nblk;
    double *band, *wbuf; long long* d_tri;
    hipMalloc(&band, ntiles * NB2 * sizeof(double)); hipMalloc(&wbuf, (size_t)w * nblk * NB2 * sizeof(double)); hipMalloc(&d_tri, tri.size() * sizeof(long long));
    // random operands (zero-filled ones read high: the matrix pipe draws less power on zeros)
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, band, ntiles * NB2, 1u); hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, wbuf, (size_t)w * nblk * NB2, 2u);
    hipMemcpy(d_tri, tri.data(), tri.size() * sizeof(long long), hipMemcpyHostToDevice);
    const unsigned grid = (unsigned)((long long)nrow * (nrow + 1) / 2);
    const double flop = (double)grid * w * 2.0 * 64 * 64 * 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < reps; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("%-60s %8.3f ms  %6.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
    };
    printf("front: %d block rows behind a group of %d block columns, %u workgroups, %.2f Gflop per launch\n", nrow, w, grid, flop / 1e9);
    {   // the forms of the update on the same operands (products and sums of these 20-bit values are exact in FP64: any difference is an indexing error)
        std::vector<double> r0(ntiles * NB2), r1(ntiles * NB2);
        hipLaunchKernelGGL(wide_kernel<0>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); hipMemcpy(r0.data(), band, r0.size() * 8, hipMemcpyDeviceToHost);
        auto check = [&](const char* name, auto launch) {
            hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, band, ntiles * NB2, 1u);
            launch(); hipMemcpy(r1.data(), band, r1.size() * 8, hipMemcpyDeviceToHost);
            double e = 0, m = 0; for (size_t q = 0; q < r0.size(); ++q) { e = std::max(e, std::fabs(r0[q] - r1[q])); m = std::max(m, std::fabs(r0[q])); }
            printf("register-staged against %s: max difference %.2e of %.2e (%s)\n", name, e, m, e <= 1e-12 * m ? "ok" : "MISMATCH");
        };
        check("LDS-DMA through the tile function of the library (parts of HK k)", [&] { hipLaunchKernelGGL(wide_kernel<1>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
        check("LDS-DMA, parts of 16 k", [&] { hipLaunchKernelGGL(wide_dma_kernel<16>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
        check("LDS-DMA, parts of 8 k", [&] { hipLaunchKernelGGL(wide_dma_kernel<8>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
        { const int nm = (nrow + 1) / 2; check("LDS-DMA, 128 x 128 macro tiles", [&] { hipLaunchKernelGGL(wide_macro_kernel, dim3((unsigned)(nm * (nm + 1) / 2)), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); }); }
        check("LDS-DMA, whole tiles", [&] { hipLaunchKernelGGL(wide_dma_kernel<64>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, band, ntiles * NB2, 1u);
    }
    time("V0r update_wide_tile through registers", [&] { hipLaunchKernelGGL(wide_kernel<0>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    time("V4 update_wide_tile_dma (LDS-DMA, half stages, one barrier per stage)", [&] { hipLaunchKernelGGL(wide_kernel<1>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    time("V5 LDS-DMA, parts of 16 k, 32 KB of LDS (four workgroups per CU)", [&] { hipLaunchKernelGGL(wide_dma_kernel<16>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    time("V5b LDS-DMA, parts of 8 k, 16 KB of LDS (eight workgroups per CU)", [&] { hipLaunchKernelGGL(wide_dma_kernel<8>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    { const int nm = (nrow + 1) / 2; time("V10 LDS-DMA, 128 x 128 macro tiles (one tile per wave), parts of 16 k, 64 KB, two workgroups per CU", [&] { hipLaunchKernelGGL(wide_macro_kernel, dim3((unsigned)(nm * (nm + 1) / 2)), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); }); }
    time("V6 LDS-DMA, parts of 32 k, 64 KB of LDS (two workgroups per CU)", [&] { hipLaunchKernelGGL(wide_dma_kernel<32>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    time("V7 LDS-DMA, whole tiles, 128 KB of LDS (one workgroup per CU)", [&] { hipLaunchKernelGGL(wide_dma_kernel<64>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    time("V0 update_wide_kernel (library)", [&] { hipLaunchKernelGGL(update_wide_kernel, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow, 0x7fffffff); });
    time("V1 no operand loads in the loop", [&] { hipLaunchKernelGGL(ub_kernel<1>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    time("V2 no parking, no barriers (MFMA + LDS operand reads)", [&] { hipLaunchKernelGGL(ub_kernel<2>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    time("V3 MFMAs on register operands only", [&] { hipLaunchKernelGGL(ub_kernel<3>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    return 0;
}
