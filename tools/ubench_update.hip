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

int main(int argc, char** argv) {
    const int nrow = argc > 1 ? atoi(argv[1]) : 150, w = argc > 2 ? atoi(argv[2]) : 8, reps = 10;
    const int nblk = nrow + w;
    std::vector<long long> tri(nblk + 1);
    for (int I = 0; I <= nblk; ++I) tri[I] = (long long)I * (I + 1) / 2;
    const size_t ntiles = (size_t)tri[nblk] + nblk;
    double *band, *wbuf; long long* d_tri;
    hipMalloc(&band, ntiles * NB2 * sizeof(double)); hipMalloc(&wbuf, (size_t)w * nblk * NB2 * sizeof(double)); hipMalloc(&d_tri, tri.size() * sizeof(long long));
    hipMemset(band, 0, ntiles * NB2 * sizeof(double)); hipMemset(wbuf, 0, (size_t)w * nblk * NB2 * sizeof(double));
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
    time("V0 update_wide_kernel (library)", [&] { hipLaunchKernelGGL(update_wide_kernel, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    time("V1 no operand loads in the loop", [&] { hipLaunchKernelGGL(ub_kernel<1>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    time("V2 no parking, no barriers (MFMA + LDS operand reads)", [&] { hipLaunchKernelGGL(ub_kernel<2>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    time("V3 MFMAs on register operands only", [&] { hipLaunchKernelGGL(ub_kernel<3>, dim3(grid), dim3(256), 0, 0, band, wbuf, (long long)nblk, d_tri, 0, w, nrow); });
    return 0;
}
