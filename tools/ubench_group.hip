// Microbenchmark (tools/, not product): the Gauss-point group step of the p = 3 element kernels (gf_gauss_loop.hpp: gauss_group) in isolation -- one wave per SIMD, the
// records of a 16-point element filled with smooth numbers (timing only, nothing is checked), four groups per "element", accumulators as the walking kernel keeps them.
// SF = 0: the 16 x 16 x 4 form of rounds 2 - 4; SF = 1 / 2: the row-side sum factorisation (polynomial / rational), accumulators parked in AGPRs.
#include <hip/hip_runtime.h>
#include <vector>
namespace gf { constexpr int GF_ASM_R_BIT = 1, GF_ASM_K_BIT = 2, GF_ASM_C_BIT = 4, GF_ASM_H_BIT = 8; }
#include "../include/goldfish_hip.h"
#include "../goldfish_amd/csrc/gf_kernels.hpp"
#include "../goldfish_amd/csrc/gf_gauss_loop.hpp"
#include <cstdio>
using namespace gf;

template <int SF, bool ALLF, int OFF = 0, bool PARKK = false>      // PARKK: the K tiles parked in AGPRs too; OFF: bit 0 no K, 1 no dR/dCP, 2 no dR/dh, 3 no body force (run-time flags of the non-ALLF instance)
__global__ __launch_bounds__(64) void group_kernel(const double* __restrict__ in, double* __restrict__ out, int nelem, long long* __restrict__ cyc) {
    constexpr int P = 3, P1 = 4, TS = 48;
    __shared__ __attribute__((aligned(16))) double s_im[16][IM_SIZE];
    __shared__ double s_tu[TS], s_tv[TS], s_pc[8], s_pad[1200];
    const int tid = threadIdx.x, x = tid & 15, kk = tid >> 4;
    for (int k = tid; k < 16 * IM_SIZE; k += 64) (&s_im[0][0])[k] = in[k % 4000] * 0.01 + 0.5;
    if (tid < TS) { s_tu[tid] = in[tid] + 0.3; s_tv[tid] = in[100 + tid] + 0.2; }
    if (tid < 8) s_pc[tid] = tid < 2 ? 1.0 + 0.3 * tid : 0.1 * tid;
    s_pad[tid] = 0.0;
    __syncthreads();
    const double* const pf = s_pc + 2; const double* const ppd = s_pc + 5;
    const RowLane L(x);
    gf_d4 accK[6], accC[9], accH[3], accB[3];
    AccReg aK[6][4], aC[9][4];
    for (int q = 0; q < 6; ++q) { accK[q] = gf_d4{0, 0, 0, 0}; if constexpr (SF != 0) for (int s = 0; s < 4; ++s) acc_init(aK[q][s]); }
    for (int q = 0; q < 9; ++q) { accC[q] = gf_d4{0, 0, 0, 0}; if constexpr (SF != 0) for (int s = 0; s < 4; ++s) acc_init(aC[q][s]); }
    for (int q = 0; q < 3; ++q) { accH[q] = gf_d4{0, 0, 0, 0}; accB[q] = gf_d4{0, 0, 0, 0}; }
    double accR[3] = {0.0, 0.0, 0.0};
    SfLane sfl; for (int k1 = 0; k1 < 3; ++k1) sfl.au[k1] = s_tu[(kk * 3 + k1) * P1 + (x & 3)];
    const int jub = x >> 2, sb = x & 3;
    const long long t0 = clock64();
    for (int e = 0; e < nelem; ++e) {
        sfl.rot = e & 3;
        const int jv = (sb - e) & 3;
        for (int grp = 0; grp < 4; ++grp) {
            const int gp = 4 * grp + kk;
            const double* im = s_im[gp];
            const double wq = im[IM_WQ];
            constexpr bool dK = !(OFF & 1), dC = !(OFF & 2), dH = !(OFF & 4), bf = !(OFF & 8);
            if constexpr (SF != 0 && PARKK) gauss_group<P, true, ALLF, SF>(L, im, wq, s_tu, s_tv, kk, grp, jub, jv, 1.0, dK, dC, dH, bf, pf, ppd, aK, aC, accH, accB, accR, sfl);
            else if constexpr (SF != 0) gauss_group<P, true, ALLF, SF>(L, im, wq, s_tu, s_tv, kk, grp, jub, jv, 1.0, dK, dC, dH, bf, pf, ppd, accK, aC, accH, accB, accR, sfl);
            else gauss_group<P, true, ALLF, SF>(L, im, wq, s_tu, s_tv, kk, grp, jub, jv, 1.0, dK, dC, dH, bf, pf, ppd, accK, accC, accH, accB, accR, sfl);
        }
        s_im[e & 15][(e * 7) % IM_SIZE] += 1e-9;
    }
    const long long t1 = clock64();
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    double s = accR[0] + accR[1] + accR[2] + s_pad[tid];
    for (int q = 0; q < 6; ++q) for (int k = 0; k < 4; ++k) s += (SF != 0 && PARKK) ? acc_get(aK[q][k]) : accK[q][k];
    for (int q = 0; q < 9; ++q) for (int k = 0; k < 4; ++k) s += SF != 0 ? acc_get(aC[q][k]) : accC[q][k];
    for (int q = 0; q < 3; ++q) for (int k = 0; k < 4; ++k) s += accH[q][k] + accB[q][k];
    out[blockIdx.x * 64 + tid] = s;
}

template <int SF, bool ALLF, int OFF = 0, bool PARKK = false> void run(const char* what, const double* din, double* dout, long long* dc) {
    const int NE = 200, NWG = 1024;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((group_kernel<SF, ALLF, OFF, PARKK>), dim3(NWG), dim3(64), 0, 0, din, dout, NE, dc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    long long c = 0; (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-70s %8.0f cycles per element (four groups; wave 0), %6.2f us per element and SIMD (grid)\n", what, (double)c / NE, 1e3 * ms / NE);
}

int main() {
    std::vector<double> h(4096);
    for (size_t k = 0; k < h.size(); ++k) h[k] = 0.001 * (double)((k * 37) % 101) - 0.05;
    double *din, *dout; long long* dc;
    (void)hipMalloc(&din, h.size() * 8); (void)hipMalloc(&dout, 1024 * 64 * 8); (void)hipMalloc(&dc, 8);
    (void)hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    run<0, true>("16 x 16 x 4 form (rounds 2 - 4), full pass", din, dout, dc);
    run<1, true>("row-side sum factorisation, polynomial patch, full pass", din, dout, dc);
    run<2, true>("row-side sum factorisation, rational patch, full pass", din, dout, dc);
    run<1, true, 0, true>("   ... polynomial, the K tiles parked in AGPRs too", din, dout, dc);
    run<2, true, 0, true>("   ... rational, the K tiles parked in AGPRs too", din, dout, dc);
    run<0, false>("16 x 16 x 4 form, run-time flags, everything", din, dout, dc);
    run<0, false, 3>("16 x 16 x 4 form: no K, no dR/dCP (front part + residual + dR/dh)", din, dout, dc);
    run<0, false, 7>("16 x 16 x 4 form: no K, no dR/dCP, no dR/dh", din, dout, dc);
    run<1, false>("SF polynomial, run-time flags, everything", din, dout, dc);
    run<1, false, 3>("SF polynomial: no K, no dR/dCP (front part + residual + dR/dh)", din, dout, dc);
    run<1, false, 7>("SF polynomial: no K, no dR/dCP, no dR/dh", din, dout, dc);
    run<1, false, 2>("SF polynomial: no dR/dCP", din, dout, dc);
    run<1, false, 1>("SF polynomial: no K", din, dout, dc);
    run<1, false, 8>("SF polynomial: no body force", din, dout, dc);
    return 0;
}
