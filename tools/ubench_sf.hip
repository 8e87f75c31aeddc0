// Microbenchmark (tools/, not product): the row-side sum-factorised contraction step of gf_gauss_loop.hpp (SF), one wave per SIMD -- what do its parts cost on the
// FP64 pipe?  Per component (15 per Gauss-point group of the full pass): T formation (19 v_fmac_f64_dpp + 3 products), 5 v_mfma_f64_4x4x4 into X[3], 12 FMAs into
// the accumulators (72 doubles in arch VGPRs).  MODE bit 0: T formation, bit 1: the products, bit 2: the accumulation; MODE 8: the 16 x 16 x 4 form of rounds 2 - 4
// (T formation + 5 v_mfma_f64_16x16x4 per component).  Output: cycles per component (s_memtime of wave 0) and the time of a grid that fills every SIMD once.
// sf_asm_kernel: the same step with the products as inline assembly (VGPR results, explicit result gap) and with the accumulators parked in AGPRs -- what the
// product kernel runs (gf_gauss_loop.hpp); mfma_order_kernel: does the ORDER of v_mfma_f64_16x16x4 with AGPR accumulators matter (it does not: 78.5 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
template <int LANE> __device__ __forceinline__ void fmac_bcast(double& t, double g, double p) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(g), "v"(p), "n"(LANE));
}
__device__ __forceinline__ void gap5(double (&t)[5]) { asm volatile("s_nop 1" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4])); }
__device__ __forceinline__ void fence15(double (&g)[15]) {
    asm volatile("s_nop 1" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]), "+v"(g[8]), "+v"(g[9]), "+v"(g[10]), "+v"(g[11]), "+v"(g[12]), "+v"(g[13]), "+v"(g[14]));
}
struct AccReg { int lo, hi; };
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void acc_zero(AccReg& a) { asm("v_accvgpr_write_b32 %0, 0" : "=a"(a.lo)); asm("v_accvgpr_write_b32 %0, 0" : "=a"(a.hi)); }
__device__ __forceinline__ double acc_get(const AccReg& a) {
    int lo, hi;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(a.lo)); asm("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(a.hi));
    return __builtin_bit_cast(double, u2{(unsigned)lo, (unsigned)hi});
}
__device__ __forceinline__ void acc_put(AccReg& a, double v) {
    const u2 w = __builtin_bit_cast(u2, v);
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a.lo) : "v"((int)w.x)); asm("v_accvgpr_write_b32 %0, %1" : "=a"(a.hi) : "v"((int)w.y));
}
__device__ __forceinline__ double mfma4_first(double a, double b) { double d; asm("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ void mfma4_acc(double& x, double a, double b) { asm("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b)); }
template <int GAP> __device__ __forceinline__ void result_gap(double (&x)[3]) {
    if constexpr (GAP == 10) asm volatile("s_nop 7\n\ts_nop 1" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
    else if constexpr (GAP == 6) asm volatile("s_nop 5" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
    else asm volatile("s_nop 1" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); } }

template <int MODE>
__global__ __launch_bounds__(64) void sf_kernel(const double* __restrict__ in, double* __restrict__ out, int ngroups, long long* __restrict__ cyc) {
    __shared__ double pad[4800];                         // 38 KB: one wave per SIMD, as the element kernel
    const int l = threadIdx.x;
    pad[l] = in[l];
    double gR[15], pb[5], psi[3], F[3], bs[3][4];
    for (int k = 0; k < 15; ++k) gR[k] = in[64 + 15 * l + k];
    for (int k = 0; k < 5; ++k) pb[k] = in[1100 + 5 * l + k];
    for (int k = 0; k < 3; ++k) { psi[k] = in[1500 + 3 * l + k]; F[k] = in[1700 + 3 * l + k]; }
    for (int k = 0; k < 3; ++k) for (int s = 0; s < 4; ++s) bs[k][s] = __builtin_bit_cast(double, (unsigned long long)__builtin_amdgcn_readfirstlane((int)(k * 4 + s + 1)) | 0x3ff0000000000000ull);
    d4 acc[15];
    for (int q = 0; q < 15; ++q) acc[q] = d4{0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int g = 0; g < ngroups; ++g) {
        fence15(gR);
        static_for<15>([&](auto q_) {
            constexpr int q = decltype(q_)::value, I = q % 3, J = (q / 3) % 3;
            double tq[5] = {1.0, 1.0, 1.0, 1.0, 1.0};
            if constexpr (MODE & 1) {
                const double nq = pb[I] * pb[J];
                static_for<5>([&](auto m_) {
                    constexpr int m = decltype(m_)::value;
                    double t = m < 2 ? 0.0 : nq * psi[m - 2];
                    fmac_bcast<3 * m + I>(t, gR[J], pb[0]);
                    fmac_bcast<3 * m + I>(t, gR[3 + J], pb[1]);
                    if constexpr (m < 2) { fmac_bcast<3 * m + I>(t, gR[6 + J], pb[2]); fmac_bcast<3 * m + I>(t, gR[9 + J], pb[3]); fmac_bcast<3 * m + I>(t, gR[12 + J], pb[4]); }
                    tq[m] = t;
                });
                gap5(tq);
            }
            if constexpr (MODE == 8 || MODE == 9) {
                static_for<5>([&](auto m_) { constexpr int m = decltype(m_)::value; acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[m % 3], tq[m], acc[q], 0, 0, 0); });
            } else {
                double X[3] = {0.0, 0.0, 0.0};
                if constexpr (MODE & 2) {
                    X[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(F[1], tq[0], X[0], 0, 0, 0);
                    X[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(F[0], tq[1], X[1], 0, 0, 0);
                    X[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(F[0], tq[3], X[2], 0, 0, 0);
                    X[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(F[2], tq[2], X[0], 0, 0, 0);
                    X[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(F[1], tq[4], X[1], 0, 0, 0);
                } else { X[0] = tq[0] + tq[2]; X[1] = tq[1] + tq[4]; X[2] = tq[3]; }
                if constexpr (MODE & 4) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[q][s] = __builtin_fma(bs[2][s], X[2], __builtin_fma(bs[1][s], X[1], __builtin_fma(bs[0][s], X[0], acc[q][s])));
                } else acc[q][0] += X[0] + X[1] + X[2];
            }
        });
        gR[g & 7] += 1e-9;
    }
    const long long t1 = clock64();
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    double s = pad[(l * 7) & 63];
    for (int q = 0; q < 15; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 64 + l] = s;
}

// The ORDER of v_mfma_f64_16x16x4 with AGPR accumulators: ORD 0 five products in a row into the same accumulator (component outer, m inner: a dependent chain),
// ORD 1 the products of one m for NQ accumulators in a row (m outer: independent accumulators, what gauss_group's SF = 0 form and the p = 4 kernels issue),
// ORD 2 component outer over PAIRS of accumulators (two interleaved chains).
template <int ORD, int NQ>
__global__ __launch_bounds__(64) void mfma_order_kernel(const double* __restrict__ in, double* __restrict__ out, int ngroups, long long* __restrict__ cyc) {
    __shared__ double pad[4800];
    const int l = threadIdx.x;
    pad[l] = in[l];
    double a[5], b[5];
    for (int k = 0; k < 5; ++k) { a[k] = in[64 + 5 * l + k]; b[k] = in[700 + 5 * l + k]; }
    d4 acc[NQ];
    for (int q = 0; q < NQ; ++q) acc[q] = d4{0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int g = 0; g < ngroups; ++g) {
        if constexpr (ORD == 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int m = 0; m < 5; ++m) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[(m + q) % 5], acc[q], 0, 0, 0);
        } else if constexpr (ORD == 1) {
#pragma unroll
            for (int m = 0; m < 5; ++m)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[(m + q) % 5], acc[q], 0, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < NQ; q += 2)
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[(m + q) % 5], acc[q], 0, 0, 0);
                    if (q + 1 < NQ) acc[q + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[(m + q + 1) % 5], acc[q + 1], 0, 0, 0);
                }
        }
        a[g & 3] += 1e-9;
    }
    const long long t1 = clock64();
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    double s = pad[(l * 7) & 63];
    for (int q = 0; q < NQ; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 64 + l] = s;
}
template <int ORD, int NQ> void run_order(const char* what, const double* din, double* dout, long long* dc) {
    const int NG = 2000, NWG = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_order_kernel<ORD, NQ>), dim3(NWG), dim3(64), 0, 0, din, dout, NG, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long c = 0; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-78s %7.1f cycles per v_mfma_f64_16x16x4 (wave 0), %6.1f ns per product and SIMD (grid)\n", what, (double)c / NG / (5.0 * NQ), 1e6 * ms / NG / (5.0 * NQ));
}

// VAR 0: asm products (VGPR results), result gap GAP, accumulators in VGPRs.  VAR 1: accumulators parked in AGPRs, slot by slot (read, 3 FMAs, write back).
// VAR 2: parked in AGPRs, the four slots read first, the twelve FMAs as four independent chains, then written back.
template <int VAR, int GAP>
__global__ __launch_bounds__(64) void sf_asm_kernel(const double* __restrict__ in, double* __restrict__ out, int ngroups, long long* __restrict__ cyc) {
    __shared__ double pad[4800];
    const int l = threadIdx.x;
    pad[l] = in[l];
    double gR[15], pb[5], psi[3], F[3], bs[3][4];
    for (int k = 0; k < 15; ++k) gR[k] = in[64 + 15 * l + k];
    for (int k = 0; k < 5; ++k) pb[k] = in[1100 + 5 * l + k];
    for (int k = 0; k < 3; ++k) { psi[k] = in[1500 + 3 * l + k]; F[k] = in[1700 + 3 * l + k]; }
    for (int k = 0; k < 3; ++k) for (int s = 0; s < 4; ++s) bs[k][s] = __builtin_bit_cast(double, (unsigned long long)__builtin_amdgcn_readfirstlane((int)(k * 4 + s + 1)) | 0x3ff0000000000000ull);
    d4 acc[15];
    AccReg ar[15][4];
    for (int q = 0; q < 15; ++q) { acc[q] = d4{0, 0, 0, 0}; if constexpr (VAR != 0) for (int s = 0; s < 4; ++s) acc_zero(ar[q][s]); }
    __syncthreads();
    const long long t0 = clock64();
    for (int g = 0; g < ngroups; ++g) {
        fence15(gR);
        static_for<15>([&](auto q_) {
            constexpr int q = decltype(q_)::value, I = q % 3, J = (q / 3) % 3;
            double tq[5];
            const double nq = pb[I] * pb[J];
            static_for<5>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                double t = m < 2 ? 0.0 : nq * psi[m - 2];
                fmac_bcast<3 * m + I>(t, gR[J], pb[0]);
                fmac_bcast<3 * m + I>(t, gR[3 + J], pb[1]);
                if constexpr (m < 2) { fmac_bcast<3 * m + I>(t, gR[6 + J], pb[2]); fmac_bcast<3 * m + I>(t, gR[9 + J], pb[3]); fmac_bcast<3 * m + I>(t, gR[12 + J], pb[4]); }
                tq[m] = t;
            });
            gap5(tq);
            double X[3];
            X[0] = mfma4_first(F[1], tq[0]); X[1] = mfma4_first(F[0], tq[1]); X[2] = mfma4_first(F[0], tq[3]);
            mfma4_acc(X[0], F[2], tq[2]); mfma4_acc(X[1], F[1], tq[4]);
            result_gap<GAP>(X);
            if constexpr (VAR == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[q][s] = __builtin_fma(bs[2][s], X[2], __builtin_fma(bs[1][s], X[1], __builtin_fma(bs[0][s], X[0], acc[q][s])));
            } else if constexpr (VAR == 1) {
#pragma unroll
                for (int s = 0; s < 4; ++s) acc_put(ar[q][s], __builtin_fma(bs[2][s], X[2], __builtin_fma(bs[1][s], X[1], __builtin_fma(bs[0][s], X[0], acc_get(ar[q][s])))));
            } else {
                double v[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] = acc_get(ar[q][s]);
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int s = 0; s < 4; ++s) v[s] = __builtin_fma(bs[k][s], X[k], v[s]);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc_put(ar[q][s], v[s]);
            }
        });
        gR[g & 7] += 1e-9;
    }
    const long long t1 = clock64();
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    double s = pad[(l * 7) & 63];
    for (int q = 0; q < 15; ++q) { s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3]; if constexpr (VAR != 0) for (int k = 0; k < 4; ++k) s += acc_get(ar[q][k]); }
    out[blockIdx.x * 64 + l] = s;
}
template <int VAR, int GAP> void run_asm(const char* what, const double* din, double* dout, long long* dc) {
    const int NG = 2000, NWG = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((sf_asm_kernel<VAR, GAP>), dim3(NWG), dim3(64), 0, 0, din, dout, NG, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long c = 0; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-78s %7.1f cycles per component (wave 0), %6.1f ns per component and SIMD (grid)\n", what, (double)c / NG / 15.0, 1e6 * ms / NG / 15.0);
}

template <int MODE> void run(const char* what, const double* din, double* dout, long long* dc) {
    const int NG = 2000, NWG = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((sf_kernel<MODE>), dim3(NWG), dim3(64), 0, 0, din, dout, NG, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long c = 0; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-78s %7.1f cycles per component (wave 0), %6.1f ns per component and SIMD (grid)\n", what, (double)c / NG / 15.0, 1e6 * ms / NG / 15.0);
}

int main() {
    std::vector<double> h(4096);
    for (size_t k = 0; k < h.size(); ++k) h[k] = 0.001 * (double)((k * 37) % 101) - 0.05;
    double *din, *dout; long long* dc;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, 1024 * 64 * 8); hipMalloc(&dc, 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    run<1>("T formation alone (19 v_fmac_f64_dpp + 3 products + 2 moves)", din, dout, dc);
    run<2>("5 v_mfma_f64_4x4x4 alone", din, dout, dc);
    run<4>("12 FMAs into the accumulators alone", din, dout, dc);
    run<3>("T formation + 5 v_mfma_f64_4x4x4", din, dout, dc);
    run<6>("5 v_mfma_f64_4x4x4 + 12 FMAs", din, dout, dc);
    run<7>("T formation + 5 v_mfma_f64_4x4x4 + 12 FMAs (the SF step)", din, dout, dc);
    run_asm<0, 10>("SF step, products as inline assembly (VGPR results), gap 10, VGPR accumulators", din, dout, dc);
    run_asm<0, 6>("   ... gap 6", din, dout, dc);
    run_asm<0, 2>("   ... gap 2 (timing only)", din, dout, dc);
    run_asm<1, 10>("SF step, accumulators parked in AGPRs, slot by slot, gap 10", din, dout, dc);
    run_asm<2, 10>("SF step, accumulators parked in AGPRs, four slots together, gap 10", din, dout, dc);
    run_asm<2, 6>("   ... gap 6", din, dout, dc);
    run_order<0, 18>("v_mfma_f64_16x16x4, 18 AGPR accumulators: five products in a row per accumulator", din, dout, dc);
    run_order<1, 18>("   ... the 18 accumulators in turn (one m at a time)", din, dout, dc);
    run_order<1, 6>("   ... 6 accumulators in turn", din, dout, dc);
    run_order<2, 18>("   ... two accumulators alternating, five products each", din, dout, dc);
    run<8>("5 v_mfma_f64_16x16x4 alone", din, dout, dc);
    run<9>("T formation + 5 v_mfma_f64_16x16x4 (the step of rounds 2 - 4)", din, dout, dc);
    return 0;
}
