// Microbenchmark (tools/, not product): does a CU-masked stream keep a latency-bound one-workgroup kernel (the solver's diagonal-tile chain) at its own speed
// while an MFMA-heavy kernel (the wide trailing update) fills the rest of the GPU?  Round 3 dropped look-ahead in the factorisation because the chain ran at a
// third of its speed next to the update workgroups it shared CUs with.  hipExtStreamCreateWithCUMask gives the chain CUs of its own.
//   chain kernel: 1 workgroup of 256 threads, a dependent sequence of LDS round trips + FP64 FMAs (~50 us alone)
//   heavy kernel: 16 384 workgroups of back-to-back v_mfma_f64_16x16x4 (~1.5 ms alone)
// Prints the chain's duration (HIP events on its stream, 20 back-to-back launches) alone, beside the heavy kernel on plain streams, and with the two streams masked
// to disjoint CU sets (chain: the first 8 CUs of the mask words' low bits; heavy: the rest), and the heavy kernel's duration in each case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void chain_kernel(double* out, int steps) {
    __shared__ double s[256];
    double v = threadIdx.x * 1e-3;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int k = 0; k < steps; ++k) {
        v = v * 1.0000001 + s[(threadIdx.x * 7 + k) & 255];
        __syncthreads();
        s[threadIdx.x] = v * 0.5;
        __syncthreads();
    }
    out[threadIdx.x] = v;
}
__global__ __launch_bounds__(256) void heavy_kernel(double* out, int iters) {
    d4 acc[4] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
    double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    double *o1, *o2; CK(hipMalloc(&o1, 256 * 8)); CK(hipMalloc(&o2, (size_t)16384 * 256 * 8));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    printf("%d CUs\n", ncu);
    auto run = [&](hipStream_t sc, hipStream_t sh, bool with_heavy, const char* what) -> int {
        hipEvent_t c0, c1, h0, h1; CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1)); CK(hipEventCreate(&h0)); CK(hipEventCreate(&h1));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            if (with_heavy) { CK(hipEventRecord(h0, sh)); for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(heavy_kernel, dim3(16384), dim3(256), 0, sh, o2, 600); CK(hipEventRecord(h1, sh)); }
            CK(hipEventRecord(c0, sc));
            for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(256), 0, sc, o1, 400);
            CK(hipEventRecord(c1, sc));
            CK(hipDeviceSynchronize());
        }
        float tc = 0, th = 0; CK(hipEventElapsedTime(&tc, c0, c1)); if (with_heavy) CK(hipEventElapsedTime(&th, h0, h1));
        printf("%-70s chain %.1f us per launch%s", what, 1e3 * tc / 20, with_heavy ? "" : "\n");
        if (with_heavy) printf(", heavy %.2f ms per launch\n", th / 4);
        return 0;
    };
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    if (run(s1, s2, false, "chain alone")) return 1;
    {   // heavy alone
        hipEvent_t h0, h1; CK(hipEventCreate(&h0)); CK(hipEventCreate(&h1));
        CK(hipEventRecord(h0, s2)); for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(heavy_kernel, dim3(16384), dim3(256), 0, s2, o2, 600); CK(hipEventRecord(h1, s2)); CK(hipDeviceSynchronize());
        float th = 0; CK(hipEventElapsedTime(&th, h0, h1)); printf("%-70s heavy %.2f ms per launch\n", "heavy alone", th / 4);
    }
    if (run(s1, s2, true, "chain beside heavy, plain streams")) return 1;
    {
        hipStream_t p1; int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CK(hipStreamCreateWithPriority(&p1, hipStreamNonBlocking, hi));
        if (run(p1, s2, true, "chain on a high-priority stream beside heavy")) return 1;
    }
    for (int reserve : {8, 16, 32}) {
        std::vector<uint32_t> mc(words, 0u), mh(words, 0u);
        for (int cu = 0; cu < ncu; ++cu) (cu < reserve ? mc : mh)[cu / 32] |= 1u << (cu % 32);
        hipStream_t m1, m2;
        hipError_t e = hipExtStreamCreateWithCUMask(&m1, words, mc.data());
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); return 0; }
        CK(hipExtStreamCreateWithCUMask(&m2, words, mh.data()));
        char what[128]; snprintf(what, sizeof what, "chain on %d reserved CUs, heavy on the other %d (CU-masked streams)", reserve, ncu - reserve);
        if (run(m1, m2, true, what)) return 1;
        if (run(m1, m2, false, "   ... the masked chain stream alone")) return 1;
    }
    return 0;
}
