// gf_solver.hip -- C ABI (include/goldfish_solver.h): block-banded L D L^T factorisation of K and solves, on the device.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC gf_solver.hip -o ../libgoldfish_solver.so      (no library dependency)
//
// Storage: control points renumbered by the caller's bandwidth-reducing order; n = 3 ncp dofs padded to nblk tiles of NB = 64
// (identity on the padding); lower block SKYLINE: block row I keeps the tiles (I, I - d), d = 0 .. Tr[I], from its first coupled
// block column on (envelope made monotone, so that the rows that reach a block column are contiguous), dense 64 x 64 row-major at
// (rowoff[I] + d) NB^2.  The penalty coupling reaches 2 (p + 1) control-point rows across an interface but only p rows inside a
// patch: the skyline holds about half the tiles of the uniform band of round 2 and the factorisation does a third of its work.
// Right-looking factorisation over block columns k, three launches per k:
//   diag_kernel     A_kk = L_kk D_k L_kk^T in LDS (one workgroup), and L_kk^-1 (unit lower) for the panel and the solves
//   panel_kernel    W_ik = A_ik L_kk^-T,  L_ik = W_ik D_k^-1           (one workgroup per tile, 64^3 product on v_mfma_f64_16x16x4)
//   update_kernel   A_ij -= W_ik L_jk^T   for k < j <= i <= k + T       (one workgroup per tile, same product)
// Solves: block forward / backward substitution with the inverted diagonal tiles (one launch per block column), then
// iterative refinement with the block-CSR K.  Reference: the MUMPS solves of GOLDFISH/utils/opt_utils.py:156-209.
#include "gf_nd_symbolic.hpp"
#include <map>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>
#include "../../include/goldfish_solver.h"

static thread_local std::string g_serr;
static int sfail(const std::string& m) { g_serr = m; return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

namespace {

#ifndef GF_UPDATE_PREFETCH
#define GF_UPDATE_PREFETCH 1        // 2: operand tiles of two columns ahead in registers -- measured at C4: 0.2487 vs 0.2519 s per factorisation (within the lease-to-lease spread): the update does not wait for its loads
#endif
constexpr int NB = 64, NB2 = NB * NB, LS = 66;        // tile edge; LDS row stride (66 doubles: the MFMA operand reads are bank-conflict free)
typedef double d4 __attribute__((ext_vector_type(4)));

// ---- 64 x 64 tile product C (+)= A B^T on the FP64 matrix pipe.  A, B in LDS (row-major, stride LS).  Wave w of the four
//      owns rows 16 w .. 16 w + 15 of C: four 16 x 16 accumulators (one per column block), 16 k-steps of 4.
//      v_mfma_f64_16x16x4: A[i][k]: lane = i + 16 k,  B[k][j]: lane = j + 16 k,  D[i][j]: lane = j + 16 (i % 4), register i / 4.
__device__ __forceinline__ void tile_abt(const double* __restrict__ sA, const double* __restrict__ sB, d4 (&acc)[4], int wave, int lane, double sign) {
    const int l16 = lane & 15, kq = lane >> 4;
    const double* pa = sA + (16 * wave + l16) * LS + kq;
    const double* pb = sB + l16 * LS + kq;
#pragma unroll 4
    for (int k0 = 0; k0 < NB; k0 += 4) {
        const double a = sign * pa[k0];
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) acc[nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb[nj * 16 * LS + k0], acc[nj], 0, 0, 0);
    }
}
__device__ __forceinline__ void load_tile(const double* __restrict__ g, double* __restrict__ s, int tid) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = 2 * (tid + 256 * q), r = idx >> 6, c = idx & 63;
        const double2 v = *reinterpret_cast<const double2*>(g + idx);
        s[r * LS + c] = v.x; s[r * LS + c + 1] = v.y;
    }
}

// K (block CSR of libgoldfish_hip, original numbering) -> lower block band in the factorisation order; identity on the padding
// value of entry ((a, i), (b, j)), b = nb[ptr + k], for the factorisation: K itself, or -- general mode (rev != null: K need not be symmetric) -- its
// symmetric part (K + K^T) / 2, the transposed entry found through rev[ptr + k] = position of a in b's neighbour list
__device__ __forceinline__ double fact_value(const double* __restrict__ valK, const long long* __restrict__ nb_ptr, const int* __restrict__ rev, long long ptr, long long deg,
                                             long long k, int b, int i, int j) {
    const double v = valK[9 * ptr + (long long)i * 3 * deg + 3 * k + j];
    if (!rev) return v;
    const long long pb = nb_ptr[b], db = nb_ptr[b + 1] - pb;
    return 0.5 * (v + valK[9 * pb + (long long)j * 3 * db + 3 * (long long)rev[ptr + k] + i]);
}
__global__ void band_fill_kernel(long long ncp, const long long* __restrict__ nb_ptr, const int* __restrict__ nb, const int* __restrict__ newi, const int* __restrict__ rev,
                                 const double* __restrict__ valK, double* __restrict__ band, const long long* __restrict__ rowoff, long long n, long long npad) {
    const long long a = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (a < ncp) {
        const long long ptr = nb_ptr[a], deg = nb_ptr[a + 1] - ptr;
        const long long pa = newi[a];
        for (long long k = lane; k < deg; k += 64) {
            const int bcp = nb[ptr + k];
            const long long pb = newi[bcp];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const long long r = 3 * pa + i, c = 3 * pb + j;
                    if (r < c) continue;
                    const long long I = r >> 6, d = I - (c >> 6);
                    band[(size_t)(rowoff[I] + d) * NB2 + (r & 63) * NB + (c & 63)] = fact_value(valK, nb_ptr, rev, ptr, deg, k, bcp, i, j);
                }
        }
    }
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < npad - n) { const long long r = n + t, I = r >> 6; band[(size_t)rowoff[I] * NB2 + (r & 63) * NB + (r & 63)] = 1.0; }
}

// diagonal tile of block column k: L D L^T (lower, no pivoting) and the inverse of the unit lower factor.
// Round 3, end: register-resident.  The first version kept the tile in LDS and walked its trailing part with a loop of run-time bounds per elimination
// step (a dependent LDS read - FMA - LDS write chain per entry): 93 us per tile, the serial chain of every front.  Now thread (i, jg) holds row i, columns
// 16 jg .. 16 jg + 15 in registers; per step p the owners of column p publish it through LDS (zero at and above the pivot, so the finished columns and
// the rows above need no predicate), one barrier, 16 FMAs; the steps are unrolled (register indices are compile-time).  The inverse of the unit lower
// factor: column c by the four lanes 4 c .. 4 c + 3 (lane q4 keeps x_m, m = q4 mod 4, in registers), row by row with a full-width dot product against
// the row of L in LDS (zeros above the diagonal: no run-time bounds), rows unrolled.
__device__ __forceinline__ void diag_body_rows(double* __restrict__ band, double* __restrict__ linv, double* __restrict__ dval, const long long* __restrict__ rowoff, int k, double* __restrict__ stat) {
    constexpr int S1 = NB + 1;
    __shared__ double sL[NB * S1];
    __shared__ __attribute__((aligned(16))) double col[2][NB];
    __shared__ double sd[NB];
    const int tid = threadIdx.x, i = tid >> 2, jg = tid & 3;
    double* A = band + (size_t)rowoff[k] * NB2;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; c += 2) { const double2 v = *reinterpret_cast<const double2*>(A + i * NB + 16 * jg + c); a[c] = v.x; a[c + 1] = v.y; }
    // Round 4: the 64 elimination steps as FOUR trips of a loop over 16 unrolled steps (the register index of the pivot column, p % 16, is compile-time inside a
    // trip; which quarter of the threads owns the column, p / 16, is the trip counter).  Fully unrolled the kernel was 72 KB of straight-line code executed once per
    // tile -- more than the instruction cache -- and took 94 us for ~25 us of dependent arithmetic; the same for the inverse below (16 unrolled rows per trip).
#pragma unroll 1
    for (int pb = 0; pb < 4; ++pb) {
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
            const int p = 16 * pb + pp;
            double* cb = col[pp & 1];
            if (jg == pb) { cb[i] = i > p ? a[pp] : 0.0; if (i == p) sd[p] = a[pp]; }
            __syncthreads();
            const double lip = cb[i] / sd[p];
#pragma unroll
            for (int c = 0; c < 16; c += 2) {
                const double2 v = *reinterpret_cast<const double2*>(cb + 16 * jg + c);
                a[c] -= lip * v.x; a[c + 1] -= lip * v.y;
            }
        }
    }
    __syncthreads();
    // a[c] for j = 16 jg + c < i: a_ij at the time column j became final = L_ij d_j
    if (tid < NB) dval[(size_t)k * NB + tid] = sd[tid];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int j = 16 * jg + c;
        const double l = j < i ? a[c] / sd[j] : 0.0;
        sL[i * S1 + j] = l;                                   // strictly lower part of L (zero on and above the diagonal: the inverse below wants that)
        a[c] = j < i ? l : (j == i ? sd[i] : 0.0);            // what the band keeps: d on the diagonal, L below, zero above
    }
#pragma unroll
    for (int c = 0; c < 16; c += 2) *reinterpret_cast<double2*>(A + i * NB + 16 * jg + c) = double2{a[c], a[c + 1]};
    // smallest / largest |d| of the tile (singularity report)
    if (tid == 0) {
        double mn = 1e300, mx = 0.0;
        for (int p = 0; p < NB; ++p) { const double d = fabs(sd[p]); mn = fmin(mn, d); mx = fmax(mx, d); }
        stat[2 * k] = mn; stat[2 * k + 1] = mx;
    }
    __syncthreads();
    // inverse of the unit lower factor: x = column c of it, x_r = delta_rc - sum_{m < r} L_rm x_m
    const int c = tid >> 2, q4 = tid & 3;
    double xr[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) xr[t] = 0.0;
    double* Li = linv + (size_t)k * NB2;
#pragma unroll 1
    for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int r = 16 * rb + rr;
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int t = 0; t < 16; t += 2) { p0 += sL[r * S1 + 4 * t + q4] * xr[t]; p1 += sL[r * S1 + 4 * (t + 1) + q4] * xr[t + 1]; }
            double part = p0 + p1;
            part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
            const double xv = (r == c ? 1.0 : 0.0) - part;         // rows above c: L's row is zero there and so are the x_m: xv = 0
            if (q4 == rr % 4) {
                Li[r * NB + c] = xv;
#pragma unroll
                for (int g = 0; g < 4; ++g) if (rb == g) xr[4 * g + rr / 4] = xv;      // x_r lives in register r / 4 = 4 rb + rr / 4: compile-time per (g, rr)
            }
        }
    }
}

// ---- Round 5: the same tile BLOCKED by 16, block updates on the matrix pipe.  The row-per-thread kernel above spends its time in 64 + 64 dependent steps that each cross
//      the workgroup (LDS publish, barrier, an IEEE division, 16 FMAs): ~ 500 cycles per step.  Here the 64 steps are four 16 x 16 diagonal blocks factored INSIDE A WAVE
//      (lane = row, the pivot column reaches the other rows through v_fmac_f64_dpp row_newbcast: no LDS, no barrier, 120 DPP FMAs; the block's inverse by substitution with
//      lane = column, another 120; all four waves do it redundantly, so nobody waits for a broadcast), the panel below a block is one 16^3 product per wave with the block's
//      inverse (W = A M^T, L = W D^-1), the trailing blocks are 16^3 products (A_ij -= W_i L_j^T), two barriers per block step.  The inverse of the unit lower factor is
//      block substitution per block column, one wave per column, and stays in registers: the accumulator layout of v_mfma_f64_16x16x4 (D[i][j]: lane j + 16 (i % 4), register
//      i / 4) IS its B-operand layout for k-step register (B[k][j]: lane j + 16 (k % 4)), so S = sum_k L_ik M_kj feeds M_ij = -M_ii S directly and M_ij feeds the next block row.
//      Results differ from the row kernel's in the last bits (other summation order, reciprocal by v_rcp_f64 + two Newton steps); deterministic.
template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
// t += (value of g in lane LANE of this lane's 16-lane row) * p.  The DPP source must not have been written by a VALU instruction in the two preceding wait states
// (dpp_fence; tools/check_dpp_hazard.py checks the code object)
template <int LANE> __device__ __forceinline__ void fmac_bcast16(double& t, double g, double p) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(g), "v"(p), "n"(LANE));
}
__device__ __forceinline__ void dpp_fence(double& g) { asm volatile("s_nop 1" : "+v"(g)); }
constexpr int MS = 17;        // row stride of the 16 x 16 blocks kept in LDS
// diagonal block pb of the tile in sT: negL[m] = -l_im (lane = row i; 0 for i <= m), x[i] = (L^-1)_ic (lane = column c), myd / mydinv = d_i, 1 / d_i of the lane's row
__device__ __forceinline__ void diag16(const double* __restrict__ sT, int pb, int l16, double (&negL)[16], double (&x)[16], double& myd, double& mydinv) {
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { const double v = sT[(16 * pb + l16) * LS + 16 * pb + c]; a[c] = c <= l16 ? v : 0.0; }
    const double one = 1.0;
    sfor<0, 16>([&](auto P) {
        constexpr int p = decltype(P)::value;
        dpp_fence(a[p]);
        double dp = 0.0;
        fmac_bcast16<p>(dp, a[p], one);                         // the pivot, from lane p
        double r = __builtin_amdgcn_rcp(dp);
        double e = __builtin_fma(-dp, r, 1.0); r = __builtin_fma(r, e, r);
        e = __builtin_fma(-dp, r, 1.0); r = __builtin_fma(r, e, r);
        if (l16 == p) { myd = dp; mydinv = r; }
        const double nl = -(a[p] * r);                          // -l_ip for the rows below the pivot
        negL[p] = l16 > p ? nl : 0.0;
        sfor<p + 1, 16>([&](auto C) { constexpr int c = decltype(C)::value; fmac_bcast16<c>(a[c], a[p], nl); });      // a_ic -= l_ip a_cp (a_cp from lane c)
    });
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = l16 == i ? 1.0 : 0.0;
    asm volatile("s_nop 1" : "+v"(negL[0]), "+v"(negL[1]), "+v"(negL[2]), "+v"(negL[3]), "+v"(negL[4]), "+v"(negL[5]), "+v"(negL[6]), "+v"(negL[7]),
                             "+v"(negL[8]), "+v"(negL[9]), "+v"(negL[10]), "+v"(negL[11]), "+v"(negL[12]), "+v"(negL[13]), "+v"(negL[14]));
    sfor<0, 15>([&](auto M_) {
        constexpr int m = decltype(M_)::value;
        sfor<m + 1, 16>([&](auto I_) { constexpr int i = decltype(I_)::value; fmac_bcast16<i>(x[i], negL[m], x[m]); });                // x_i -= l_im x_m (l_im from lane i)
    });
}
// block column J of the inverse of the unit lower factor (one wave): M_JJ from sMb, M_iJ = -M_ii sum_{k = J}^{i - 1} L_ik M_kJ, written to Li (64 x 64, row-major; zero blocks above)
template <int J> __device__ __forceinline__ void inv_block_column(const double* __restrict__ sT, double (*sMb)[16 * MS], double* __restrict__ Li, int l16, int kq) {
    d4 Mk[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) Mk[J][ks] = sMb[J][(4 * ks + kq) * MS + l16];
    sfor<J + 1, 4>([&](auto I_) {
        constexpr int i = decltype(I_)::value;
        d4 S = {0, 0, 0, 0};
        sfor<J, i>([&](auto K_) {
            constexpr int kk = decltype(K_)::value;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) S = __builtin_amdgcn_mfma_f64_16x16x4f64(sT[(16 * i + l16) * LS + 16 * kk + 4 * ks + kq], Mk[kk][ks], S, 0, 0, 0);
        });
        d4 R = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) R = __builtin_amdgcn_mfma_f64_16x16x4f64(-sMb[i][l16 * MS + 4 * ks + kq], S[ks], R, 0, 0, 0);
        Mk[i] = R;
    });
    sfor<0, 4>([&](auto I_) {
        constexpr int i = decltype(I_)::value;
#pragma unroll
        for (int t = 0; t < 4; ++t) Li[(16 * i + 4 * t + kq) * NB + 16 * J + l16] = i < J ? 0.0 : Mk[i < J ? J : i][t];
    });
}
constexpr int SMEM_DOUBLES = 2 * NB * LS;      // LDS of every tile kernel: two operand tiles (67 584 B); the blocked diagonal tile needs 58 240 B of it
constexpr int DIAG_LDS_DOUBLES = NB * LS + (4 + 4 + 3) * 16 * MS + NB;
static_assert(DIAG_LDS_DOUBLES <= SMEM_DOUBLES, "diag_body_blocked lives in the tile kernels' LDS");
__device__ __forceinline__ void diag_body_blocked(double* __restrict__ band, double* __restrict__ linv, double* __restrict__ dval, const long long* __restrict__ rowoff, int k, double* __restrict__ stat,
                                                  double* __restrict__ smem) {
    double* sT = smem;
    double (*sMw)[16 * MS] = reinterpret_cast<double (*)[16 * MS]>(smem + NB * LS);
    double (*sMb)[16 * MS] = sMw + 4;
    double (*sW)[16 * MS] = sMb + 4;
    double* sd = smem + NB * LS + 11 * 16 * MS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
    double* A = band + (size_t)rowoff[k] * NB2;
    {
        const int i = tid >> 2, jg = tid & 3;
#pragma unroll
        for (int c = 0; c < 16; c += 2) { const double2 v = *reinterpret_cast<const double2*>(A + i * NB + 16 * jg + c); sT[i * LS + 16 * jg + c] = v.x; sT[i * LS + 16 * jg + c + 1] = v.y; }
    }
    __syncthreads();
    double negL[16], x[16], myd = 0.0, mydinv = 0.0;
#pragma unroll 1
    for (int pb = 0; pb < 4; ++pb) {
        diag16(sT, pb, l16, negL, x, myd, mydinv);
        if (kq == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sMw[wave][i * MS + l16] = x[i];
            if (wave == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) sMb[pb][i * MS + l16] = x[i];
                sd[16 * pb + l16] = myd;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int ib = pb + 1 + wave;
        if (ib < 4) {                                            // panel block (ib, pb): W = A M^T (kept for the trailing update), L = W D^-1 (in place)
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sT[(16 * ib + l16) * LS + 16 * pb + 4 * ks + kq], sMw[wave][l16 * MS + 4 * ks + kq], acc, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) { const int r = 4 * t + kq; sW[wave][r * MS + l16] = acc[t]; sT[(16 * ib + r) * LS + 16 * pb + l16] = acc[t] * mydinv; }
        }
        __syncthreads();
        if (wave == 0 && kq == 0) {                              // what the band keeps of the diagonal block: d on the diagonal, L below, zero above (nobody reads the block again)
#pragma unroll
            for (int c = 0; c < 16; ++c) sT[(16 * pb + l16) * LS + 16 * pb + c] = c < l16 ? -negL[c] : (c == l16 ? myd : 0.0);
        }
        const int nrem = 3 - pb, npairs = nrem * (nrem + 1) / 2;
        for (int n = wave; n < npairs; n += 4) {                 // trailing blocks (ib2 >= jb > pb): A -= W_ib2 L_jb^T
            int gi = 0, gj = n;
            while (gj > gi) { gj -= gi + 1; ++gi; }
            const int ib2 = pb + 1 + gi, jb = pb + 1 + gj;
            d4 acc;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = sT[(16 * ib2 + 4 * t + kq) * LS + 16 * jb + l16];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-sW[gi][l16 * MS + 4 * ks + kq], sT[(16 * jb + l16) * LS + 16 * pb + 4 * ks + kq], acc, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) sT[(16 * ib2 + 4 * t + kq) * LS + 16 * jb + l16] = acc[t];
        }
        __syncthreads();
    }
    double* Li = linv + (size_t)k * NB2;
    if (wave == 0) inv_block_column<0>(sT, sMb, Li, l16, kq);
    else if (wave == 1) inv_block_column<1>(sT, sMb, Li, l16, kq);
    else if (wave == 2) inv_block_column<2>(sT, sMb, Li, l16, kq);
    else inv_block_column<3>(sT, sMb, Li, l16, kq);
    if (tid < NB) dval[(size_t)k * NB + tid] = sd[tid];
    if (tid == 64) {                                             // smallest / largest |d| of the tile (singularity report)
        double mn = 1e300, mx = 0.0;
        for (int p = 0; p < NB; ++p) { const double d = fabs(sd[p]); mn = fmin(mn, d); mx = fmax(mx, d); }
        stat[2 * k] = mn; stat[2 * k + 1] = mx;
    }
    {
        const int i = tid >> 2, jg = tid & 3;
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            const bool up = jg > (i >> 4);
            *reinterpret_cast<double2*>(A + i * NB + 16 * jg + c) = double2{up ? 0.0 : sT[i * LS + 16 * jg + c], up ? 0.0 : sT[i * LS + 16 * jg + c + 1]};
        }
    }
}
#ifndef GF_DIAG_BLOCKED
#define GF_DIAG_BLOCKED 1
#endif
__device__ __forceinline__ void diag_body(double* __restrict__ band, double* __restrict__ linv, double* __restrict__ dval, const long long* __restrict__ rowoff, int k, double* __restrict__ stat,
                                          double* __restrict__ smem) {
#if GF_DIAG_BLOCKED
    diag_body_blocked(band, linv, dval, rowoff, k, stat, smem);
#else
    diag_body_rows(band, linv, dval, rowoff, k, stat);
#endif
}
#define GF_TILE_SMEM __shared__ __attribute__((aligned(16))) double smem[SMEM_DOUBLES]
#define GF_DIAG_SMEM __shared__ __attribute__((aligned(16))) double smem[DIAG_LDS_DOUBLES]
__global__ __launch_bounds__(256) void diag_kernel(double* __restrict__ band, double* __restrict__ linv, double* __restrict__ dval, const long long* __restrict__ rowoff, int k, double* __restrict__ stat) {
    GF_DIAG_SMEM;
    diag_body(band, linv, dval, rowoff, k, stat, smem);
}
// Round 5: the workgroup that applies the LAST update to a diagonal tile factors it on the spot (its update is in global memory, the tile kernels' LDS is free again) --
// the chain of a block column is panel -> narrow update (+ next diagonal tile) instead of diagonal tile -> panel -> narrow update: one dependent launch less per column.
struct DiagNext { double* linv; double* dval; double* stat; int on; };

// panel tile i = k + 1 + blockIdx.x: W = A_ik L_kk^-T (to wbuf), L_ik = W D_k^-1 (in place)
__device__ __forceinline__ void panel_body(double* __restrict__ band, const double* __restrict__ linv, const double* __restrict__ dval, double* __restrict__ wbuf, const long long* __restrict__ rowoff, int k, int g,
                                           double* __restrict__ smem) {
    double *sA = smem, *sB = smem + NB * LS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* A = band + (size_t)(rowoff[k + 1 + g] + (g + 1)) * NB2;
    load_tile(A, sA, tid); load_tile(linv + (size_t)k * NB2, sB, tid);
    __syncthreads();
    d4 acc[4] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
    tile_abt(sA, sB, acc, wave, lane, 1.0);
    double* W = wbuf + (size_t)g * NB2;
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
        const int c = 16 * nj + (lane & 15);
        const double di = 1.0 / dval[(size_t)k * NB + c];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int r = 16 * wave + 4 * rg + (lane >> 4);
            W[r * NB + c] = acc[nj][rg]; A[r * NB + c] = acc[nj][rg] * di;
        }
    }
}

// trailing tile (i, j), k < j <= i: A_ij -= W_ik L_jk^T
// linear index -> (gi >= gj) over a lower triangle of tiles
__device__ __forceinline__ void tri_index(int bidx, int& gi, int& gj) {
    gi = (int)((sqrt(8.0 * bidx + 1.0) - 1.0) * 0.5);
    while ((gi + 1) * (gi + 2) / 2 <= bidx) ++gi;
    while (gi * (gi + 1) / 2 > bidx) --gi;
    gj = bidx - gi * (gi + 1) / 2;
}
__device__ __forceinline__ void update_tile(double* __restrict__ band, const double* __restrict__ wbuf, const long long* __restrict__ rowoff, int k, int gi, int gj, double* __restrict__ smem) {
    double *sA = smem, *sB = smem + NB * LS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = k + 1 + gi, j = k + 1 + gj;
    load_tile(wbuf + (size_t)gi * NB2, sA, tid);
    load_tile(band + (size_t)(rowoff[j] + (gj + 1)) * NB2, sB, tid);
    double* C = band + (size_t)(rowoff[i] + (gi - gj)) * NB2;
    d4 acc[4];
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[nj][rg] = C[(16 * wave + 4 * rg + (lane >> 4)) * NB + 16 * nj + (lane & 15)];
    __syncthreads();
    tile_abt(sA, sB, acc, wave, lane, -1.0);
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) C[(16 * wave + 4 * rg + (lane >> 4)) * NB + 16 * nj + (lane & 15)] = acc[nj][rg];
}
// trailing update behind a GROUP of w block columns k0 .. k0 + w - 1 (all factored, panels W_c in wbuf + c wstride tiles): tile (i, j), i >= j >= k0 + w,
//     A_ij -= sum_c W_i,k0+c L_j,k0+c^T
// one read-modify-write of the target tile per group instead of per column (the single-column update moves 64 KB of HBM per 64^3 product: bandwidth bound);
// the next column's operand tiles are requested into registers before the MFMAs of the current one.
__device__ __forceinline__ void fetch_tile(const double* __restrict__ g, double2 (&r)[8], int tid) {
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = *reinterpret_cast<const double2*>(g + 2 * (tid + 256 * q));
}
__device__ __forceinline__ void park_tile(const double2 (&r)[8], double* __restrict__ s, int tid) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int idx = 2 * (tid + 256 * q), rr = idx >> 6, c = idx & 63; s[rr * LS + c] = r[q].x; s[rr * LS + c + 1] = r[q].y; }
}
__device__ __forceinline__ void update_wide_tile(double* __restrict__ band, const double* __restrict__ wbuf, long long wstride, const long long* __restrict__ rowoff, int k0, int w, int gi, int gj,
                                                 double* __restrict__ smem) {
    double *sA = smem, *sB = smem + NB * LS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = k0 + w + gi, j = k0 + w + gj;
    double* C = band + (size_t)(rowoff[i] + (gi - gj)) * NB2;
#if GF_UPDATE_PREFETCH == 2
    // operand tiles of TWO columns ahead in registers: the loads of column c + 2 are issued when column c's tiles have been parked, so that a load has the MFMAs of
    // two columns (2 x 4096 cycles of the wave) to arrive instead of one
    double2 ra[2][8], rb[2][8];
    auto fetch = [&](int c, double2 (&xa)[8], double2 (&xb)[8]) {
        const int k = k0 + c;
        fetch_tile(wbuf + (size_t)((long long)c * wstride + (i - (k + 1))) * NB2, xa, tid);
        fetch_tile(band + (size_t)(rowoff[j] + (j - k)) * NB2, xb, tid);
    };
    fetch(0, ra[0], rb[0]);
    if (w > 1) fetch(1, ra[1], rb[1]);
    d4 acc[4];
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[nj][rg] = C[(16 * wave + 4 * rg + (lane >> 4)) * NB + 16 * nj + (lane & 15)];
    for (int c = 0; c < w; c += 2) {
        park_tile(ra[0], sA, tid); park_tile(rb[0], sB, tid);
        __syncthreads();
        if (c + 2 < w) fetch(c + 2, ra[0], rb[0]);
        tile_abt(sA, sB, acc, wave, lane, -1.0);
        __syncthreads();
        if (c + 1 < w) {
            park_tile(ra[1], sA, tid); park_tile(rb[1], sB, tid);
            __syncthreads();
            if (c + 3 < w) fetch(c + 3, ra[1], rb[1]);
            tile_abt(sA, sB, acc, wave, lane, -1.0);
            __syncthreads();
        }
    }
#else
    double2 ra[8], rb[8];
    fetch_tile(wbuf + (size_t)(i - (k0 + 1)) * NB2, ra, tid);
    fetch_tile(band + (size_t)(rowoff[j] + (j - k0)) * NB2, rb, tid);
    d4 acc[4];
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[nj][rg] = C[(16 * wave + 4 * rg + (lane >> 4)) * NB + 16 * nj + (lane & 15)];
    for (int c = 0; c < w; ++c) {
        park_tile(ra, sA, tid); park_tile(rb, sB, tid);
        __syncthreads();
        if (c + 1 < w) {
            const int k = k0 + c + 1;
            fetch_tile(wbuf + (size_t)((c + 1) * wstride + (i - (k + 1))) * NB2, ra, tid);
            fetch_tile(band + (size_t)(rowoff[j] + (j - k)) * NB2, rb, tid);
        }
        tile_abt(sA, sB, acc, wave, lane, -1.0);
        __syncthreads();
    }
#endif
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) C[(16 * wave + 4 * rg + (lane >> 4)) * NB + 16 * nj + (lane & 15)] = acc[nj][rg];
}
// ---- Round 5: the same update with the operand tiles brought in by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write parking), in HALF tiles
//      of 32 k (two buffers of 32 KB: the same 64 KB of LDS, two workgroups per CU) so that the loads of stage s + 1 are in flight under the products of stage s and
//      a stage costs ONE barrier.  The DMA writes a wave's 64 x 16 bytes linearly (four rows of 256 B), so the bank-conflict-free image is made on the SOURCE side:
//      the lane that fills 16-byte slot p of row r fetches chunk p ^ (r & 15) of that row, and the operand reads apply the same XOR.  A lane reads both k of a chunk
//      with one ds_read_b128 (k-step t of a round takes component t: the k order inside a stage is permuted identically for both operands).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
// rows 16 wave .. 16 wave + 15 of one operand's part tile (k = HKT h .. HKT h + HKT - 1 of the 64 x 64 row-major tile g) -> dst (64 rows x HKT doubles): HKT / 8 pieces of 1 KB per wave
// AUX: cache policy bits of the load (0: default; 16 = sc1: the load does not hit in the CU's L1 -- for tiles that this workgroup has read, rewritten in place and reads again)
template <int HKT, int AUX = 0> __device__ __forceinline__ void glds_part(const double* __restrict__ g, double* __restrict__ dst, int h, int wave, int lane) {
    constexpr int CPR = HKT / 2, RPP = 64 / CPR, NP = 16 / RPP;   // 16-byte chunks per row, rows per piece, pieces per wave
    const int rr = lane / CPR, p = lane % CPR;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int r = 16 * wave + RPP * q + rr;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(g + r * NB + HKT * h + 2 * (p ^ (r & (CPR - 1)))), (lds_void_t*)(dst + (16 * wave + RPP * q) * HKT), 16, 0, AUX);
    }
}
// acc (+ / -)= sum over ncol pairs of 64 x 64 tiles (A_c, B_c) of A_c B_c^T; src(c, gA, gB) names pair c.  Stage s = (pair s / SPC, part s % SPC) lives in buffer s % 2.
// SCALE: the A operand of pair c is multiplied by sdv[64 c + k] on its way into the product (W = L D formed on the fly from the L tiles: "W-less" updates below)
template <int HKT, bool NEG, bool SCALE = false, int AUX = 0, int AUXB = AUX, class Src> __device__ __forceinline__ void mma_pairs_dma(Src&& src, int ncol, d4 (&acc)[4], double* __restrict__ smem, const double* __restrict__ sdv = nullptr) {
    constexpr int PART = NB * HKT, STAGE = 2 * PART, CPR = HKT / 2, SPC = NB / HKT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
    auto issue = [&](int s) {
        const double *gA, *gB; src(s / SPC, gA, gB);
        double* buf = smem + (s & 1) * STAGE;
        glds_part<HKT, AUX>(gA, buf, s % SPC, wave, lane);
        glds_part<HKT, AUXB>(gB, buf + PART, s % SPC, wave, lane);
    };
    issue(0);
    const int ns = SPC * ncol;
    for (int s = 0; s < ns; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of stage s have landed (and whatever the caller loaded before)
        __builtin_amdgcn_s_barrier();                             // ... everybody's; and everybody is done with stage s - 1, whose buffer the next loads overwrite
        asm volatile("" ::: "memory");
        if (s + 1 < ns) issue(s + 1);
        const double* bufA = smem + (s & 1) * STAGE + (16 * wave + l16) * HKT;
        const double* bufB = smem + (s & 1) * STAGE + PART + l16 * HKT;
#pragma unroll
        for (int rd = 0; rd < HKT / 8; ++rd) {
            const int off = 2 * ((4 * rd + kq) ^ (l16 & (CPR - 1)));
            double2 a = *reinterpret_cast<const double2*>(bufA + off);
            if constexpr (SCALE) {
                const double2 dv = *reinterpret_cast<const double2*>(sdv + 64 * (s / SPC) + HKT * (s % SPC) + 2 * (4 * rd + kq));
                a.x *= dv.x; a.y *= dv.y;
            }
            double2 b[4];
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) b[nj] = *reinterpret_cast<const double2*>(bufB + nj * 16 * HKT + off);
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) acc[nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -a.x : a.x, b[nj].x, acc[nj], 0, 0, 0);
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) acc[nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -a.y : a.y, b[nj].y, acc[nj], 0, 0, 0);
        }
    }
}
// assign: the target tile holds nothing yet (a Schur-block tile at the front's first wide update, "lazy S" below): C = - sum instead of C -= sum, the tile is not read
// WL ("W-less"): wbuf is the front's dval (d of block column k at wbuf + 64 k); the A operand of column k is the tile L_ik itself, scaled by d on its way into the
// product, instead of the copy W_ik = L_ik D_k the panel kernels otherwise keep -- the panels write one tile less, and A and B operands come from the same tiles
// (smem: the two stage buffers + 8 x 64 doubles for the d of the group's columns)
template <int HKT, bool WL = false, int AUX = 0, int AUXB = AUX> __device__ __forceinline__ void update_wide_tile_dma(double* __restrict__ band, const double* __restrict__ wbuf, long long wstride, const long long* __restrict__ rowoff, int k0, int w,
                                                                        int gi, int gj, double* __restrict__ smem, bool assign = false) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
    const int i = k0 + w + gi, j = k0 + w + gj;
    double* C = band + (size_t)(rowoff[i] + (gi - gj)) * NB2;
    d4 acc[4];
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[nj][rg] = assign ? 0.0 : C[(16 * wave + 4 * rg + kq) * NB + 16 * nj + l16];
    double* sdv = smem + 4 * NB * HKT;
    if constexpr (WL) {
        for (int q = tid; q < 64 * w; q += 256) sdv[q] = wbuf[(size_t)64 * k0 + q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the loop's barriers are raw s_barrier (they wait for the DMA, vmcnt): this wave's LDS writes must have landed before it passes the first one
    }
    mma_pairs_dma<HKT, true, WL, AUX, AUXB>([&](int c, const double*& gA, const double*& gB) {
        const int k = k0 + c;
        gA = WL ? band + (size_t)(rowoff[i] + (i - k)) * NB2 : wbuf + (size_t)((long long)c * wstride + (i - (k + 1))) * NB2;
        gB = band + (size_t)(rowoff[j] + (j - k)) * NB2;
    }, w, acc, smem, sdv);
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) C[(16 * wave + 4 * rg + kq) * NB + 16 * nj + l16] = acc[nj][rg];
}
// ---- 128 x 128 macro tiles: one workgroup updates the 2 x 2 target tiles (2 mi + {0, 1}, 2 mj + {0, 1}), one tile per WAVE (64 accumulator doubles per lane), from two W panels and two L
//      panels staged by LDS-DMA as above (parts of 16 k: 2 x 128 rows x 16 k per stage, two stages = 64 KB, two workgroups per CU).  Per tile product that is half the operand
//      bytes across the L2 <-> fabric interface (the wide updates move 368 KB per 64 KB target tile, profiles/r05_solver_traffic_before_xcd.txt), 8 instead of 20 ds_read_b128 and one
//      barrier instead of four per 32 MFMAs of a wave.  Tiles above the diagonal and beyond the last block row are not computed (their wave still loads and synchronises).
template <int HKT, bool WL = false> __device__ __forceinline__ void update_wide_macro_dma(double* __restrict__ band, const double* __restrict__ wbuf, long long wstride, const long long* __restrict__ rowoff, int k0, int w,
                                                                         int mi, int mj, int nrow, double* __restrict__ smem, int jassign = 0x7fffffff) {
    constexpr int PART = 2 * NB * HKT, STAGE = 2 * PART, CPR = HKT / 2, RPP = 64 / CPR, NP = 32 / RPP, SPC = NB / HKT;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, kq = lane >> 4, wi = wave >> 1, wj = wave & 1;
    const int gi = 2 * mi + wi, gj = 2 * mj + wj;
    const bool valid = gi < nrow && gj <= gi;
    int it[2], jt[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { it[t] = k0 + w + min(2 * mi + t, nrow - 1); jt[t] = k0 + w + min(2 * mj + t, nrow - 1); }
    double* C = band + (size_t)(rowoff[k0 + w + min(gi, nrow - 1)] + (valid ? gi - gj : 0)) * NB2;
    auto issue = [&](int s) {                                    // this wave's rows 32 wave .. 32 wave + 31 of the two 128-row operand parts of stage s
        const int c = s / SPC, h = s % SPC, k = k0 + c;
        double* buf = smem + (s & 1) * STAGE;
        const int rr = lane / CPR, p = lane % CPR;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int r = 32 * wave + RPP * q + rr, t = r >> 6, tr = r & 63;
            const double* gA = WL ? band + (size_t)(rowoff[it[t]] + (it[t] - k)) * NB2 : wbuf + (size_t)((long long)c * wstride + (it[t] - (k + 1))) * NB2;
            const double* gB = band + (size_t)(rowoff[jt[t]] + (jt[t] - k)) * NB2;
            __builtin_amdgcn_global_load_lds((glb_void_t*)(gA + tr * NB + HKT * h + 2 * (p ^ (r & (CPR - 1)))), (lds_void_t*)(buf + (32 * wave + RPP * q) * HKT), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void_t*)(gB + tr * NB + HKT * h + 2 * (p ^ (r & (CPR - 1)))), (lds_void_t*)(buf + PART + (32 * wave + RPP * q) * HKT), 16, 0, 0);
        }
    };
    double* sdv = smem + 2 * STAGE;                               // W-less: the d of the group's block columns (update_wide_tile_dma)
    if constexpr (WL) {
        for (int q = tid; q < 64 * w; q += 256) sdv[q] = wbuf[(size_t)64 * k0 + q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // as in update_wide_tile_dma
    }
    issue(0);
    d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[a][b][t] = (valid && k0 + w + gj < jassign) ? C[(16 * a + 4 * t + kq) * NB + 16 * b + l16] : 0.0;
    const int ns = SPC * w;
    for (int s = 0; s < ns; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 1 < ns) issue(s + 1);
        if (valid) {
            const double* bufA = smem + (s & 1) * STAGE + (64 * wi + l16) * HKT;
            const double* bufB = smem + (s & 1) * STAGE + PART + (64 * wj + l16) * HKT;
#pragma unroll
            for (int rd = 0; rd < HKT / 8; ++rd) {
                const int off = 2 * ((4 * rd + kq) ^ (l16 & (CPR - 1)));
                double2 av[4], bv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) av[a] = *reinterpret_cast<const double2*>(bufA + a * 16 * HKT + off);
                if constexpr (WL) {
                    const double2 dv = *reinterpret_cast<const double2*>(sdv + 64 * (s / SPC) + HKT * (s % SPC) + 2 * (4 * rd + kq));
#pragma unroll
                    for (int a = 0; a < 4; ++a) { av[a].x *= dv.x; av[a].y *= dv.y; }
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) bv[b] = *reinterpret_cast<const double2*>(bufB + b * 16 * HKT + off);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[a].x, bv[b].x, acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[a].y, bv[b].y, acc[a][b], 0, 0, 0);
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int t = 0; t < 4; ++t) C[(16 * a + 4 * t + kq) * NB + 16 * b + l16] = acc[a][b][t];
    }
}
// panel tile in the same form: W = A_ik L_kk^-T (to wbuf), L_ik = W D_k^-1 (in place: the tile's last part has landed in LDS before anything is stored)
template <int HKT, bool WL = false, int AUX = 0> __device__ __forceinline__ void panel_body_dma(double* __restrict__ band, const double* __restrict__ linv, const double* __restrict__ dval, double* __restrict__ wbuf,
                                                                  const long long* __restrict__ rowoff, int k, int g, double* __restrict__ smem) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double* A = band + (size_t)(rowoff[k + 1 + g] + (g + 1)) * NB2;
    d4 acc[4] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
    mma_pairs_dma<HKT, false, false, AUX>([&](int, const double*& gA, const double*& gB) { gA = A; gB = linv + (size_t)k * NB2; }, 1, acc, smem);
    double* W = WL ? nullptr : wbuf + (size_t)g * NB2;
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
        const int c = 16 * nj + (lane & 15);
        const double di = 1.0 / dval[(size_t)k * NB + c];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int r = 16 * wave + 4 * rg + (lane >> 4);
            if constexpr (!WL) W[r * NB + c] = acc[nj][rg];
            A[r * NB + c] = acc[nj][rg] * di;
        }
    }
}
// parts of 16 k: two buffers of 16 KB -- FOUR workgroups per CU (the kernels that call this keep their LDS and registers small: no fused diagonal tile).  One MI355X, 150 block
// rows behind 8 block columns, random operands (tools/ubench_update.hip, profiles/r05_ubench_update.txt): through registers 48.6 TFLOP/s; LDS-DMA with parts of 32 k (64 KB,
// two workgroups per CU) 56.0, parts of 16 k 62.9; whole tiles (128 KB, one workgroup per CU) 47.1
constexpr int HK = 16;
static_assert(4 * NB * HK * sizeof(double) <= 65536, "the LDS-DMA destination (M0 base + lane offset) reaches 64 KB: the whole-tile variant (128 KB) failed the index check of tools/ubench_update.hip");
#ifndef GF_UPDATE_DMA
#define GF_UPDATE_DMA 1        // 0: operand tiles through registers (update_wide_tile), the form of rounds 3 - 4
#endif
#ifndef GF_WLESS_BATCH
#define GF_WLESS_BATCH 1       // the level-batched small fronts (two-launch sub-groups, GF_SOLVER_BLOCKCHAIN bit 1) keep no W = L D panels: their updates scale L by d on the fly
#endif
#ifndef GF_WLESS_BIG
#define GF_WLESS_BIG 0         // 1: the large fronts too (panel_kernel is handed no W buffer, narrow / mid / wide / macro updates get the front's dval in the W slot): correct
                               // (same tests), and measured without gain -- C4, same box: 0.1968 s with, 0.1955 s without (profiles/r05_solver_wless_ab.txt): their panels are 4.6 GB
#endif
#ifndef GF_WIDE_WAVES
#define GF_WIDE_WAVES 4        // workgroups of a wide update per CU.  Three would leave 64 KB of LDS and a third of the registers of every CU free for the chain kernels of the
#endif                         // other fronts (panel 32 KB; narrow update + diagonal tile 58 KB): their launches then take 15 / 30 us instead of 65 - 190 us beside the wide updates, but the
                               // wide updates lose more than the chains gain -- C4, same box, median of 7: 0.2238 s with three, 0.2210 s with four (profiles/r05_solver_variants_ab.txt)
#if GF_UPDATE_DMA
#define GF_UPDATE_WIDE_TILE update_wide_tile_dma<HK>
#define GF_WIDE_SMEM __shared__ __attribute__((aligned(16))) double smem[4 * NB * HK + 8 * NB]
#define GF_WIDE_ATTR __attribute__((amdgpu_waves_per_eu(1, GF_WIDE_WAVES)))
#ifndef GF_LEAN_CHAIN
#define GF_LEAN_CHAIN 1        // 0: panel and narrow update through registers with two whole operand tiles in LDS (67.6 KB), the form of rounds 3 - 4
#endif
#if GF_LEAN_CHAIN
#define GF_PANEL_SMEM __shared__ __attribute__((aligned(16))) double smem[4 * NB * HK]
#define GF_PANEL_BODY panel_body_dma<HK>
#define GF_NARROW_SMEM __shared__ __attribute__((aligned(16))) double smem[DIAG_LDS_DOUBLES > 4 * NB * HK ? DIAG_LDS_DOUBLES : 4 * NB * HK]
#define GF_UPDATE_TILE(band, wbuf, rowoff, k, gi, gj, smem) update_wide_tile_dma<HK>(band, wbuf, 0, rowoff, k, 1, gi, gj, smem)
#else
#define GF_PANEL_SMEM GF_TILE_SMEM
#define GF_PANEL_BODY panel_body
#define GF_NARROW_SMEM GF_TILE_SMEM
#define GF_UPDATE_TILE update_tile
#endif
#else
#undef GF_LEAN_CHAIN
#define GF_LEAN_CHAIN 0
#define GF_UPDATE_WIDE_TILE update_wide_tile
#define GF_WIDE_SMEM GF_TILE_SMEM
#define GF_WIDE_ATTR
#define GF_PANEL_SMEM GF_TILE_SMEM
#define GF_PANEL_BODY panel_body
#define GF_NARROW_SMEM GF_TILE_SMEM
#define GF_UPDATE_TILE update_tile
#endif
__global__ __launch_bounds__(256) void panel_kernel(double* __restrict__ band, const double* __restrict__ linv, const double* __restrict__ dval, double* __restrict__ wbuf, const long long* __restrict__ rowoff, int k) {
    GF_PANEL_SMEM;
#if GF_UPDATE_DMA && GF_LEAN_CHAIN
    if (!wbuf) { panel_body_dma<HK, true>(band, linv, dval, nullptr, rowoff, k, (int)blockIdx.x, smem); return; }       // W-less front (the skyline keeps its W)
#endif
    GF_PANEL_BODY(band, linv, dval, wbuf, rowoff, k, (int)blockIdx.x, smem);
}

// the diagonal tile of block column kn has had its last update (by this workgroup, in global memory): factor it
__device__ __forceinline__ void diag_next(double* __restrict__ band, const long long* __restrict__ rowoff, int kn, const DiagNext& dn, double* __restrict__ smem) {
    __syncthreads();                                              // the update's stores (this workgroup's) are visible to all its threads; its LDS operands are dead
    diag_body(band, dn.linv, dn.dval, rowoff, kn, dn.stat, smem);
}
__global__ __launch_bounds__(256) void update_kernel(double* __restrict__ band, const double* __restrict__ wbuf, const long long* __restrict__ rowoff, int k, int ni, DiagNext dn) {
    GF_NARROW_SMEM;
    int gi, gj; tri_index((int)blockIdx.x, gi, gj);
    if (gi >= ni) return;
    GF_UPDATE_TILE(band, wbuf, rowoff, k, gi, gj, smem);
    if (dn.on && blockIdx.x == 0) diag_next(band, rowoff, k + 1, dn, smem);
}
// the same update restricted to the trailing columns k + 1 .. k + nin (the rest of a panel group): blockIdx = (row gi, column gj < nin)
__global__ __launch_bounds__(256) void update_narrow_kernel(double* __restrict__ band, const double* __restrict__ wbuf, const long long* __restrict__ rowoff, int k, DiagNext dn) {
    GF_NARROW_SMEM;
    if (blockIdx.x < blockIdx.y) return;
#if GF_UPDATE_DMA && GF_LEAN_CHAIN && GF_WLESS_BIG
    update_wide_tile_dma<HK, true>(band, wbuf, 0, rowoff, k, 1, (int)blockIdx.x, (int)blockIdx.y, smem);                  // wbuf: the front's dval
#else
    GF_UPDATE_TILE(band, wbuf, rowoff, k, (int)blockIdx.x, (int)blockIdx.y, smem);
#endif
    if (dn.on && blockIdx.x == 0 && blockIdx.y == 0) diag_next(band, rowoff, k + 1, dn, smem);       // tile (k + 1, k + 1): block column k was its last update inside the panel group
}
// ---- Round 5, last step: a SUB-GROUP of a panel group (sg <= 4 block columns ks .. ks + sg - 1) in TWO launches instead of two or three per block column.
//      subgroup_block (one workgroup per front): the sg x sg block triangle on the diagonal -- diagonal tile, the panels below it inside the triangle, their updates of the
//      triangle's remaining tiles, next diagonal tile, ... (right-looking, one tile operation after the other: with the 13 us diagonal tile that chain is ~ 90 us for four
//      columns; in round 3, with the 93 us tile, the same idea lost).  subgroup_row (one workgroup per block row below the triangle): the row's sg panels, each behind the
//      row's own lazy update from the sub-group's earlier columns (left-looking: W_i,c' is this workgroup's, L_k,c' the triangle's) -- a row's tiles are read once and its W / L
//      written once, where panel + narrow updates re-read and re-wrote them per column.  Both use the tile kernels' device functions; a workgroup sees its own global writes
//      behind __syncthreads().
// The tile operations of ONE workgroup hand tiles over through global memory, and a tile may be read, rewritten in place and read again (a panel tile: A_ik in, L_ik out, L_ik as an
// operand of the next update).  Found the hard way (round 5: one Newton solve in thirty rejected by its backward error, every second run of one test): the second LDS-DMA read of
// such a tile can be served by the line the FIRST read left in the CU's L1 -- the plain stores in between do not touch it.  The sub-group kernels therefore load with sc1
// (GF_SC1: the load does not hit in L1) where they read a tile again that they have read before through the DMA and rewritten: everything in subgroup_block (its cost is nothing),
// the A operand of subgroup_row's lazy update.  Ten of ten runs clean either way; an L1 invalidate behind every tile operation instead cost 10 %, sc1 on every load of both kernels 2 %.
#ifndef GF_SC1_BITS
#define GF_SC1_BITS 16             // -DGF_SC1_BITS=0 rebuilds the hazard (tests/test_gpu_fullsize.py::test_repeated_factorisations_of_a_quarter_of_c4_give_the_same_bits then fails)
#endif
constexpr int GF_SC1 = GF_SC1_BITS;
template <bool WL> __device__ __forceinline__ void subgroup_block(double* __restrict__ band, double* __restrict__ linv, double* __restrict__ dval, double* __restrict__ stat, const long long* __restrict__ tri,
                                               double* __restrict__ wbuf, long long wstride, int k0, int ks, int sg, int nblk_t, double* __restrict__ smem) {
    for (int c = 0; c < sg; ++c) {
        const int k = ks + c;
        diag_body(band, linv, dval, tri, k, stat, smem);
        __syncthreads();
        double* wb = wbuf + (size_t)(k - k0) * wstride * NB2;
        const int rin = min(ks + sg, nblk_t) - (k + 1);           // rows of the triangle below block column k
        if constexpr (WL) {                                       // (the d of block column k: dval + 64 k, written by diag_body above)
            for (int g = 0; g < rin; ++g) { panel_body_dma<HK, true, GF_SC1>(band, linv, dval, nullptr, tri, k, g, smem); __syncthreads(); }
            for (int gj = 0; gj < rin; ++gj)
                for (int gi = gj; gi < rin; ++gi) { update_wide_tile_dma<HK, true, GF_SC1>(band, dval, 0, tri, k, 1, gi, gj, smem); __syncthreads(); }
        } else {
            for (int g = 0; g < rin; ++g) { panel_body_dma<HK, false, GF_SC1>(band, linv, dval, wb, tri, k, g, smem); __syncthreads(); }
            for (int gj = 0; gj < rin; ++gj)
                for (int gi = gj; gi < rin; ++gi) { update_wide_tile_dma<HK, false, GF_SC1>(band, wb, 0, tri, k, 1, gi, gj, smem); __syncthreads(); }
        }
    }
}
template <bool WL> __device__ __forceinline__ void subgroup_row(double* __restrict__ band, const double* __restrict__ linv, const double* __restrict__ dval, const long long* __restrict__ tri,
                                             double* __restrict__ wbuf, long long wstride, int k0, int ks, int sg, int i, double* __restrict__ smem) {
    for (int c = 0; c < sg; ++c) {
        const int k = ks + c;
        if (c > 0) {                                              // A_ik -= sum over the sub-group's earlier columns k' of W_ik' L_kk'^T
            if constexpr (WL) update_wide_tile_dma<HK, true, GF_SC1, 0>(band, dval, 0, tri, ks, c, i - k, 0, smem);      // A: this workgroup's own L_ik' (read as A_ik', rewritten); B: the triangle's L_kk' (another launch)
            else GF_UPDATE_WIDE_TILE(band, wbuf + (size_t)(ks - k0) * wstride * NB2, wstride, tri, ks, c, i - k, 0, smem);
            __syncthreads();
        }
        if constexpr (WL) panel_body_dma<HK, true>(band, linv, dval, nullptr, tri, k, i - (k + 1), smem);
        else GF_PANEL_BODY(band, linv, dval, wbuf + (size_t)(k - k0) * wstride * NB2, tri, k, i - (k + 1), smem);
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void subgroup_block_kernel(double* __restrict__ band, double* __restrict__ linv, double* __restrict__ dval, double* __restrict__ stat, const long long* __restrict__ tri,
                                                             double* __restrict__ wbuf, long long wstride, int k0, int ks, int sg, int nblk_t) {
    GF_NARROW_SMEM;
    subgroup_block<false>(band, linv, dval, stat, tri, wbuf, wstride, k0, ks, sg, nblk_t, smem);
}
__global__ __launch_bounds__(256) GF_WIDE_ATTR void subgroup_row_kernel(double* __restrict__ band, const double* __restrict__ linv, const double* __restrict__ dval, const long long* __restrict__ tri,
                                                                        double* __restrict__ wbuf, long long wstride, int k0, int ks, int sg) {
    GF_WIDE_SMEM;
    subgroup_row<false>(band, linv, dval, tri, wbuf, wstride, k0, ks, sg, ks + sg + (int)blockIdx.x, smem);
}
// the same update restricted to the first ncol trailing block columns (blockIdx = (row gi, column gj < ncol)): before a SUB-GROUP of a panel group starts, its columns
// receive the products of all earlier panels of the group in one read-modify-write; the narrow updates then stay inside the sub-group.  A tile of the group's j-th column
// is rewritten 1 + (j mod 4) times instead of j times (groups of 8, sub-groups of 4): the narrow updates are bound by exactly that traffic.
__global__ __launch_bounds__(256) GF_WIDE_ATTR void update_mid_kernel(double* __restrict__ band, const double* __restrict__ wbuf, long long wstride, const long long* __restrict__ rowoff, int k0, int w) {
    GF_WIDE_SMEM;
    if (blockIdx.x < blockIdx.y) return;
#if GF_UPDATE_DMA && GF_LEAN_CHAIN && GF_WLESS_BIG
    update_wide_tile_dma<HK, true>(band, wbuf, 0, rowoff, k0, w, (int)blockIdx.x, (int)blockIdx.y, smem);
#else
    GF_UPDATE_WIDE_TILE(band, wbuf, wstride, rowoff, k0, w, (int)blockIdx.x, (int)blockIdx.y, smem);
#endif
}
// jassign: target tiles in block columns >= jassign are ASSIGNED (the front's Schur block at its first wide update; INT_MAX: none)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void update_wide_macro_kernel(double* __restrict__ band, const double* __restrict__ wbuf, long long wstride,
                                                                                                      const long long* __restrict__ rowoff, int k0, int w, int nrow, int jassign) {
    __shared__ __attribute__((aligned(16))) double smem[8 * NB * HK + 8 * NB];
    int mi, mj; tri_index((int)blockIdx.x, mi, mj);
    update_wide_macro_dma<HK, GF_LEAN_CHAIN && GF_WLESS_BIG>(band, wbuf, wstride, rowoff, k0, w, mi, mj, nrow, smem, jassign);
}
__global__ __launch_bounds__(256) GF_WIDE_ATTR void update_wide_kernel(double* __restrict__ band, const double* __restrict__ wbuf, long long wstride, const long long* __restrict__ rowoff, int k0, int w, int nrow,
                                                                       int jassign) {
    GF_WIDE_SMEM;
    int gi, gj; tri_index((int)blockIdx.x, gi, gj);
    if (gi >= nrow) return;
#if GF_UPDATE_DMA
    update_wide_tile_dma<HK, GF_LEAN_CHAIN && GF_WLESS_BIG>(band, wbuf, wstride, rowoff, k0, w, gi, gj, smem, k0 + w + gj >= jassign);
#else
    GF_UPDATE_WIDE_TILE(band, wbuf, wstride, rowoff, k0, w, gi, gj, smem);
#endif
}

// forward substitution, block column k: y_k = L_kk^-1 b_k (every workgroup; workgroup 0 keeps it), b_{k+g} -= L_{k+g,k} y_k (workgroup g >= 1)
__global__ __launch_bounds__(256) void fwd_kernel(const double* __restrict__ band, const double* __restrict__ linv, double* __restrict__ b, double* __restrict__ y, const long long* __restrict__ rowoff, int k) {
    __shared__ double sb[NB], sy[NB];
    const int tid = threadIdx.x, r = tid >> 2, q4 = tid & 3, g = blockIdx.x;
    if (tid < NB) sb[tid] = b[(size_t)k * NB + tid];
    __syncthreads();
    {
        const double* L = linv + (size_t)k * NB2 + r * NB + 16 * q4;
        double part = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) part += L[c] * sb[16 * q4 + c];
        part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
        if (q4 == 0) sy[r] = part;
    }
    __syncthreads();
    if (g == 0) { if (tid < NB) y[(size_t)k * NB + tid] = sy[tid]; return; }
    const double* L = band + (size_t)(rowoff[k + g] + g) * NB2 + r * NB + 16 * q4;
    double part = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) part += L[c] * sy[16 * q4 + c];
    part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
    if (q4 == 0) b[(size_t)(k + g) * NB + r] -= part;
}
// backward substitution, block column k: x_k = L_kk^-T z_k, z_{k-g} -= L_{k,k-g}^T x_k (workgroup g >= 1)
__global__ __launch_bounds__(256) void bwd_kernel(const double* __restrict__ band, const double* __restrict__ linv, double* __restrict__ z, double* __restrict__ x, const long long* __restrict__ rowoff, int k) {
    __shared__ double sz[NB], sx[NB], sp[4][NB];
    const int tid = threadIdx.x, c = tid & 63, rq = tid >> 6, g = blockIdx.x;
    if (tid < NB) sz[tid] = z[(size_t)k * NB + tid];
    __syncthreads();
    {
        const double* L = linv + (size_t)k * NB2 + (16 * rq) * NB + c;
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) part += L[r * NB] * sz[16 * rq + r];
        sp[rq][c] = part;
    }
    __syncthreads();
    if (tid < NB) sx[tid] = sp[0][tid] + sp[1][tid] + sp[2][tid] + sp[3][tid];
    __syncthreads();
    if (g == 0) { if (tid < NB) x[(size_t)k * NB + tid] = sx[tid]; return; }
    const double* L = band + (size_t)(rowoff[k] + g) * NB2 + (16 * rq) * NB + c;
    double part = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) part += L[r * NB] * sx[16 * rq + r];
    __syncthreads();
    sp[rq][c] = part;
    __syncthreads();
    if (tid < NB) z[(size_t)(k - g) * NB + tid] -= sp[0][tid] + sp[1][tid] + sp[2][tid] + sp[3][tid];
}

// ---- substitutions of a dense front, a GROUP of w <= 4 block columns per launch (one launch per block column is latency: 6 us each, 1 727 columns in the
//      large fronts of C4).  Every workgroup solves the group's w x w block triangle itself (a few 64 x 64 matrix-vector products on tiles that sit in L2:
//      cheaper than a second launch or a cross-workgroup hand-over), workgroup 0 keeps the result, workgroup g >= 1 applies it to one block row outside.
//      NR right-hand sides per launch (round 4): the sweeps read the factors once for all of them -- every tile entry is loaded once and multiplied into NR
//      sums (the sweeps are bound by the 73 GB of factors at C4, not by the arithmetic).  Vec<NR>: the NR vectors of one kind (one workspace per right-hand side).
template <int NR> struct Vec { double* p[NR]; };
template <int NR> __device__ __forceinline__ void mv_row(const double* __restrict__ L, const double (*v)[NB], int vs, int r, int q4, double (&out)[NR]) {      // (L v_j)_r, four lanes per row; v_j = v[j * vs]
    const double* Lr = L + r * NB + 16 * q4;
    double l[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) l[c] = Lr[c];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        double part = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) part += l[c] * v[j * vs][16 * q4 + c];
        part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
        out[j] = part;
    }
}
template <int NR>
__global__ __launch_bounds__(256) void fwd_group_kernel(const double* __restrict__ band, const double* __restrict__ linv, Vec<NR> b, Vec<NR> y, const long long* __restrict__ tri, int k0, int w) {
    __shared__ double sb[NR * 4][NB], sy[NR * 4][NB];                  // [j][block column of the group]
    const int tid = threadIdx.x, r = tid >> 2, q4 = tid & 3, g = blockIdx.x;
#pragma unroll
    for (int j = 0; j < NR; ++j) for (int q = tid; q < w * NB; q += 256) sb[4 * j + (q >> 6)][q & 63] = b.p[j][(size_t)k0 * NB + q];
    __syncthreads();
    double out[NR];
    for (int c = 0; c < w; ++c) {
        mv_row<NR>(linv + (size_t)(k0 + c) * NB2, sb + c, 4, r, q4, out);
        if (q4 == 0) {
#pragma unroll
            for (int j = 0; j < NR; ++j) sy[4 * j + c][r] = out[j];
        }
        __syncthreads();
        for (int c2 = c + 1; c2 < w; ++c2) {
            mv_row<NR>(band + (size_t)(tri[k0 + c2] + (c2 - c)) * NB2, sy + c, 4, r, q4, out);
            if (q4 == 0) {
#pragma unroll
                for (int j = 0; j < NR; ++j) sb[4 * j + c2][r] -= out[j];
            }
        }
        __syncthreads();
    }
    if (g == 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) for (int q = tid; q < w * NB; q += 256) y.p[j][(size_t)k0 * NB + q] = sy[4 * j + (q >> 6)][q & 63];
        return;
    }
    const int I = k0 + w + (g - 1);
    double part[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) part[j] = 0.0;
    for (int c = 0; c < w; ++c) {
        mv_row<NR>(band + (size_t)(tri[I] + (I - (k0 + c))) * NB2, sy + c, 4, r, q4, out);
#pragma unroll
        for (int j = 0; j < NR; ++j) part[j] += out[j];
    }
    if (q4 == 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) b.p[j][(size_t)I * NB + r] -= part[j];
    }
}
// (L^T v_j)_c over the 16 rows 16 rq .. 16 rq + 15 of a tile: partial sums of the four row quarters, to be added by the caller
template <int NR> __device__ __forceinline__ void mvt_part(const double* __restrict__ L, const double (*v)[NB], int vs, int c, int rq, double (&out)[NR]) {
    const double* Lc = L + (16 * rq) * NB + c;
    double l[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) l[rr] = Lc[rr * NB];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        double part = 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) part += l[rr] * v[j * vs][16 * rq + rr];
        out[j] = part;
    }
}
template <int NR>
__global__ __launch_bounds__(256) void bwd_group_kernel(const double* __restrict__ band, const double* __restrict__ linv, Vec<NR> z, Vec<NR> x, const long long* __restrict__ tri, int k0, int w) {
    __shared__ double sz[NR * 4][NB], sx[NR * 4][NB], sp[NR * 4][NB];  // sz, sx: [j][block column]; sp: [j][row quarter]
    const int tid = threadIdx.x, c = tid & 63, rq = tid >> 6, g = blockIdx.x;
#pragma unroll
    for (int j = 0; j < NR; ++j) for (int q = tid; q < w * NB; q += 256) sz[4 * j + (q >> 6)][q & 63] = z.p[j][(size_t)k0 * NB + q];
    __syncthreads();
    double out[NR];
    for (int cc = w - 1; cc >= 0; --cc) {
        mvt_part<NR>(linv + (size_t)(k0 + cc) * NB2, sz + cc, 4, c, rq, out);
#pragma unroll
        for (int j = 0; j < NR; ++j) sp[4 * j + rq][c] = out[j];
        __syncthreads();
        if (tid < NB) {
#pragma unroll
            for (int j = 0; j < NR; ++j) sx[4 * j + cc][tid] = sp[4 * j][tid] + sp[4 * j + 1][tid] + sp[4 * j + 2][tid] + sp[4 * j + 3][tid];
        }
        __syncthreads();
        for (int c2 = cc - 1; c2 >= 0; --c2) {
            mvt_part<NR>(band + (size_t)(tri[k0 + cc] + (cc - c2)) * NB2, sx + cc, 4, c, rq, out);
#pragma unroll
            for (int j = 0; j < NR; ++j) sp[4 * j + rq][c] = out[j];
            __syncthreads();
            if (tid < NB) {
#pragma unroll
                for (int j = 0; j < NR; ++j) sz[4 * j + c2][tid] -= sp[4 * j][tid] + sp[4 * j + 1][tid] + sp[4 * j + 2][tid] + sp[4 * j + 3][tid];
            }
            __syncthreads();
        }
    }
    if (g == 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) for (int q = tid; q < w * NB; q += 256) x.p[j][(size_t)k0 * NB + q] = sx[4 * j + (q >> 6)][q & 63];
        return;
    }
    const int J = k0 - g;
    double part[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) part[j] = 0.0;
    for (int cc = 0; cc < w; ++cc) {
        mvt_part<NR>(band + (size_t)(tri[k0 + cc] + (k0 + cc - J)) * NB2, sx + cc, 4, c, rq, out);
#pragma unroll
        for (int j = 0; j < NR; ++j) part[j] += out[j];
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) sp[4 * j + rq][c] = part[j];
    __syncthreads();
    if (tid < NB) {
#pragma unroll
        for (int j = 0; j < NR; ++j) z.p[j][(size_t)J * NB + tid] -= sp[4 * j][tid] + sp[4 * j + 1][tid] + sp[4 * j + 2][tid] + sp[4 * j + 3][tid];
    }
}

__global__ void permute_in_kernel(long long ncp, const int* __restrict__ newi, const double* __restrict__ src, double* __restrict__ dst) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 3 * ncp) dst[3 * (long long)newi[t / 3] + t % 3] = src[t];
}
__global__ void permute_out_kernel(long long ncp, const int* __restrict__ newi, const double* __restrict__ src, double* __restrict__ dst, int add) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 3 * ncp) { const double v = src[3 * (long long)newi[t / 3] + t % 3]; dst[t] = add ? dst[t] + v : v; }
}
__global__ void scale_kernel(long long n, const double* __restrict__ d, const double* __restrict__ y, double* __restrict__ z) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) z[t] = y[t] / d[t];
}
// r = b - K x (block CSR, original numbering): one wave per control point, fixed reduction order
__global__ __launch_bounds__(256) void residual_kernel(long long ncp, const long long* __restrict__ nb_ptr, const int* __restrict__ nb, const double* __restrict__ val,
                                                       const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ r) {
    const long long a = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (a >= ncp) return;
    const long long ptr = nb_ptr[a], deg = nb_ptr[a + 1] - ptr;
    const double* v = val + 9 * ptr;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (long long k = lane; k < deg; k += 64) {
        const long long col = 3LL * nb[ptr + k];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double xv = x[col + j];
            s0 += v[3 * k + j] * xv; s1 += v[3 * deg + 3 * k + j] * xv; s2 += v[6 * deg + 3 * k + j] * xv;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    if (lane == 0) { r[3 * a] = b[3 * a] - s0; r[3 * a + 1] = b[3 * a + 1] - s1; r[3 * a + 2] = b[3 * a + 2] - s2; }
}
// r = b - K^T x through the reverse index (general mode): entry ((b, j), (a, i)) lies in b's rows at a's position rev[ptr + k]
__global__ __launch_bounds__(256) void residual_t_kernel(long long ncp, const long long* __restrict__ nb_ptr, const int* __restrict__ nb, const int* __restrict__ rev,
                                                         const double* __restrict__ val, const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ r) {
    const long long a = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (a >= ncp) return;
    const long long ptr = nb_ptr[a], deg = nb_ptr[a + 1] - ptr;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (long long k = lane; k < deg; k += 64) {
        const long long bc = nb[ptr + k], pb = nb_ptr[bc], db = nb_ptr[bc + 1] - pb, kb = rev[ptr + k];
        const double* v = val + 9 * pb + 3 * kb;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double xv = x[3 * bc + j];
            s0 += v[j * 3 * db] * xv; s1 += v[j * 3 * db + 1] * xv; s2 += v[j * 3 * db + 2] * xv;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_down(s0, off, 64); s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    if (lane == 0) { r[3 * a] = b[3 * a] - s0; r[3 * a + 1] = b[3 * a + 1] - s1; r[3 * a + 2] = b[3 * a + 2] - s2; }
}
// reverse index of the (symmetric) block pattern: rev[ptr_a + k] = position of a in the neighbour list of b = nb[ptr_a + k]; -1 (and *bad = 1) if absent
__global__ void rev_index_kernel(long long nent, long long ncp, const long long* __restrict__ nb_ptr, const int* __restrict__ nb, int* __restrict__ rev, int* __restrict__ bad) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nent) return;
    long long lo = 0, hi = ncp - 1;                                   // row a of entry e: last a with nb_ptr[a] <= e
    while (lo < hi) { const long long mid = (lo + hi + 1) >> 1; if (nb_ptr[mid] <= e) lo = mid; else hi = mid - 1; }
    const int a = (int)lo, b = nb[e];
    int pos = -1;
    for (long long q = nb_ptr[b]; q < nb_ptr[b + 1]; ++q) if (nb[q] == a) { pos = (int)(q - nb_ptr[b]); break; }
    rev[e] = pos;
    if (pos < 0) *bad = 1;
}
// sum of squares in fixed order: per-block partials, summed on the host
__global__ __launch_bounds__(256) void sumsq_kernel(long long n, const double* __restrict__ v, double* __restrict__ part) {
    __shared__ double s[256];
    double acc = 0.0;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long long)gridDim.x * 256) acc += v[t] * v[t];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}


// ================================================================================================================================
// Nested-dissection multifrontal mode (goldfish_amd/_nd.py gives the fronts): every front is a dense symmetric matrix in the same lower
// tile layout as a skyline with a full envelope -- tile (I, J) at (tri[I] + I - J) NB^2, tri[I] = I (I + 1) / 2 -- so diag_kernel /
// panel_kernel / update_kernel / fwd_kernel / bwd_kernel run on it unchanged (band = the front's base, rowoff = tri).  A front
// eliminates its first nblk_e block columns; the trailing tiles then hold its Schur complement, which is added into the parent front.
struct Front { long long tile_off, kbase, elim_off, bnd_off; int nblk_e, nblk_t, ne_cp, nb_cp, ne_pad, parent, pad0, pad1; };

__device__ __forceinline__ int nd_dofpos(const Front& F, int p, int i) { return p < F.ne_cp ? 3 * p + i : F.ne_pad + 3 * (p - F.ne_cp) + i; }
__device__ __forceinline__ size_t nd_entry(const Front& F, const long long* __restrict__ tri, int R, int C) {      // R >= C
    const int I = R >> 6, J = C >> 6;
    return (size_t)(F.tile_off + tri[I] + (I - J)) * NB2 + (size_t)(R & 63) * NB + (C & 63);
}
// position of control point c in front t: its place among the eliminated ones, or behind them in the boundary list (sorted by elimination order)
__device__ __forceinline__ int nd_pos(const Front& F, int t, int c, const int* __restrict__ front_of, const long long* __restrict__ order, const int* __restrict__ bnd) {
    if (front_of[c] == t) return (int)(order[c] - F.elim_off);
    const long long oc = order[c];
    int lo = 0, hi = F.nb_cp - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (order[bnd[F.bnd_off + mid]] < oc) lo = mid + 1; else hi = mid; }
    return F.ne_cp + lo;
}
// K (block CSR, original numbering) -> fronts: the block (a, b) belongs to the front that eliminates the earlier of the two, and is stored
// there when a's position is not in front of b's (lower triangle); one wave per control point a
// row_ok (may be nullptr = every row holds values): a handle on ONE RANK'S K of a sharded model (gfs_set_row_mask) has values in the rows of the control points that
// rank owns only; the block (a, b) of a pair whose later control point b is a ghost there is then written from a's row, transposed (K is symmetric) -- a rank's
// subtrees never need a row another rank assembled.
__global__ void nd_scatter_kernel(long long ncp, const long long* __restrict__ nb_ptr, const int* __restrict__ nb, const int* __restrict__ rev, const double* __restrict__ valK,
                                  const Front* __restrict__ fronts, const int* __restrict__ front_of, const long long* __restrict__ order, const int* __restrict__ bnd,
                                  const long long* __restrict__ tri, double* __restrict__ arena, const unsigned char* __restrict__ row_ok) {
    const long long a = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (a >= ncp) return;
    if (row_ok && !row_ok[a]) return;
    const long long ptr = nb_ptr[a], deg = nb_ptr[a + 1] - ptr;
    const long long oa = order[a];
    for (long long k = lane; k < deg; k += 64) {
        const int b = nb[ptr + k];
        const int t = order[b] < oa ? front_of[b] : front_of[a];
        if (t < 0) continue;                        // partial handle (gfs_create_nd_partial): the entry belongs to a front of another handle
        const Front F = fronts[t];
        const int pa = nd_pos(F, t, (int)a, front_of, order, bnd), pb = nd_pos(F, t, b, front_of, order, bnd);
        if (pa < pb) {
            if (!row_ok || row_ok[b]) continue;     // b's own row writes the block
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {       // K[(b, j), (a, i)] = K[(a, i), (b, j)]: row b is not here, a's is
                    const int R = nd_dofpos(F, pb, j), C = nd_dofpos(F, pa, i);
                    arena[nd_entry(F, tri, R, C)] = fact_value(valK, nb_ptr, rev, ptr, deg, k, b, i, j);
                }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int R = nd_dofpos(F, pa, i), C = nd_dofpos(F, pb, j);
                if (R < C) continue;
                arena[nd_entry(F, tri, R, C)] = fact_value(valK, nb_ptr, rev, ptr, deg, k, b, i, j);
            }
    }
}
// identity on the padding of the eliminated part of every front (one workgroup per front)
// the Schur complement of a front after its partial factorisation (the boundary x boundary tiles of its arena) packed as the lower triangle of (nblk_t - nblk_e)
// block rows -- the layout of a stub front's arena: tile (I', J') at tri[I'] + (I' - J'); workgroup = one tile
__global__ __launch_bounds__(256) void nd_export_schur_kernel(Front F, const long long* __restrict__ tri, const double* __restrict__ arena, double* __restrict__ dst) {
    int gi, gj; tri_index((int)blockIdx.x, gi, gj);
    const double* src = arena + (size_t)(F.tile_off + tri[gi + F.nblk_e] + (gi - gj)) * NB2;
    double* out = dst + (size_t)(tri[gi] + (gi - gj)) * NB2;
    for (int q = threadIdx.x; q < NB2; q += 256) out[q] = src[q];
}
__global__ void nd_pad_kernel(const Front* __restrict__ fronts, const long long* __restrict__ tri, double* __restrict__ arena) {
    const Front F = fronts[blockIdx.x];
    for (int r = 3 * F.ne_cp + threadIdx.x; r < F.ne_pad; r += blockDim.x) arena[nd_entry(F, tri, r, r)] = 1.0;
}
// Schur complement of child c (its trailing tiles) added into its parent: one workgroup per lower tile of the child's boundary block
__global__ __launch_bounds__(256) void nd_extend_add_kernel(const Front* __restrict__ fronts, int c, const int* __restrict__ pmap, const long long* __restrict__ tri, double* __restrict__ arena) {
    const Front Fc = fronts[c]; const Front Fp = fronts[Fc.parent];
    int gi = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((gi + 1) * (gi + 2) / 2 <= (int)blockIdx.x) ++gi;
    while (gi * (gi + 1) / 2 > (int)blockIdx.x) --gi;
    const int gj = blockIdx.x - gi * (gi + 1) / 2;
    const int I = Fc.nblk_e + gi, J = Fc.nblk_e + gj;
    const double* src = arena + (size_t)(Fc.tile_off + tri[I] + (I - J)) * NB2;
    const int nbd = 3 * Fc.nb_cp;
    for (int q = threadIdx.x; q < NB2; q += 256) {
        const int rr = q >> 6, cc = q & 63;
        const int rl = 64 * gi + rr, cl = 64 * gj + cc;              // boundary dof indices of the child
        if (rl >= nbd || cl >= nbd || rl < cl) continue;
        const int pr = pmap[Fc.bnd_off + rl / 3], pc = pmap[Fc.bnd_off + cl / 3];
        const int R = nd_dofpos(Fp, pr, rl % 3), C = nd_dofpos(Fp, pc, cl % 3);       // the map is monotone: R >= C
        arena[nd_entry(Fp, tri, R, C)] += src[q];
    }
}
// ---- level-batched factorisation of the small fronts: the fronts of one tree height are independent, so block column k of ALL of them is one diag / panel /
//      update launch (blockIdx.y = front of the level's list, sorted by eliminated block columns so that the fronts that still have a column k are a prefix;
//      blockIdx.x beyond a front's own panel / trailing block exits) -- the diagonal tile's serial chain is paid per (height, k) instead of per (front, k).
__global__ __launch_bounds__(256) void nd_diag_batch_kernel(const Front* __restrict__ fronts, const int* __restrict__ list, const long long* __restrict__ tri, double* __restrict__ arena,
                                                            double* __restrict__ linv, double* __restrict__ dval, double* __restrict__ stat, int k) {
    GF_DIAG_SMEM;
    const Front F = fronts[list[blockIdx.x]];
    diag_body(arena + (size_t)F.tile_off * NB2, linv + (size_t)F.kbase * NB2, dval + (size_t)F.kbase * NB, tri, k, stat + 2 * F.kbase, smem);
}
// wofs: tile offset of the front's panel scratch (WP slots of nblk_t - 1 tiles each); c = k - k0: slot of block column k inside its panel group
__global__ __launch_bounds__(256) void nd_panel_batch_kernel(const Front* __restrict__ fronts, const int* __restrict__ list, const long long* __restrict__ wofs, const long long* __restrict__ tri,
                                                             double* __restrict__ arena, const double* __restrict__ linv, const double* __restrict__ dval, double* __restrict__ wbuf, int k, int c) {
    GF_PANEL_SMEM;
    const Front F = fronts[list[blockIdx.y]];
    if ((int)blockIdx.x >= F.nblk_t - 1 - k) return;
    GF_PANEL_BODY(arena + (size_t)F.tile_off * NB2, linv + (size_t)F.kbase * NB2, dval + (size_t)F.kbase * NB, wbuf + (size_t)(wofs[blockIdx.y] + (long long)c * (F.nblk_t - 1)) * NB2, tri, k,
               (int)blockIdx.x, smem);
}
// block column k updates the remaining columns of its panel group (k0 .. k0 + w - 1, w = min(WP, nblk_e - k0) per front): blockIdx = (row gi, column gj, front)
__global__ __launch_bounds__(256) void nd_update_narrow_batch_kernel(const Front* __restrict__ fronts, const int* __restrict__ list, const long long* __restrict__ wofs,
                                                                     const long long* __restrict__ tri, double* __restrict__ arena, const double* __restrict__ wbuf, int k, int k0, int WP,
                                                                     int send, DiagNext dn) {       // send: end of the sub-group of k, relative to k0
    GF_NARROW_SMEM;
    const Front F = fronts[list[blockIdx.z]];
    const int ni = F.nblk_t - 1 - k, w = min(min(WP, F.nblk_e - k0), send), nin = k0 + w - 1 - k;
    if ((int)blockIdx.x >= ni || (int)blockIdx.y >= nin || blockIdx.x < blockIdx.y) return;
    double* band = arena + (size_t)F.tile_off * NB2;
    GF_UPDATE_TILE(band, wbuf + (size_t)(wofs[blockIdx.z] + (long long)(k - k0) * (F.nblk_t - 1)) * NB2, tri, k, (int)blockIdx.x, (int)blockIdx.y, smem);
    if (dn.on && blockIdx.x == 0 && blockIdx.y == 0)
        diag_next(band, tri, k + 1, DiagNext{dn.linv + (size_t)F.kbase * NB2, dn.dval + (size_t)F.kbase * NB, dn.stat + 2 * F.kbase, 1}, smem);
}
// trailing update behind the panel group that starts at k0, all its block columns at once (update_wide_tile): one read-modify-write of a target tile per group
// instead of per column -- the single-column batched update was HBM bound (64 KB per 64^3 product: 143 of the 406 ms of a C4 factorisation)
// sub-group start inside the panel group at k0 (update_mid_kernel): wprev panels are done, the next ncol block columns get their products; blockIdx = (row gi, column gj, front)
__global__ __launch_bounds__(256) GF_WIDE_ATTR void nd_update_mid_batch_kernel(const Front* __restrict__ fronts, const int* __restrict__ list, const long long* __restrict__ wofs,
                                                                               const long long* __restrict__ tri, double* __restrict__ arena, const double* __restrict__ wbuf, int k0, int wprev, int ncol, int WP) {
    GF_WIDE_SMEM;
    const Front F = fronts[list[blockIdx.z]];
    const int w = min(WP, F.nblk_e - k0), nc = min(ncol, w - wprev), nrow = F.nblk_t - (k0 + wprev);
    if ((int)blockIdx.y >= nc || (int)blockIdx.x >= nrow || blockIdx.x < blockIdx.y) return;
#if GF_UPDATE_DMA && GF_WLESS_BATCH
    update_wide_tile_dma<HK, true>(arena + (size_t)F.tile_off * NB2, wbuf + (size_t)F.kbase * NB, 0, tri, k0, wprev, (int)blockIdx.x, (int)blockIdx.y, smem);      // wbuf: the handle's dval here
#else
    GF_UPDATE_WIDE_TILE(arena + (size_t)F.tile_off * NB2, wbuf + (size_t)wofs[blockIdx.z] * NB2, F.nblk_t - 1, tri, k0, wprev, (int)blockIdx.x, (int)blockIdx.y, smem);
#endif
}
// the sub-group kernels over the fronts of a tree height (blockIdx.x of the block kernel / blockIdx.y of the row kernel = front of the level's list)
__global__ __launch_bounds__(256) void nd_subgroup_block_batch_kernel(const Front* __restrict__ fronts, const int* __restrict__ list, const long long* __restrict__ wofs, const long long* __restrict__ tri,
                                                                      double* __restrict__ arena, double* __restrict__ linv, double* __restrict__ dval, double* __restrict__ stat, double* __restrict__ wbuf,
                                                                      int k0, int cs, int SG, int WP) {
    GF_NARROW_SMEM;
    const Front F = fronts[list[blockIdx.x]];
    const int sg = min(SG, min(WP, F.nblk_e - k0) - cs);
    if (sg <= 0) return;
    subgroup_block<GF_WLESS_BATCH != 0>(arena + (size_t)F.tile_off * NB2, linv + (size_t)F.kbase * NB2, dval + (size_t)F.kbase * NB, stat + 2 * F.kbase, tri, wbuf + (size_t)wofs[blockIdx.x] * NB2,
                                        F.nblk_t - 1, k0, k0 + cs, sg, F.nblk_t, smem);
}
__global__ __launch_bounds__(256) GF_WIDE_ATTR void nd_subgroup_row_batch_kernel(const Front* __restrict__ fronts, const int* __restrict__ list, const long long* __restrict__ wofs,
                                                                                 const long long* __restrict__ tri, double* __restrict__ arena, const double* __restrict__ linv,
                                                                                 const double* __restrict__ dval, double* __restrict__ wbuf, int k0, int cs, int SG, int WP) {
    GF_WIDE_SMEM;
    const Front F = fronts[list[blockIdx.y]];
    const int sg = min(SG, min(WP, F.nblk_e - k0) - cs), i = k0 + cs + sg + (int)blockIdx.x;
    if (sg <= 0 || i >= F.nblk_t) return;
    subgroup_row<GF_WLESS_BATCH != 0>(arena + (size_t)F.tile_off * NB2, linv + (size_t)F.kbase * NB2, dval + (size_t)F.kbase * NB, tri, wbuf + (size_t)wofs[blockIdx.y] * NB2, F.nblk_t - 1, k0, k0 + cs, sg, i,
                                      smem);
}
__global__ __launch_bounds__(256) GF_WIDE_ATTR void nd_update_wide_batch_kernel(const Front* __restrict__ fronts, const int* __restrict__ list, const long long* __restrict__ wofs,
                                                                   const long long* __restrict__ tri, double* __restrict__ arena, const double* __restrict__ wbuf, int k0, int WP, int lazy) {
    GF_WIDE_SMEM;
    const Front F = fronts[list[blockIdx.y]];
    const int w = min(WP, F.nblk_e - k0), nrow = F.nblk_t - (k0 + w);
    if ((long long)blockIdx.x >= (long long)nrow * (nrow + 1) / 2) return;
    int gi, gj; tri_index((int)blockIdx.x, gi, gj);
    double* band = arena + (size_t)F.tile_off * NB2;
#if GF_UPDATE_DMA
#if GF_WLESS_BATCH
    update_wide_tile_dma<HK, true>(band, wbuf + (size_t)F.kbase * NB, 0, tri, k0, w, gi, gj, smem, lazy && k0 == 0 && k0 + w + gj >= F.nblk_e);                        // wbuf: the handle's dval here
#else
    update_wide_tile_dma<HK>(band, wbuf + (size_t)wofs[blockIdx.y] * NB2, F.nblk_t - 1, tri, k0, w, gi, gj, smem, lazy && k0 == 0 && k0 + w + gj >= F.nblk_e);
#endif
#else
    GF_UPDATE_WIDE_TILE(band, wbuf + (size_t)wofs[blockIdx.y] * NB2, F.nblk_t - 1, tri, k0, w, gi, gj, smem);
#endif
}
// Schur complements of a list of children (no two of the same parent in one launch: one writer per entry) added into their parents
// part: 0 = every entry; 1 = the entries that land in the parent's ELIMINATED block columns (before the parent is factored); 2 = those that land in its Schur block
// (behind the parent's wide updates: "lazy S")
__global__ __launch_bounds__(256) void nd_extend_add_batch_kernel(const Front* __restrict__ fronts, const int* __restrict__ list, const int* __restrict__ pmap, const long long* __restrict__ tri,
                                                                  double* __restrict__ arena, int part) {
    const int c = list[blockIdx.y];
    const Front Fc = fronts[c]; const Front Fp = fronts[Fc.parent];
    const long long nbb = Fc.nblk_t - Fc.nblk_e;
    if ((long long)blockIdx.x >= nbb * (nbb + 1) / 2) return;
    int gi = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((gi + 1) * (gi + 2) / 2 <= (int)blockIdx.x) ++gi;
    while (gi * (gi + 1) / 2 > (int)blockIdx.x) --gi;
    const int gj = blockIdx.x - gi * (gi + 1) / 2;
    const int I = Fc.nblk_e + gi, J = Fc.nblk_e + gj;
    const double* src = arena + (size_t)(Fc.tile_off + tri[I] + (I - J)) * NB2;
    const int nbd = 3 * Fc.nb_cp;
    // the tile's 64 rows and 64 columns in the parent's numbering, once per tile (before round 4 every entry looked its two positions up again).  Measured without
    // effect on the C4 factorisation (0.2471 vs 0.2468 s): the kernel is bound by its read-modify-write traffic (24 bytes per entry), not by the index arithmetic
    __shared__ int sR[NB], sC[NB]; __shared__ long long sRow[NB];
    if (threadIdx.x < 2 * NB) {
        const int t = threadIdx.x & 63, l = 64 * (threadIdx.x < NB ? gi : gj) + t;
        int pos = -1;
        if (l < nbd) pos = nd_dofpos(Fp, pmap[Fc.bnd_off + l / 3], l % 3);
        if (threadIdx.x < NB) { sR[t] = pos; sRow[t] = pos >= 0 ? (Fp.tile_off + tri[pos >> 6] + (pos >> 6)) * (long long)NB2 + (long long)(pos & 63) * NB : 0; }
        else sC[t] = pos;
    }
    __syncthreads();
    const int nep = Fp.ne_pad;
    if (part) {                                                        // does any column of this tile land in the wanted part?  (the map is monotone in the tile's columns)
        int lo = 0x7fffffff, hi = -1;
        for (int cc = 0; cc < NB; ++cc) { const int C = sC[cc]; if (C >= 0) { lo = min(lo, C); hi = max(hi, C); } }
        if (part == 1 ? lo >= nep : hi < nep) return;
    }
    for (int q = threadIdx.x; q < NB2; q += 256) {
        const int rr = q >> 6, cc = q & 63;
        const int R = sR[rr], C = sC[cc];
        if (R < 0 || C < 0 || (gi == gj && rr < cc)) continue;          // the map is monotone: R >= C for an entry of the lower triangle
        if (part && (part == 1) != (C < nep)) continue;
        arena[sRow[rr] - (long long)(C >> 6) * NB2 + (C & 63)] += src[q];
    }
}
// "lazy S": the factor storage is cleared only where something is ADDED before it is written -- the tiles of a front's eliminated block columns (K's entries, the
// children's contributions); its Schur block is first written by its first wide update (assign) and receives the children's contributions behind it.  One workgroup per
// (front, block row): the row's tiles in eliminated columns are the LAST min(I + 1, nblk_e) of the row (tile (I, J) sits at tri[I] + I - J).
__global__ __launch_bounds__(256) void nd_zero_lpart_kernel(const Front* __restrict__ fronts, const int2* __restrict__ rows, const long long* __restrict__ tri, double* __restrict__ arena) {
    const int2 fr = rows[blockIdx.x];
    const Front F = fronts[fr.x];
    const int I = fr.y, nt = min(I + 1, F.nblk_e);
    double2* p = reinterpret_cast<double2*>(arena + (size_t)(F.tile_off + tri[I] + (I + 1 - nt)) * NB2);
    const long long n2 = (long long)nt * NB2 / 2;
    for (long long q = threadIdx.x; q < n2; q += 256) p[q] = double2{0.0, 0.0};
}
// front-local right-hand side: the eliminated dofs from the global vector (original numbering), zeros on the padding and the boundary part
template <int NR> __global__ void nd_gather_rhs_kernel(Front F, const int* __restrict__ elim, Vec<NR> b, Vec<NR> w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 64 * F.nblk_t) return;
    const long long src = t < 3 * F.ne_cp ? 3 * (long long)elim[F.elim_off + t / 3] + t % 3 : -1;
#pragma unroll
    for (int j = 0; j < NR; ++j) w.p[j][t] = src >= 0 ? b.p[j][src] : 0.0;
}
// the boundary updates a child left behind (fbnd, per front: 3 doubles per boundary control point) pulled into the parent's local vector: the
// parent adds its children one after the other (fixed order, no two writers: bitwise reproducible, and sibling subtrees may run concurrently)
template <int NR> __global__ void nd_pull_child_kernel(Front Fc, Front Fp, const int* __restrict__ pmap, Vec<NR> fbnd, Vec<NR> w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * Fc.nb_cp) return;
    const int dst = nd_dofpos(Fp, pmap[Fc.bnd_off + t / 3], t % 3);
#pragma unroll
    for (int j = 0; j < NR; ++j) w.p[j][dst] += fbnd.p[j][3 * Fc.bnd_off + t];
}
// after the forward substitution of a front: y of the eliminated dofs to the global y, the updated boundary part to the front's own buffer
template <int NR> __global__ void nd_scatter_fwd_kernel(Front F, const int* __restrict__ elim, Vec<NR> wy, Vec<NR> wb, Vec<NR> y, Vec<NR> fbnd) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 3 * F.ne_cp) {
        const long long dst = 3 * (long long)elim[F.elim_off + t / 3] + t % 3;
#pragma unroll
        for (int j = 0; j < NR; ++j) y.p[j][dst] = wy.p[j][t];
    }
    if (t < 3 * F.nb_cp) {
#pragma unroll
        for (int j = 0; j < NR; ++j) fbnd.p[j][3 * F.bnd_off + t] = wb.p[j][F.ne_pad + t];
    }
}
// before the backward substitution of a front: z = D^-1 y on the eliminated dofs, x of the boundary dofs (ancestors: already known)
template <int NR> __global__ void nd_gather_bwd_kernel(Front F, const int* __restrict__ elim, const int* __restrict__ bnd, Vec<NR> y, Vec<NR> x,
                                                       const double* __restrict__ dval, Vec<NR> wz, Vec<NR> wx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 64 * F.nblk_t) return;
    if (t < F.ne_pad) {
        const bool in = t < 3 * F.ne_cp;
        const long long src = in ? 3 * (long long)elim[F.elim_off + t / 3] + t % 3 : 0;
        const double d = in ? dval[F.kbase * NB + t] : 1.0;
#pragma unroll
        for (int j = 0; j < NR; ++j) { wz.p[j][t] = in ? y.p[j][src] / d : 0.0; wx.p[j][t] = 0.0; }
    } else {
        const int q = t - F.ne_pad; const bool in = q < 3 * F.nb_cp;
        const long long src = in ? 3 * (long long)bnd[F.bnd_off + q / 3] + q % 3 : 0;
#pragma unroll
        for (int j = 0; j < NR; ++j) { wx.p[j][t] = in ? x.p[j][src] : 0.0; wz.p[j][t] = 0.0; }
    }
}
// z_J -= sum over the boundary block rows I of L_IJ^T x_I  (workgroup J < nblk_e)
template <int NR>
__global__ __launch_bounds__(256) void nd_bwd_bnd_kernel(const double* __restrict__ band, const long long* __restrict__ tri, int nblk_e, int nblk_t, Vec<NR> wx, Vec<NR> wz) {
    __shared__ double sx[NR][NB], sp[NR * 4][NB];
    const int tid = threadIdx.x, c = tid & 63, rq = tid >> 6, J = blockIdx.x;
    double acc[NR], out[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[j] = 0.0;
    for (int I = nblk_e; I < nblk_t; ++I) {
        __syncthreads();
        if (tid < NB) {
#pragma unroll
            for (int j = 0; j < NR; ++j) sx[j][tid] = wx.p[j][(size_t)I * NB + tid];
        }
        __syncthreads();
        mvt_part<NR>(band + (size_t)(tri[I] + (I - J)) * NB2, sx, 1, c, rq, out);
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[j] += out[j];
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) sp[4 * j + rq][c] = acc[j];
    __syncthreads();
    if (tid < NB) {
#pragma unroll
        for (int j = 0; j < NR; ++j) wz.p[j][(size_t)J * NB + tid] -= sp[4 * j][tid] + sp[4 * j + 1][tid] + sp[4 * j + 2][tid] + sp[4 * j + 3][tid];
    }
}
// ---- whole-front substitutions for the small fronts (most block columns of a model sit in fronts of a few dozen blocks: one launch per block column
//      makes a solve launch bound).  One workgroup per front, the front-local vectors (NR right-hand sides) in LDS, the tiles streamed once; all fronts of one
//      tree height in one launch (they are independent), heights in ascending (forward) / descending (backward) order on one stream.
template <int NR>
__global__ __launch_bounds__(256) void nd_fwd_front_kernel(const Front* __restrict__ fronts, const int* __restrict__ list, const int* __restrict__ kid_off, const int* __restrict__ kid,
                                                           const long long* __restrict__ tri, const double* __restrict__ arena, const double* __restrict__ linv, const int* __restrict__ elim,
                                                           const int* __restrict__ pmap, Vec<NR> gb, Vec<NR> gy, Vec<NR> fbnd) {
    extern __shared__ double sw[];                                  // [NR][64 nblk_t] front-local right-hand sides, then sy [NR][64]
    const int t = list[blockIdx.x], tid = threadIdx.x;
    const Front F = fronts[t];
    if (F.ne_cp == 0) return;                                       // stub front of a partial handle: its boundary contribution comes from outside (gfs_set_fbnd)
    const int nloc = 64 * F.nblk_t;
    double* sy = sw + NR * nloc;
    for (int q = tid; q < nloc; q += 256) {
        const long long src = q < 3 * F.ne_cp ? 3 * (long long)elim[F.elim_off + q / 3] + q % 3 : -1;
#pragma unroll
        for (int j = 0; j < NR; ++j) sw[j * nloc + q] = src >= 0 ? gb.p[j][src] : 0.0;
    }
    __syncthreads();
    for (int ci = kid_off[t]; ci < kid_off[t + 1]; ++ci) {           // children one after the other: fixed order
        const Front Fc = fronts[kid[ci]];
        for (int q = tid; q < 3 * Fc.nb_cp; q += 256) {
            const int dst = nd_dofpos(F, pmap[Fc.bnd_off + q / 3], q % 3);
#pragma unroll
            for (int j = 0; j < NR; ++j) sw[j * nloc + dst] += fbnd.p[j][3 * Fc.bnd_off + q];
        }
        __syncthreads();
    }
    const double* band = arena + (size_t)F.tile_off * NB2;
    const int r = tid >> 2, q4 = tid & 3;
    for (int k = 0; k < F.nblk_e; ++k) {
        {
            const double* L = linv + (size_t)(F.kbase + k) * NB2 + r * NB + 16 * q4;
            double l[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) l[c] = L[c];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                double part = 0.0;
#pragma unroll
                for (int c = 0; c < 16; ++c) part += l[c] * sw[j * nloc + 64 * k + 16 * q4 + c];
                part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
                if (q4 == 0) sy[64 * j + r] = part;
            }
        }
        __syncthreads();
        for (int I = k + 1; I < F.nblk_t; ++I) {
            const double* L = band + (size_t)(tri[I] + (I - k)) * NB2 + r * NB + 16 * q4;
            double l[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) l[c] = L[c];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                double part = 0.0;
#pragma unroll
                for (int c = 0; c < 16; ++c) part += l[c] * sy[64 * j + 16 * q4 + c];
                part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
                if (q4 == 0) sw[j * nloc + 64 * I + r] -= part;
            }
        }
        if (tid < NB) {
            const int q = 64 * k + tid;
            if (q < 3 * F.ne_cp) {
                const long long dst = 3 * (long long)elim[F.elim_off + q / 3] + q % 3;
#pragma unroll
                for (int j = 0; j < NR; ++j) gy.p[j][dst] = sy[64 * j + tid];
            }
        }
        __syncthreads();
    }
    for (int q = tid; q < 3 * F.nb_cp; q += 256) {
#pragma unroll
        for (int j = 0; j < NR; ++j) fbnd.p[j][3 * F.bnd_off + q] = sw[j * nloc + F.ne_pad + q];
    }
}
template <int NR>
__global__ __launch_bounds__(256) void nd_bwd_front_kernel(const Front* __restrict__ fronts, const int* __restrict__ list, const long long* __restrict__ tri, const double* __restrict__ arena,
                                                           const double* __restrict__ linv, const double* __restrict__ dval, const int* __restrict__ elim, const int* __restrict__ bnd,
                                                           Vec<NR> gy, Vec<NR> gx) {
    extern __shared__ double sw[];                                  // [NR][64 nblk_t]: z on the eliminated blocks (becomes x), x on the boundary blocks; then sp [NR][4][64]
    const int t = list[blockIdx.x], tid = threadIdx.x;
    const Front F = fronts[t];
    if (F.ne_cp == 0) return;                                       // stub front: nothing to solve for
    const int nloc = 64 * F.nblk_t;
    double (*sp)[NB] = reinterpret_cast<double (*)[NB]>(sw + NR * nloc);
    for (int q = tid; q < nloc; q += 256) {
        if (q < F.ne_pad) {
            const bool in = q < 3 * F.ne_cp;
            const long long src = in ? 3 * (long long)elim[F.elim_off + q / 3] + q % 3 : 0;
            const double d = in ? dval[F.kbase * NB + q] : 1.0;
#pragma unroll
            for (int j = 0; j < NR; ++j) sw[j * nloc + q] = in ? gy.p[j][src] / d : 0.0;
        } else {
            const int qb = q - F.ne_pad; const bool in = qb < 3 * F.nb_cp;
            const long long src = in ? 3 * (long long)bnd[F.bnd_off + qb / 3] + qb % 3 : 0;
#pragma unroll
            for (int j = 0; j < NR; ++j) sw[j * nloc + q] = in ? gx.p[j][src] : 0.0;
        }
    }
    __syncthreads();
    const double* band = arena + (size_t)F.tile_off * NB2;
    const int c = tid & 63, rq = tid >> 6;
    // z_J -= sum over the rows I below (boundary rows, then the eliminated rows already solved) of L_IJ^T x_I, J descending; then x_J = L_JJ^-T z_J
    for (int J = F.nblk_e - 1; J >= 0; --J) {
        double acc[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[j] = 0.0;
        for (int I = J + 1; I < F.nblk_t; ++I) {
            const double* L = band + (size_t)(tri[I] + (I - J)) * NB2 + (16 * rq) * NB + c;
            double l[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) l[r] = L[r * NB];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j] += l[r] * sw[j * nloc + 64 * I + 16 * rq + r];
            }
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) sp[4 * j + rq][c] = acc[j];
        __syncthreads();
        if (tid < NB) {
#pragma unroll
            for (int j = 0; j < NR; ++j) sw[j * nloc + 64 * J + tid] -= sp[4 * j][tid] + sp[4 * j + 1][tid] + sp[4 * j + 2][tid] + sp[4 * j + 3][tid];
        }
        __syncthreads();
        {
            const double* L = linv + (size_t)(F.kbase + J) * NB2 + (16 * rq) * NB + c;
            double l[16], part[NR];
#pragma unroll
            for (int r = 0; r < 16; ++r) l[r] = L[r * NB];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                part[j] = 0.0;
#pragma unroll
                for (int r = 0; r < 16; ++r) part[j] += l[r] * sw[j * nloc + 64 * J + 16 * rq + r];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NR; ++j) sp[4 * j + rq][c] = part[j];
        }
        __syncthreads();
        if (tid < NB) {
#pragma unroll
            for (int j = 0; j < NR; ++j) sw[j * nloc + 64 * J + tid] = sp[4 * j][tid] + sp[4 * j + 1][tid] + sp[4 * j + 2][tid] + sp[4 * j + 3][tid];
        }
        __syncthreads();
    }
    for (int q = tid; q < 3 * F.ne_cp; q += 256) {
        const long long dst = 3 * (long long)elim[F.elim_off + q / 3] + q % 3;
#pragma unroll
        for (int j = 0; j < NR; ++j) gx.p[j][dst] = sw[j * nloc + q];
    }
}
template <int NR> __global__ void nd_scatter_bwd_kernel(Front F, const int* __restrict__ elim, Vec<NR> wx, Vec<NR> x) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * F.ne_cp) return;
    const long long dst = 3 * (long long)elim[F.elim_off + t / 3] + t % 3;
#pragma unroll
    for (int j = 0; j < NR; ++j) x.p[j][dst] = wx.p[j][t];
}

}  // namespace

struct gfs_handle {
    int device = 0; hipStream_t stream = nullptr;
    long long ncp = 0, n = 0, npad = 0, nblk = 0; int T = 0; long long bw = 0;       // T: largest number of tiles left of the diagonal in a block row
    std::vector<int> Tr, nik;                    // per block row: tiles left of the diagonal; per block column: block rows below that reach it
    long long* rowoff = nullptr; long long ntiles = 0;
    long long* nb_ptr = nullptr; int* nb = nullptr; int* newi = nullptr; const double* valK = nullptr;
    double *band = nullptr, *linv = nullptr, *dval = nullptr, *wbuf = nullptr, *stat = nullptr;
    double *vb = nullptr, *vy = nullptr, *vz = nullptr, *vx = nullptr, *vr = nullptr, *vsol = nullptr, *vrhs = nullptr, *part = nullptr;
    std::vector<void*> allocs; long long bytes = 0; bool factored = false, small_pivot = false;
    int* d_rev = nullptr; bool general = false;     // general mode (gfs_set_general): K need not be symmetric
    long long nnz9 = 0; double normK = 0.0, backward_error = 0.0;     // Frobenius norm of the factored K; backward error of the last solve
    // nested-dissection multifrontal mode
    bool nd = false; std::vector<Front> fronts; std::vector<std::vector<int>> kids; Front* d_fronts = nullptr;
    int *d_elim = nullptr, *d_bnd = nullptr, *d_pmap = nullptr, *d_front_of = nullptr; long long* d_order = nullptr; long long* d_tri = nullptr;
    long long nbe_tot = 0; int max_blk = 0; double nd_flops = 0.0;
    double *gy = nullptr, *gb = nullptr, *gx = nullptr, *fbnd = nullptr;
    // independent subtrees run on their own streams (their fronts are small: a single stream leaves the device idle); the fronts above
    // them ("top") follow on the main stream.  Per stream: W tiles of a panel, front-local vectors
    static constexpr int NS = 8;
    hipStream_t st[NS] = {}; hipEvent_t ev[NS] = {}, ev_main = nullptr, ev_norm = nullptr; double* part_k = nullptr;
    double *s_wbuf[NS + 1] = {}, *s_b[NS + 1] = {}, *s_y[NS + 1] = {}, *s_z[NS + 1] = {}, *s_x[NS + 1] = {};
    std::vector<std::vector<int>> sub;       // sub[s]: fronts of the subtrees assigned to stream s, in post-order
    std::vector<int> top;                    // the remaining fronts, in post-order
    // the sweeps are ~1e5 small launches with a fixed structure: captured once into HIP graphs (all streams), replayed per factorisation / substitution
    hipGraphExec_t g_factor = nullptr, g_solve = nullptr; bool use_graph = true;
    std::map<void*, int> graph_calls; int graph_after = 3;        // direct launches for the first graph_after calls of every sweep (GF_SOLVER_GRAPH_AFTER)
    // partial handle (gfs_create_nd_partial: a sub-forest of the elimination tree -- a rank's own subtrees, or the top of the tree above stub fronts that stand for the
    // subtrees of other ranks): Schur complements of stub fronts come from device buffers (stub_src, copied in by gfs_refactor), the sweeps run in halves
    bool partial = false; std::vector<const double*> stub_src; hipGraphExec_t g_fwd = nullptr, g_bwd = nullptr;
    // substitutions: fronts by tree height; the small ones of a height in one launch (nd_fwd_front_kernel / nd_bwd_front_kernel), the large ones per block column
    static constexpr int FUSE_MAX_BLK = 96;  // 64 x 96 doubles = 48 KB of LDS for the front-local vector
    struct Level { int off_small, n_small, max_blk; std::vector<int> big; };
    std::vector<Level> levels; int *d_lvl_list = nullptr, *d_kid_off = nullptr, *d_kid = nullptr;
    // factorisation by tree height: extend-add rounds (the r-th children of all fronts of the height), the small fronts column by column in batched launches,
    // the large fronts per block column on the side streams
    struct Round { int off, n; long long max_tiles; };
    struct FLevel { int off = 0, n = 0; std::vector<int> nk, max_ni; std::vector<Round> rounds; std::vector<int> big; };
    std::vector<FLevel> flevels; int *d_flist = nullptr, *d_ealist = nullptr; long long* d_fwofs = nullptr; double* bwbuf = nullptr; int batch_blk = 96, batch_panel_w = 8, panel_w = 8;       // panel_w: block columns per trailing update (C4: 4 -> 8: 0.257 -> 0.249 s, the target tile is read and written once per group)
    // substitution workspaces: [0] aliases the handle's own buffers and stream; [1 ..] are created by the first multi-right-hand-side solve, one stream each, so
    // that the sweeps of several right-hand sides (latency-bound chains of small launches) run next to each other (gfs_solve_multi)
    struct SolveWs { hipStream_t stream = nullptr; double *gb = nullptr, *gy = nullptr, *gx = nullptr, *fbnd = nullptr, *sb = nullptr, *sy = nullptr, *sz = nullptr, *sx = nullptr,
                     *vr = nullptr, *vsol = nullptr, *vrhs = nullptr, *part = nullptr; hipGraphExec_t g_solve[3] = {nullptr, nullptr, nullptr};
                     // round 5: the large fronts of one tree height run side by side on the handle's side streams -- front-local vectors per side stream, fork / join events
                     double* big = nullptr; long long big_len = 0; hipEvent_t ev_fork = nullptr, ev_join[8] = {};
                     double* scratch(int set, int which) const { return set < 0 ? (which == 0 ? sb : which == 1 ? sy : which == 2 ? sz : sx) : big + (size_t)(4 * set + which) * (size_t)big_len; } };
    static constexpr int RHS_BLOCK = 3;      // right-hand sides per pass over the factors (the front-local vectors of the small fronts sit in LDS: 3 x 48 KB)
    static constexpr int MAX_RHS = 8;
    std::vector<SolveWs> ws; long long ws_front_len = 0, ws_bnd_len = 0;
    unsigned char* d_row_ok = nullptr;            // gfs_set_row_mask: rows of d_valK that hold values (a rank's own rows of a sharded K); nullptr = all
    bool sweep_streams = true;                    // GF_SOLVER_SWEEP_STREAMS=0: the large fronts of a substitution one after the other on the sweep's stream
    bool prepared = false;                        // gfs_prepare_refactor has cleared the factor storage for the next gfs_refactor
    bool lazy_s = true; int2* d_zrows = nullptr; long long n_zrows = 0;      // GF_SOLVER_LAZY_S=0: the whole factor storage is cleared and every Schur block is read-modified-written from the start
    int block_chain = 2;                      // GF_SOLVER_BLOCKCHAIN (bit 0: large fronts, bit 1: level-batched small fronts): two launches per sub-group (subgroup_block / subgroup_row) instead of diagonal tile / panel /
                                                  // narrow update per block column.  C4, same box: none 0.2133 s, small fronts only 0.2115 s (default), large fronts only 0.2172 s (their triangle's sixteen tile
                                                  // operations run one after the other in one workgroup, where the per-column launches spread them over the device): profiles/r05_solver_blockchain_ab.txt
    int macro_min_rows = 64;                      // GF_SOLVER_MACRO_ROWS: wide updates of the large fronts with at least this many block rows run in 128 x 128 macro tiles (alone the
                                                  // two forms are equal from 100 block rows on and the macro form loses below 60; C4: 0.2178 -> 0.2152 s; profiles/r05_solver_macro_ab.txt)
    int subgroup = 4;                             // GF_SOLVER_SUBGROUP: block columns per sub-group of a panel group (0: none)
    bool fuse_diag = true;                        // GF_SOLVER_FUSE_DIAG=0: every diagonal tile in a launch of its own (the chain before round 5)
    bool lds_raised[3] = {false, false, false};   // hipFuncAttributeMaxDynamicSharedMemorySize of the NR-right-hand-side sweep kernels raised on this handle's device
    template <class Tp> Tp* dalloc(size_t cnt) {
        void* p = nullptr; const size_t nb_ = (cnt ? cnt : 1) * sizeof(Tp);
        HIPCHK(hipMalloc(&p, nb_)); allocs.push_back(p); bytes += (long long)nb_; return (Tp*)p;
    }
    template <class Tp> Tp* up(const Tp* src, size_t cnt) { Tp* p = dalloc<Tp>(cnt); if (cnt) HIPCHK(hipMemcpy(p, src, cnt * sizeof(Tp), hipMemcpyHostToDevice)); return p; }
};

static double norm2(gfs_handle* h, const double* v, long long n) {
    const int nblk = 240;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(256), 0, h->stream, n, v, h->part);
    double p[240];
    HIPCHK(hipMemcpyAsync(p, h->part, sizeof(p), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    long double s = 0; for (int k = 0; k < nblk; ++k) s += p[k];
    return std::sqrt((double)s);
}

// x (original numbering, device) = (L D L^T)^-1 rhs (original numbering, device); add: x += instead
__global__ void nd_out_kernel(long long n, const double* __restrict__ src, double* __restrict__ dst, int add) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = add ? dst[t] + src[t] : src[t];
}
// ---- multifrontal mode: work of one front on a stream with that stream's scratch (index NS = the main stream's)
#define GF_BIG_WLESS (GF_UPDATE_DMA && GF_LEAN_CHAIN && GF_WLESS_BIG)
static void nd_factor_front(gfs_handle* h, int t, hipStream_t st, int si, bool extend_add = true, bool lazy = false) {
    const Front& F = h->fronts[t];
    if (extend_add) for (int c : h->kids[t]) {
        const long long nbb = h->fronts[c].nblk_t - h->fronts[c].nblk_e;
        if (nbb > 0) hipLaunchKernelGGL(nd_extend_add_kernel, dim3((unsigned)(nbb * (nbb + 1) / 2)), dim3(256), 0, st, h->d_fronts, c, h->d_pmap, h->d_tri, h->band);
    }
    double* band = h->band + (size_t)F.tile_off * NB2;
    double* linv = h->linv + (size_t)F.kbase * NB2; double* dval = h->dval + (size_t)F.kbase * NB; double* stat = h->stat + 2 * F.kbase;
    const int WP = std::max(h->panel_w, 1); const long long wstride = h->max_blk;
    // (Measured and dropped, round 3: look-ahead -- the bulk of a group's wide update on a partner stream while this stream goes on with the next group's
    //  diag / panel / narrow chain: 0.473 instead of 0.407 s at C4.  The chain's one-workgroup diagonal tile runs at a third of its speed next to the
    //  MFMA-heavy update workgroups it shares a CU with, so the chain does not get shorter and the split update costs a launch more per group.)
    // (Look-ahead -- the bulk of a group's wide update on a partner stream under the next group's chain -- was measured twice and dropped twice.  Round 3: 0.473 against 0.407 s, the
    //  chain's diagonal tile ran at a third of its speed beside the update workgroups.  Round 5, with the 13 us tile and the LDS-DMA update: same products in the same order through
    //  update_mid_kernel for the next group's columns + an offset wide update on the partner, two sets of panel buffers: direct launches 0.2198 s without, 0.2189 s with it, the captured
    //  graph 0.2158 s -- and the ladder of dependencies between the two streams cannot be captured: hipGraphInstantiate of ROCm 7.2 walks every path through it (stack overflow; 270 GB
    //  of host memory with an unlimited stack).  profiles/r05_solver_lookahead_ab.txt; the code is in the history.)
    if ((h->block_chain & 1) && GF_UPDATE_DMA && GF_LEAN_CHAIN && h->subgroup > 0 && h->subgroup <= 4) {     // two launches per sub-group (subgroup_block / subgroup_row)
        const int SG = h->subgroup;
        double* wset = h->s_wbuf[si];
        const double* wsrc = GF_BIG_WLESS ? (const double*)dval : (const double*)wset;      // the W-less mid / wide / macro kernels read L and d; the sub-group kernels of this path keep their W
        for (int k0 = 0; k0 < F.nblk_e; k0 += WP) {
            const int w = std::min(WP, F.nblk_e - k0);
            for (int cs = 0; cs < w; cs += SG) {
                const int ks = k0 + cs, sg = std::min(SG, w - cs), nbelow = F.nblk_t - (ks + sg);
                if (cs > 0) hipLaunchKernelGGL(update_mid_kernel, dim3(F.nblk_t - ks, sg), dim3(256), 0, st, band, wsrc, wstride, h->d_tri, k0, cs);
                hipLaunchKernelGGL(subgroup_block_kernel, dim3(1), dim3(256), 0, st, band, linv, dval, stat, h->d_tri, wset, wstride, k0, ks, sg, F.nblk_t);
                if (nbelow > 0) hipLaunchKernelGGL(subgroup_row_kernel, dim3(nbelow), dim3(256), 0, st, band, linv, dval, h->d_tri, wset, wstride, k0, ks, sg);
            }
            const int nrow = F.nblk_t - (k0 + w), jassign = (lazy && k0 == 0) ? F.nblk_e : 0x7fffffff;
            if (nrow >= h->macro_min_rows) {
                const long long nm = (nrow + 1) / 2;
                hipLaunchKernelGGL(update_wide_macro_kernel, dim3((unsigned)(nm * (nm + 1) / 2)), dim3(256), 0, st, band, wsrc, wstride, h->d_tri, k0, w, nrow, jassign);
            } else if (nrow > 0)
                hipLaunchKernelGGL(update_wide_kernel, dim3((unsigned)((long long)nrow * (nrow + 1) / 2)), dim3(256), 0, st, band, wsrc, wstride, h->d_tri, k0, w, nrow, jassign);
        }
        return;
    }
    bool have_diag = false;
    for (int k0 = 0; k0 < F.nblk_e; k0 += WP) {                       // groups of WP block columns: one wide trailing update per group
        const int w = std::min(WP, F.nblk_e - k0);
        double* wset = h->s_wbuf[si];
        const double* wsrc = GF_BIG_WLESS ? (const double*)dval : (const double*)wset;      // what the updates get in their W slot
        const int SG = h->subgroup > 0 ? h->subgroup : WP;
        for (int c = 0; c < w; ++c) {
            const int send = std::min((c / SG + 1) * SG, w);              // the sub-group of column c ends here (relative to k0)
            const int k = k0 + c, ni = F.nblk_t - 1 - k, nin = k0 + send - 1 - k;
            double* wb = wset + (size_t)c * wstride * NB2;
            if (c > 0 && c % SG == 0)                                     // a sub-group starts: its columns get the products of the group's earlier panels in one pass
                hipLaunchKernelGGL(update_mid_kernel, dim3(F.nblk_t - k, send - c), dim3(256), 0, st, band, wsrc, wstride, h->d_tri, k0, c);
            // inside a sub-group the diagonal tile of a block column has been factored by the workgroup that applied its last update (DiagNext); the sub-group's first
            // column gets a launch (the wide and mid updates stay small: 32 KB of LDS, four workgroups per CU).  A lean narrow update (32 KB, no fused tile) at the tree
            // heights where several large fronts run side by side was measured without effect (profiles/r05_solver_fuse_maxf_ab.txt) and removed.
            if (!have_diag) hipLaunchKernelGGL(diag_kernel, dim3(1), dim3(256), 0, st, band, linv, dval, h->d_tri, k, stat);
            have_diag = false;
            if (ni > 0) hipLaunchKernelGGL(panel_kernel, dim3(ni), dim3(256), 0, st, band, linv, dval, GF_BIG_WLESS ? (double*)nullptr : wb, h->d_tri, k);
            if (w == 1 && ni > 0 && !(lazy && k0 == 0) && !GF_BIG_WLESS) {
                have_diag = h->fuse_diag && k + 1 < F.nblk_e;
                hipLaunchKernelGGL(update_kernel, dim3((unsigned)((long long)ni * (ni + 1) / 2)), dim3(256), 0, st, band, wb, h->d_tri, k, ni, DiagNext{linv, dval, stat, have_diag ? 1 : 0});
            } else if (nin > 0) {
                have_diag = h->fuse_diag;
                hipLaunchKernelGGL(update_narrow_kernel, dim3(ni, nin), dim3(256), 0, st, band, GF_BIG_WLESS ? (const double*)dval : (const double*)wb, h->d_tri, k,
                                   DiagNext{linv, dval, stat, have_diag ? 1 : 0});
            }
        }
        const int nrow = F.nblk_t - (k0 + w), jassign = (lazy && k0 == 0) ? F.nblk_e : 0x7fffffff;
        const bool wide = w > 1 || (lazy && k0 == 0) || GF_BIG_WLESS;     // (a single-column group is otherwise served by update_kernel above)
        if (GF_UPDATE_DMA && wide && nrow >= h->macro_min_rows) {
            const long long nm = (nrow + 1) / 2;
            hipLaunchKernelGGL(update_wide_macro_kernel, dim3((unsigned)(nm * (nm + 1) / 2)), dim3(256), 0, st, band, wsrc, wstride, h->d_tri, k0, w, nrow, jassign);
        } else if (wide && nrow > 0)
            hipLaunchKernelGGL(update_wide_kernel, dim3((unsigned)((long long)nrow * (nrow + 1) / 2)), dim3(256), 0, st, band, wsrc, wstride, h->d_tri, k0, w, nrow, jassign);
    }
}
// the NR vectors of one kind out of NR workspaces (one workspace per right-hand side)
#define GF_VEC(member) ([&] { Vec<NR> v_; for (int j_ = 0; j_ < NR; ++j_) v_.p[j_] = W[j_]->member; return v_; }())
#define GF_VECS(which) ([&] { Vec<NR> v_; for (int j_ = 0; j_ < NR; ++j_) v_.p[j_] = W[j_]->scratch(set, which); return v_; }())
template <int NR> static void nd_forward_front(gfs_handle* h, int t, const gfs_handle::SolveWs* const (&W)[NR], hipStream_t st, int set) {
    const Front& F = h->fronts[t];
    if (F.ne_cp == 0) return;                                       // stub front of a partial handle
    const double* band = h->band + (size_t)F.tile_off * NB2;
    const unsigned gl = (unsigned)((64 * F.nblk_t + 255) / 256);
    const Vec<NR> gb = GF_VEC(gb), gy = GF_VEC(gy), fbnd = GF_VEC(fbnd), sb = GF_VECS(0), sy = GF_VECS(1);
    hipLaunchKernelGGL(nd_gather_rhs_kernel<NR>, dim3(gl), dim3(256), 0, st, F, h->d_elim, gb, sb);
    for (int c : h->kids[t]) {
        const Front& Fc = h->fronts[c];
        if (Fc.nb_cp > 0) hipLaunchKernelGGL(nd_pull_child_kernel<NR>, dim3((unsigned)((3 * Fc.nb_cp + 255) / 256)), dim3(256), 0, st, Fc, F, h->d_pmap, fbnd, sb);
    }
    for (int k0 = 0; k0 < F.nblk_e; k0 += 4) {
        const int w = std::min(4, F.nblk_e - k0);
        hipLaunchKernelGGL(fwd_group_kernel<NR>, dim3(1 + F.nblk_t - (k0 + w)), dim3(256), 0, st, band, h->linv + (size_t)F.kbase * NB2, sb, sy, h->d_tri, k0, w);
    }
    hipLaunchKernelGGL(nd_scatter_fwd_kernel<NR>, dim3(gl), dim3(256), 0, st, F, h->d_elim, sy, sb, gy, fbnd);
}
template <int NR> static void nd_backward_front(gfs_handle* h, int t, const gfs_handle::SolveWs* const (&W)[NR], hipStream_t st, int set) {
    const Front& F = h->fronts[t];
    if (F.ne_cp == 0) return;
    const double* band = h->band + (size_t)F.tile_off * NB2;
    const unsigned gl = (unsigned)((64 * F.nblk_t + 255) / 256);
    const Vec<NR> gy = GF_VEC(gy), gx = GF_VEC(gx), sz = GF_VECS(2), sx = GF_VECS(3);
    hipLaunchKernelGGL(nd_gather_bwd_kernel<NR>, dim3(gl), dim3(256), 0, st, F, h->d_elim, h->d_bnd, gy, gx, h->dval, sz, sx);
    if (F.nblk_t > F.nblk_e && F.nblk_e > 0)
        hipLaunchKernelGGL(nd_bwd_bnd_kernel<NR>, dim3(F.nblk_e), dim3(256), 0, st, band, h->d_tri, F.nblk_e, F.nblk_t, sx, sz);
    for (int k0 = ((F.nblk_e - 1) / 4) * 4; k0 >= 0; k0 -= 4) {
        const int w = std::min(4, F.nblk_e - k0);
        hipLaunchKernelGGL(bwd_group_kernel<NR>, dim3(1 + k0), dim3(256), 0, st, band, h->linv + (size_t)F.kbase * NB2, sz, sx, h->d_tri, k0, w);
    }
    hipLaunchKernelGGL(nd_scatter_bwd_kernel<NR>, dim3(gl), dim3(256), 0, st, F, h->d_elim, sx, gx);
}
// bottom-up sweep: the independent subtrees on their streams (forked behind the main stream's earlier work), then the top fronts on the main stream
template <class Fn> static void nd_sweep_up(gfs_handle* h, Fn&& fn) {
    constexpr int NS = gfs_handle::NS;
    HIPCHK(hipEventRecord(h->ev_main, h->stream));
    for (int s = 0; s < NS; ++s) {
        if (h->sub[s].empty()) continue;
        HIPCHK(hipStreamWaitEvent(h->st[s], h->ev_main, 0));
        for (int t : h->sub[s]) fn(h, t, h->st[s], s);
        HIPCHK(hipEventRecord(h->ev[s], h->st[s]));
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev[s], 0));
    }
    for (int t : h->top) fn(h, t, h->stream, NS);
}
// top-down sweep: the top fronts in reverse on the main stream, then the subtrees in reverse on their streams; joined on the main stream
template <class Fn> static void nd_sweep_down(gfs_handle* h, Fn&& fn) {
    constexpr int NS = gfs_handle::NS;
    for (auto it = h->top.rbegin(); it != h->top.rend(); ++it) fn(h, *it, h->stream, NS);
    HIPCHK(hipEventRecord(h->ev_main, h->stream));
    for (int s = 0; s < NS; ++s) {
        if (h->sub[s].empty()) continue;
        HIPCHK(hipStreamWaitEvent(h->st[s], h->ev_main, 0));
        for (auto it = h->sub[s].rbegin(); it != h->sub[s].rend(); ++it) fn(h, *it, h->st[s], s);
        HIPCHK(hipEventRecord(h->ev[s], h->st[s]));
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev[s], 0));
    }
}
// factorisation sweep by tree height
// what the batched mid / wide updates get in their W slot: the W panels, or (W-less build) the handle's dval
#if GF_UPDATE_DMA && GF_WLESS_BATCH
#define GF_BATCH_WSRC(h) ((const double*)(h)->dval)
#else
#define GF_BATCH_WSRC(h) ((const double*)(h)->bwbuf)
#endif
static void nd_factor_levels(gfs_handle* h) {
    constexpr int NS = gfs_handle::NS;
    for (const auto& L : h->flevels) {
        const bool lazy = h->lazy_s && GF_UPDATE_DMA;
        for (const auto& R : L.rounds)
            hipLaunchKernelGGL(nd_extend_add_batch_kernel, dim3((unsigned)R.max_tiles, (unsigned)R.n), dim3(256), 0, h->stream, h->d_fronts, h->d_ealist + R.off, h->d_pmap, h->d_tri, h->band,
                               lazy ? 1 : 0);
        int used = 0;
        if (!L.big.empty()) {
            HIPCHK(hipEventRecord(h->ev_main, h->stream));
            used = std::min<int>(NS, (int)L.big.size());
            for (int s = 0; s < used; ++s) HIPCHK(hipStreamWaitEvent(h->st[s], h->ev_main, 0));
            for (size_t i = 0; i < L.big.size(); ++i) { const int s = (int)(i % NS); nd_factor_front(h, L.big[i], h->st[s], s, false, lazy); }
        }
        const int WP = std::max(h->batch_panel_w, 1), kmax = (int)L.nk.size();
        const bool blockchain = (h->block_chain & 2) && GF_UPDATE_DMA && GF_LEAN_CHAIN && h->subgroup > 0 && h->subgroup <= 4;
        for (int k0 = 0; blockchain && k0 < kmax; k0 += WP) {            // panel groups in sub-groups of two launches each (subgroup_block / subgroup_row), as nd_factor_front
            const int SG = h->subgroup;
            for (int cs = 0; cs < WP && k0 + cs < kmax; cs += SG) {
                const int ks = k0 + cs, nk = L.nk[ks], mni = L.max_ni[ks];
                if (cs > 0)
                    hipLaunchKernelGGL(nd_update_mid_batch_kernel, dim3(mni + 1, std::min(SG, WP - cs), nk), dim3(256), 0, h->stream, h->d_fronts, h->d_flist + L.off, h->d_fwofs + L.off,
                                       h->d_tri, h->band, GF_BATCH_WSRC(h), k0, cs, SG, WP);
                hipLaunchKernelGGL(nd_subgroup_block_batch_kernel, dim3(nk), dim3(256), 0, h->stream, h->d_fronts, h->d_flist + L.off, h->d_fwofs + L.off, h->d_tri, h->band, h->linv, h->dval,
                                   h->stat, h->bwbuf, k0, cs, SG, WP);
                if (mni > 0)
                    hipLaunchKernelGGL(nd_subgroup_row_batch_kernel, dim3(mni, nk), dim3(256), 0, h->stream, h->d_fronts, h->d_flist + L.off, h->d_fwofs + L.off, h->d_tri, h->band, h->linv,
                                       h->dval, h->bwbuf, k0, cs, SG, WP);
            }
            const int mni0 = L.max_ni[k0];
            if (mni0 > 0)
                hipLaunchKernelGGL(nd_update_wide_batch_kernel, dim3((unsigned)((long long)mni0 * (mni0 + 1) / 2), L.nk[k0]), dim3(256), 0, h->stream, h->d_fronts, h->d_flist + L.off,
                                   h->d_fwofs + L.off, h->d_tri, h->band, GF_BATCH_WSRC(h), k0, WP, lazy ? 1 : 0);
        }
        for (int k0 = 0; !blockchain && k0 < kmax; k0 += WP) {           // panel groups, as nd_factor_front does for one front
            const int SG = h->subgroup > 0 ? h->subgroup : WP;
            for (int c = 0; c < WP && k0 + c < kmax; ++c) {
                const int k = k0 + c, nk = L.nk[k], mni = L.max_ni[k];
                if (c > 0 && c % SG == 0)                                // a sub-group starts (nd_factor_front)
                    hipLaunchKernelGGL(nd_update_mid_batch_kernel, dim3(mni + 1, std::min(SG, WP - c), nk), dim3(256), 0, h->stream, h->d_fronts, h->d_flist + L.off, h->d_fwofs + L.off,
                                       h->d_tri, h->band, GF_BATCH_WSRC(h), k0, c, SG, WP);
                if (c % SG == 0 || !h->fuse_diag)
                    hipLaunchKernelGGL(nd_diag_batch_kernel, dim3(nk), dim3(256), 0, h->stream, h->d_fronts, h->d_flist + L.off, h->d_tri, h->band, h->linv, h->dval, h->stat, k);
                if (mni <= 0) continue;
                hipLaunchKernelGGL(nd_panel_batch_kernel, dim3(mni, nk), dim3(256), 0, h->stream, h->d_fronts, h->d_flist + L.off, h->d_fwofs + L.off, h->d_tri, h->band, h->linv, h->dval,
                                   h->bwbuf, k, c);
                if (c + 1 < WP && (c + 1) % SG != 0)
                    hipLaunchKernelGGL(nd_update_narrow_batch_kernel, dim3(mni, std::min(SG - 1 - c % SG, WP - 1 - c), nk), dim3(256), 0, h->stream, h->d_fronts, h->d_flist + L.off,
                                       h->d_fwofs + L.off, h->d_tri, h->band, h->bwbuf, k, k0, WP, (c / SG + 1) * SG, DiagNext{h->linv, h->dval, h->stat, h->fuse_diag ? 1 : 0});
            }
            const int mni0 = L.max_ni[k0];                                // >= the trailing rows of every front of the group
            if (mni0 > 0)
                hipLaunchKernelGGL(nd_update_wide_batch_kernel, dim3((unsigned)((long long)mni0 * (mni0 + 1) / 2), L.nk[k0]), dim3(256), 0, h->stream, h->d_fronts, h->d_flist + L.off,
                                   h->d_fwofs + L.off, h->d_tri, h->band, GF_BATCH_WSRC(h), k0, WP, lazy ? 1 : 0);
        }
        for (int s = 0; s < used; ++s) { HIPCHK(hipEventRecord(h->ev[s], h->st[s])); HIPCHK(hipStreamWaitEvent(h->stream, h->ev[s], 0)); }
        if (lazy) for (const auto& R : L.rounds)                         // the children's contributions to the Schur blocks of this height's fronts, behind their wide updates
            hipLaunchKernelGGL(nd_extend_add_batch_kernel, dim3((unsigned)R.max_tiles, (unsigned)R.n), dim3(256), 0, h->stream, h->d_fronts, h->d_ealist + R.off, h->d_pmap, h->d_tri, h->band, 2);
    }
}
// run `body` (kernel launches, event record / wait on h->stream and the side streams) through a graph captured at the first call
template <class Body> static void nd_run_captured(gfs_handle* h, hipGraphExec_t* exec, Body&& body, hipStream_t cs = nullptr) {
    if (!cs) cs = h->stream;
    if (!h->use_graph) { body(); return; }
    // the first calls of a sweep launch directly: capturing + instantiating a graph of a few thousand nodes costs 0.3 - 0.6 s per sweep at C4 and pays back 9 ms per
    // factorisation -- a session of a handful of solves never gets there (C4: the first solve 2.6 -> 1.4 s)
    if (!*exec && h->graph_calls[(void*)exec]++ < h->graph_after) { body(); return; }
    if (!*exec) {
        hipGraph_t g = nullptr;
        // thread-local capture: only this thread's calls are checked against the capture, and a failure inside body() must not leave the stream
        // (and the forked side streams) capturing -- the caller's fallback (host LU: hipMemcpy of K on this thread, gfs_destroy) would fail on
        // exactly the path it exists for.  On an exception the capture is ended, the partial graph dropped, graphs switched off.
        HIPCHK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        try { body(); }
        catch (...) {
            (void)hipStreamEndCapture(cs, &g);
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            h->use_graph = false;
            throw;
        }
        HIPCHK(hipStreamEndCapture(cs, &g));
        const hipError_t ei = hipGraphInstantiate(exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ei != hipSuccess) { *exec = nullptr; h->use_graph = false; throw std::runtime_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(ei)); }
    }
    HIPCHK(hipGraphLaunch(*exec, cs));
}
// the large fronts of one tree height (independent of each other): side by side on the handle's side streams, each with the front-local vectors of its stream, forked
// from and joined to the sweep's stream (round 5: their group kernels are chains of dependent 20 us launches -- 16.3 of a sweep's 37 ms at C4 ran one after the other)
// returns the number of side streams to join (nd_big_join, behind whatever the caller runs on st beside them: the small fronts of the same height)
template <int NR> static int nd_big_fronts(gfs_handle* h, const std::vector<int>& big, const gfs_handle::SolveWs* const (&W)[NR], hipStream_t st, bool backward) {
    constexpr int NS = gfs_handle::NS;
    const int nb = (int)big.size();
    const bool side = h->sweep_streams && nb >= 2 && W[0]->ev_fork && h->st[0];
    for (int j = 0; side && j < NR; ++j) if (!W[j]->big) throw std::runtime_error("substitution workspace without side-stream vectors");
    if (!side) {
        for (int i = 0; i < nb; ++i) { const int t = backward ? big[nb - 1 - i] : big[i]; if (backward) nd_backward_front<NR>(h, t, W, st, -1); else nd_forward_front<NR>(h, t, W, st, -1); }
        return 0;
    }
    const int used = std::min(NS, nb);
    HIPCHK(hipEventRecord(W[0]->ev_fork, st));
    for (int s = 0; s < used; ++s) HIPCHK(hipStreamWaitEvent(h->st[s], W[0]->ev_fork, 0));
    for (int i = 0; i < nb; ++i) {
        const int t = backward ? big[nb - 1 - i] : big[i], s = i % NS;
        if (backward) nd_backward_front<NR>(h, t, W, h->st[s], s); else nd_forward_front<NR>(h, t, W, h->st[s], s);
    }
    return used;
}
static void nd_big_join(gfs_handle* h, const gfs_handle::SolveWs* W0, hipStream_t st, int used) {
    for (int s = 0; s < used; ++s) { HIPCHK(hipEventRecord(W0->ev_join[s], h->st[s])); HIPCHK(hipStreamWaitEvent(st, W0->ev_join[s], 0)); }
}
// the two halves of a substitution: forward over the tree heights ascending, backward descending (W: one workspace per right-hand side, st: the stream)
template <int NR> static void nd_forward_all(gfs_handle* h, const gfs_handle::SolveWs* const (&W)[NR], hipStream_t st) {
    const Vec<NR> gb = GF_VEC(gb), gy = GF_VEC(gy), fbnd = GF_VEC(fbnd);
    for (const auto& L : h->levels) {
        const int used = nd_big_fronts<NR>(h, L.big, W, st, false);
        if (L.n_small > 0)
            hipLaunchKernelGGL(nd_fwd_front_kernel<NR>, dim3(L.n_small), dim3(256), (size_t)NR * (64 * L.max_blk + 64) * sizeof(double), st, h->d_fronts, h->d_lvl_list + L.off_small,
                               h->d_kid_off, h->d_kid, h->d_tri, h->band, h->linv, h->d_elim, h->d_pmap, gb, gy, fbnd);
        nd_big_join(h, W[0], st, used);
    }
}
template <int NR> static void nd_backward_all(gfs_handle* h, const gfs_handle::SolveWs* const (&W)[NR], hipStream_t st) {
    const Vec<NR> gy = GF_VEC(gy), gx = GF_VEC(gx);
    for (auto it = h->levels.rbegin(); it != h->levels.rend(); ++it) {
        const auto& L = *it;
        const int used = nd_big_fronts<NR>(h, L.big, W, st, true);
        if (L.n_small > 0)
            hipLaunchKernelGGL(nd_bwd_front_kernel<NR>, dim3(L.n_small), dim3(256), (size_t)NR * (64 * L.max_blk + 4 * 64) * sizeof(double), st, h->d_fronts, h->d_lvl_list + L.off_small,
                               h->d_tri, h->band, h->linv, h->dval, h->d_elim, h->d_bnd, gy, gx);
        nd_big_join(h, W[0], st, used);
    }
}
// multifrontal substitutions of NR right-hand sides in one pass over the factors; vectors in the original numbering.  Workspace W[j] holds right-hand side j's vectors;
// everything runs on W[0]'s stream; the captured graph (the sweeps are ~1e4 launches of fixed structure) belongs to W[0] and is keyed by NR -- the caller always
// groups the same workspaces (solve_dev_impl: right-hand sides 3 c .. 3 c + NR - 1), so the pointers baked into the graph stay valid.
// rhs[j] == nullptr: the right-hand side of workspace j is not refreshed and x[j] not written (a right-hand side of the group that has already converged rides along:
// the sweeps are bound by the factor bytes, an idle slot costs nothing, and the group's graph stays the same).
template <int NR> static void substitute_nd(gfs_handle* h, gfs_handle::SolveWs* const (&Wm)[NR], const double* const (&rhs)[NR], double* const (&x)[NR], int add) {
    const gfs_handle::SolveWs* W[NR];
    for (int j = 0; j < NR; ++j) W[j] = Wm[j];
    hipStream_t st = W[0]->stream;
    for (int j = 0; j < NR; ++j) if (rhs[j]) HIPCHK(hipMemcpyAsync(W[j]->gb, rhs[j], h->n * sizeof(double), hipMemcpyDeviceToDevice, st));
    const gfs_handle::SolveWs* const (&Wc)[NR] = W;
    if (NR > 1) {                                                                 // NR front-local vectors of up to FUSE_MAX_BLK blocks: more than the 64 KB a launch gets by default
        // per handle, not per process: the attribute belongs to the CURRENT DEVICE's copy of the kernel, and a process may hold handles on several GPUs
        if (!h->lds_raised[NR - 1]) {
            const int lim = 160 * 1024;
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nd_fwd_front_kernel<NR>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nd_bwd_front_kernel<NR>), hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            h->lds_raised[NR - 1] = true;
        }
    }
    nd_run_captured(h, &Wm[0]->g_solve[NR - 1], [&] { nd_forward_all<NR>(h, Wc, st); nd_backward_all<NR>(h, Wc, st); }, st);
    for (int j = 0; j < NR; ++j) if (rhs[j]) hipLaunchKernelGGL(nd_out_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, st, h->n, W[j]->gx, x[j], add);
    HIPCHK(hipGetLastError());
}
#undef GF_VEC
#undef GF_VECS
static void substitute_nd(gfs_handle* h, gfs_handle::SolveWs& W, const double* rhs, double* x, int add) {
    gfs_handle::SolveWs* const Wm[1] = {&W}; const double* const r[1] = {rhs}; double* const xx[1] = {x};
    substitute_nd<1>(h, Wm, r, xx, add);
}

// front-local vectors per side stream and the fork / join events of a workspace (nested-dissection handles)
static void solve_ws_side(gfs_handle* h, gfs_handle::SolveWs& W) {
    if (!h->nd || !h->st[0] || h->ws_front_len <= 0) return;
    W.big_len = h->ws_front_len; W.big = h->dalloc<double>((size_t)4 * gfs_handle::NS * (size_t)W.big_len);
    HIPCHK(hipEventCreateWithFlags(&W.ev_fork, hipEventDisableTiming));
    for (int s = 0; s < gfs_handle::NS; ++s) HIPCHK(hipEventCreateWithFlags(&W.ev_join[s], hipEventDisableTiming));
}
// workspace k of the handle: 0 aliases the handle's own vectors and stream, the others are allocated on first use (nested-dissection mode only)
static gfs_handle::SolveWs& solve_ws(gfs_handle* h, int k) {
    constexpr int NS = gfs_handle::NS;
    if (h->ws.empty()) {
        h->ws.reserve(gfs_handle::MAX_RHS + 1);        // the graphs of a workspace are keyed by the address of its g_solve slots (graph_calls): the vector must never reallocate
        gfs_handle::SolveWs W;
        W.stream = h->stream; W.gb = h->gb; W.gy = h->gy; W.gx = h->gx; W.fbnd = h->fbnd; W.sb = h->s_b[NS]; W.sy = h->s_y[NS]; W.sz = h->s_z[NS]; W.sx = h->s_x[NS];
        W.vr = h->vr; W.vsol = h->vsol; W.vrhs = h->vrhs; W.part = h->part;
        solve_ws_side(h, W);
        h->ws.push_back(W);
    }
    while ((int)h->ws.size() <= k) {
        if (!h->nd) throw std::runtime_error("gfs_solve_multi: concurrent right-hand sides need the nested-dissection mode");
        gfs_handle::SolveWs W;
        HIPCHK(hipStreamCreate(&W.stream));
        W.gb = h->dalloc<double>(h->n); W.gy = h->dalloc<double>(h->n); W.gx = h->dalloc<double>(h->n);
        W.fbnd = h->dalloc<double>((size_t)std::max<long long>(h->ws_bnd_len, 1));
        W.sb = h->dalloc<double>((size_t)h->ws_front_len); W.sy = h->dalloc<double>((size_t)h->ws_front_len); W.sz = h->dalloc<double>((size_t)h->ws_front_len); W.sx = h->dalloc<double>((size_t)h->ws_front_len);
        W.vr = h->dalloc<double>(h->n); W.vsol = h->dalloc<double>(h->n); W.vrhs = h->dalloc<double>(h->n); W.part = h->dalloc<double>(256);
        HIPCHK(hipMemsetAsync(W.gy, 0, h->n * sizeof(double), W.stream)); HIPCHK(hipMemsetAsync(W.gx, 0, h->n * sizeof(double), W.stream));
        HIPCHK(hipStreamSynchronize(W.stream));
        solve_ws_side(h, W);
        h->ws.push_back(W);
    }
    return h->ws[k];
}

static void substitute(gfs_handle* h, const double* rhs, double* x, int add) {
    if (h->nd) { substitute_nd(h, solve_ws(h, 0), rhs, x, add); return; }
    const unsigned g3 = (unsigned)((3 * h->ncp + 255) / 256), gp = (unsigned)((h->npad + 255) / 256);
    HIPCHK(hipMemsetAsync(h->vb, 0, h->npad * sizeof(double), h->stream));
    hipLaunchKernelGGL(permute_in_kernel, dim3(g3), dim3(256), 0, h->stream, h->ncp, h->newi, rhs, h->vb);
    for (long long k = 0; k < h->nblk; ++k) {
        hipLaunchKernelGGL(fwd_kernel, dim3(h->nik[k] + 1), dim3(256), 0, h->stream, h->band, h->linv, h->vb, h->vy, h->rowoff, (int)k);
    }
    hipLaunchKernelGGL(scale_kernel, dim3(gp), dim3(256), 0, h->stream, h->npad, h->dval, h->vy, h->vz);
    for (long long k = h->nblk - 1; k >= 0; --k) {
        hipLaunchKernelGGL(bwd_kernel, dim3(h->Tr[k] + 1), dim3(256), 0, h->stream, h->band, h->linv, h->vz, h->vx, h->rowoff, (int)k);
    }
    hipLaunchKernelGGL(permute_out_kernel, dim3(g3), dim3(256), 0, h->stream, h->ncp, h->newi, h->vx, x, add);
    HIPCHK(hipGetLastError());
}

extern "C" {

const char* gfs_last_error(void) { return g_serr.c_str(); }

int gfs_create(int device, int64_t ncp, const int64_t* nb_ptr, const int32_t* nb, const int32_t* new_index, const double* d_valK, gfs_handle** out) {
    if (!out || !nb_ptr || !nb || !new_index || !d_valK) return sfail("gfs_create: null argument");
    *out = nullptr;
    if (ncp <= 0) return sfail("gfs_create: no control points");
    gfs_handle* h = nullptr;
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("gfs_create: no HIP device visible (libgoldfish_solver has no CPU fallback)");
        if (device < 0 || device >= ndev) throw std::runtime_error("gfs_create: device index out of range");
        {   // new_index must be a permutation; the neighbour relation gives the bandwidth
            std::vector<char> seen(ncp, 0);
            for (int64_t a = 0; a < ncp; ++a) { const int32_t p = new_index[a]; if (p < 0 || p >= ncp || seen[p]) throw std::runtime_error("gfs_create: new_index is not a permutation"); seen[p] = 1; }
        }
        long long bwcp = 0;
        for (int64_t a = 0; a < ncp; ++a) for (int64_t k = nb_ptr[a]; k < nb_ptr[a + 1]; ++k) {
            if (nb[k] < 0 || nb[k] >= ncp) throw std::runtime_error("gfs_create: neighbour index out of range");
            bwcp = std::max<long long>(bwcp, std::llabs((long long)new_index[a] - new_index[nb[k]]));
        }
        h = new gfs_handle(); h->device = device;
        if (const char* e_ = getenv("GF_SOLVER_FUSE_DIAG")) h->fuse_diag = atoi(e_) != 0;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreate(&h->stream));
        h->ncp = ncp; h->n = 3 * ncp; h->nblk = (h->n + NB - 1) / NB; h->npad = h->nblk * NB;
        h->bw = 3 * bwcp + 2;
        {   // envelope: first coupled block column of every block row, made monotone (the rows that reach block column k are then k + 1 .. k + nik[k])
            std::vector<long long> f(h->nblk);
            for (long long I = 0; I < h->nblk; ++I) f[I] = I;
            for (int64_t a = 0; a < ncp; ++a) {
                long long mn = new_index[a];
                for (int64_t k = nb_ptr[a]; k < nb_ptr[a + 1]; ++k) mn = std::min<long long>(mn, new_index[nb[k]]);
                const long long cb = (3 * mn) >> 6;
                for (int i = 0; i < 3; ++i) { const long long I = (3 * (long long)new_index[a] + i) >> 6; f[I] = std::min(f[I], cb); }
            }
            for (long long I = h->nblk - 2; I >= 0; --I) f[I] = std::min(f[I], f[I + 1]);
            h->Tr.resize(h->nblk); h->nik.assign(h->nblk, 0);
            std::vector<long long> off(h->nblk + 1, 0);
            h->T = 0;
            for (long long I = 0; I < h->nblk; ++I) { h->Tr[I] = (int)(I - f[I]); h->T = std::max(h->T, h->Tr[I]); off[I + 1] = off[I] + h->Tr[I] + 1; }
            h->ntiles = off[h->nblk];
            long long r = 0;
            for (long long k = 0; k < h->nblk; ++k) { r = std::max(r, k); while (r + 1 < h->nblk && f[r + 1] <= k) ++r; h->nik[k] = (int)(r - k); }
            h->rowoff = nullptr;
            const double gbs = ((double)h->ntiles + h->nblk + h->T + 1) * NB2 * 8.0 / 1e9;
            size_t freeb = 0, totb = 0; HIPCHK(hipMemGetInfo(&freeb, &totb));
            if (gbs * 1e9 > 0.92 * (double)freeb)
                throw std::runtime_error("gfs_create: the factor skyline needs " + std::to_string(gbs) + " GB (" + std::to_string(h->n) + " dofs, half bandwidth " + std::to_string(h->bw) +
                                         "), more than the free device memory");
            h->rowoff = h->up(off.data(), off.size());
        }
        std::vector<long long> ptr(nb_ptr, nb_ptr + ncp + 1);
        h->nb_ptr = h->up(ptr.data(), ptr.size()); h->nb = h->up(nb, (size_t)nb_ptr[ncp]); h->newi = h->up(new_index, (size_t)ncp);
        h->valK = d_valK; h->nnz9 = 9 * (long long)nb_ptr[ncp];
        h->band = h->dalloc<double>((size_t)h->ntiles * NB2);
        h->linv = h->dalloc<double>((size_t)h->nblk * NB2);
        h->dval = h->dalloc<double>((size_t)h->npad); h->stat = h->dalloc<double>((size_t)2 * h->nblk);
        h->wbuf = h->dalloc<double>((size_t)std::max(h->T, 1) * NB2);       // W tiles of one panel: at most T rows reach a block column
        h->vb = h->dalloc<double>(h->npad); h->vy = h->dalloc<double>(h->npad); h->vz = h->dalloc<double>(h->npad); h->vx = h->dalloc<double>(h->npad);
        h->vr = h->dalloc<double>(h->npad); h->vsol = h->dalloc<double>(h->npad); h->vrhs = h->dalloc<double>(h->npad); h->part = h->dalloc<double>(256);
        HIPCHK(hipDeviceSynchronize());
    } catch (const std::exception& ex) {
        if (h) gfs_destroy(h);
        return sfail(ex.what());
    }
    *out = h;
    return 0;
}

// Nested-dissection multifrontal mode: the fronts come from the host's symbolic phase (goldfish_amd/_nd.py); all arrays are host pointers.
//   elim [ncp], elim_off [nfronts + 1]: control points eliminated by every front (post-order); bnd, bnd_off: boundary control points per front (ascending
//   elimination order); parent [nfronts] (-1: root); order [ncp]: position in the elimination order; front_of [ncp]; pmap [size of bnd]: position of every
//   boundary control point of a front in its PARENT's numbering (eliminated control points first, then the parent's boundary).
static int create_nd_impl(int device, int64_t ncp, const int64_t* nb_ptr, const int32_t* nb, const double* d_valK, int64_t nfronts, const int64_t* elim, const int64_t* elim_off,
                          const int64_t* bnd, const int64_t* bnd_off, const int64_t* parent, const int64_t* order, const int64_t* front_of, const int64_t* pmap, bool partial, gfs_handle** out) {
    if (!out || !nb_ptr || !nb || !d_valK || !elim || !elim_off || !bnd_off || !parent || !order || !front_of) return sfail("gfs_create_nd: null argument");
    *out = nullptr;
    if (ncp <= 0 || nfronts <= 0) return sfail("gfs_create_nd: empty model");
    gfs_handle* h = nullptr;
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("gfs_create_nd: no HIP device visible (libgoldfish_solver has no CPU fallback)");
        if (device < 0 || device >= ndev) throw std::runtime_error("gfs_create_nd: device index out of range");
        const int64_t nelim = elim_off[nfronts];
        if (nelim > ncp || (!partial && nelim != ncp)) throw std::runtime_error("gfs_create_nd: the fronts do not eliminate every control point exactly once");
        {
            std::vector<char> seen(ncp, 0);
            for (int64_t q = 0; q < nelim; ++q) {
                const int64_t a = elim[q];
                if (a < 0 || a >= ncp || seen[a] || order[a] != q) throw std::runtime_error("gfs_create_nd: elim / order are not a consistent permutation");
                seen[a] = 1;
            }
        }
        h = new gfs_handle(); h->device = device; h->nd = true;
        if (const char* e_ = getenv("GF_SOLVER_GRAPH")) h->use_graph = std::string(e_) != "0";
        if (const char* e_ = getenv("GF_SOLVER_GRAPH_AFTER")) h->graph_after = std::max(0, atoi(e_));
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreate(&h->stream));
        h->ncp = ncp; h->n = 3 * ncp; h->npad = h->n; h->nblk = 0; h->bw = 0;
        h->fronts.resize(nfronts); h->kids.assign(nfronts, {});
        long long tiles = 0, kb = 0; int maxb = 0; double fl = 0.0;
        for (int64_t t = 0; t < nfronts; ++t) {
            Front& F = h->fronts[t];
            F.elim_off = elim_off[t]; F.bnd_off = bnd_off[t];
            F.ne_cp = (int)(elim_off[t + 1] - elim_off[t]); F.nb_cp = (int)(bnd_off[t + 1] - bnd_off[t]);
            if (F.ne_cp < 0 || (F.ne_cp == 0 && !partial)) throw std::runtime_error("gfs_create_nd: a front eliminates nothing");
            F.nblk_e = (3 * F.ne_cp + NB - 1) / NB; F.ne_pad = F.nblk_e * NB;
            F.nblk_t = F.nblk_e + (3 * F.nb_cp + NB - 1) / NB;
            F.parent = (int)parent[t]; F.pad0 = F.pad1 = 0;
            if (F.parent >= 0) { if (F.parent <= t || F.parent >= nfronts) throw std::runtime_error("gfs_create_nd: the fronts are not in post-order"); h->kids[F.parent].push_back((int)t); }
            else if (F.nb_cp != 0 && !partial) throw std::runtime_error("gfs_create_nd: a root front has a boundary");
            F.tile_off = tiles; F.kbase = kb;
            tiles += (long long)F.nblk_t * (F.nblk_t + 1) / 2; kb += F.nblk_e; maxb = std::max(maxb, F.nblk_t);
            for (int k = 0; k < F.nblk_e; ++k) { const double r = F.nblk_t - 1 - k; fl += 2.0 * NB * NB * NB * (r + r * (r + 1) / 2) + 2.0 * NB * NB * NB / 3; }
        }
        h->ntiles = tiles; h->nbe_tot = kb; h->max_blk = maxb; h->nd_flops = fl; h->T = maxb;
        const double gbs = ((double)tiles + kb + maxb) * NB2 * 8.0 / 1e9;
        size_t freeb = 0, totb = 0; HIPCHK(hipMemGetInfo(&freeb, &totb));
        if (gbs * 1e9 > 0.92 * (double)freeb) throw std::runtime_error("gfs_create_nd: the fronts need " + std::to_string(gbs) + " GB, more than the free device memory");
        std::vector<long long> ptr(nb_ptr, nb_ptr + ncp + 1), tri(maxb + 1), ord(order, order + ncp);
        for (int I = 0; I <= maxb; ++I) tri[I] = (long long)I * (I + 1) / 2;
        std::vector<int> e32(elim, elim + nelim), fo32(front_of, front_of + ncp), b32, pm32;
        if (e32.empty()) e32.push_back(0);
        h->partial = partial; h->stub_src.assign(nfronts, nullptr);
        const int64_t nbnd = bnd_off[nfronts];
        if (nbnd > 0) { if (!bnd || !pmap) throw std::runtime_error("gfs_create_nd: bnd / pmap missing"); b32.assign(bnd, bnd + nbnd); pm32.assign(pmap, pmap + nbnd); }
        for (int64_t t = 0; t < nfronts; ++t) {               // the map into the parent must be strictly increasing and inside the parent
            const Front& F = h->fronts[t];
            if (F.parent < 0) continue;
            const Front& P = h->fronts[F.parent];
            for (int q = 0; q < F.nb_cp; ++q) {
                const int v = pm32[F.bnd_off + q];
                if (v < 0 || v >= P.ne_cp + P.nb_cp || (q > 0 && v <= pm32[F.bnd_off + q - 1])) throw std::runtime_error("gfs_create_nd: boundary map into the parent front is not monotone");
            }
        }
        h->nb_ptr = h->up(ptr.data(), ptr.size()); h->nb = h->up(nb, (size_t)nb_ptr[ncp]);
        h->d_tri = h->up(tri.data(), tri.size()); h->d_order = h->up(ord.data(), ord.size());
        h->d_elim = h->up(e32.data(), e32.size()); h->d_front_of = h->up(fo32.data(), fo32.size());
        h->d_bnd = h->up(b32.data(), b32.size()); h->d_pmap = h->up(pm32.data(), pm32.size());
        h->d_fronts = h->up(h->fronts.data(), h->fronts.size());
        {   // (front, block row) pairs whose tiles in eliminated block columns are cleared before a factorisation ("lazy S": clear_factor_storage)
            std::vector<int2> zr;
            for (size_t t = 0; t < h->fronts.size(); ++t) if (h->fronts[t].nblk_e > 0) for (int I = 0; I < h->fronts[t].nblk_t; ++I) zr.push_back(int2{(int)t, I});
            h->n_zrows = (long long)zr.size();
            if (!zr.empty()) h->d_zrows = h->up(zr.data(), zr.size());
        }
        h->valK = d_valK; h->nnz9 = 9 * (long long)nb_ptr[ncp];
        h->band = h->dalloc<double>((size_t)tiles * NB2);
        h->linv = h->dalloc<double>((size_t)kb * NB2);
        h->dval = h->dalloc<double>((size_t)kb * NB); h->stat = h->dalloc<double>((size_t)2 * kb);
        if (const char* e = std::getenv("GF_SOLVER_PANEL_W")) h->panel_w = h->batch_panel_w = std::max(1, std::min(8, std::atoi(e)));
        if (const char* e = std::getenv("GF_SOLVER_FUSE_DIAG")) h->fuse_diag = std::atoi(e) != 0;
        if (const char* e = std::getenv("GF_SOLVER_SUBGROUP")) h->subgroup = std::max(0, std::min(8, std::atoi(e)));
        if (const char* e = std::getenv("GF_SOLVER_LAZY_S")) h->lazy_s = std::atoi(e) != 0;
        if (const char* e = std::getenv("GF_SOLVER_BLOCKCHAIN")) h->block_chain = std::atoi(e);
        if (const char* e = std::getenv("GF_SOLVER_MACRO_ROWS")) h->macro_min_rows = std::max(2, std::atoi(e));
        if (const char* e = std::getenv("GF_SOLVER_SWEEP_STREAMS")) h->sweep_streams = std::atoi(e) != 0;
        if (const char* e = std::getenv("GF_SOLVER_BATCH_PANEL_W")) h->batch_panel_w = std::max(1, std::min(8, std::atoi(e)));      // panel groups of the level-batched small fronts
        {   // independent subtrees for the side streams: split the largest subtree (by factorisation work) until there are enough of them
            constexpr int NS = gfs_handle::NS;
            std::vector<double> work(nfronts, 0.0); std::vector<int> cnt(nfronts, 1);
            for (int64_t t = 0; t < nfronts; ++t) {
                const Front& F = h->fronts[t];
                for (int k = 0; k < F.nblk_e; ++k) { const double r = F.nblk_t - 1 - k; work[t] += 1.0 + r + r * (r + 1) / 2; }
            }
            for (int64_t t = 0; t < nfronts; ++t) if (h->fronts[t].parent >= 0) { work[h->fronts[t].parent] += work[t]; cnt[h->fronts[t].parent] += cnt[t]; }   // post-order: children first
            std::vector<int> roots, topset;
            for (int64_t t = 0; t < nfronts; ++t) if (h->fronts[t].parent < 0) roots.push_back((int)t);
            std::vector<int> cand = roots;
            while ((int)cand.size() < 6 * NS) {
                int best = -1;
                for (int i = 0; i < (int)cand.size(); ++i) if (!h->kids[cand[i]].empty() && (best < 0 || work[cand[i]] > work[cand[best]])) best = i;
                if (best < 0) break;
                const int t = cand[best];
                cand.erase(cand.begin() + best); topset.push_back(t);
                for (int c : h->kids[t]) cand.push_back(c);
            }
            std::vector<char> is_top(nfronts, 0);
            for (int t : topset) is_top[t] = 1;
            // greedy assignment of the subtrees to the streams (largest first); a subtree = the post-order range [t - cnt[t] + 1, t]
            std::sort(cand.begin(), cand.end(), [&](int x, int y) { return work[x] > work[y]; });
            std::vector<double> load(NS, 0.0); std::vector<std::vector<int>> subroots(NS);
            for (int t : cand) { const int s_ = (int)(std::min_element(load.begin(), load.end()) - load.begin()); load[s_] += work[t]; subroots[s_].push_back(t); }
            h->sub.assign(NS, {});
            for (int s_ = 0; s_ < NS; ++s_) {
                std::sort(subroots[s_].begin(), subroots[s_].end());
                for (int t : subroots[s_]) for (int q = t - cnt[t] + 1; q <= t; ++q) h->sub[s_].push_back(q);
            }
            for (int64_t t = 0; t < nfronts; ++t) if (is_top[t]) h->top.push_back((int)t);
            for (int s_ = 0; s_ < NS; ++s_) { HIPCHK(hipStreamCreate(&h->st[s_])); HIPCHK(hipEventCreateWithFlags(&h->ev[s_], hipEventDisableTiming)); }
            HIPCHK(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->ev_norm, hipEventDisableTiming)); h->part_k = h->dalloc<double>(256);
            const size_t fl_ = (size_t)maxb * NB;
            h->ws_front_len = (long long)fl_;
            for (int s_ = 0; s_ <= NS; ++s_) {
                h->s_wbuf[s_] = h->dalloc<double>((size_t)std::max(maxb, 1) * NB2 * std::max(h->panel_w, 1));
                h->s_b[s_] = h->dalloc<double>(fl_); h->s_y[s_] = h->dalloc<double>(fl_); h->s_z[s_] = h->dalloc<double>(fl_); h->s_x[s_] = h->dalloc<double>(fl_);
            }
            h->wbuf = h->s_wbuf[NS];
        }
        {   // tree heights and the lists of the whole-front substitution kernels
            std::vector<int> height(nfronts, 0), koff(nfronts + 1, 0), kflat;
            for (int64_t t = 0; t < nfronts; ++t) { for (int c : h->kids[t]) { height[t] = std::max(height[t], height[c] + 1); kflat.push_back(c); } koff[t + 1] = (int)kflat.size(); }
            int hmax = 0; for (int64_t t = 0; t < nfronts; ++t) hmax = std::max(hmax, height[t]);
            h->levels.assign(hmax + 1, {});
            std::vector<std::vector<int>> small(hmax + 1);
            const int fuse_max = getenv("GF_SOLVER_FUSE_MAX_BLK") ? std::max(0, std::min(atoi(getenv("GF_SOLVER_FUSE_MAX_BLK")), (int)gfs_handle::FUSE_MAX_BLK)) : gfs_handle::FUSE_MAX_BLK;      // test switch: small models through the large-front kernels
            for (int64_t t = 0; t < nfronts; ++t) (h->fronts[t].nblk_t <= fuse_max ? small[height[t]] : h->levels[height[t]].big).push_back((int)t);
            // A whole-front kernel streams its front with ONE workgroup: fine while a height has hundreds of fronts, not near the top of the small fronts' heights -- at C4 the
            // three heights with 61 / 21 / 4 fronts of up to 96 blocks took 4.7 of the forward kernel's 8.3 ms (profiles/r05_solver_c4_timeline.txt).  Where a height has few small
            // fronts, the larger of them go with the large fronts (group kernels: a workgroup per block row, side by side on the side streams).
            const int few_max = getenv("GF_SOLVER_FEW_FRONTS") ? atoi(getenv("GF_SOLVER_FEW_FRONTS")) : 24, few_blk = getenv("GF_SOLVER_FEW_BLK") ? atoi(getenv("GF_SOLVER_FEW_BLK")) : 32;
            for (int l = 0; l <= hmax; ++l) if ((int)small[l].size() <= few_max) {
                std::vector<int> keep;
                for (int t : small[l]) (h->fronts[t].nblk_t > few_blk ? h->levels[l].big : keep).push_back(t);
                small[l].swap(keep);
            }
            std::vector<int> flat;
            for (int l = 0; l <= hmax; ++l) {
                auto& L = h->levels[l]; L.off_small = (int)flat.size(); L.n_small = (int)small[l].size(); L.max_blk = 1;
                for (int t : small[l]) { flat.push_back(t); L.max_blk = std::max(L.max_blk, h->fronts[t].nblk_t); }
            }
            if (kflat.empty()) kflat.push_back(0);
            if (flat.empty()) flat.push_back(0);
            h->d_lvl_list = h->up(flat.data(), flat.size()); h->d_kid_off = h->up(koff.data(), koff.size()); h->d_kid = h->up(kflat.data(), kflat.size());
        }
        {   // factorisation levels (GF_SOLVER_BATCH_BLK: fronts of at most that many blocks are factored in batched launches per tree height; 0: per front on streams)
            if (const char* e = std::getenv("GF_SOLVER_BATCH_BLK")) h->batch_blk = std::atoi(e);
            std::vector<int> height(nfronts, 0);
            for (int64_t t = 0; t < nfronts; ++t) for (int c : h->kids[t]) height[t] = std::max(height[t], height[c] + 1);
            int hmax = 0; for (int64_t t = 0; t < nfronts; ++t) hmax = std::max(hmax, height[t]);
            h->flevels.assign(hmax + 1, {});
            std::vector<std::vector<int>> small(hmax + 1), all(hmax + 1);
            for (int64_t t = 0; t < nfronts; ++t) {
                all[height[t]].push_back((int)t);
                (h->fronts[t].nblk_t <= h->batch_blk ? small[height[t]] : h->flevels[height[t]].big).push_back((int)t);
            }
            std::vector<int> flist, ealist; std::vector<long long> wofs; long long wmax = 1;
            for (int l = 0; l <= hmax; ++l) {
                auto& L = h->flevels[l];
                std::stable_sort(small[l].begin(), small[l].end(), [&](int x, int y) { return h->fronts[x].nblk_e > h->fronts[y].nblk_e; });
                L.off = (int)flist.size(); L.n = (int)small[l].size();
                long long w = 0;
                for (int t : small[l]) { flist.push_back(t); wofs.push_back(w); w += (long long)std::max(h->batch_panel_w, 1) * std::max(h->fronts[t].nblk_t - 1, 1); }
                wmax = std::max(wmax, w);
                const int kmax = L.n ? h->fronts[small[l][0]].nblk_e : 0;
                L.nk.assign(kmax, 0); L.max_ni.assign(kmax, 0);
                for (int t : small[l]) for (int k = 0; k < h->fronts[t].nblk_e; ++k) { ++L.nk[k]; L.max_ni[k] = std::max(L.max_ni[k], h->fronts[t].nblk_t - 1 - k); }
                // extend-add rounds: round r = the r-th child (with a boundary) of every front of this height
                size_t maxkids = 0; for (int t : all[l]) maxkids = std::max(maxkids, h->kids[t].size());
                for (size_t r = 0; r < maxkids; ++r) {
                    gfs_handle::Round R{(int)ealist.size(), 0, 0};
                    for (int t : all[l]) if (r < h->kids[t].size()) {
                        const int c = h->kids[t][r]; const long long nbb = h->fronts[c].nblk_t - h->fronts[c].nblk_e;
                        if (nbb <= 0) continue;
                        ealist.push_back(c); ++R.n; R.max_tiles = std::max(R.max_tiles, nbb * (nbb + 1) / 2);
                    }
                    if (R.n > 0) L.rounds.push_back(R);
                }
            }
            if (flist.empty()) { flist.push_back(0); wofs.push_back(0); }
            if (ealist.empty()) ealist.push_back(0);
            h->d_flist = h->up(flist.data(), flist.size()); h->d_fwofs = h->up(wofs.data(), wofs.size()); h->d_ealist = h->up(ealist.data(), ealist.size());
            h->bwbuf = h->dalloc<double>((size_t)wmax * NB2);
        }
        h->fbnd = h->dalloc<double>((size_t)std::max<int64_t>(3 * nbnd, 1));
        h->ws_bnd_len = (long long)std::max<int64_t>(3 * nbnd, 1);
        h->gy = h->dalloc<double>(h->n); h->gb = h->dalloc<double>(h->n); h->gx = h->dalloc<double>(h->n);
        h->vr = h->dalloc<double>(h->n); h->vsol = h->dalloc<double>(h->n); h->vrhs = h->dalloc<double>(h->n); h->part = h->dalloc<double>(256);
        HIPCHK(hipMemsetAsync(h->gy, 0, h->n * sizeof(double), h->stream)); HIPCHK(hipMemsetAsync(h->gx, 0, h->n * sizeof(double), h->stream));
        HIPCHK(hipDeviceSynchronize());
    } catch (const std::exception& ex) {
        if (h) gfs_destroy(h);
        return sfail(ex.what());
    }
    *out = h;
    return 0;
}

int gfs_create_nd(int device, int64_t ncp, const int64_t* nb_ptr, const int32_t* nb, const double* d_valK, int64_t nfronts, const int64_t* elim, const int64_t* elim_off,
                  const int64_t* bnd, const int64_t* bnd_off, const int64_t* parent, const int64_t* order, const int64_t* front_of, const int64_t* pmap, gfs_handle** out) {
    return create_nd_impl(device, ncp, nb_ptr, nb, d_valK, nfronts, elim, elim_off, bnd, bnd_off, parent, order, front_of, pmap, false, out);
}
int gfs_create_nd_partial(int device, int64_t ncp, const int64_t* nb_ptr, const int32_t* nb, const double* d_valK, int64_t nfronts, const int64_t* elim, const int64_t* elim_off,
                          const int64_t* bnd, const int64_t* bnd_off, const int64_t* parent, const int64_t* order, const int64_t* front_of, const int64_t* pmap, gfs_handle** out) {
    return create_nd_impl(device, ncp, nb_ptr, nb, d_valK, nfronts, elim, elim_off, bnd, bnd_off, parent, order, front_of, pmap, true, out);
}

void gfs_destroy(gfs_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (void* p : h->allocs) (void)hipFree(p);
    for (int s_ = 0; s_ < gfs_handle::NS; ++s_) { if (h->ev[s_]) (void)hipEventDestroy(h->ev[s_]); if (h->st[s_]) (void)hipStreamDestroy(h->st[s_]); }
    if (h->ev_main) (void)hipEventDestroy(h->ev_main);
    if (h->ev_norm) (void)hipEventDestroy(h->ev_norm);
    if (h->g_factor) (void)hipGraphExecDestroy(h->g_factor);
    if (h->g_solve) (void)hipGraphExecDestroy(h->g_solve);
    if (h->g_fwd) (void)hipGraphExecDestroy(h->g_fwd);
    if (h->g_bwd) (void)hipGraphExecDestroy(h->g_bwd);
    for (size_t k = 0; k < h->ws.size(); ++k) {
        for (auto& g : h->ws[k].g_solve) if (g) (void)hipGraphExecDestroy(g);
        if (k > 0 && h->ws[k].stream) (void)hipStreamDestroy(h->ws[k].stream);
        if (h->ws[k].ev_fork) (void)hipEventDestroy(h->ws[k].ev_fork);
        for (auto& e : h->ws[k].ev_join) if (e) (void)hipEventDestroy(e);
    }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// what a factorisation adds into before it writes it: everything, or ("lazy S", nested dissection through nd_factor_levels) the fronts' eliminated block columns only
static void clear_factor_storage(gfs_handle* h) {
    if (h->nd && h->lazy_s && GF_UPDATE_DMA && h->batch_blk > 0 && h->d_zrows) {
        if (h->n_zrows > 0) hipLaunchKernelGGL(nd_zero_lpart_kernel, dim3((unsigned)h->n_zrows), dim3(256), 0, h->stream, h->d_fronts, h->d_zrows, h->d_tri, h->band);
    } else HIPCHK(hipMemsetAsync(h->band, 0, (size_t)h->ntiles * NB2 * sizeof(double), h->stream));
}
int gfs_prepare_refactor(gfs_handle* h) {
    if (!h) return sfail("gfs_prepare_refactor: null handle");
    try {
        HIPCHK(hipSetDevice(h->device));
        h->factored = false;
        clear_factor_storage(h);
        h->prepared = true;
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}
int gfs_refactor(gfs_handle* h) {
    if (!h) return sfail("gfs_refactor: null handle");
    try {
        HIPCHK(hipSetDevice(h->device));
        h->factored = false;
        // |K|_F (the backward errors' denominator) reads K's 9 nnz doubles once: on a side stream under the factorisation instead of behind it (2.1 ms at C4)
        const bool norm_aside = h->nd && h->st[0] && h->ev_norm && h->part_k;
        if (norm_aside) {
            HIPCHK(hipEventRecord(h->ev_norm, h->stream));
            HIPCHK(hipStreamWaitEvent(h->st[0], h->ev_norm, 0));
            hipLaunchKernelGGL(sumsq_kernel, dim3(240), dim3(256), 0, h->st[0], h->nnz9, h->valK, h->part_k);
            HIPCHK(hipEventRecord(h->ev_norm, h->st[0]));
        }
        if (!h->prepared) clear_factor_storage(h);
        h->prepared = false;
        if (h->nd) {
            hipLaunchKernelGGL(nd_scatter_kernel, dim3((unsigned)((h->ncp * 64 + 255) / 256)), dim3(256), 0, h->stream, h->ncp, h->nb_ptr, h->nb, h->general ? h->d_rev : nullptr, h->valK, h->d_fronts, h->d_front_of, h->d_order,
                               h->d_bnd, h->d_tri, h->band, h->d_row_ok);
            const int nf = (int)h->fronts.size();
            hipLaunchKernelGGL(nd_pad_kernel, dim3(nf), dim3(64), 0, h->stream, h->d_fronts, h->d_tri, h->band);
            if (h->partial) for (int t = 0; t < nf; ++t) if (h->stub_src[t]) {         // a stub front's tiles = the Schur complement another handle exported (one contiguous triangle)
                const Front& F = h->fronts[t];
                HIPCHK(hipMemcpyAsync(h->band + (size_t)F.tile_off * NB2, h->stub_src[t], (size_t)F.nblk_t * (F.nblk_t + 1) / 2 * NB2 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            }
            if (h->batch_blk > 0) nd_run_captured(h, &h->g_factor, [&] { nd_factor_levels(h); });
            else nd_run_captured(h, &h->g_factor, [&] { nd_sweep_up(h, [](gfs_handle* hh, int t, hipStream_t st, int si) { nd_factor_front(hh, t, st, si); }); });
        } else {
        hipLaunchKernelGGL(band_fill_kernel, dim3((unsigned)((h->ncp * 64 + 255) / 256)), dim3(256), 0, h->stream, h->ncp, h->nb_ptr, h->nb, h->newi, h->general ? h->d_rev : nullptr, h->valK, h->band, h->rowoff, h->n, h->npad);
        bool have_diag = false;                                          // the diagonal tile of this block column was factored by the update that completed it
        for (long long k = 0; k < h->nblk; ++k) {
            const int ni = h->nik[k];
            if (!have_diag) hipLaunchKernelGGL(diag_kernel, dim3(1), dim3(256), 0, h->stream, h->band, h->linv, h->dval, h->rowoff, (int)k, h->stat);
            have_diag = false;
            if (ni > 0) {
                hipLaunchKernelGGL(panel_kernel, dim3(ni), dim3(256), 0, h->stream, h->band, h->linv, h->dval, h->wbuf, h->rowoff, (int)k);
                have_diag = h->fuse_diag && k + 1 < h->nblk;              // block column k is the last one that reaches tile (k + 1, k + 1)
                hipLaunchKernelGGL(update_kernel, dim3((unsigned)((long long)ni * (ni + 1) / 2)), dim3(256), 0, h->stream, h->band, h->wbuf, h->rowoff, (int)k, ni,
                                   DiagNext{h->linv, h->dval, h->stat, have_diag ? 1 : 0});
            }
        }
        }
        HIPCHK(hipGetLastError());
        const long long ncol = h->nd ? h->nbe_tot : h->nblk;
        std::vector<double> st(2 * ncol);
        double pk[240];
        HIPCHK(hipMemcpyAsync(st.data(), h->stat, st.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (norm_aside) {
            HIPCHK(hipStreamWaitEvent(h->stream, h->ev_norm, 0));
            HIPCHK(hipMemcpyAsync(pk, h->part_k, sizeof(pk), hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        double mn = 1e300, mx = 0.0;
        for (long long k = 0; k < ncol; ++k) { mn = std::min(mn, st[2 * k]); mx = std::max(mx, st[2 * k + 1]); }
        if (!(mn == mn) || !(mx == mx) || !std::isfinite(mx) || mn == 0.0) throw std::runtime_error("gfs_refactor: zero or non-finite pivot (K is singular for this ordering without pivoting)");
        h->small_pivot = mn < 1e-14 * mx;
        if (norm_aside) { long double a = 0; for (int q = 0; q < 240; ++q) a += pk[q]; h->normK = std::sqrt((double)a); }      // the same partial sums in the same order as norm2
        else h->normK = norm2(h, h->valK, h->nnz9);
        h->factored = true;
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}

int gfs_set_general(gfs_handle* h, int nonsymmetric) {
    if (!h) return sfail("gfs_set_general: null handle");
    try {
        HIPCHK(hipSetDevice(h->device));
        if (nonsymmetric && !h->d_rev) {
            long long nent = 0;
            HIPCHK(hipMemcpy(&nent, h->nb_ptr + h->ncp, sizeof(long long), hipMemcpyDeviceToHost));
            h->d_rev = h->dalloc<int>((size_t)nent);
            int* d_bad = h->dalloc<int>(1); int bad = 0;
            HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(int), h->stream));
            hipLaunchKernelGGL(rev_index_kernel, dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, h->stream, nent, h->ncp, h->nb_ptr, h->nb, h->d_rev, d_bad);
            HIPCHK(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            if (bad) throw std::runtime_error("gfs_set_general: the block pattern of K is not symmetric");
        }
        if (h->general != (nonsymmetric != 0)) h->factored = false;       // the factors belong to the other mode
        h->general = nonsymmetric != 0;
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}

// One or several right-hand sides (device pointers, [nrhs][n]).  The refinement runs in lockstep rounds: the substitutions and residuals of all right-hand
// sides still being refined are issued on their own streams (workspace r), then the norms are read back in one synchronisation per round.  The sweeps are
// chains of small dependent launches (latency, not bandwidth: the device is nearly idle during one), so k right-hand sides cost little more than one.
static int solve_dev_impl(gfs_handle* h, const double* d_b, double* d_x, int max_refine, double* rel_residual, int transpose, int nrhs = 1) {
    if (!h || !d_b || !d_x) return sfail("gfs_solve: null argument");
    if (!h->factored) return sfail("gfs_solve: no factorisation (call gfs_refactor)");
    if (nrhs < 1 || nrhs > gfs_handle::MAX_RHS) return sfail("gfs_solve_multi: 1 .. " + std::to_string(gfs_handle::MAX_RHS) + " right-hand sides");
    transpose = transpose && h->general;                                  // symmetric K: the same system
    try {
        HIPCHK(hipSetDevice(h->device));
        const bool conc = h->nd && nrhs > 1;                              // skyline mode: one after the other on the handle's stream (small models)
        const unsigned gcp = (unsigned)((h->ncp * 64 + 255) / 256);
        const int nblk = 240;
        constexpr int RB = gfs_handle::RHS_BLOCK;
        std::vector<double> nb_(nrhs, 0.0), best(nrhs, -1.0), nx(nrhs, 0.0);
        std::vector<char> active(nrhs, 1);
        std::vector<double> host((size_t)nrhs * 256, 0.0), hostx((size_t)nrhs * 256, 0.0);
        // several right-hand sides: groups of RB share one pass over the factors (substitute_nd<NR>), the groups run next to each other; everything of a group
        // is issued on the stream of the group's first workspace
        auto W = [&](int r) -> gfs_handle::SolveWs& { return solve_ws(h, conc ? r : 0); };
        auto S = [&](int r) -> hipStream_t { return conc ? solve_ws(h, (r / RB) * RB).stream : h->stream; };
        if (conc) { HIPCHK(hipStreamSynchronize(h->stream)); for (int r = 1; r < nrhs; ++r) (void)W(r); }
        auto sumsq_async = [&](int r, const double* v) {
            gfs_handle::SolveWs& w = W(r);
            hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(256), 0, S(r), h->n, v, w.part);
            HIPCHK(hipMemcpyAsync(host.data() + (size_t)r * 256, w.part, nblk * sizeof(double), hipMemcpyDeviceToHost, S(r)));
        };
        auto sumsq_get = [&](int r) { long double a = 0; for (int k = 0; k < nblk; ++k) a += host[(size_t)r * 256 + k]; return std::sqrt((double)a); };
        auto sync_all = [&]() { if (conc) { for (int r = 0; r < nrhs; r += RB) HIPCHK(hipStreamSynchronize(S(r))); } else HIPCHK(hipStreamSynchronize(h->stream)); };
        // substitutions of the right-hand sides with want[r] != 0: x_r (+)= K^-1 rhs_r, rhs_r = first ? d_b_r : W(r).vr, x_r = W(r).vsol
        auto subst_all = [&](const std::vector<char>& want, bool first) {
            if (!h->nd) { substitute(h, first ? d_b : W(0).vr, W(0).vsol, first ? 0 : 1); return; }
            for (int r0 = 0; r0 < nrhs; r0 += RB) {
                const int nr = std::min(RB, nrhs - r0);
                bool any = false; for (int j = 0; j < nr; ++j) any = any || want[r0 + j];
                if (!any) continue;
                auto rhs_of = [&](int r) -> const double* { return !want[r] ? nullptr : (first ? d_b + (size_t)r * h->n : W(r).vr); };
                auto x_of = [&](int r) -> double* { return !want[r] ? nullptr : W(r).vsol; };
                if (nr == 1) { gfs_handle::SolveWs* const Wm[1] = {&W(r0)}; const double* const rh[1] = {rhs_of(r0)}; double* const xx[1] = {x_of(r0)}; substitute_nd<1>(h, Wm, rh, xx, first ? 0 : 1); }
                else if (nr == 2) { gfs_handle::SolveWs* const Wm[2] = {&W(r0), &W(r0 + 1)}; const double* const rh[2] = {rhs_of(r0), rhs_of(r0 + 1)}; double* const xx[2] = {x_of(r0), x_of(r0 + 1)};
                                    substitute_nd<2>(h, Wm, rh, xx, first ? 0 : 1); }
                else { gfs_handle::SolveWs* const Wm[3] = {&W(r0), &W(r0 + 1), &W(r0 + 2)}; const double* const rh[3] = {rhs_of(r0), rhs_of(r0 + 1), rhs_of(r0 + 2)};
                       double* const xx[3] = {x_of(r0), x_of(r0 + 1), x_of(r0 + 2)}; substitute_nd<3>(h, Wm, rh, xx, first ? 0 : 1); }
            }
        };
        if (!conc && nrhs > 1) {                                          // sequential fall-back: each right-hand side through the single-vector path
            double bw = 0.0;
            for (int r = 0; r < nrhs; ++r) {
                double rr = 0.0;
                if (solve_dev_impl(h, d_b + (size_t)r * h->n, d_x + (size_t)r * h->n, max_refine, &rr, transpose, 1)) return 1;
                if (rel_residual) rel_residual[r] = rr;
                bw = std::max(bw, h->backward_error);
            }
            h->backward_error = bw;
            return 0;
        }
        subst_all(active, true);
        for (int r = 0; r < nrhs; ++r) sumsq_async(r, d_b + (size_t)r * h->n);
        sync_all();
        for (int r = 0; r < nrhs; ++r) nb_[r] = sumsq_get(r);
        for (int itr = 0; itr <= max_refine; ++itr) {
            bool any = false;
            for (int r = 0; r < nrhs; ++r) if (active[r]) {
                gfs_handle::SolveWs& w = W(r);
                const double* b = d_b + (size_t)r * h->n;
                if (transpose) hipLaunchKernelGGL(residual_t_kernel, dim3(gcp), dim3(256), 0, S(r), h->ncp, h->nb_ptr, h->nb, h->d_rev, h->valK, b, w.vsol, w.vr);
                else hipLaunchKernelGGL(residual_kernel, dim3(gcp), dim3(256), 0, S(r), h->ncp, h->nb_ptr, h->nb, h->valK, b, w.vsol, w.vr);
                sumsq_async(r, w.vr);
                if (itr == 0) {                                           // |x| of the unrefined solution: is it already at round-off?
                    hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(256), 0, S(r), h->n, w.vsol, w.vrhs);       // vrhs is free until the first correction
                    HIPCHK(hipMemcpyAsync(hostx.data() + (size_t)r * 256, w.vrhs, nblk * sizeof(double), hipMemcpyDeviceToHost, S(r)));
                }
                any = true;
            }
            if (!any) break;
            sync_all();
            std::vector<char> correct(nrhs, 0);
            for (int r = 0; r < nrhs; ++r) if (active[r]) {
                gfs_handle::SolveWs& w = W(r);
                const double nr = sumsq_get(r);
                if (itr == 0) {
                    // a solve whose normwise backward error |b - K x| / (|K|_F |x| + |b|) is already a tenth of the unit round-off cannot be improved by refinement
                    // in working precision: every further sweep would read the factors twice for nothing (C4: 73 GB per sweep -- the solves are bound by that,
                    // and the stopping rule below needs two more sweeps to find out that nothing halves any more)
                    long double a = 0; for (int k = 0; k < nblk; ++k) a += hostx[(size_t)r * 256 + k];
                    const double nx0 = std::sqrt((double)a);
                    static const double early = getenv("GF_SOLVER_EARLY_STOP") ? atof(getenv("GF_SOLVER_EARLY_STOP")) : 0.0;      // off by default: at C4 the sweeps after the first still gain a decade of residual (profiles/r04_device_solver_bench.txt); callers that do not need it pass max_refine = 0
                    if (nr <= early * (h->normK * nx0 + nb_[r])) { best[r] = nr; active[r] = 0; continue; }
                }
                // symmetric mode: refinement only polishes round-off, so a step that does not halve the residual ends it; general mode: the refinement IS the solver
                // for the skew part and contracts by |S^-1 (K - S)|, which may be anything below one -- it goes on while the residual drops at all
                if (best[r] >= 0.0 && !(nr < (h->general ? 0.95 : 0.5) * best[r])) {      // the last correction did not help: keep the previous iterate
                    if (nr >= best[r]) HIPCHK(hipMemcpyAsync(w.vsol, w.vrhs, h->n * sizeof(double), hipMemcpyDeviceToDevice, S(r))); else best[r] = nr;
                    active[r] = 0;
                    continue;
                }
                best[r] = nr;
                if (itr == max_refine || nr == 0.0) { active[r] = 0; continue; }
                HIPCHK(hipMemcpyAsync(w.vrhs, w.vsol, h->n * sizeof(double), hipMemcpyDeviceToDevice, S(r)));     // previous iterate
                correct[r] = 1;
            }
            subst_all(correct, false);                                    // the corrections of a group in one pass over the factors
        }
        for (int r = 0; r < nrhs; ++r) {
            HIPCHK(hipMemcpyAsync(d_x + (size_t)r * h->n, W(r).vsol, h->n * sizeof(double), hipMemcpyDeviceToDevice, S(r)));
            sumsq_async(r, d_x + (size_t)r * h->n);
        }
        sync_all();
        // normwise backward error |b - K x| / (|K|_F |x| + |b|): the measure a backward-stable solve keeps at round-off level whatever
        // the conditioning (|b - K x| / |b| alone has a floor of eps cond(K)); several right-hand sides: the largest
        double bw = 0.0;
        for (int r = 0; r < nrhs; ++r) {
            nx[r] = sumsq_get(r);
            if (rel_residual) rel_residual[r] = nb_[r] > 0.0 ? best[r] / nb_[r] : best[r];
            const double den = h->normK * nx[r] + nb_[r];
            bw = std::max(bw, den > 0.0 ? best[r] / den : best[r]);
        }
        h->backward_error = bw;
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}

int gfs_solve_dev(gfs_handle* h, const double* d_b, double* d_x, int max_refine, double* rel_residual) { return solve_dev_impl(h, d_b, d_x, max_refine, rel_residual, 0); }
int gfs_solve_transposed_dev(gfs_handle* h, const double* d_b, double* d_x, int max_refine, double* rel_residual) { return solve_dev_impl(h, d_b, d_x, max_refine, rel_residual, 1); }

static int solve_host_impl(gfs_handle* h, const double* b, double* x, int max_refine, double* rel_residual, int transpose, int nrhs = 1) {
    if (!h || !b || !x) return sfail("gfs_solve: null argument");
    if (nrhs < 1 || nrhs > gfs_handle::MAX_RHS) return sfail("gfs_solve_multi: 1 .. " + std::to_string(gfs_handle::MAX_RHS) + " right-hand sides");
    try {
        HIPCHK(hipSetDevice(h->device));
        double *db = nullptr, *dx = nullptr;
        const size_t nb_ = (size_t)nrhs * h->n * sizeof(double);
        HIPCHK(hipMalloc(&db, nb_));
        if (hipMalloc(&dx, nb_) != hipSuccess) { (void)hipFree(db); throw std::runtime_error("gfs_solve: out of device memory"); }
        int rc = 1;
        if (hipMemcpy(db, b, nb_, hipMemcpyHostToDevice) == hipSuccess) {
            rc = solve_dev_impl(h, db, dx, max_refine, rel_residual, transpose, nrhs);
            if (!rc && hipMemcpy(x, dx, nb_, hipMemcpyDeviceToHost) != hipSuccess) rc = sfail("gfs_solve: copy to the host failed");
        } else rc = sfail("gfs_solve: copy to the device failed");
        (void)hipFree(db); (void)hipFree(dx);
        return rc;
    } catch (const std::exception& ex) { return sfail(ex.what()); }
}

int gfs_solve(gfs_handle* h, const double* b, double* x, int max_refine, double* rel_residual) { return solve_host_impl(h, b, x, max_refine, rel_residual, 0); }
int gfs_solve_transposed(gfs_handle* h, const double* b, double* x, int max_refine, double* rel_residual) { return solve_host_impl(h, b, x, max_refine, rel_residual, 1); }

int gfs_solve_multi(gfs_handle* h, int nrhs, const double* b, double* x, int max_refine, double* rel_residual, int transpose) { return solve_host_impl(h, b, x, max_refine, rel_residual, transpose, nrhs); }
int gfs_solve_multi_dev(gfs_handle* h, int nrhs, const double* d_b, double* d_x, int max_refine, double* rel_residual, int transpose) { return solve_dev_impl(h, d_b, d_x, max_refine, rel_residual, transpose, nrhs); }

// ---- symbolic phase on the host (gf_nd_symbolic.hpp): no device needed
struct gfs_symbolic { gfnd::Symbolic S; };
int gfs_symbolic_create(int64_t ncp, const int64_t* nb_ptr, const int32_t* nb, const double* coords, int dim, int64_t leaf, double cut_window, int threads, gfs_symbolic** out) {
    if (!out || !nb_ptr || !nb || !coords) return sfail("gfs_symbolic_create: null argument");
    *out = nullptr;
    try {
        gfs_symbolic* s = new gfs_symbolic();
        try { s->S = gfnd::nested_dissection(ncp, nb_ptr, nb, coords, dim, leaf, cut_window, threads); }
        catch (...) { delete s; throw; }
        *out = s;
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}
void gfs_symbolic_sizes(const gfs_symbolic* s, int64_t* nfronts, int64_t* nbnd) {
    if (nfronts) *nfronts = s ? s->S.nfronts : 0;
    if (nbnd) *nbnd = s ? (int64_t)s->S.bnd.size() : 0;
}
void gfs_symbolic_copy(const gfs_symbolic* s, int64_t* elim, int64_t* elim_off, int64_t* bnd, int64_t* bnd_off, int64_t* parent, int64_t* order, int64_t* front_of, int64_t* pmap) {
    if (!s) return;
    auto cp = [](const std::vector<int64_t>& v, int64_t* dst) { if (dst && !v.empty()) std::copy(v.begin(), v.end(), dst); };
    cp(s->S.elim, elim); cp(s->S.elim_off, elim_off); cp(s->S.bnd, bnd); cp(s->S.bnd_off, bnd_off); cp(s->S.parent, parent); cp(s->S.order, order); cp(s->S.front_of, front_of); cp(s->S.pmap, pmap);
}
void gfs_symbolic_destroy(gfs_symbolic* s) { delete s; }

// ---- partial handles: the pieces a distributed factorisation is put together from (goldfish_amd/_dsolver.py)
static int partial_front(gfs_handle* h, int64_t front, const char* who) {
    if (!h || !h->nd) return sfail(std::string(who) + ": needs a nested-dissection handle");
    if (front < 0 || front >= (int64_t)h->fronts.size()) return sfail(std::string(who) + ": front index out of range");
    return 0;
}
int64_t gfs_schur_doubles(gfs_handle* h, int64_t front) {
    if (partial_front(h, front, "gfs_schur_doubles")) return -1;
    const Front& F = h->fronts[front]; const long long nbb = F.nblk_t - F.nblk_e;
    return (int64_t)(nbb * (nbb + 1) / 2 * NB2);
}
int gfs_export_schur(gfs_handle* h, int64_t front, double* d_buf) {
    if (partial_front(h, front, "gfs_export_schur")) return 1;
    if (!h->factored) return sfail("gfs_export_schur: no factorisation (call gfs_refactor)");
    if (!d_buf) return sfail("gfs_export_schur: null buffer");
    try {
        HIPCHK(hipSetDevice(h->device));
        const Front& F = h->fronts[front]; const long long nbb = F.nblk_t - F.nblk_e;
        if (nbb > 0) hipLaunchKernelGGL(nd_export_schur_kernel, dim3((unsigned)(nbb * (nbb + 1) / 2)), dim3(256), 0, h->stream, F, h->d_tri, h->band, d_buf);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}
int gfs_set_schur_source(gfs_handle* h, int64_t front, const double* d_buf) {
    if (partial_front(h, front, "gfs_set_schur_source")) return 1;
    if (!h->partial || h->fronts[front].ne_cp != 0) return sfail("gfs_set_schur_source: not a stub front of a partial handle");
    h->stub_src[front] = d_buf; h->factored = false;
    return 0;
}
// boundary contributions of root fronts: `n` fronts packed one after the other (3 doubles per boundary control point, in the order of each front's boundary list);
// asynchronous copies on the handle's sweep stream, one synchronisation per call
static int fbnd_copy(gfs_handle* h, int64_t n, const int64_t* fronts, double* d_buf, bool get, const char* who) {
    if (!h || !h->nd) return sfail(std::string(who) + ": needs a nested-dissection handle");
    if (!h->partial) return sfail(std::string(who) + ": needs a partial handle (gfs_create_nd_partial)");
    if (!d_buf || (n > 0 && !fronts)) return sfail(std::string(who) + ": null argument");
    if (get && !h->factored) return sfail(std::string(who) + ": no factorisation (call gfs_refactor)");
    for (int64_t k = 0; k < n; ++k) if (fronts[k] < 0 || fronts[k] >= (int64_t)h->fronts.size()) return sfail(std::string(who) + ": front index out of range");
    try {
        HIPCHK(hipSetDevice(h->device));
        gfs_handle::SolveWs& W0 = solve_ws(h, 0);
        size_t off = 0;
        for (int64_t k = 0; k < n; ++k) {
            const Front& F = h->fronts[fronts[k]];
            const size_t len = (size_t)3 * F.nb_cp;
            if (len) {
                if (get) HIPCHK(hipMemcpyAsync(d_buf + off, W0.fbnd + 3 * F.bnd_off, len * sizeof(double), hipMemcpyDeviceToDevice, W0.stream));
                else HIPCHK(hipMemcpyAsync(W0.fbnd + 3 * F.bnd_off, d_buf + off, len * sizeof(double), hipMemcpyDeviceToDevice, W0.stream));
            }
            off += len;
        }
        HIPCHK(hipStreamSynchronize(W0.stream));
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}
// rows of d_valK that hold values: mask [ncp] (host), 1 = the control point's three rows are assembled in this K (a rank's owned rows of a sharded model), 0 = not
// (ghost rows).  The scatter of a partial handle then reads a block whose later control point has no row from the earlier one's row, transposed.  nullptr: all rows.
int gfs_set_row_mask(gfs_handle* h, const unsigned char* mask) {
    if (!h || !h->nd) return sfail("gfs_set_row_mask: needs a nested-dissection handle");
    try {
        HIPCHK(hipSetDevice(h->device));
        if (!mask) { h->d_row_ok = nullptr; h->factored = false; return 0; }
        if (!h->d_row_ok) h->d_row_ok = h->dalloc<unsigned char>((size_t)h->ncp);
        HIPCHK(hipMemcpy(h->d_row_ok, mask, (size_t)h->ncp, hipMemcpyHostToDevice));
        h->factored = false;
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}
int gfs_get_fbnd(gfs_handle* h, int64_t front, double* d_out) { return fbnd_copy(h, 1, &front, d_out, true, "gfs_get_fbnd"); }
int gfs_set_fbnd(gfs_handle* h, int64_t front, const double* d_in) { return fbnd_copy(h, 1, &front, const_cast<double*>(d_in), false, "gfs_set_fbnd"); }
int gfs_get_fbnd_packed(gfs_handle* h, int64_t n, const int64_t* fronts, double* d_out) { return fbnd_copy(h, n, fronts, d_out, true, "gfs_get_fbnd_packed"); }
int gfs_set_fbnd_packed(gfs_handle* h, int64_t n, const int64_t* fronts, const double* d_in) { return fbnd_copy(h, n, fronts, const_cast<double*>(d_in), false, "gfs_set_fbnd_packed"); }
double* gfs_x_ptr(gfs_handle* h) { return (h && h->nd) ? h->gx : nullptr; }
// forward half: y of this handle's eliminated dofs and the boundary contributions of its fronts from the right-hand side d_b (3 * ncp doubles, original numbering);
// backward half: x of this handle's eliminated dofs into the vector gfs_x_ptr points at, whose entries at the boundary control points of the root fronts (eliminated by
// another handle) must have been written before.  One right-hand side, the handle's own workspace and stream; both calls return when the device is done.
int gfs_forward_dev(gfs_handle* h, const double* d_b) {
    if (!h || !h->nd || !d_b) return sfail("gfs_forward_dev: needs a nested-dissection handle and a right-hand side");
    if (!h->factored) return sfail("gfs_forward_dev: no factorisation (call gfs_refactor)");
    try {
        HIPCHK(hipSetDevice(h->device));
        gfs_handle::SolveWs& W0 = solve_ws(h, 0);
        const gfs_handle::SolveWs* const W[1] = {&W0};
        HIPCHK(hipMemcpyAsync(W0.gb, d_b, h->n * sizeof(double), hipMemcpyDeviceToDevice, W0.stream));
        nd_run_captured(h, &h->g_fwd, [&] { nd_forward_all<1>(h, W, W0.stream); }, W0.stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(W0.stream));
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}
int gfs_backward_dev(gfs_handle* h) {
    if (!h || !h->nd) return sfail("gfs_backward_dev: needs a nested-dissection handle");
    if (!h->factored) return sfail("gfs_backward_dev: no factorisation (call gfs_refactor)");
    try {
        HIPCHK(hipSetDevice(h->device));
        gfs_handle::SolveWs& W0 = solve_ws(h, 0);
        const gfs_handle::SolveWs* const W[1] = {&W0};
        nd_run_captured(h, &h->g_bwd, [&] { nd_backward_all<1>(h, W, W0.stream); }, W0.stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(W0.stream));
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}

int gfs_info(gfs_handle* h, double info[8]) {
    if (!h || !info) return sfail("gfs_info: null argument");
    info[0] = (double)h->bw; info[1] = (double)h->nblk; info[2] = (double)(h->T + 1); info[3] = (double)h->bytes;
    double fl = h->nd ? h->nd_flops : 0.0;
    for (long long k = 0; !h->nd && k < h->nblk; ++k) { const double ni = (double)h->nik[k]; fl += 2.0 * NB * NB * NB * (ni + ni * (ni + 1) / 2) + 2.0 * NB * NB * NB / 3; }
    info[4] = fl; info[5] = h->small_pivot ? 1.0 : 0.0; info[6] = h->backward_error; info[7] = h->normK;
    return 0;
}

}  // extern "C"
