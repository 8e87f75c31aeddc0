// gf_solver.hip -- C ABI (include/goldfish_solver.h): device-resident re-factorisation and solves with K (rocSOLVER csrrf).
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC gf_solver.hip -o ../libgoldfish_solver.so -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/goldfish_solver.h"

static thread_local std::string g_serr;
static int sfail(const std::string& m) { g_serr = m; return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define RBCHK(x) do { rocblas_status s_ = (x); if (s_ != rocblas_status_success) throw std::runtime_error(std::string(#x) + ": rocblas status " + std::to_string((int)s_)); } while (0)

struct gfs_handle {
    int device = 0; hipStream_t stream = nullptr; rocblas_handle rb = nullptr; rocsolver_rfinfo rf = nullptr;
    rocblas_int n = 0, nnzA = 0, nnzT = 0;
    rocblas_int *ptrA = nullptr, *indA = nullptr, *ptrT = nullptr, *indT = nullptr, *pivP = nullptr, *pivQ = nullptr;
    double *valA = nullptr, *valT = nullptr, *B = nullptr;
    std::vector<void*> allocs; long long bytes = 0;
    template <class T> T* up(const T* src, size_t cnt) {
        void* p = nullptr; HIPCHK(hipMalloc(&p, (cnt ? cnt : 1) * sizeof(T))); allocs.push_back(p); bytes += (long long)(cnt * sizeof(T));
        if (src && cnt) HIPCHK(hipMemcpy(p, src, cnt * sizeof(T), hipMemcpyHostToDevice));
        return (T*)p;
    }
};

extern "C" {

const char* gfs_last_error(void) { return g_serr.c_str(); }

int gfs_create(int device, int64_t n, int64_t nnzA, const int32_t* ptrA, const int32_t* indA, const double* d_valA,
               int64_t nnzT, const int32_t* ptrT, const int32_t* indT, const double* valT,
               const int32_t* pivP, const int32_t* pivQ, gfs_handle** out) {
    if (!out || !ptrA || !indA || !d_valA || !ptrT || !indT || !valT || !pivP || !pivQ) return sfail("gfs_create: null argument");
    *out = nullptr;
    if (n <= 0 || nnzA <= 0 || nnzT <= 0 || n >= (int64_t(1) << 31) || nnzA >= (int64_t(1) << 31) || nnzT >= (int64_t(1) << 31))
        return sfail("gfs_create: sizes must be positive and fit 32-bit indices (rocSOLVER csrrf)");
    gfs_handle* h = nullptr;
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) throw std::runtime_error("gfs_create: no such HIP device");
        h = new gfs_handle(); h->device = device;
        HIPCHK(hipSetDevice(device));
        const bool verbose = getenv("GFS_VERBOSE") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) { if (verbose) { auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "[gfs_create] %s %.2f s\n", what, std::chrono::duration<double>(t1 - t0).count()); t0 = t1; } };
        HIPCHK(hipStreamCreate(&h->stream));
        RBCHK(rocblas_create_handle(&h->rb));
        lap("rocblas_create_handle");
        RBCHK(rocblas_set_stream(h->rb, h->stream));
        h->n = (rocblas_int)n; h->nnzA = (rocblas_int)nnzA; h->nnzT = (rocblas_int)nnzT;
        h->ptrA = h->up(ptrA, n + 1); h->indA = h->up(indA, nnzA); h->valA = const_cast<double*>(d_valA);
        h->ptrT = h->up(ptrT, n + 1); h->indT = h->up(indT, nnzT); h->valT = h->up(valT, nnzT);
        h->pivP = h->up(pivP, n); h->pivQ = h->up(pivQ, n);
        h->B = h->up<double>(nullptr, n);
        lap("uploads");
        RBCHK(rocsolver_create_rfinfo(&h->rf, h->rb));
        lap("rocsolver_create_rfinfo");
        RBCHK(rocsolver_dcsrrf_analysis(h->rb, h->n, 1, h->nnzA, h->ptrA, h->indA, h->valA, h->nnzT, h->ptrT, h->indT, h->valT, h->pivP, h->pivQ, h->B, h->n, h->rf));
        HIPCHK(hipStreamSynchronize(h->stream));
        lap("rocsolver_dcsrrf_analysis");
    } catch (const std::exception& ex) { if (h) gfs_destroy(h); return sfail(ex.what()); }
    *out = h;
    return 0;
}

void gfs_destroy(gfs_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->rf) (void)rocsolver_destroy_rfinfo(h->rf);
    if (h->rb) (void)rocblas_destroy_handle(h->rb);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int gfs_refactor(gfs_handle* h) {
    if (!h) return sfail("gfs_refactor: null handle");
    try {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipDeviceSynchronize());                    // the assembly that produced the new values ran on another stream
        RBCHK(rocsolver_dcsrrf_refactlu(h->rb, h->n, h->nnzA, h->ptrA, h->indA, h->valA, h->nnzT, h->ptrT, h->indT, h->valT, h->pivP, h->pivQ, h->rf));
        HIPCHK(hipStreamSynchronize(h->stream));
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}

int gfs_solve(gfs_handle* h, const double* b, double* x) {
    if (!h || !b || !x) return sfail("gfs_solve: null argument");
    try {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipMemcpyAsync(h->B, b, (size_t)h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        RBCHK(rocsolver_dcsrrf_solve(h->rb, h->n, 1, h->nnzT, h->ptrT, h->indT, h->valT, h->pivP, h->pivQ, h->B, h->n, h->rf));
        HIPCHK(hipMemcpyAsync(x, h->B, (size_t)h->n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    } catch (const std::exception& ex) { return sfail(ex.what()); }
    return 0;
}

int64_t gfs_nnz_factors(gfs_handle* h) { return h ? h->nnzT : 0; }
int64_t gfs_device_bytes(gfs_handle* h) { return h ? h->bytes : 0; }

}  // extern "C"
