// gf_lib.hip -- C ABI (include/goldfish_hip.h) of the MI355X shell assembly + sensitivity path.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC gf_lib.hip -o ../libgoldfish_hip.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace gf {
constexpr int GF_ASM_R_BIT = 1, GF_ASM_K_BIT = 2, GF_ASM_C_BIT = 4, GF_ASM_H_BIT = 8;
}
#include "../../include/goldfish_hip.h"
#include "gf_kernels.hpp"
#include "gf_element_mfma.hpp"
#include "gf_element_mfma4.hpp"
#include "gf_element_rec.hpp"
#include "gf_element_rec4.hpp"
#include "gf_penalty_row16.hpp"
#include "gf_penalty_point16.hpp"
#include "gf_extra_loads.hpp"

using namespace gf;

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct Chunk { int p0, p1; long long e0, e1, a0, a1; };

struct gf_handle {
    int device = 0; hipStream_t stream = nullptr;
    HostModel H;
    std::vector<void*> allocs; long long bytes = 0;
    DevModel M{}; DevPenalty Q{};
    double *d_cp4 = nullptr, *d_u = nullptr, *d_h = nullptr, *d_R = nullptr, *d_blk = nullptr, *d_pbuf = nullptr;
    double* d_val[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double *d_x = nullptr, *d_y = nullptr;           // staging for host-pointer gf_apply / per-element partial sums
    const long long *d_elem_off = nullptr, *d_if_off = nullptr; double* d_red = nullptr; long long nred = 0;   // per-patch / per-interface reductions on the device
    double* d_dxi = nullptr; long long dxi_doubles = 0;   // gf_penalty_dxi: block buffer kept between calls
    int fun_owner = -1;                              // which entry wrote d_fun last (0 = gf_functionals)
    std::vector<double> fun_wp, fun_vp;              // per-patch W_int / volume of the last gf_functionals call (gf_functionals_per_patch)
    double* d_many = nullptr;                        // gf_apply_many: five more vectors of ndof doubles (allocated on first use)
    double* d_pt_nu2 = nullptr;                                        // second derivatives of the basis at the mortar vertices (gf_penalty_dxi)
    double *d_fun = nullptr, *d_pen_en = nullptr, *d_ve = nullptr;    // functional gradients [11*total_cp], penalty energies [npts]
    long long* d_pl_dof = nullptr; double* d_pl_val = nullptr;
    std::vector<Chunk> chunks;
    std::vector<Chunk> rchunks;                      // p = 4 row-record path: its own chunks of patches (== chunks unless hybrid)
    bool hybrid4 = false;                            // p = 4 default: Newton passes (R, K) through the row records, passes with dR/dCP / dR/dh through element blocks
    std::vector<hipEvent_t> ev0, ev1; int ev_n = 0;   // element-kernel timing
    bool assembled[5] = {false, false, false, false, false};
    bool rec = false;                                 // p = 2, 3, MFMA path, default: walking kernel that stores row records + kl_gather_rec_kernel (gf_element_rec.hpp); GF_ASSEMBLY=block: one block per element + row gather
    int sumfact = 2;                // p = 3 walking kernel (GF_SUMFACT): 0 the 16 x 16 x 4 products for every item, 1 row-side sum factorisation on polynomial patches, 2 on rational ones too
    const WalkItem* d_rec_items = nullptr; const int* d_rec_order = nullptr; const RecCp* d_rec_cp = nullptr; double* d_rec = nullptr; long long rec_doubles = 0;
    bool rec4 = false; const RecCp4* d_rec_cp4 = nullptr;   // p = 4, default: three walks that store row records + kl_gather_rec4_kernel (gf_element_rec4.hpp); GF_ASSEMBLY=block: element blocks
    bool mfma = true;                                 // p = 3: contraction on the FP64 matrix pipe (GF_ELEMENT=valu selects the VALU kernel)
    const int *d_rev_s = nullptr, *d_rev_c = nullptr;
    const int* d_load_cps = nullptr;                  // control points of the follower pressure / edge tractions (gf_extra_loads.hpp)
    bool gather1 = true;                              // element-block path: one-wave gather for p <= 3, four-wave gather for p = 4
    int pen_maxdeg = 0;                               // largest neighbour count of an interface control point
    bool pen16 = true;                                // p = 2, 3: pen_point16_kernel + pen_row16_kernel (16 lanes per mortar vertex / per visit); p = 4 and GF_PENALTY=owner: pen_point_kernel + pen_owner_kernel

    template <class T> T* dalloc(size_t n) {
        void* p = nullptr; const size_t nb = (n > 0 ? n : 1) * sizeof(T);
        HIPCHK(hipMalloc(&p, nb)); allocs.push_back(p); bytes += (long long)nb; return (T*)p;
    }
    template <class T> T* upload(const std::vector<T>& v) {
        T* p = dalloc<T>(v.size());
        if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return p;
    }
};

extern "C" {

int gf_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
const char* gf_last_error(void) { return g_err.c_str(); }

int gf_create(const gf_model_desc* desc, int device, gf_handle** out) {
    if (!desc || !out) return fail("gf_create: null argument");
    *out = nullptr;
    gf_handle* h = nullptr;
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("gf_create: no HIP device visible (libgoldfish_hip has no CPU fallback)");
        if (device < 0 || device >= ndev) throw std::runtime_error("gf_create: device index out of range");
        h = new gf_handle(); h->device = device;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreate(&h->stream));
        h->H.build(desc);
        if (const char* s = getenv("GF_ELEMENT")) h->mfma = std::string(s) != "valu";
        h->gather1 = h->H.degree <= 3;                    // p = 4: 25 elements x 75-wide rows per control point are bandwidth bound either way (57.9 vs 57.6 ms per step)
        HostModel& H = h->H;
        {
            // p = 2, 3 on the matrix pipe: row records + record gather (gf_element_rec.hpp) by default; GF_ASSEMBLY=block: one block per
            // element + row gather (the cross-check path of the tests)
            bool want_rec = h->mfma && H.degree <= 4;
            if (const char* s = getenv("GF_ASSEMBLY")) want_rec = want_rec && std::string(s) != "block";
            int seg = 0;                                      // whole strips unless the model is small (HostModel::build_rec); GF_REC_SEG: elements per work item
            if (const char* s = getenv("GF_REC_SEG")) seg = std::max(1, atoi(s));
            if (const char* s = getenv("GF_SUMFACT")) h->sumfact = std::max(0, std::min(2, atoi(s)));
            H.tick("penalty owner lists, visit records");
            if (want_rec) { H.build_rec(seg); h->rec = H.degree <= 3; h->rec4 = H.degree == 4; H.tick("row-record tables"); }
            // p = 4: each pass kind on the path that is faster for it (same-lease A/B on one GPU's share of C5, profiles/r04_c5share_*: Newton pass 21.3 ms through
            // the records against 24.2 through element blocks, full pass 60.0 against 54.4).  GF_ASSEMBLY=rec: records for every pass (least memory and traffic),
            // =block: element blocks for every pass
            h->hybrid4 = h->rec4;
            if (const char* s = getenv("GF_ASSEMBLY")) h->hybrid4 = h->hybrid4 && std::string(s) != "rec";

        }
        std::vector<long long> nbs(H.nb_ptr_s.begin(), H.nb_ptr_s.end()), nbc(H.nb_ptr_c.begin(), H.nb_ptr_c.end());
        h->d_cp4 = h->dalloc<double>(4 * H.total_cp); h->d_u = h->dalloc<double>(H.ndof); h->d_h = h->dalloc<double>(H.total_cp); h->d_R = h->dalloc<double>(H.ndof);
        {   // weights into the 4th slot, rest zero until gf_set_cp
            std::vector<double> c4(4 * H.total_cp, 0.0);
            for (long long a = 0; a < H.total_cp; ++a) c4[4 * a + 3] = H.weights[a];
            HIPCHK(hipMemcpy(h->d_cp4, c4.data(), c4.size() * sizeof(double), hipMemcpyHostToDevice));
            HIPCHK(hipMemset(h->d_u, 0, H.ndof * sizeof(double))); HIPCHK(hipMemset(h->d_h, 0, H.total_cp * sizeof(double)));
        }
        DevModel& M = h->M;
        M.patches = h->upload(H.patches); M.tab = h->upload(H.tab); M.ints = h->upload(H.ints);
        M.elem_patch = h->upload(H.elem_patch); M.cp_patch = h->upload(H.cp_patch); M.edesc = h->upload(H.elem_desc); M.cpdesc = h->upload(H.cp_desc); M.nb_meta = h->upload(H.nb_meta);
        h->d_rev_s = h->upload(H.nb_rev_s); h->d_rev_c = h->upload(H.nb_rev_c);
        h->d_load_cps = h->upload(H.load_cps);
        M.cp4 = h->d_cp4; M.u = h->d_u; M.h = h->d_h; M.zero = h->upload(H.zero);
        M.nb_ptr_s = h->upload(nbs); M.nb_s = h->upload(H.nb_s); M.nb_ptr_c = h->upload(nbc); M.nb_c = h->upload(H.nb_c);
        M.total_cp = H.total_cp; M.nelem = H.nelem;
        std::vector<long long> pld(H.pl_dof.begin(), H.pl_dof.end());
        h->d_pl_dof = h->upload(pld); h->d_pl_val = h->upload(H.pl_val);
        // CSR value arrays
        const long long nnzc = H.nb_ptr_c[H.total_cp], nnzs = H.nb_ptr_s[H.total_cp];
        h->d_val[GF_MAT_K] = h->dalloc<double>(9 * nnzc);
        for (int f = 0; f < 3; ++f) h->d_val[GF_MAT_DRDCP0 + f] = h->dalloc<double>(3 * nnzc);
        h->d_val[GF_MAT_DRDH] = h->dalloc<double>(3 * nnzs);
        HIPCHK(hipMemset(h->d_val[GF_MAT_K], 0, 9 * nnzc * sizeof(double)));
        for (int f = 0; f < 3; ++f) HIPCHK(hipMemset(h->d_val[GF_MAT_DRDCP0 + f], 0, 3 * nnzc * sizeof(double)));
        HIPCHK(hipMemset(h->d_val[GF_MAT_DRDH], 0, 3 * nnzs * sizeof(double)));
        HIPCHK(hipMemset(h->d_R, 0, H.ndof * sizeof(double)));
        h->d_x = h->dalloc<double>(H.ndof); h->d_y = h->dalloc<double>(H.ndof);
        h->d_fun = h->dalloc<double>(11 * H.total_cp); h->d_pen_en = h->dalloc<double>(H.npts); h->d_ve = h->dalloc<double>(H.nelem);
        HIPCHK(hipMemset(h->d_fun, 0, 11 * H.total_cp * sizeof(double)));
        {
            std::vector<long long> eo(H.np + 1), io(H.if_off.begin(), H.if_off.end());
            for (int s = 0; s < H.np; ++s) eo[s] = H.patches[s].elem_off;
            eo[H.np] = H.nelem;
            h->d_elem_off = h->upload(eo); h->d_if_off = h->upload(io);
            h->nred = std::max<long long>(H.np, H.ni) + 1; h->d_red = h->dalloc<double>(3 * (size_t)h->nred);
        }
        // penalty
        DevPenalty& Q = h->Q;
        std::vector<unsigned char> pen_row(H.total_cp, 0);
        Q.npts = H.npts;
        if (H.npts > 0) {
            std::vector<long long> rp(H.row_ptr.begin(), H.row_ptr.end()), ep(H.ent_ptr.begin(), H.ent_ptr.end());
            Q.pt_iface = h->upload(H.pt_iface); Q.pt_base = h->upload(H.pt_base); Q.pt_nu = h->upload(H.pt_nu);
            Q.pt_tau = h->upload(H.pt_tau); Q.pt_wt = h->upload(H.pt_wt); Q.if_patch = h->upload(H.if_patch); Q.if_alpha = h->upload(H.if_alpha);
            Q.entries = h->upload(H.pen_entries); Q.ent_ptr = h->upload(ep); Q.row_cp = h->upload(H.row_cp);
            Q.slots = H.degree <= 4 ? h->upload(H.pen_slots) : nullptr;
            h->pen16 = H.degree <= 4;                      // p = 4: the vertex records by pen_point_kernel, the rows by the 32-lane form of pen_row16_kernel (round 5)
            if (const char* s = getenv("GF_PENALTY")) h->pen16 = h->pen16 && std::string(s) != "owner";
            Q.nrow_groups = (long long)rp.size() - 1;
            for (long long g = 0; g + 1 < (long long)rp.size(); ++g) if (rp[g + 1] > rp[g]) pen_row[H.row_items[rp[g]].a] = 1;
            h->d_pbuf = h->dalloc<double>((size_t)H.npts * PB_STRIDE);
            for (const PenRowItem& it : H.row_items)
                h->pen_maxdeg = std::max(h->pen_maxdeg, (int)(H.nb_ptr_c[it.a + 1] - H.nb_ptr_c[it.a]));
            for (int i = 0; i < H.ni; ++i) if (H.if_patch[2 * i] == H.if_patch[2 * i + 1]) throw std::runtime_error("gf_create: self-interfaces (both sides on one patch) are not supported");
            for (const PenRowItem& it : H.row_items)
                if (H.nb_ptr_c[it.a + 1] - H.nb_ptr_c[it.a] > PEN_MAXDEG)
                    throw std::runtime_error("gf_create: a control point couples to more than " + std::to_string(PEN_MAXDEG) + " neighbours (PEN_MAXDEG)");
        }
        M.pen_row = h->upload(pen_row);
        for (long long a = 0; a < H.total_cp; ++a)
            if (H.nb_ptr_c[a + 1] - H.nb_ptr_c[a] > GATHER_MAXMETA) throw std::runtime_error("gf_create: a control point has more than " + std::to_string(GATHER_MAXMETA) + " neighbours (GATHER_MAXMETA)");
        // element-block scratch, chunked over whole patches
        const int P = H.degree, NB = (P + 1) * (P + 1), ND = 3 * NB;
        const bool recs = h->rec || h->rec4;
        // doubles of scratch per element: element blocks, or (row-record paths) the residual entries + the element's share of the records
        const long long rec_sz = h->rec4 ? (h->hybrid4 ? Rec4Cfg<9>::SZ : Rec4Cfg<21>::SZ) : RecCfg<true>::SZ;
        const long long full_blk = 2LL * ND * ND + (long long)ND * NB + ND;
        const long long blk_doubles = (recs && !h->hybrid4) ? (long long)ND : full_blk;
        // p = 4 records: up to 120 GB of row records per chunk of patches, capped by half of the free device memory -- full C5 (117 GB of records) then runs as ONE chunk on
        // an empty MI355X (288 GB): chunks cost launch tails (3 chunks: 468 ms per pass, 8 chunks: 486 ms, same lease); GF_SCRATCH_GB overrides.
        // Hybrid: the Newton-pass records (9 of 21 values per pair: 50 GB at C5) within a quarter of the free memory, the element blocks of the full pass within 30 % of it
        double budget_gb = 40.0, rec_budget_gb = 1e9;
        if (h->rec4) {
            size_t fb = 0, tb = 0; double free_gb = 240.0;
            if (hipMemGetInfo(&fb, &tb) == hipSuccess) free_gb = (double)fb / 1e9;
            if (h->hybrid4) { rec_budget_gb = std::min(60.0, 0.25 * free_gb); budget_gb = std::min(80.0, 0.30 * free_gb); }
            else { rec_budget_gb = std::min(120.0, 0.5 * free_gb); budget_gb = rec_budget_gb; }
        }
        if (const char* s = getenv("GF_SCRATCH_GB")) { budget_gb = atof(s); if (h->rec4) rec_budget_gb = budget_gb; }
        if (h->rec) budget_gb = 1e9;                      // p = 2, 3 records: one chunk
        // chunks of whole patches within a budget of ``per_elem`` doubles per element
        auto make_chunks = [&](double gb, double per_elem, std::vector<Chunk>& out, long long& biggest, long long& biggest_items) {
            const long long max_elems = std::max<long long>(1, (long long)(gb * 1e9 / (per_elem * 8.0)));
            biggest = 0; biggest_items = 0;
            for (int s = 0; s < H.n_owned;) {
                Chunk c; c.p0 = s; c.e0 = H.patches[s].elem_off; c.a0 = H.patches[s].cp_off;
                long long ne = 0;
                while (s < H.n_owned) {
                    const long long pe = (long long)H.patches[s].nelu * H.patches[s].nelv;
                    if (ne > 0 && ne + pe > max_elems) break;
                    ne += pe; ++s;
                }
                c.p1 = s; c.e1 = c.e0 + ne; c.a1 = (s < H.np) ? H.patches[s].cp_off : H.total_cp;
                out.push_back(c); biggest = std::max(biggest, ne);
                if (recs) {
                    const long long i0 = H.rec_patch[c.p0].item_off, i1 = c.p1 < H.n_owned ? H.rec_patch[c.p1].item_off : (long long)H.rec_items.size();
                    biggest_items = std::max(biggest_items, i1 - i0);
                }
            }
        };
        long long biggest = 0, biggest_items = 0;
        // p = 4 records: ~rec_sz doubles per element (one record row per element row and strip)
        if (h->rec4 && !h->hybrid4) { make_chunks(rec_budget_gb, double(rec_sz + ND), h->chunks, biggest, biggest_items); h->rchunks = h->chunks; }
        else {
            make_chunks(budget_gb, double(blk_doubles), h->chunks, biggest, biggest_items);
            if (h->hybrid4) { long long b2 = 0; make_chunks(rec_budget_gb, double(rec_sz + ND), h->rchunks, b2, biggest_items); biggest = std::max(biggest, (b2 * ND + blk_doubles - 1) / blk_doubles); }
        }
        long long scratch_doubles = biggest * blk_doubles;
        if (recs) {
            scratch_doubles = std::max<long long>(scratch_doubles, H.nelem * (long long)(11 * NB + 2));   // the functionals' element blocks (FunCfg::STRIDE) share the scratch
            h->d_rec_items = h->upload(H.rec_items);
            h->d_rec_order = h->upload(H.rec_order);
            if (h->rec4) h->d_rec_cp4 = h->upload(H.rec_cp4); else h->d_rec_cp = h->upload(H.rec_cp);
            h->rec_doubles = biggest_items * H.rec_rows * rec_sz;
            h->d_rec = h->dalloc<double>((size_t)h->rec_doubles);
        }
        h->d_blk = h->dalloc<double>((size_t)scratch_doubles);
        h->ev0.resize(64); h->ev1.resize(64);
        for (int k = 0; k < 64; ++k) { HIPCHK(hipEventCreate(&h->ev0[k])); HIPCHK(hipEventCreate(&h->ev1[k])); }
        HIPCHK(hipDeviceSynchronize());
        H.tick("device allocations, uploads, memsets");
        if (const char* s = getenv("GF_SETUP_TIMING")) if (std::string(s) == "1") {
            double tot = 0; for (const auto& t : H.timing) tot += t.second;
            fprintf(stderr, "gf_create: %.1f ms (%lld control points, %lld elements, %lld mortar vertices)\n", tot, (long long)H.total_cp, (long long)H.nelem, (long long)H.npts);
            for (const auto& t : H.timing) fprintf(stderr, "  %9.1f ms  %s\n", t.second, t.first.c_str());
        }
    } catch (const std::exception& ex) {
        if (h) gf_destroy(h);
        return fail(ex.what());
    }
    *out = h;
    return 0;
}

void gf_destroy(gf_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (void* p : h->allocs) (void)hipFree(p);
    for (auto e : h->ev0) if (e) (void)hipEventDestroy(e);
    for (auto e : h->ev1) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int64_t gf_total_cp(const gf_handle* h) { return h->H.total_cp; }
int64_t gf_num_dofs(const gf_handle* h) { return h->H.ndof; }
int64_t gf_num_elements(const gf_handle* h) { return h->H.nelem; }
int64_t gf_num_gauss_points(const gf_handle* h) { return h->H.ngp; }
int64_t gf_num_mortar_points(const gf_handle* h) { return h->H.npts; }
int64_t gf_device_bytes(const gf_handle* h) { return h->bytes; }

__global__ void strided_set_kernel(double* dst, int stride, int off, const double* src, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i * stride + off] = src[i];
}
static int set_strided(gf_handle* h, double* dst, int stride, int off, const double* src, int64_t n, int64_t expect, const char* who) {
    if (!h || !src) return fail(std::string(who) + ": null argument");
    if (n != expect) return fail(std::string(who) + ": array length " + std::to_string(n) + " != expected " + std::to_string(expect));
    try {
        HIPCHK(hipSetDevice(h->device));
        if (stride == 1) HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        else {
            HIPCHK(hipMemcpyAsync(h->d_x, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(strided_set_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, dst, stride, off, h->d_x, (long long)n);
        }
        HIPCHK(hipStreamSynchronize(h->stream));
    } catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}
int gf_set_cp(gf_handle* h, int field, const double* cp, int64_t n) {
    if (field < 0 || field > 2) return fail("gf_set_cp: field must be 0, 1 or 2");
    return set_strided(h, h->d_cp4, 4, field, cp, n, h->H.total_cp, "gf_set_cp");
}
int gf_set_thickness(gf_handle* h, const double* v, int64_t n) { return set_strided(h, h->d_h, 1, 0, v, n, h->H.total_cp, "gf_set_thickness"); }
int gf_set_u(gf_handle* h, const double* v, int64_t n) { return set_strided(h, h->d_u, 1, 0, v, n, h->H.ndof, "gf_set_u"); }

int64_t gf_nnz(const gf_handle* h, int which) {
    const HostModel& H = h->H;
    if (which == GF_MAT_K) return 9 * H.nb_ptr_c[H.total_cp];
    if (which == GF_MAT_DRDH) return 3 * H.nb_ptr_s[H.total_cp];
    if (which >= GF_MAT_DRDCP0 && which <= GF_MAT_DRDCP2) return 3 * H.nb_ptr_c[H.total_cp];
    return -1;
}
// control-point-level pattern of K (neighbour lists incl. the control point itself, ascending): what the solver's symbolic phase works on -- 9 times smaller than
// gf_pattern(GF_MAT_K), from which goldfish_amd/_solver.py: control_point_graph would otherwise recover it
int64_t gf_cp_graph_size(const gf_handle* h) { return h ? (int64_t)h->H.nb_c.size() : -1; }
int gf_cp_graph(const gf_handle* h, int64_t* nb_ptr, int32_t* nb) {
    if (!h || !nb_ptr || !nb) return fail("gf_cp_graph: null argument");
    const HostModel& H = h->H;
    std::copy(H.nb_ptr_c.begin(), H.nb_ptr_c.end(), nb_ptr);
    std::copy(H.nb_c.begin(), H.nb_c.end(), nb);
    return 0;
}
int gf_pattern(const gf_handle* h, int which, int64_t* rowptr, int32_t* col) {
    if (!h || !rowptr || !col) return fail("gf_pattern: null argument");
    if (which < 0 || which > 4) return fail("gf_pattern: unknown matrix id");
    const HostModel& H = h->H;
    const std::vector<int64_t>& ptr = which == GF_MAT_DRDH ? H.nb_ptr_s : H.nb_ptr_c; const std::vector<int>& nb = which == GF_MAT_DRDH ? H.nb_s : H.nb_c;
    const int bw = which == GF_MAT_K ? 3 : 1; int64_t pos = 0; rowptr[0] = 0;
    for (int64_t a = 0; a < H.total_cp; ++a) for (int i = 0; i < 3; ++i) {
        for (int64_t k = ptr[a]; k < ptr[a + 1]; ++k) for (int j = 0; j < bw; ++j) col[pos++] = nb[k] * bw + j;
        rowptr[3 * a + i + 1] = pos;
    }
    return 0;
}

}  // extern "C"

// Penalty kernels of one pass: the vertex records, then the rows of the interface control points are WRITTEN (the gather adds the
// shell part afterwards).
template <int P> static int run_penalty(gf_handle* h, int flags) {
    hipStream_t st = h->stream;
    const HostModel& H = h->H;
    const int pen = (H.npts > 0 && (flags & (GF_ASM_R | GF_ASM_K | GF_ASM_DRDCP))) ? 1 : 0;
    if (pen) {
        const int mode = !(flags & (GF_ASM_K | GF_ASM_DRDCP)) ? 1 : (!(flags & GF_ASM_DRDCP) ? 2 : (!(flags & GF_ASM_K) ? 3 : 0));
        bool done = false;
        if constexpr (P <= 3) {
            if (h->pen16) { hipLaunchKernelGGL(pen_point16_kernel<P>, dim3((unsigned)((H.npts + 3) / 4)), dim3(64), 0, st, h->M, h->Q, h->d_pbuf, mode); done = true; }
        }
        if (!done) hipLaunchKernelGGL(pen_point_kernel<P>, dim3((unsigned)((H.npts + 63) / 64)), dim3(64), 0, st, h->M, h->Q, h->d_pbuf, mode);
    }
    if (pen) {
        const dim3 grid((unsigned)(((h->Q.nrow_groups + 7) / 8) * 8)), blk64(64);       // multiple of 8: XCD-contiguous group ranges
        if constexpr (P <= 4) {
            if (h->pen16 && h->Q.slots) {
                const size_t lds1 = (size_t)h->pen_maxdeg * 9 * sizeof(double);          // accumulators of one matrix per row
#define GF_PEN16(WC, WK) hipLaunchKernelGGL((pen_row16_kernel<P, WC, WK>), grid, blk64, ((WC) && (WK)) ? 2 * lds1 : lds1, st, h->M, h->Q, flags, h->d_pbuf, h->d_R, h->d_val[0], h->d_val[1], h->d_val[2], h->d_val[3])
                if (!(flags & GF_ASM_DRDCP)) GF_PEN16(false, true);
                else if (!(flags & GF_ASM_K)) GF_PEN16(true, false);
                else GF_PEN16(true, true);
#undef GF_PEN16
                return pen;
            }
        }
        const int sl = (h->pen_maxdeg + 63) / 64;         // neighbour slots per lane, register resident
#define GF_PEN_LAUNCH(SL, WC, WK) hipLaunchKernelGGL((pen_owner_kernel<P, SL, WC, WK>), grid, blk64, 0, st, h->M, h->Q, flags, h->pen_maxdeg, h->d_pbuf, h->d_R, \
                                                     h->d_val[0], h->d_val[1], h->d_val[2], h->d_val[3])
#define GF_PEN_SLOTS(WC, WK) do { if (sl <= 2) GF_PEN_LAUNCH(2, WC, WK); else if (sl == 3) GF_PEN_LAUNCH(3, WC, WK); else GF_PEN_LAUNCH(5, WC, WK); } while (0)
        if (!(flags & GF_ASM_DRDCP)) GF_PEN_SLOTS(false, true);                 // Newton pass
        else if (!(flags & GF_ASM_K)) GF_PEN_SLOTS(true, false);                // linearize right after a Newton solve (K is current)
        else GF_PEN_SLOTS(true, true);
#undef GF_PEN_SLOTS
#undef GF_PEN_LAUNCH
    }
    return pen;
}

// follower pressure and dead edge tractions: added to the rows the gather has written (and to the gathered residual)
template <int P> static void run_extra_loads(gf_handle* h, int flags) {
    const int n = (int)h->H.load_cps.size();
    if (n == 0 || !(flags & (GF_ASM_R | GF_ASM_K | GF_ASM_DRDCP))) return;
    hipLaunchKernelGGL(kl_extra_loads_kernel<P>, dim3((unsigned)n), dim3(64), 0, h->stream, h->M, h->d_load_cps, n, flags, h->d_R, h->d_val[0], h->d_val[1], h->d_val[2], h->d_val[3]);
}

static void finish_residual(gf_handle* h) {
    const HostModel& H = h->H;
    const long long npl = (long long)H.pl_dof.size();
    if (npl > 0) hipLaunchKernelGGL(residual_finish_kernel, dim3((unsigned)((npl + 255) / 256)), dim3(256), 0, h->stream, (long long)H.ndof, h->M.zero, npl, h->d_pl_dof, h->d_pl_val, h->d_R);
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((H.ndof + 255) / 256)), dim3(256), 0, h->stream, (long long)H.ndof, h->M.zero, h->d_R);
}

// Row-record path (gf_element_rec.hpp): penalty rows first (written), one launch of the walking kernel over all work items, then the
// record gather adds the shell part per control point.
template <int P> static void run_assemble_rec(gf_handle* h, int flags) {
    constexpr int PW = P == 2 ? 2 : 3;
    const HostModel& H = h->H;
    const bool mats = (flags & ~GF_ASM_R) != 0;
    const int pen = run_penalty<P>(h, flags);
    const RecOut O{h->d_rec, h->d_blk, H.rec_rows};
    const int slot = h->ev_n % 64, n = (int)H.rec_items.size();
    HIPCHK(hipEventRecord(h->ev0[slot], h->stream));
    constexpr int ALL_BITS = GF_ASM_R | GF_ASM_K | GF_ASM_DRDCP | GF_ASM_DRDH;
    // p = 3: the row-side sum-factorised walk (gf_gauss_loop.hpp: SfLane), one launch per kind of patch (polynomial: one product per derivative, rational: nine
    // (m, k2) pairs); p = 2 and GF_SUMFACT = 0: the 16 x 16 x 4 products of rounds 2 - 4, one launch over all items.  (The two launches one behind the other:
    // side by side on two streams -- one pool of work items, one partly filled last round -- measured 13.47 against 13.21 ms at C4, same box.)
    auto walk = [&](auto sf_, const int* order, int cnt) {
        constexpr int SF = decltype(sf_)::value;
        if (cnt <= 0) return;
        if ((flags & ALL_BITS) == ALL_BITS) hipLaunchKernelGGL((kl_element_rec_kernel<PW, true, true, SF>), dim3((unsigned)cnt), dim3(64), 0, h->stream, h->M, h->d_rec_items, order, flags, O);
        else if (flags & GF_ASM_DRDCP) hipLaunchKernelGGL((kl_element_rec_kernel<PW, true, false, SF>), dim3((unsigned)cnt), dim3(64), 0, h->stream, h->M, h->d_rec_items, order, flags, O);
        else hipLaunchKernelGGL((kl_element_rec_kernel<PW, false, false, SF>), dim3((unsigned)cnt), dim3(64), 0, h->stream, h->M, h->d_rec_items, order, flags, O);
    };
    if constexpr (PW == 3 && GF_SUMFACT_BUILD != 0) {
        if (h->sumfact == 0) walk(std::integral_constant<int, 0>{}, nullptr, n);
        else {
            walk(std::integral_constant<int, 1>{}, h->d_rec_order, H.rec_npoly);
            if (h->sumfact == 2) walk(std::integral_constant<int, 2>{}, h->d_rec_order + H.rec_npoly, n - H.rec_npoly);
            else walk(std::integral_constant<int, 0>{}, h->d_rec_order + H.rec_npoly, n - H.rec_npoly);
        }
    } else walk(std::integral_constant<int, 0>{}, nullptr, n);
    HIPCHK(hipEventRecord(h->ev1[slot], h->stream));
    h->ev_n++;
    const Chunk& c = h->chunks[0];
    const long long ne = c.e1 - c.e0, na = c.a1 - c.a0;
    auto gather = [&](const int* list, long long cnt, long long a0, long long a1, int pen_add) {
        if (cnt <= 0) return;
        const dim3 grid((unsigned)(((cnt + 7) / 8) * 8));
        if (flags & GF_ASM_DRDCP) hipLaunchKernelGGL((kl_gather_rec_kernel<PW, true>), grid, dim3(64), 0, h->stream, h->M, a0, a1, list, flags, h->d_rec, H.rec_rows, h->d_rec_cp,
                                                     h->d_val[0], h->d_val[1], h->d_val[2], h->d_val[3], h->d_val[4], pen_add);
        else hipLaunchKernelGGL((kl_gather_rec_kernel<PW, false>), grid, dim3(64), 0, h->stream, h->M, a0, a1, list, flags, h->d_rec, H.rec_rows, h->d_rec_cp,
                                h->d_val[0], h->d_val[1], h->d_val[2], h->d_val[3], h->d_val[4], pen_add);
    };
    if (mats) gather(nullptr, na, c.a0, c.a1, pen);
    if (flags & GF_ASM_R) hipLaunchKernelGGL(kl_rgather_kernel<P>, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, h->stream, h->M, c.a0, c.a1, c.e0, ne, h->d_blk, h->d_R, pen, 3 * (P + 1) * (P + 1), 0);
    run_extra_loads<P>(h, flags);
    if (flags & GF_ASM_R) finish_residual(h);
    HIPCHK(hipGetLastError());
}

// p = 4 row-record path (gf_element_rec4.hpp): penalty rows first (written), then per chunk of patches the walks -- PASS 0 (R + K), PASS 1, 2 (dR/dCP +
// dR/dh per b tile) -- over the chunk's work items and the record gather of the chunk's control points.
static void run_assemble_rec4(gf_handle* h, int flags) {
    const HostModel& H = h->H;
    const bool mats = (flags & ~GF_ASM_R) != 0, full = (flags & (GF_ASM_DRDCP | GF_ASM_DRDH)) != 0;
    const int pen = run_penalty<4>(h, flags);
    for (const Chunk& c : h->rchunks) {
        const long long i0 = H.rec_patch[c.p0].item_off, i1 = c.p1 < H.n_owned ? H.rec_patch[c.p1].item_off : (long long)H.rec_items.size();
        const unsigned n = (unsigned)(i1 - i0);
        const Rec4Out O{h->d_rec, h->d_blk, H.rec_rows, c.e0};
        const WalkItem* items = h->d_rec_items + i0;
        const int slot = h->ev_n % 64;
        HIPCHK(hipEventRecord(h->ev0[slot], h->stream));
        if (flags & (GF_ASM_R | GF_ASM_K)) {
            if (full) hipLaunchKernelGGL((kl_element_rec4_kernel<0, 21>), dim3(n), dim3(64), 0, h->stream, h->M, items, flags, O);
            else hipLaunchKernelGGL((kl_element_rec4_kernel<0, 9>), dim3(n), dim3(64), 0, h->stream, h->M, items, flags, O);
        }
        if (full) {
            hipLaunchKernelGGL((kl_element_rec4_kernel<1, 21>), dim3(n), dim3(64), 0, h->stream, h->M, items, flags, O);
            hipLaunchKernelGGL((kl_element_rec4_kernel<2, 21>), dim3(n), dim3(64), 0, h->stream, h->M, items, flags, O);
        }
        HIPCHK(hipEventRecord(h->ev1[slot], h->stream));
        h->ev_n++;
        const long long ne = c.e1 - c.e0, na = c.a1 - c.a0;
        if (mats && na > 0) {
            const dim3 grid((unsigned)(((na + 7) / 8) * 8));
            const long long row_base = i0 * H.rec_rows;
            if (full) hipLaunchKernelGGL((kl_gather_rec4_kernel<21>), grid, dim3(64), 0, h->stream, h->M, c.a0, c.a1, flags, h->d_rec, row_base, h->d_rec_cp4,
                                         h->d_val[0], h->d_val[1], h->d_val[2], h->d_val[3], h->d_val[4], pen);
            else hipLaunchKernelGGL((kl_gather_rec4_kernel<9>), grid, dim3(64), 0, h->stream, h->M, c.a0, c.a1, flags, h->d_rec, row_base, h->d_rec_cp4,
                                    h->d_val[0], h->d_val[1], h->d_val[2], h->d_val[3], h->d_val[4], pen);
        }
        if (flags & GF_ASM_R) hipLaunchKernelGGL(kl_rgather_kernel<4>, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, h->stream, h->M, c.a0, c.a1, c.e0, ne, h->d_blk, h->d_R, pen, 75, 0);
    }
    run_extra_loads<4>(h, flags);
    if (flags & GF_ASM_R) finish_residual(h);
    HIPCHK(hipGetLastError());
}

template <int P> static void run_assemble(gf_handle* h, int flags) {
    using Cfg = ElemCfg<P>;
    const HostModel& H = h->H;
    // Penalty rows first: pen_owner_kernel WRITES the rows of the interface control points, the gather adds the shell part to
    // them (no read-modify-write pass over those rows afterwards).  Running the penalty kernels on a second stream was measured
    // and dropped: next to the element kernel they cost it LDS occupancy (17.7 -> 22.8 ms), next to the gather both slow down
    // by what the overlap saves (profiles/r01_v8_*).
    if ((P == 2 || P == 3) && h->rec) { run_assemble_rec<P>(h, flags); return; }
    if (P == 4 && h->rec4 && !(h->hybrid4 && (flags & (GF_ASM_DRDCP | GF_ASM_DRDH)))) { run_assemble_rec4(h, flags); return; }
    const int pen = run_penalty<P>(h, flags);
    for (size_t ci = 0; ci < h->chunks.size(); ++ci) {
        const Chunk& c = h->chunks[ci];
        const long long ne = c.e1 - c.e0, na = c.a1 - c.a0;
        double* const blk = h->d_blk;
        hipStream_t gs = h->stream;
        const int slot = h->ev_n % 64;
        HIPCHK(hipEventRecord(h->ev0[slot], h->stream));
        if ((P == 3 || P == 2) && h->mfma) {
            if (flags & GF_ASM_DRDCP) hipLaunchKernelGGL((kl_element_mfma_kernel<(P == 2 ? 2 : 3), true>), dim3((unsigned)ne), dim3(64), 0, h->stream, h->M, (int)c.e0, flags, blk);
            else hipLaunchKernelGGL((kl_element_mfma_kernel<(P == 2 ? 2 : 3), false>), dim3((unsigned)ne), dim3(64), 0, h->stream, h->M, (int)c.e0, flags, blk);
        }
        else if (P == 4 && h->mfma) {
            if (flags & GF_ASM_DRDCP) hipLaunchKernelGGL(kl_element_mfma4_kernel<true>, dim3((unsigned)ne), dim3(64), 0, h->stream, h->M, (int)c.e0, flags, blk);
            else hipLaunchKernelGGL(kl_element_mfma4_kernel<false>, dim3((unsigned)ne), dim3(64), 0, h->stream, h->M, (int)c.e0, flags, blk);
        }
        else hipLaunchKernelGGL(kl_element_kernel<P>, dim3((unsigned)ne), dim3(Cfg::NT), 0, h->stream, h->M, (int)c.e0, flags, blk);
        HIPCHK(hipEventRecord(h->ev1[slot], h->stream));
        h->ev_n++;
        if (flags == GF_ASM_R)
            hipLaunchKernelGGL(kl_rgather_kernel<P>, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, gs, h->M, c.a0, c.a1, c.e0, ne, blk, h->d_R, pen, Cfg::BLK, Cfg::OFF_R);
        else if (!(flags & GF_ASM_DRDCP) && h->gather1)      // one wave per control point; without dR/dCP (Newton pass) the leaner instance
            hipLaunchKernelGGL((kl_gather1_kernel<P, false>), dim3((unsigned)na), dim3(64), 0, gs, h->M, c.a0, c.e0, ne, flags, blk,
                               h->d_val[0], h->d_val[1], h->d_val[2], h->d_val[3], h->d_val[4], h->d_R, pen);
        else if (h->gather1)
            hipLaunchKernelGGL((kl_gather1_kernel<P, true>), dim3((unsigned)na), dim3(64), 0, gs, h->M, c.a0, c.e0, ne, flags, blk,
                               h->d_val[0], h->d_val[1], h->d_val[2], h->d_val[3], h->d_val[4], h->d_R, pen);
        else
        hipLaunchKernelGGL(kl_gather_kernel<P>, dim3((unsigned)na), dim3(256), 0, gs, h->M, c.a0, c.e0, ne, flags, blk,
                           h->d_val[0], h->d_val[1], h->d_val[2], h->d_val[3], h->d_val[4], h->d_R, pen);
    }
    run_extra_loads<P>(h, flags);
    if (flags & GF_ASM_R) finish_residual(h);
    HIPCHK(hipGetLastError());
}

template <int P> static void run_functionals(gf_handle* h, int apply_bcs) {
    const size_t stride = (size_t)FunCfg<P>::STRIDE;
    for (const Chunk& c : h->chunks) {
        const long long ne = c.e1 - c.e0, na = c.a1 - c.a0, nthr = std::max(ne, na);
        constexpr int NE = 64 / ((P + 1) * (P + 1));                     // elements per wave
        hipLaunchKernelGGL((kl_pointfun_kernel<P, 0>), dim3((unsigned)((ne + NE - 1) / NE)), dim3(64), 0, h->stream, h->M, (int)c.e0, (int)ne, StressCfg{}, h->d_blk, stride);
        hipLaunchKernelGGL(kl_fgather_kernel<P>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, h->stream, h->M, c.a0, c.a1, c.e0, ne, apply_bcs,
                           h->d_blk, stride, h->d_fun, h->d_x, h->d_y);
    }
    const HostModel& H = h->H;
    if (H.npts > 0) {
        bool done = false;
        if constexpr (P <= 3) { if (h->pen16) { hipLaunchKernelGGL(pen_point16_kernel<P>, dim3((unsigned)((H.npts + 3) / 4)), dim3(64), 0, h->stream, h->M, h->Q, h->d_pbuf, 1); done = true; } }
        if (!done) hipLaunchKernelGGL(pen_point_kernel<P>, dim3((unsigned)((H.npts + 63) / 64)), dim3(64), 0, h->stream, h->M, h->Q, h->d_pbuf, 1);
        hipLaunchKernelGGL(pen_energy_kernel, dim3((unsigned)((H.npts + 255) / 256)), dim3(256), 0, h->stream, (long long)H.npts, h->d_pbuf, h->d_pen_en);
    }
    HIPCHK(hipGetLastError());
}

template <int P> static void run_compliance(gf_handle* h, int apply_bcs) {
    const size_t stride = (size_t)FunCfg<P>::STRIDE;
    // forces were staged in d_y; the per-element partials of kl_fgather_kernel go to d_x (C_e) and d_fun + 0 (unused V_e slot reuses d_pen_en-safe d_x tail)
    double* d_forces = h->d_y;
    for (const Chunk& c : h->chunks) {
        const long long ne = c.e1 - c.e0, na = c.a1 - c.a0, nthr = std::max(ne, na);
        hipLaunchKernelGGL(kl_compliance_kernel<P>, dim3((unsigned)ne), dim3(64), 0, h->stream, h->M, (int)c.e0, d_forces, h->d_blk, stride);
        hipLaunchKernelGGL(kl_fgather_kernel<P>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, h->stream, h->M, c.a0, c.a1, c.e0, ne, apply_bcs,
                           h->d_blk, stride, h->d_fun, h->d_x, h->d_ve);
    }
    HIPCHK(hipGetLastError());
}

template <int P> static void run_regu(gf_handle* h, const StressCfg& S) {
    const size_t stride = (size_t)FunCfg<P>::STRIDE;
    constexpr int NE = 64 / ((P + 1) * (P + 1));
    for (const Chunk& c : h->chunks) {
        const long long ne = c.e1 - c.e0, na = c.a1 - c.a0, nthr = std::max(ne, na);
        hipLaunchKernelGGL((kl_pointfun_kernel<P, 2>), dim3((unsigned)((ne + NE - 1) / NE)), dim3(64), 0, h->stream, h->M, (int)c.e0, (int)ne, S, h->d_blk, stride);
        hipLaunchKernelGGL(kl_fgather_kernel<P>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, h->stream, h->M, c.a0, c.a1, c.e0, ne, 0,
                           h->d_blk, stride, h->d_fun, h->d_x, h->d_ve);
    }
    HIPCHK(hipGetLastError());
}

template <int P> static void run_stress(gf_handle* h, const StressCfg& S, int apply_bcs) {
    const size_t stride = (size_t)FunCfg<P>::STRIDE;
    for (const Chunk& c : h->chunks) {
        const long long ne = c.e1 - c.e0, na = c.a1 - c.a0, nthr = std::max(ne, na);
        constexpr int NE = 64 / ((P + 1) * (P + 1));
        hipLaunchKernelGGL((kl_pointfun_kernel<P, 1>), dim3((unsigned)((ne + NE - 1) / NE)), dim3(64), 0, h->stream, h->M, (int)c.e0, (int)ne, S, h->d_blk, stride);
        hipLaunchKernelGGL(kl_fgather_kernel<P>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, h->stream, h->M, c.a0, c.a1, c.e0, ne, apply_bcs,
                           h->d_blk, stride, h->d_fun, h->d_x, h->d_ve);
    }
    HIPCHK(hipGetLastError());
}

// per-patch sums of two per-element arrays (and the per-patch maximum of the second): a few numbers to the host instead of nelem-long arrays
static void patch_sums(gf_handle* h, const double* a, const double* b, std::vector<double>& sa, std::vector<double>& sb, std::vector<double>* mb) {
    const int np = h->H.np;
    hipLaunchKernelGGL(seg_reduce_kernel, dim3(np), dim3(256), 0, h->stream, h->d_elem_off, a, b, h->d_red, b ? h->d_red + h->nred : nullptr, (b && mb) ? h->d_red + 2 * h->nred : nullptr);
    sa.assign(np, 0.0); sb.assign(np, 0.0);
    HIPCHK(hipMemcpyAsync(sa.data(), h->d_red, np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (b) HIPCHK(hipMemcpyAsync(sb.data(), h->d_red + h->nred, np * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (b && mb) { mb->assign(np, 0.0); HIPCHK(hipMemcpyAsync(mb->data(), h->d_red + 2 * h->nred, np * sizeof(double), hipMemcpyDeviceToHost, h->stream)); }
}

extern "C" {

int gf_assemble(gf_handle* h, int flags) {
    if (!h) return fail("gf_assemble: null handle");
    if (flags == 0 || (flags & ~GF_ASM_ALL)) return fail("gf_assemble: bad flags");
    try {
        HIPCHK(hipSetDevice(h->device));
        switch (h->H.degree) {
            case 2: run_assemble<2>(h, flags); break;
            case 3: run_assemble<3>(h, flags); break;
            case 4: run_assemble<4>(h, flags); break;
            default: throw std::runtime_error("gf_assemble: unsupported degree");
        }
        if (flags & GF_ASM_K) h->assembled[GF_MAT_K] = true;
        if (flags & GF_ASM_DRDCP) h->assembled[1] = h->assembled[2] = h->assembled[3] = true;
        if (flags & GF_ASM_DRDH) h->assembled[GF_MAT_DRDH] = true;
    } catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}

int gf_sync(gf_handle* h) {
    if (!h) return fail("gf_sync: null handle");
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(std::string("gf_sync: ") + hipGetErrorString(e));
    return 0;
}

int gf_get_residual(gf_handle* h, double* R, int64_t n) {
    if (!h || !R) return fail("gf_get_residual: null argument");
    if (n != h->H.ndof) return fail("gf_get_residual: wrong length");
    try { HIPCHK(hipMemcpyAsync(R, h->d_R, n * sizeof(double), hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream)); }
    catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}
int gf_get_values(gf_handle* h, int which, double* vals, int64_t n) {
    if (!h || !vals) return fail("gf_get_values: null argument");
    if (which < 0 || which > 4) return fail("gf_get_values: unknown matrix id");
    if (n != gf_nnz(h, which)) return fail("gf_get_values: wrong length");
    if (!h->assembled[which]) return fail("gf_get_values: matrix has not been assembled");
    try { HIPCHK(hipMemcpyAsync(vals, h->d_val[which], n * sizeof(double), hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream)); }
    catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}

int gf_apply_dev(gf_handle* h, int which, int transpose, const double* x, double* y) {
    if (!h || !x || !y) return fail("gf_apply: null argument");
    if (which < 0 || which > 4) return fail("gf_apply: unknown matrix id");
    if (!h->assembled[which]) return fail("gf_apply: matrix has not been assembled");
    const long long* ptr = which == GF_MAT_DRDH ? h->M.nb_ptr_s : h->M.nb_ptr_c; const int* nb = which == GF_MAT_DRDH ? h->M.nb_s : h->M.nb_c;
    const int bw = which == GF_MAT_K ? 3 : 1; const long long ncp = h->H.total_cp;
    const unsigned grid_cp = (unsigned)((ncp * 64 + 255) / 256);
    // K is symmetric including its Dirichlet treatment (rows+cols zeroed, unit diagonal): K^T x = K x, so the
    // transposed product uses the atomic-free row kernel as well (bitwise reproducible adjoint products with K)
    // (not on a shard: ghost rows are not assembled there, so the local K is not symmetric)
    if (which == GF_MAT_K && (!transpose || (h->H.n_owned == h->H.np && h->H.symmetric_K))) hipLaunchKernelGGL(csr_apply_kernel<3>, dim3(grid_cp), dim3(256), 0, h->stream, ncp, ptr, nb, h->d_val[which], x, y);
    else if (!transpose) hipLaunchKernelGGL(csr_apply_kernel<1>, dim3(grid_cp), dim3(256), 0, h->stream, ncp, ptr, nb, h->d_val[which], x, y);
    else if (bw == 1) hipLaunchKernelGGL(csr_apply_tdet_kernel<1>, dim3(grid_cp), dim3(256), 0, h->stream, ncp, ptr, nb, which == GF_MAT_DRDH ? h->d_rev_s : h->d_rev_c, h->d_val[which], x, y);
    else hipLaunchKernelGGL(csr_apply_tdet_kernel<3>, dim3(grid_cp), dim3(256), 0, h->stream, ncp, ptr, nb, h->d_rev_c, h->d_val[which], x, y);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("gf_apply: ") + hipGetErrorString(e));
    return 0;
}
int gf_apply(gf_handle* h, int which, int transpose, const double* x, int64_t nx, double* y, int64_t ny) {
    if (!h || !x || !y) return fail("gf_apply: null argument");
    if (which < 0 || which > 4) return fail("gf_apply: unknown matrix id");
    const int64_t nrow = h->H.ndof, ncol = which == GF_MAT_K ? h->H.ndof : h->H.total_cp;
    const int64_t ex = transpose ? nrow : ncol, ey = transpose ? ncol : nrow;
    if (nx != ex || ny != ey) return fail("gf_apply: vector lengths do not match the matrix shape");
    try {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipMemcpyAsync(h->d_x, x, nx * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_y, y, ny * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (gf_apply_dev(h, which, transpose, h->d_x, h->d_y)) return 1;
        HIPCHK(hipMemcpyAsync(y, h->d_y, ny * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    } catch (const std::exception& ex2) { return fail(ex2.what()); }
    return 0;
}

int gf_apply_many(gf_handle* h, int transpose, int nmat, const int* which, const double* const* xs, double* const* ys) {
    if (!h || !which || !xs || !ys) return fail("gf_apply_many: null argument");
    if (nmat < 1 || nmat > 5) return fail("gf_apply_many: 1 to 5 matrices");
    for (int m = 0; m < nmat; ++m) {
        if (which[m] < 0 || which[m] > 4) return fail("gf_apply_many: unknown matrix id");
        if (!h->assembled[which[m]]) return fail("gf_apply_many: matrix has not been assembled");
        if (!(transpose ? ys[m] : xs[m])) return fail("gf_apply_many: null vector");
    }
    if (!(transpose ? xs[0] : ys[0])) return fail("gf_apply_many: null vector");
    try {
        HIPCHK(hipSetDevice(h->device));
        const long long nd = h->H.ndof, ncp = h->H.total_cp;
        if (!h->d_many) h->d_many = h->dalloc<double>(5 * (size_t)nd);
        auto len = [&](int w) { return w == GF_MAT_K ? nd : ncp; };
        if (!transpose) {
            HIPCHK(hipMemcpyAsync(h->d_y, ys[0], nd * sizeof(double), hipMemcpyHostToDevice, h->stream));
            for (int m = 0; m < nmat; ++m) HIPCHK(hipMemcpyAsync(h->d_many + (size_t)m * nd, xs[m], len(which[m]) * sizeof(double), hipMemcpyHostToDevice, h->stream));
            for (int m = 0; m < nmat; ++m) if (gf_apply_dev(h, which[m], 0, h->d_many + (size_t)m * nd, h->d_y)) return 1;
            HIPCHK(hipMemcpyAsync(ys[0], h->d_y, nd * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        } else {
            HIPCHK(hipMemcpyAsync(h->d_x, xs[0], nd * sizeof(double), hipMemcpyHostToDevice, h->stream));
            for (int m = 0; m < nmat; ++m) HIPCHK(hipMemcpyAsync(h->d_many + (size_t)m * nd, ys[m], len(which[m]) * sizeof(double), hipMemcpyHostToDevice, h->stream));
            for (int m = 0; m < nmat; ++m) if (gf_apply_dev(h, which[m], 1, h->d_x, h->d_many + (size_t)m * nd)) return 1;
            for (int m = 0; m < nmat; ++m) HIPCHK(hipMemcpyAsync(ys[m], h->d_many + (size_t)m * nd, len(which[m]) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(hipStreamSynchronize(h->stream));
    } catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}

void* gf_device_ptr(gf_handle* h, int which) {
    if (!h) return nullptr;
    switch (which) {
        case GF_BUF_CP: return h->d_cp4; case GF_BUF_U: return h->d_u; case GF_BUF_H: return h->d_h; case GF_BUF_R: return h->d_R;
        case GF_BUF_VAL_K: return h->d_val[0]; case GF_BUF_VAL_C0: return h->d_val[1]; case GF_BUF_VAL_C1: return h->d_val[2];
        case GF_BUF_VAL_C2: return h->d_val[3]; case GF_BUF_VAL_H: return h->d_val[4];
    }
    return nullptr;
}

double gf_kernel_ms(gf_handle* h, int* n_launches) {
    if (!h) return 0.0;
    (void)hipStreamSynchronize(h->stream);
    const int n = h->ev_n < 64 ? h->ev_n : 64; double tot = 0.0;
    for (int k = 0; k < n; ++k) { float ms = 0.f; if (hipEventElapsedTime(&ms, h->ev0[k], h->ev1[k]) == hipSuccess) tot += ms; }
    if (n_launches) *n_launches = n;
    h->ev_n = 0;
    return n > 0 ? tot / n : 0.0;
}

void* gf_stream(gf_handle* h) { return h ? (void*)h->stream : nullptr; }

int gf_assembly_path(const gf_handle* h) { return !h ? -1 : (h->rec4 ? (h->hybrid4 ? 6 : 5) : (h->rec ? 4 : (h->mfma ? 0 : 3))); }

int gf_get_functional_gradient(gf_handle* h, int field, double* out, int64_t n) {
    if (!h || !out) return fail("gf_get_functional_gradient: null argument");
    if (field < 0 || field > 4) return fail("gf_get_functional_gradient: field must be 0..4");
    if (h->fun_owner != 0) return fail("gf_get_functional_gradient: the gradient buffer does not hold the fields of a gf_functionals call");
    const long long T = h->H.total_cp;
    const long long off[5] = {0, 3 * T, 6 * T, 7 * T, 10 * T}, len[5] = {3 * T, 3 * T, T, 3 * T, T};
    if (n != len[field]) return fail("gf_get_functional_gradient: wrong length");
    try { HIPCHK(hipMemcpyAsync(out, h->d_fun + off[field], n * sizeof(double), hipMemcpyDeviceToHost, h->stream)); HIPCHK(hipStreamSynchronize(h->stream)); }
    catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}

int gf_compliance(gf_handle* h, const double* forces, int64_t nf, double* C, double* dCdu, double* dCdcp, int apply_bcs) {
    if (!h || !forces || !C) return fail("gf_compliance: null argument");
    if (nf != 3 * (int64_t)h->H.np) return fail("gf_compliance: forces must hold 3 values per patch");
    try {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipMemcpyAsync(h->d_y, forces, nf * sizeof(double), hipMemcpyHostToDevice, h->stream));   // d_y: >= 3*n_patches doubles
        h->fun_owner = 1;
        switch (h->H.degree) {
            case 2: run_compliance<2>(h, apply_bcs); break;
            case 3: run_compliance<3>(h, apply_bcs); break;
            case 4: run_compliance<4>(h, apply_bcs); break;
            default: throw std::runtime_error("gf_compliance: unsupported degree");
        }
        const HostModel& H = h->H; const long long T = H.total_cp;
        std::vector<double> ce, unused;
        patch_sums(h, h->d_x, nullptr, ce, unused, nullptr);
        if (dCdu) HIPCHK(hipMemcpyAsync(dCdu, h->d_fun, 3 * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (dCdcp) HIPCHK(hipMemcpyAsync(dCdcp, h->d_fun + 7 * T, 3 * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        long double acc = 0; for (int s = 0; s < H.n_owned; ++s) acc += ce[s];      // owned patches only (fixed order)
        *C = (double)acc;
    } catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}

int gf_shape_regu(gf_handle* h, int field, const double* cp0, int64_t ncp, const double* coef, int64_t nc, double* value, double* dcp) {
    if (!h || !cp0 || !coef || !value) return fail("gf_shape_regu: null argument");
    if (field < 0 || field > 2) return fail("gf_shape_regu: field must be 0, 1 or 2");
    if (ncp != (int64_t)h->H.total_cp || nc != (int64_t)h->H.np) return fail("gf_shape_regu: cp0 must hold total_cp values and coef one value per patch");
    try {
        HIPCHK(hipSetDevice(h->device));
        const HostModel& H = h->H; const long long T = H.total_cp;
        HIPCHK(hipMemcpyAsync(h->d_y, coef, nc * sizeof(double), hipMemcpyHostToDevice, h->stream));          // d_y, d_x: >= ndof doubles each
        HIPCHK(hipMemcpyAsync(h->d_y + H.np, cp0, ncp * sizeof(double), hipMemcpyHostToDevice, h->stream));
        StressCfg S{}; S.m_list = h->d_y; S.cp0 = h->d_y + H.np; S.field = field;
        h->fun_owner = 3;
        switch (H.degree) {
            case 2: run_regu<2>(h, S); break;
            case 3: run_regu<3>(h, S); break;
            case 4: run_regu<4>(h, S); break;
            default: throw std::runtime_error("gf_shape_regu: unsupported degree");
        }
        std::vector<double> ve, unused;
        patch_sums(h, h->d_x, nullptr, ve, unused, nullptr);
        if (dcp) HIPCHK(hipMemcpyAsync(dcp, h->d_fun + 3 * T, 3 * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        long double acc = 0; for (int s = 0; s < H.n_owned; ++s) acc += ve[s];
        *value = (double)acc;
    } catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}

int gf_stress_forms(gf_handle* h, int mode, double rho, const double* m_list, int64_t nm, int surf, int measure,
                    double* forms, double* vmax, double* dIdu, double* dIdcp, double* dIdh, int apply_bcs) {
    if (!h || !m_list) return fail("gf_stress_forms: null argument");
    if (nm != (int64_t)h->H.np) return fail("gf_stress_forms: m_list must hold one value per patch");
    if ((mode != 0 && mode != 1) || (measure != 0 && measure != 1) || surf < -1 || surf > 1) return fail("gf_stress_forms: bad mode / measure / surf");
    for (int64_t s = 0; s < nm; ++s) if (!(m_list[s] > 0.0)) return fail("gf_stress_forms: m_list entries must be positive");
    try {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipMemcpyAsync(h->d_y, m_list, nm * sizeof(double), hipMemcpyHostToDevice, h->stream));   // d_y: >= 3*n_patches doubles
        StressCfg S; S.mode = mode; S.measure = measure; S.rho = rho; S.sgn = (double)surf; S.m_list = h->d_y;
        h->fun_owner = 2;
        switch (h->H.degree) {
            case 2: run_stress<2>(h, S, apply_bcs); break;
            case 3: run_stress<3>(h, S, apply_bcs); break;
            case 4: run_stress<4>(h, S, apply_bcs); break;
            default: throw std::runtime_error("gf_stress_forms: unsupported degree");
        }
        const HostModel& H = h->H; const long long T = H.total_cp;
        std::vector<double> ip, sp_, mp;
        patch_sums(h, h->d_x, h->d_ve, ip, sp_, &mp);
        if (dIdu) HIPCHK(hipMemcpyAsync(dIdu, h->d_fun, 3 * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (dIdcp) HIPCHK(hipMemcpyAsync(dIdcp, h->d_fun + 3 * T, 3 * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (dIdh) HIPCHK(hipMemcpyAsync(dIdh, h->d_fun + 6 * T, T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        // fixed-order host sums per patch; the ghost patches of a shard are not evaluated here (their owner reports them): 0
        for (int s = 0; s < H.np; ++s) {
            if (forms) forms[s] = s < H.n_owned ? ip[s] : 0.0;
            if (vmax) vmax[s] = s < H.n_owned ? mp[s] : 0.0;
        }
    } catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}

int gf_penalty_dxi_range(gf_handle* h, int64_t v_first, int64_t v_count, double* blocks, int64_t n, int32_t* windows, int64_t nw) {
    if (!h || !blocks) return fail("gf_penalty_dxi: null argument");
    const HostModel& H = h->H;
    const int NB = (H.degree + 1) * (H.degree + 1);
    if (v_first < 0 || v_count < 0 || v_first + v_count > (int64_t)H.npts) return fail("gf_penalty_dxi: vertex range outside the model's mortar vertices");
    const int64_t need = v_count * 6 * 2 * NB * 3;
    if (n != need) return fail("gf_penalty_dxi: blocks must hold (vertices) * 6 * 2 * (p+1)^2 * 3 doubles");
    if (windows && nw != 4 * v_count) return fail("gf_penalty_dxi: windows must hold 4 ints per mortar vertex");
    if (v_count == 0) return 0;
    try {
        HIPCHK(hipSetDevice(h->device));
        if (!h->d_pt_nu2) {                                  // uploaded on first use: only moving-intersection problems need it
            h->d_pt_nu2 = h->dalloc<double>(H.pt_nu2.size());
            HIPCHK(hipMemcpy(h->d_pt_nu2, H.pt_nu2.data(), H.pt_nu2.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        if (h->dxi_doubles < need) {                           // kept between calls (no hipMalloc / hipFree per call); a larger range replaces the buffer
            if (h->d_dxi) {
                HIPCHK(hipStreamSynchronize(h->stream));
                h->allocs.erase(std::remove(h->allocs.begin(), h->allocs.end(), (void*)h->d_dxi), h->allocs.end());
                (void)hipFree(h->d_dxi); h->bytes -= h->dxi_doubles * (long long)sizeof(double); h->d_dxi = nullptr; h->dxi_doubles = 0;
            }
            h->d_dxi = h->dalloc<double>((size_t)need); h->dxi_doubles = need;
        }
        double* d_out = h->d_dxi;
        const long long nt = (long long)v_count * 6;
        switch (H.degree) {
            case 2: hipLaunchKernelGGL(pen_dxi_kernel<2>, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, h->stream, h->M, h->Q, h->d_pt_nu2, d_out, (long long)v_first, (long long)v_count); break;
            case 3: hipLaunchKernelGGL(pen_dxi_kernel<3>, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, h->stream, h->M, h->Q, h->d_pt_nu2, d_out, (long long)v_first, (long long)v_count); break;
            case 4: hipLaunchKernelGGL(pen_dxi_kernel<4>, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, h->stream, h->M, h->Q, h->d_pt_nu2, d_out, (long long)v_first, (long long)v_count); break;
            default: throw std::runtime_error("gf_penalty_dxi: unsupported degree");
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(blocks, d_out, need * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) throw std::runtime_error(std::string("gf_penalty_dxi: ") + hipGetErrorString(e));
        if (windows) for (int64_t k = 0; k < 4 * v_count; ++k) windows[k] = H.pt_base[4 * v_first + k];
    } catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}
// Moving intersections: new parametric coordinates of the mortar vertices of ONE interface.  When every vertex stays in its knot spans (same support windows:
// the coupling pattern, the visit lists and every index table stay valid) only the vertex tables -- basis values / derivatives, curve tangents, quadrature
// weights -- of that interface are re-evaluated and uploaded; returns 2, changing nothing, when a vertex crossed a knot line (the caller re-creates the model).
int gf_update_interface(gf_handle* h, int iface, const double* xi, const double* tau, const double* wt, int64_t npts_if) {
    if (!h || !xi || !tau || !wt) return fail("gf_update_interface: null argument");
    HostModel& H = h->H;
    if (iface < 0 || iface >= H.ni) return fail("gf_update_interface: interface index out of range");
    const int64_t v0 = H.if_off[iface], n = H.if_off[iface + 1] - v0;
    if (npts_if != n) return fail("gf_update_interface: the interface has " + std::to_string(n) + " mortar vertices, got " + std::to_string(npts_if));
    try {
        HIPCHK(hipSetDevice(h->device));
        const int NB = (H.degree + 1) * (H.degree + 1);
        std::vector<double> nu((size_t)n * 2 * 3 * NB), nu2((size_t)n * 2 * 3 * NB);
        for (int64_t k = 0; k < n; ++k) for (int sd = 0; sd < 2; ++sd) {
            int win[2];
            H.eval_mortar_vertex(H.if_patch[2 * iface + sd], xi[4 * k + 2 * sd], xi[4 * k + 2 * sd + 1], win, &nu[((size_t)k * 2 + sd) * 3 * NB], &nu2[((size_t)k * 2 + sd) * 3 * NB]);
            if (win[0] != H.pt_base[4 * (v0 + k) + 2 * sd] || win[1] != H.pt_base[4 * (v0 + k) + 2 * sd + 1]) return 2;      // a vertex left its knot spans
        }
        std::copy(nu.begin(), nu.end(), H.pt_nu.begin() + (size_t)v0 * 2 * 3 * NB);
        std::copy(nu2.begin(), nu2.end(), H.pt_nu2.begin() + (size_t)v0 * 2 * 3 * NB);
        std::copy(tau, tau + 2 * n, H.pt_tau.begin() + 2 * v0);
        std::copy(wt, wt + n, H.pt_wt.begin() + v0);
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(const_cast<double*>(h->Q.pt_nu) + (size_t)v0 * 2 * 3 * NB, nu.data(), nu.size() * sizeof(double), hipMemcpyHostToDevice));
        if (h->d_pt_nu2) HIPCHK(hipMemcpy(h->d_pt_nu2 + (size_t)v0 * 2 * 3 * NB, nu2.data(), nu2.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(const_cast<double*>(h->Q.pt_tau) + 2 * v0, tau, 2 * n * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(const_cast<double*>(h->Q.pt_wt) + v0, wt, n * sizeof(double), hipMemcpyHostToDevice));
        for (int w = 0; w < 5; ++w) h->assembled[w] = false;
    } catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}

// reverse-mode product with dR/dxi on the device: out[v][dir] = sum over the owned, non-Dirichlet rows of block[v][dir] * lam (pen_dxi_kernel<P, true>)
int gf_penalty_dxi_rev(gf_handle* h, int64_t v_first, int64_t v_count, const double* lam, int64_t nlam, double* out, int64_t nout) {
    if (!h || !lam || !out) return fail("gf_penalty_dxi_rev: null argument");
    const HostModel& H = h->H;
    if (v_first < 0 || v_count < 0 || v_first + v_count > (int64_t)H.npts) return fail("gf_penalty_dxi_rev: vertex range outside the model's mortar vertices");
    if (nlam != (int64_t)H.ndof || nout != 6 * v_count) return fail("gf_penalty_dxi_rev: lam must hold ndof values, out 6 per mortar vertex");
    if (v_count == 0) return 0;
    try {
        HIPCHK(hipSetDevice(h->device));
        if (!h->d_pt_nu2) {
            h->d_pt_nu2 = h->dalloc<double>(H.pt_nu2.size());
            HIPCHK(hipMemcpy(h->d_pt_nu2, H.pt_nu2.data(), H.pt_nu2.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        if (!h->d_many) h->d_many = h->dalloc<double>(5 * (size_t)H.ndof);      // d_many: >= 6 * npts doubles?  (npts * 6 <= 5 ndof is checked below)
        if (6 * v_count > 5 * (int64_t)H.ndof) return fail("gf_penalty_dxi_rev: more mortar vertices than the staging buffer holds");
        HIPCHK(hipMemcpyAsync(h->d_x, lam, nlam * sizeof(double), hipMemcpyHostToDevice, h->stream));
        const long long nt = (long long)v_count * 6;
        const dim3 grid((unsigned)((nt + 63) / 64));
        switch (H.degree) {
            case 2: hipLaunchKernelGGL((pen_dxi_kernel<2, true>), grid, dim3(64), 0, h->stream, h->M, h->Q, h->d_pt_nu2, h->d_many, (long long)v_first, (long long)v_count, h->d_x, (long long)H.owned_cp); break;
            case 3: hipLaunchKernelGGL((pen_dxi_kernel<3, true>), grid, dim3(64), 0, h->stream, h->M, h->Q, h->d_pt_nu2, h->d_many, (long long)v_first, (long long)v_count, h->d_x, (long long)H.owned_cp); break;
            case 4: hipLaunchKernelGGL((pen_dxi_kernel<4, true>), grid, dim3(64), 0, h->stream, h->M, h->Q, h->d_pt_nu2, h->d_many, (long long)v_first, (long long)v_count, h->d_x, (long long)H.owned_cp); break;
            default: throw std::runtime_error("gf_penalty_dxi_rev: unsupported degree");
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(out, h->d_many, nout * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) throw std::runtime_error(std::string("gf_penalty_dxi_rev: ") + hipGetErrorString(e));
    } catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}

int gf_penalty_dxi(gf_handle* h, double* blocks, int64_t n, int32_t* windows, int64_t nw) {
    if (!h) return fail("gf_penalty_dxi: null argument");
    return gf_penalty_dxi_range(h, 0, (int64_t)h->H.npts, blocks, n, windows, nw);
}

#if defined(GF_STAMPS) || defined(GF_STAMPS_PEN)
// diagnostic build only: cycle sums per kernel section (lane 0 of every wave), then reset
int gf_debug_stamps(unsigned long long out[8]) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess;
}
#endif

int gf_functionals(gf_handle* h, double out[3], double* dWdu, double* dWdcp, double* dWdh, double* dVdcp, double* dVdh, int apply_bcs) {
    if (!h || !out) return fail("gf_functionals: null argument");
    try {
        HIPCHK(hipSetDevice(h->device));
        h->fun_owner = 0;
        switch (h->H.degree) {
            case 2: run_functionals<2>(h, apply_bcs); break;
            case 3: run_functionals<3>(h, apply_bcs); break;
            case 4: run_functionals<4>(h, apply_bcs); break;
            default: throw std::runtime_error("gf_functionals: unsupported degree");
        }
        const HostModel& H = h->H; const long long T = H.total_cp;
        std::vector<double> wp, vp, pi(H.ni, 0.0);
        patch_sums(h, h->d_x, h->d_y, wp, vp, nullptr);
        if (H.npts > 0) {       // per-interface sums of the vertex energies
            hipLaunchKernelGGL(seg_reduce_kernel, dim3(H.ni), dim3(256), 0, h->stream, h->d_if_off, h->d_pen_en, (const double*)nullptr, h->d_red + 2 * h->nred, (double*)nullptr, (double*)nullptr);
            HIPCHK(hipMemcpyAsync(pi.data(), h->d_red + 2 * h->nred, H.ni * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        }
        if (dWdu) HIPCHK(hipMemcpyAsync(dWdu, h->d_fun, 3 * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (dWdcp) HIPCHK(hipMemcpyAsync(dWdcp, h->d_fun + 3 * T, 3 * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (dWdh) HIPCHK(hipMemcpyAsync(dWdh, h->d_fun + 6 * T, T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (dVdcp) HIPCHK(hipMemcpyAsync(dVdcp, h->d_fun + 7 * T, 3 * T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        if (dVdh) HIPCHK(hipMemcpyAsync(dVdh, h->d_fun + 10 * T, T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        // fixed-order host sums of the per-element / per-vertex partials (owned elements only)
        long double W = 0, V = 0, Wp = 0;
        for (int s = 0; s < H.n_owned; ++s) { W += wp[s]; V += vp[s]; }           // owned patches, fixed order
        // an interface cut by the partition is present on both ranks: its energy is counted by the owner of side A
        for (int i = 0; i < H.ni; ++i) if (H.if_patch[2 * i] < H.n_owned) Wp += pi[i];
        out[0] = (double)W; out[1] = (double)V; out[2] = (double)Wp;
        h->fun_wp = wp; h->fun_vp = vp;
        for (int s = H.n_owned; s < H.np; ++s) { h->fun_wp[s] = 0.0; h->fun_vp[s] = 0.0; }      // ghost patches of a shard: reported by their owner
    } catch (const std::exception& ex) { return fail(ex.what()); }
    return 0;
}

int gf_functionals_per_patch(gf_handle* h, double* W_patch, double* V_patch, int64_t np) {
    if (!h) return fail("gf_functionals_per_patch: null handle");
    if (np != (int64_t)h->H.np) return fail("gf_functionals_per_patch: arrays must hold one value per patch");
    if (h->fun_wp.size() != (size_t)np) return fail("gf_functionals_per_patch: no gf_functionals call has been made on this handle");
    for (int64_t s = 0; s < np; ++s) { if (W_patch) W_patch[s] = h->fun_wp[s]; if (V_patch) V_patch[s] = h->fun_vp[s]; }
    return 0;
}

}  // extern "C"
