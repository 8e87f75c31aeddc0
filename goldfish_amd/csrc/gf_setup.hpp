// gf_setup.hpp -- host-side preprocessing of a gf_model_desc into the flat tables the
// kernels read: knot-span elements, 1-D Gauss/basis tables, control-point -> element ranges,
// neighbour (CSR block) lists, mortar-point basis values and the deterministic
// "owner" lists of the penalty coupling.  Pure C++ (no HIP calls).
//
// Reference counterparts: tIGAr ExtractedSpline construction + PENGoLINS
// create_transfer_matrix_list (nonmatching_opt.py:589-612, 1124-1136) -- here the
// "transfer matrices" A0/A1 are just the rational basis values/first derivatives at the
// mortar vertices, and the FE<->IGA extraction does not exist.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>
#include <thread>
#include <mutex>
#include <cstdlib>
#include "../../include/goldfish_model.h"

namespace gf {

constexpr int MAXP = 4;
constexpr int MAXNB = (MAXP + 1) * (MAXP + 1);

struct PatchDev {            // POD mirrored on the device
    int p, q, nu, nv, nelu, nelv;
    int tabu, tabv;          // offsets into tab[] : [nel][ng][3][deg+1]
    int wu, wv;              // offsets into tab[] : [nel][ng]
    int spu, spv;            // offsets into ints[] : span index per element
    int c2u, c2v;            // offsets into ints[] : [n][2] first/last element touching CP index
    long long cp_off, elem_off;
    double E, nu_, f[3], pd[3];   // pd != 0: load per unit projected area (gf_model_desc.load_proj)
    double press, et[12];         // follower pressure; dead edge tractions [2 d + side][3] (gf_model_desc.pressure / edge_traction; gf_extra_loads.hpp)
};

// per-element descriptor: one load instead of the elem_patch -> patch -> span-table chain of dependent scalar loads
struct ElemDesc { int patch, g0, nu, tabu, tabv, wu, wv, pad; };   // g0: global id of the element's first control point; offsets into tab[]

// per-control-point descriptor of the gather: element range, neighbour box and local indices in one load instead of the
// cp_patch -> patch -> c2u/c2v -> span-table chain of dependent loads
struct CpDesc { int patch, ia, ja, eu0, neu, ev0, nev, i0, j0, i1, j1, nelu; long long e00; int bu[5], bv[5]; };   // bu[k] = first CP index of element eu0 + k

// one work item of the walking kernel (gf_element_rec.hpp): the elements ev0 .. ev0 + nel - 1 of the strip eu of a patch
struct WalkItem { int patch, eu, ev0, nel, seg, cls, iu0, pad; };
// row-record path (gf_element_rec.hpp), per patch: first work item (items in the order strip-major, segment), segments per strip, and
// (offsets into ints[]) the segment of every element row / the first element row of every segment (nseg + 1 entries)
struct RecPatch { int item_off, nseg, seg_of, ev0_of; };
// per control point: the work items that hold pairs of it, in summation order (strips ascending, segments ascending) -- one load
// instead of the chain cp_desc -> c2v -> segment tables -> span table.  row = item * rec_rows + (ja - first row of the item): the
// row record of ja in that item; info = (iu0 - i0) | (ia - iu0) << 8 | pm << 16, pm bit d: the rows ja and ja - 3 + d share an element
// of the item (only then the pair (ja, ja - 3 + d) is present in its records)
struct RecCp { int nit, flags; struct { int row; unsigned info; } it[8]; };   // flags: bits 0-2 Dirichlet dofs of the control point
// p = 4 (gf_element_rec4.hpp): a control point lies in up to 5 strips x 2 segments; pm bit d: the rows ja and ja - 4 + d share an element of the item
struct RecCp4 { int nit, flags; struct { int row; unsigned info; } it[10]; };

struct PenRowItem { int a, code, lo, hi; };           // code = iface*2 + s
// one (owned control point, mortar vertex) visit of the penalty row kernel: everything the kernel needs to address the
// vertex record and both support windows without dependent index loads
struct PenEntry { int v, sal, baseA, baseB, pA, pB, pad0, pad1; };   // sal = side << 8 | local index; base = iu0 | iv0 << 16

inline void gauss_legendre(int n, double* x, double* w) {
    for (int i = 0; i < n; ++i) {
        double z = std::cos(M_PI * (i + 0.75) / (n + 0.5)), pp = 0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1, p2 = 0;
            for (int j = 1; j <= n; ++j) { double p3 = p2; p2 = p1; p1 = ((2.0 * j - 1) * z * p2 - (j - 1.0) * p3) / j; }
            pp = n * (z * p1 - p2) / (z * z - 1);
            double dz = p1 / pp; z -= dz;
            if (std::fabs(dz) < 1e-16) break;
        }
        x[n - 1 - i] = z; w[n - 1 - i] = 2 / ((1 - z * z) * pp * pp);
    }
}

inline int find_span(int n, int p, const double* U, double xi) {
    if (xi >= U[n]) { int i = n - 1; while (i > p && U[i] >= U[i + 1]) --i; return i; }
    if (xi <= U[p]) { int i = p; while (i < n - 1 && U[i] >= U[i + 1]) ++i; return i; }
    int lo = p, hi = n, mid = (lo + hi) / 2;
    while (xi < U[mid] || xi >= U[mid + 1]) { if (xi < U[mid]) hi = mid; else lo = mid; mid = (lo + hi) / 2; }
    return mid;
}

// values, first and second derivatives of the p+1 non-zero B-spline basis functions
inline void basis_ders(int span, double xi, int p, const double* U, double ders[3][MAXP + 1]) {
    double ndu[MAXP + 1][MAXP + 1], left[MAXP + 1], right[MAXP + 1], a[2][MAXP + 1];
    ndu[0][0] = 1;
    for (int j = 1; j <= p; ++j) {
        left[j] = xi - U[span + 1 - j]; right[j] = U[span + j] - xi;
        double saved = 0;
        for (int r = 0; r < j; ++r) {
            ndu[j][r] = right[r + 1] + left[j - r];
            double temp = ndu[r][j - 1] / ndu[j][r];
            ndu[r][j] = saved + right[r + 1] * temp; saved = left[j - r] * temp;
        }
        ndu[j][j] = saved;
    }
    for (int j = 0; j <= p; ++j) ders[0][j] = ndu[j][p];
    for (int k = 1; k <= 2; ++k) for (int j = 0; j <= p; ++j) ders[k][j] = 0;
    const int nd = p < 2 ? p : 2;
    for (int r = 0; r <= p; ++r) {
        int s1 = 0, s2 = 1; a[0][0] = 1;
        for (int k = 1; k <= nd; ++k) {
            double d = 0; const int rk = r - k, pk = p - k;
            if (r >= k) { a[s2][0] = a[s1][0] / ndu[pk + 1][rk]; d = a[s2][0] * ndu[rk][pk]; }
            const int j1 = rk >= -1 ? 1 : -rk, j2 = (r - 1 <= pk) ? k - 1 : p - r;
            for (int j = j1; j <= j2; ++j) { a[s2][j] = (a[s1][j] - a[s1][j - 1]) / ndu[pk + 1][rk + j]; d += a[s2][j] * ndu[rk + j][pk]; }
            if (r <= pk) { a[s2][k] = -a[s1][k - 1] / ndu[pk + 1][r]; d += a[s2][k] * ndu[r][pk]; }
            ders[k][r] = d; std::swap(s1, s2);
        }
    }
    double r = p;
    for (int k = 1; k <= nd; ++k) { for (int j = 0; j <= p; ++j) ders[k][j] *= r; r *= (p - k); }
}


// Index ranges [0, n) cut into contiguous chunks, one thread per chunk (GF_SETUP_THREADS, default: the hardware threads, at most 16); the first exception of any
// chunk is rethrown on the caller's thread.  The set-up loops below write disjoint entries per index, so the result does not depend on the thread count.
template <class Fn> inline void parallel_chunks(int64_t n, Fn&& fn, int64_t min_chunk = 1024) {
    int nt = 1;
    if (const char* e = std::getenv("GF_SETUP_THREADS")) nt = std::atoi(e);
    else { nt = (int)std::thread::hardware_concurrency(); if (nt > 16) nt = 16; }
    if (nt < 1) nt = 1;
    if ((int64_t)nt > n / min_chunk) nt = (int)std::max<int64_t>(1, n / min_chunk);
    if (nt <= 1) { fn((int64_t)0, n); return; }
    std::vector<std::thread> th; std::mutex mu; std::string err; bool failed = false;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        th.emplace_back([&, lo, hi] {
            try { fn(lo, hi); }
            catch (const std::exception& ex) { std::lock_guard<std::mutex> g(mu); if (!failed) { failed = true; err = ex.what(); } }
        });
    }
    for (auto& x : th) x.join();
    if (failed) throw std::runtime_error(err);
}

struct HostModel {
    int np = 0, n_owned = 0, degree = 0, ni = 0;
    int64_t owned_cp = 0;               // control points of the owned patches (first owned_cp ids)
    int64_t total_cp = 0, ndof = 0, nelem = 0, ngp = 0, npts = 0;
    std::vector<PatchDev> patches;
    std::vector<double> tab;            // 1-D tables
    std::vector<int> ints;              // spans + cp->element ranges
    std::vector<int> elem_patch;        // [nelem]
    std::vector<ElemDesc> elem_desc;    // [nelem]
    std::vector<CpDesc> cp_desc;        // [total_cp]
    std::vector<WalkItem> rec_items; std::vector<RecPatch> rec_patch; std::vector<RecCp> rec_cp; std::vector<RecCp4> rec_cp4; int rec_rows = 0;   // row-record path
    void build_rec(int seg_len);
    std::vector<int> rec_order; int rec_npoly = 0;      // work items of the polynomial patches first, then those of the rational ones (WalkItem::pad), each in natural order
    void eval_mortar_vertex(int patch, double xu, double xv, int win[2], double* nu, double* nu2) const;
    std::vector<int> cp_patch;          // [total_cp]
    std::vector<double> weights;
    std::vector<unsigned char> zero;    // [ndof]
    std::vector<int64_t> pl_dof; std::vector<double> pl_val;
    bool symmetric_K = true;            // false: a follower pressure adds its (non-symmetric) load stiffness to K
    std::vector<int> load_cps;          // owned control points that carry a follower pressure or an edge traction (kl_extra_loads_kernel)
    // set-up phases and their wall time in ms (printed by gf_create when GF_SETUP_TIMING=1)
    std::vector<std::pair<std::string, double>> timing;
    std::chrono::steady_clock::time_point t_last = std::chrono::steady_clock::now();
    void tick(const char* what) {
        const auto t = std::chrono::steady_clock::now();
        timing.push_back({what, std::chrono::duration<double, std::milli>(t - t_last).count()}); t_last = t;
    }
    // neighbour lists (CP level): shell only / shell + coupling
    std::vector<int64_t> nb_ptr_s, nb_ptr_c; std::vector<int> nb_s, nb_c;
    std::vector<int> nb_rev_s, nb_rev_c;   // per neighbour entry (a, k) with b = nb[k]: position of a in b's list (the relation is symmetric): fixed-order transposed products
    std::vector<unsigned short> nb_meta;   // per nb_c entry: bits 0-6 box slot of the neighbour (127: coupling-only column), 7-9 Dirichlet flags of its dofs, 10 self
    // mortar points
    std::vector<double> knots; std::vector<int64_t> knot_off;   // copies of the desc's knot vectors (gf_update_interface re-evaluates mortar vertices)
    std::vector<int> pt_iface;          // [npts]
    std::vector<int> pt_base;           // [npts][2][2] (iu0, iv0) per side
    std::vector<double> pt_nu;          // [npts][2][3][NB] rational value/d1/d2
    std::vector<double> pt_nu2;         // [npts][2][3][NB] rational second derivatives uu/vv/uv (moving intersections, dR/dxi)
    std::vector<double> pt_tau, pt_wt;  // [npts][2], [npts]
    std::vector<int> if_patch; std::vector<double> if_alpha; std::vector<int64_t> if_off;
    // deterministic owner lists
    std::vector<PenRowItem> row_items; std::vector<int64_t> row_ptr;      // groups by CP a
    std::vector<PenEntry> pen_entries; std::vector<int64_t> ent_ptr; std::vector<int> row_cp;   // per group: its visits in fixed order
    std::vector<unsigned short> pen_slots;   // [visit][side][LW] (LW = 16 for p <= 3, 32 for p = 4) index of the window's control points in the row's coupled neighbour list

    void build(const gf_model_desc* D);
};

inline void HostModel::build(const gf_model_desc* D) {
    t_last = std::chrono::steady_clock::now();
    np = D->n_patches;
    if (np <= 0) throw std::runtime_error("gf_create: model has no patches");
    degree = D->degree[0];
    if (degree < 2 || degree > MAXP) throw std::runtime_error("gf_create: degree must be 2.." + std::to_string(MAXP) + " (KL shells need C1)");
    total_cp = D->cp_off[np]; ndof = 3 * total_cp;
    n_owned = (D->n_owned_patches > 0 && D->n_owned_patches < np) ? D->n_owned_patches : np;
    owned_cp = D->cp_off[n_owned];
    if (total_cp >= (int64_t(1) << 31) / 3) throw std::runtime_error("gf_create: too many control points for 32-bit column ids");
    patches.resize(np); cp_patch.resize(total_cp);
    weights.assign(D->weights, D->weights + total_cp);
    nelem = 0; ngp = 0;
    for (int s = 0; s < np; ++s) {
        PatchDev& P = patches[s];
        P.p = D->degree[2 * s]; P.q = D->degree[2 * s + 1];
        if (P.p != degree || P.q != degree) throw std::runtime_error("gf_create: all patches must share one degree p = q (got a mixed-degree model)");
        P.nu = D->ncp[2 * s]; P.nv = D->ncp[2 * s + 1];
        P.cp_off = D->cp_off[s]; P.E = D->young[s]; P.nu_ = D->poisson[s];
        if (D->cp_off[s + 1] - D->cp_off[s] != int64_t(P.nu) * P.nv) throw std::runtime_error("gf_create: cp_off inconsistent with ncp");
        for (int k = 0; k < 3; ++k) { P.f[k] = D->body_force ? D->body_force[3 * s + k] : 0.0; P.pd[k] = D->load_proj ? D->load_proj[3 * s + k] : 0.0; }
        P.press = D->pressure ? D->pressure[s] : 0.0;
        for (int k = 0; k < 12; ++k) P.et[k] = D->edge_traction ? D->edge_traction[12 * s + k] : 0.0;
        if (P.press != 0.0) symmetric_K = false;
        for (int64_t a = P.cp_off; a < D->cp_off[s + 1]; ++a) cp_patch[a] = s;
        for (int d = 0; d < 2; ++d) {
            const int p = d ? P.q : P.p, n = d ? P.nv : P.nu;
            const double* U = D->knots + D->knot_off[2 * s + d];
            if (D->knot_off[2 * s + d + 1] - D->knot_off[2 * s + d] != n + p + 1) throw std::runtime_error("gf_create: knot vector length != n + p + 1");
            std::vector<int> sp;
            for (int i = p; i < n; ++i) if (U[i + 1] > U[i]) sp.push_back(i);
            const int nel = (int)sp.size(), ng = p + 1;
            double gx[MAXP + 2], gw[MAXP + 2]; gauss_legendre(ng, gx, gw);
            const int t0 = (int)tab.size(); tab.resize(t0 + size_t(nel) * ng * 3 * (p + 1));
            const int w0 = (int)tab.size(); tab.resize(w0 + size_t(nel) * ng);
            for (int e = 0; e < nel; ++e) for (int g = 0; g < ng; ++g) {
                const double a = U[sp[e]], b = U[sp[e] + 1], xi = 0.5 * (a + b) + 0.5 * (b - a) * gx[g];
                double ders[3][MAXP + 1]; basis_ders(sp[e], xi, p, U, ders);
                for (int k = 0; k < 3; ++k) for (int j = 0; j <= p; ++j) tab[t0 + ((size_t(e) * ng + g) * 3 + k) * (p + 1) + j] = ders[k][j];
                tab[w0 + e * ng + g] = 0.5 * (b - a) * gw[g];
            }
            const int s0 = (int)ints.size(); ints.insert(ints.end(), sp.begin(), sp.end());
            const int c0 = (int)ints.size(); ints.resize(c0 + 2 * n);
            for (int i = 0; i < n; ++i) {            // elements whose support contains CP index i: span in [i, i+p]
                int lo = nel, hi = -1;
                for (int e = 0; e < nel; ++e) if (sp[e] >= i && sp[e] <= i + p) { lo = std::min(lo, e); hi = std::max(hi, e); }
                ints[c0 + 2 * i] = lo; ints[c0 + 2 * i + 1] = hi;
            }
            if (d) { P.nelv = nel; P.tabv = t0; P.wv = w0; P.spv = s0; P.c2v = c0; }
            else   { P.nelu = nel; P.tabu = t0; P.wu = w0; P.spu = s0; P.c2u = c0; }
        }
        P.elem_off = nelem; nelem += int64_t(P.nelu) * P.nelv;
        if (s < n_owned) ngp += int64_t(P.nelu) * P.nelv * (P.p + 1) * (P.q + 1);
    }
    if (nelem >= (int64_t(1) << 31)) throw std::runtime_error("gf_create: too many elements");
    elem_patch.resize(nelem);
    for (int s = 0; s < np; ++s) std::fill(elem_patch.begin() + patches[s].elem_off, elem_patch.begin() + patches[s].elem_off + int64_t(patches[s].nelu) * patches[s].nelv, s);
    elem_desc.resize(nelem);
    for (int s = 0; s < np; ++s) {
        const PatchDev& P = patches[s];
        const int p1 = P.p + 1, q1 = P.q + 1;
        for (int ev = 0; ev < P.nelv; ++ev) for (int eu = 0; eu < P.nelu; ++eu) {
            const int iu0 = ints[P.spu + eu] - P.p, iv0 = ints[P.spv + ev] - P.q;
            elem_desc[P.elem_off + eu + int64_t(ev) * P.nelu] = {s, int(P.cp_off + iu0 + int64_t(iv0) * P.nu), P.nu, P.tabu + eu * p1 * 3 * p1, P.tabv + ev * q1 * 3 * q1,
                                                                  P.wu + eu * p1, P.wv + ev * q1, 0};
        }
    }
    zero.assign(ndof, 0);
    if ((D->n_zero_dofs > 0 && !D->zero_dofs) || (D->n_point_loads > 0 && (!D->pl_dof || !D->pl_val))) throw std::runtime_error("gf_create: a count is positive but its array is NULL");
    for (int64_t k = 0; k < D->n_zero_dofs; ++k) {
        if (D->zero_dofs[k] < 0 || D->zero_dofs[k] >= ndof) throw std::runtime_error("gf_create: zero_dofs out of range");
        zero[D->zero_dofs[k]] = 1;
    }
    {   // point loads: duplicates summed here (fixed order) so that the device applies one value per dof
        std::vector<std::pair<int64_t, double>> pl;
        for (int64_t k = 0; k < D->n_point_loads; ++k) {
            if (D->pl_dof[k] < 0 || D->pl_dof[k] >= ndof) throw std::runtime_error("gf_create: pl_dof out of range");
            pl.push_back({D->pl_dof[k], D->pl_val[k]});
        }
        std::stable_sort(pl.begin(), pl.end(), [](const std::pair<int64_t, double>& a, const std::pair<int64_t, double>& b) { return a.first < b.first; });
        for (const auto& e : pl) {
            if (!pl_dof.empty() && pl_dof.back() == e.first) pl_val.back() += e.second;
            else { pl_dof.push_back(e.first); pl_val.push_back(e.second); }
        }
    }

    tick("patch tables, elements, Dirichlet, point loads");
    // ---- mortar points ------------------------------------------------------------
    ni = D->n_interfaces;
    const int NB = (degree + 1) * (degree + 1);
    knot_off.assign(D->knot_off, D->knot_off + 2 * np + 1); knots.assign(D->knots, D->knots + knot_off[2 * np]);
    if (ni > 0 && (!D->if_patch || !D->if_off || !D->if_xi || !D->if_tau || !D->if_wt || !D->if_alpha)) throw std::runtime_error("gf_create: n_interfaces > 0 but an interface array is NULL");
    npts = ni > 0 ? D->if_off[ni] : 0;
    if_patch.assign(D->if_patch, D->if_patch + 2 * ni); if_alpha.assign(D->if_alpha, D->if_alpha + 2 * ni);
    if (ni > 0) if_off.assign(D->if_off, D->if_off + ni + 1); else if_off.assign(1, 0);     // the interface arrays may be NULL when there is none
    pt_iface.resize(npts); pt_base.resize(4 * npts); pt_nu.assign(size_t(npts) * 2 * 3 * NB, 0.0); pt_nu2.assign(size_t(npts) * 2 * 3 * NB, 0.0);
    pt_tau.assign(D->if_tau, D->if_tau + 2 * npts); pt_wt.assign(D->if_wt, D->if_wt + npts);
    for (int i = 0; i < ni; ++i) {
        for (int sd = 0; sd < 2; ++sd) if (if_patch[2 * i + sd] < 0 || if_patch[2 * i + sd] >= np) throw std::runtime_error("gf_create: if_patch out of range");
        for (int64_t v = if_off[i]; v < if_off[i + 1]; ++v) {
            pt_iface[v] = i;
            for (int sd = 0; sd < 2; ++sd) {
                int win[2];
                eval_mortar_vertex(if_patch[2 * i + sd], D->if_xi[4 * v + 2 * sd], D->if_xi[4 * v + 2 * sd + 1], win, &pt_nu[(size_t(v) * 2 + sd) * 3 * NB], &pt_nu2[(size_t(v) * 2 + sd) * 3 * NB]);
                pt_base[4 * v + 2 * sd] = win[0]; pt_base[4 * v + 2 * sd + 1] = win[1];
            }
        }
    }

    // ---- per (interface, side): CP -> bounding range of mortar points -------------------
    struct CpRange { int cp, lo, hi; };
    std::vector<std::vector<CpRange>> ranges(2 * size_t(ni));
    {
        std::vector<int> lo(total_cp, -1), hi(total_cp, -1);
        for (int i = 0; i < ni; ++i) for (int sd = 0; sd < 2; ++sd) {
            const PatchDev& P = patches[if_patch[2 * i + sd]];
            std::vector<int> touched;
            for (int64_t v = if_off[i]; v < if_off[i + 1]; ++v)
                for (int jv = 0; jv <= P.q; ++jv) for (int ju = 0; ju <= P.p; ++ju) {
                    const int64_t a = P.cp_off + (pt_base[4 * v + 2 * sd] + ju) + int64_t(pt_base[4 * v + 2 * sd + 1] + jv) * P.nu;
                    if (lo[a] < 0) { lo[a] = (int)v; touched.push_back((int)a); }
                    hi[a] = (int)v;
                }
            std::sort(touched.begin(), touched.end());
            for (int a : touched) { ranges[2 * i + sd].push_back({a, lo[a], hi[a]}); lo[a] = hi[a] = -1; }
        }
    }
    tick("mortar vertices: spans, basis values, control-point ranges");
    // coupling partners per CP and the block / row owner lists
    // two control points couple iff some mortar vertex has both in its support windows (the ranges are
    // only bounding ranges when an intersection curve is not monotone in the control net)
    auto in_window = [&](int64_t v, int itf, int sd, int cp) {
        const PatchDev& P = patches[if_patch[2 * itf + sd]];
        const int l = int(cp - P.cp_off), di = l % P.nu - pt_base[4 * v + 2 * sd], dj = l / P.nu - pt_base[4 * v + 2 * sd + 1];
        return di >= 0 && di <= P.p && dj >= 0 && dj <= P.q;
    };
    auto cosupport = [&](int itf, int s, const CpRange& A, int t, const CpRange& B) {
        for (int64_t v = std::max(A.lo, B.lo); v <= std::min(A.hi, B.hi); ++v)
            if (in_window(v, itf, s, A.cp) && in_window(v, itf, t, B.cp)) return true;
        return false;
    };
    // coupling lists: the control points of both sides whose windows share a mortar vertex.  The pairs of an interface are found in parallel (the test walks the
    // common vertex range), then appended per control point in interface order (a corner control point lies on two interfaces)
    std::vector<std::vector<int>> extra(total_cp);
    {
        std::vector<std::vector<std::pair<int, int>>> found((size_t)ni);
        parallel_chunks(ni, [&](int64_t i0, int64_t i1) {
            for (int i = (int)i0; i < (int)i1; ++i) for (int s = 0; s < 2; ++s) for (int t = 0; t < 2; ++t)
                for (const CpRange& A : ranges[2 * i + s]) for (const CpRange& B : ranges[2 * i + t])
                    if (A.lo <= B.hi && B.lo <= A.hi && cosupport(i, s, A, t, B)) found[(size_t)i].push_back({(int)A.cp, (int)B.cp});
        }, 1);
        for (int i = 0; i < ni; ++i) for (const auto& ab : found[(size_t)i]) extra[(size_t)ab.first].push_back(ab.second);
    }
    tick("coupling pairs of the interfaces");
    nb_ptr_s.assign(total_cp + 1, 0); nb_ptr_c.assign(total_cp + 1, 0);
    // per control point: its box of shell neighbours (the control points of the elements it lies in), merged with the coupling lists; sizes first, then the
    // prefix sums, then the lists -- both passes over the patches in parallel (every control point writes its own entries)
    auto for_cps = [&](int pass) {
        parallel_chunks(np, [&](int64_t s0, int64_t s1) {
            std::vector<int> box;
            for (int64_t s = s0; s < s1; ++s) {
                const PatchDev& P = patches[s];
                const int *spu = &ints[P.spu], *spv = &ints[P.spv], *c2u = &ints[P.c2u], *c2v = &ints[P.c2v];
                for (int j = 0; j < P.nv; ++j) for (int i = 0; i < P.nu; ++i) {
                    const int64_t a = P.cp_off + i + int64_t(j) * P.nu;
                    box.clear();
                    if (c2u[2 * i + 1] >= 0 && c2v[2 * j + 1] >= 0) {
                        const int i0 = spu[c2u[2 * i]] - P.p, i1 = spu[c2u[2 * i + 1]], j0 = spv[c2v[2 * j]] - P.q, j1 = spv[c2v[2 * j + 1]];
                        for (int jj = j0; jj <= j1; ++jj) for (int ii = i0; ii <= i1; ++ii) box.push_back(int(P.cp_off + ii + int64_t(jj) * P.nu));
                    }
                    if (pass == 0) {
                        nb_ptr_s[a + 1] = (int64_t)box.size();
                        std::vector<int>& ex = extra[a];
                        if (!ex.empty()) {
                            ex.insert(ex.end(), box.begin(), box.end());
                            std::sort(ex.begin(), ex.end()); ex.erase(std::unique(ex.begin(), ex.end()), ex.end());
                            nb_ptr_c[a + 1] = (int64_t)ex.size();
                        } else nb_ptr_c[a + 1] = (int64_t)box.size();
                    } else {
                        std::copy(box.begin(), box.end(), nb_s.begin() + nb_ptr_s[a]);
                        const std::vector<int>& src = extra[a].empty() ? box : extra[a];
                        std::copy(src.begin(), src.end(), nb_c.begin() + nb_ptr_c[a]);
                    }
                }
            }
        }, 1);
    };
    for_cps(0);
    for (int64_t a = 0; a < total_cp; ++a) { nb_ptr_s[a + 1] += nb_ptr_s[a]; nb_ptr_c[a + 1] += nb_ptr_c[a]; }
    nb_s.resize(nb_ptr_s[total_cp]); nb_c.resize(nb_ptr_c[total_cp]);
    for_cps(1);
    {   // reverse indices (lists are sorted ascending)
        auto build_rev = [&](const std::vector<int64_t>& ptr, const std::vector<int>& nb, std::vector<int>& rev) {
            rev.assign(nb.size(), 0);
            parallel_chunks(total_cp, [&](int64_t a0, int64_t a1) {
                for (int64_t a = a0; a < a1; ++a) for (int64_t k = ptr[a]; k < ptr[a + 1]; ++k) {
                    const int b = nb[k];
                    const int* lo = nb.data() + ptr[b]; const int* hi = nb.data() + ptr[b + 1];
                    const int* it = std::lower_bound(lo, hi, (int)a);
                    if (it == hi || *it != (int)a) throw std::runtime_error("gf_create: neighbour relation is not symmetric");
                    rev[k] = int(it - lo);
                }
            });
        };
        build_rev(nb_ptr_s, nb_s, nb_rev_s); build_rev(nb_ptr_c, nb_c, nb_rev_c);
    }
    cp_desc.assign(total_cp, CpDesc{});
    parallel_chunks(np, [&](int64_t s0_, int64_t s1_) { for (int s = (int)s0_; s < (int)s1_; ++s) {
        const PatchDev& P = patches[s];
        const int *spu = &ints[P.spu], *spv = &ints[P.spv], *c2u = &ints[P.c2u], *c2v = &ints[P.c2v];
        for (int j = 0; j < P.nv; ++j) for (int i = 0; i < P.nu; ++i) {
            CpDesc& c = cp_desc[P.cp_off + i + int64_t(j) * P.nu];
            const int eu0 = c2u[2 * i], eu1 = c2u[2 * i + 1], ev0 = c2v[2 * j], ev1 = c2v[2 * j + 1];
            const bool has = eu1 >= eu0 && ev1 >= ev0;
            c.patch = s; c.ia = i; c.ja = j; c.eu0 = has ? eu0 : 0; c.neu = has ? eu1 - eu0 + 1 : 0; c.ev0 = has ? ev0 : 0; c.nev = has ? ev1 - ev0 + 1 : 0;
            c.i0 = has ? spu[eu0] - P.p : 0; c.j0 = has ? spv[ev0] - P.q : 0; c.i1 = has ? spu[eu1] : -1; c.j1 = has ? spv[ev1] : -1;
            c.nelu = P.nelu; c.e00 = P.elem_off + c.eu0 + int64_t(c.ev0) * P.nelu;
            if (c.neu > 5 || c.nev > 5) throw std::runtime_error("gf_create: a control point lies in more than 5 knot spans per direction");
            for (int k = 0; k < 5; ++k) { c.bu[k] = k < c.neu ? spu[eu0 + k] - P.p : 0; c.bv[k] = k < c.nev ? spv[ev0 + k] - P.q : 0; }
        }
    } }, 1);
    tick("neighbour lists (shell, coupling), reverse indices, control-point descriptors");
    // per-entry metadata of the coupling lists: what the gather's write phase would otherwise derive from dependent loads
    nb_meta.assign(nb_c.size(), 0);
    parallel_chunks(np, [&](int64_t s0_, int64_t s1_) { for (int s = (int)s0_; s < (int)s1_; ++s) {
        const PatchDev& P = patches[s];
        const int *spu = &ints[P.spu], *spv = &ints[P.spv], *c2u = &ints[P.c2u], *c2v = &ints[P.c2v];
        for (int j = 0; j < P.nv; ++j) for (int i = 0; i < P.nu; ++i) {
            const int64_t a = P.cp_off + i + int64_t(j) * P.nu;
            const bool has = c2u[2 * i + 1] >= 0 && c2v[2 * j + 1] >= 0;
            const int i0 = has ? spu[c2u[2 * i]] - P.p : 0, i1 = has ? spu[c2u[2 * i + 1]] : -1, j0 = has ? spv[c2v[2 * j]] - P.q : 0, j1 = has ? spv[c2v[2 * j + 1]] : -1;
            const int wbox = i1 - i0 + 1;
            for (int64_t k = nb_ptr_c[a]; k < nb_ptr_c[a + 1]; ++k) {
                const int64_t b = nb_c[k];
                int slot = 127;
                if (b >= P.cp_off && b < P.cp_off + int64_t(P.nu) * P.nv) {
                    const int lb = int(b - P.cp_off), ib = lb % P.nu, jb = lb / P.nu;
                    if (ib >= i0 && ib <= i1 && jb >= j0 && jb <= j1) slot = (ib - i0) + (jb - j0) * wbox;
                }
                if (slot != 127 && slot > 126) throw std::runtime_error("gf_create: neighbour box too large");
                nb_meta[k] = (unsigned short)(slot | (zero[3 * b] ? 128 : 0) | (zero[3 * b + 1] ? 256 : 0) | (zero[3 * b + 2] ? 512 : 0) | (b == a ? 1024 : 0));
            }
        }
    } }, 1);
    // control points of the loads that run behind the gather (gf_extra_loads.hpp): every control point of a pressurised patch, the
    // edge row of a loaded edge
    {
        std::vector<unsigned char> mark(owned_cp > 0 ? owned_cp : 1, 0);
        for (int s = 0; s < n_owned; ++s) {
            const PatchDev& P = patches[s];
            if (P.press != 0.0) for (int64_t a = P.cp_off; a < P.cp_off + int64_t(P.nu) * P.nv; ++a) mark[a] = 1;
            for (int e = 0; e < 4; ++e) {
                if (P.et[3 * e] == 0.0 && P.et[3 * e + 1] == 0.0 && P.et[3 * e + 2] == 0.0) continue;
                const int d = e >> 1, side = e & 1;
                if (d == 0) { const int i = side ? P.nu - 1 : 0; for (int j = 0; j < P.nv; ++j) mark[P.cp_off + i + int64_t(j) * P.nu] = 1; }
                else { const int j = side ? P.nv - 1 : 0; for (int i = 0; i < P.nu; ++i) mark[P.cp_off + i + int64_t(j) * P.nu] = 1; }
            }
        }
        load_cps.clear();
        for (int64_t a = 0; a < owned_cp; ++a) if (mark[a]) load_cps.push_back((int)a);
    }
    tick("per-entry metadata of the coupling lists");
    // owner lists: (interface, side, vertex range) items grouped by owned control point
    {
        std::vector<PenRowItem> rows;
        for (int i = 0; i < ni; ++i) for (int s = 0; s < 2; ++s) for (const CpRange& A : ranges[2 * i + s]) if (A.cp < owned_cp) rows.push_back({A.cp, 2 * i + s, A.lo, A.hi});
        std::stable_sort(rows.begin(), rows.end(), [](const PenRowItem& x, const PenRowItem& y) { return x.a < y.a; });
        row_items = rows; row_ptr.clear(); row_ptr.push_back(0);
        for (size_t k = 1; k <= rows.size(); ++k) if (k == rows.size() || rows[k].a != rows[k - 1].a) row_ptr.push_back((int64_t)k);
        // the visits of a row group (an owned control point a): every mortar vertex of its ranges whose window holds a.  Two passes over the groups, both in
        // parallel: count, prefix sums, fill (a visit costs 32 binary searches in a's neighbour list: 1.4 of the 2.7 s of gf_create at C4 when done serially)
        const int64_t ngroups = (int64_t)row_ptr.size() - 1;
        const bool slots = degree <= 4;
        const int slot_w = degree <= 3 ? 16 : 32;               // lanes of a visit's row in pen_row16_kernel
        auto visits = [&](int64_t g, auto&& emit) {
            const int a = rows[row_ptr[g]].a;
            for (int64_t it = row_ptr[g]; it < row_ptr[g + 1]; ++it) {
                const PenRowItem& I = rows[it];
                const int itf = I.code >> 1, sd = I.code & 1;
                const PatchDev& P = patches[if_patch[2 * itf + sd]];
                const int l = int(a - P.cp_off), ia = l % P.nu, ja = l / P.nu;
                for (int v = I.lo; v <= I.hi; ++v) {
                    const int di = ia - pt_base[4 * v + 2 * sd], dj = ja - pt_base[4 * v + 2 * sd + 1];
                    if (di < 0 || di > P.p || dj < 0 || dj > P.q) continue;
                    emit(a, itf, sd, v, di + dj * (P.p + 1));
                }
            }
        };
        ent_ptr.assign((size_t)ngroups + 1, 0); row_cp.assign((size_t)ngroups, 0);
        parallel_chunks(ngroups, [&](int64_t g0, int64_t g1) {
            for (int64_t g = g0; g < g1; ++g) {
                int64_t n = 0;
                visits(g, [&](int, int, int, int, int) { ++n; });
                ent_ptr[(size_t)g + 1] = n; row_cp[(size_t)g] = rows[row_ptr[g]].a;
            }
        });
        for (int64_t g = 0; g < ngroups; ++g) ent_ptr[(size_t)g + 1] += ent_ptr[(size_t)g];
        pen_entries.assign((size_t)ent_ptr[(size_t)ngroups], PenEntry{});
        pen_slots.assign(slots ? (size_t)ent_ptr[(size_t)ngroups] * 2 * slot_w : 0, (unsigned short)0xFFFF);
        parallel_chunks(ngroups, [&](int64_t g0, int64_t g1) {
            for (int64_t g = g0; g < g1; ++g) {
                int64_t e = ent_ptr[(size_t)g];
                visits(g, [&](int a, int itf, int sd, int v, int al) {
                    for (int k = 0; k < 4; ++k) if (pt_base[4 * v + k] < 0 || pt_base[4 * v + k] > 32767) throw std::runtime_error("gf_create: patch too large for the packed mortar windows");
                    pen_entries[(size_t)e] = PenEntry{v, (sd << 8) | al, pt_base[4 * v] | (pt_base[4 * v + 1] << 16), pt_base[4 * v + 2] | (pt_base[4 * v + 3] << 16),
                                                      if_patch[2 * itf], if_patch[2 * itf + 1], 0, 0};
                    if (slots) {                                     // where the window's control points sit in a's coupled neighbour list (sorted by id)
                        const int p1 = degree + 1;
                        const int* nb0 = nb_c.data() + nb_ptr_c[a]; const int* nb1 = nb_c.data() + nb_ptr_c[a + 1];
                        for (int t = 0; t < 2; ++t) {
                            const PatchDev& Pt = patches[if_patch[2 * itf + t]];
                            for (int c = 0; c < p1 * p1; ++c) {
                                const int64_t b = Pt.cp_off + (pt_base[4 * v + 2 * t] + c % p1) + int64_t(pt_base[4 * v + 2 * t + 1] + c / p1) * Pt.nu;
                                const int* it2 = std::lower_bound(nb0, nb1, (int)b);
                                if (it2 == nb1 || *it2 != (int)b) throw std::runtime_error("gf_create: a mortar vertex couples control points that are not neighbours");
                                pen_slots[((size_t)e * 2 + t) * slot_w + c] = (unsigned short)(it2 - nb0);
                            }
                        }
                    }
                    ++e;
                });
            }
        });
    }
    tick("owner lists of the penalty rows, visit records, window slots");
}

// Support window (first control-point indices) and rational basis values / first / second derivatives of patch `patch` at the parametric point (xu, xv):
// nu [3][NB] = R, R_u, R_v; nu2 [3][NB] = R_uu, R_vv, R_uv (quotient rule on the tensor-product B-spline values)
inline void HostModel::eval_mortar_vertex(int s, double xu, double xv, int win[2], double* o, double* o2) const {
    const PatchDev& P = patches[s];
    const int NB = (degree + 1) * (degree + 1);
    const double* Uu = knots.data() + knot_off[2 * s]; const double* Uv = knots.data() + knot_off[2 * s + 1];
    const int su = find_span(P.nu, P.p, Uu, xu), sv = find_span(P.nv, P.q, Uv, xv);
    double du[3][MAXP + 1], dv[3][MAXP + 1]; basis_ders(su, xu, P.p, Uu, du); basis_ders(sv, xv, P.q, Uv, dv);
    double N[6][MAXNB], W[6] = {0, 0, 0, 0, 0, 0};
    for (int jv = 0; jv <= P.q; ++jv) for (int ju = 0; ju <= P.p; ++ju) {
        const int a = ju + jv * (P.p + 1);
        const double w = weights[P.cp_off + (su - P.p + ju) + int64_t(sv - P.q + jv) * P.nu];
        N[0][a] = du[0][ju] * dv[0][jv]; N[1][a] = du[1][ju] * dv[0][jv]; N[2][a] = du[0][ju] * dv[1][jv];
        N[3][a] = du[2][ju] * dv[0][jv]; N[4][a] = du[0][ju] * dv[2][jv]; N[5][a] = du[1][ju] * dv[1][jv];
        for (int k = 0; k < 6; ++k) W[k] += N[k][a] * w;
    }
    win[0] = su - P.p; win[1] = sv - P.q;
    for (int a = 0; a < NB; ++a) {
        const double R = N[0][a] / W[0];
        const double R1 = (N[1][a] - R * W[1]) / W[0], R2 = (N[2][a] - R * W[2]) / W[0];
        o[a] = R; o[NB + a] = R1; o[2 * NB + a] = R2;
        o2[a] = (N[3][a] - 2 * R1 * W[1] - R * W[3]) / W[0];
        o2[NB + a] = (N[4][a] - 2 * R2 * W[2] - R * W[4]) / W[0];
        o2[2 * NB + a] = (N[5][a] - R1 * W[2] - R2 * W[1] - R * W[5]) / W[0];
    }
}

// Tables of the row-record path: work items in natural order (patch, strip, segment) -- nothing depends on launch order --, per
// patch the segment of every element row, and the number of record rows an item can touch.  Called before the tables are uploaded.
inline void HostModel::build_rec(int seg_len) {
    if (degree < 2 || degree > 4) throw std::runtime_error("build_rec: p = 2, 3, 4 only");
    const int P1 = degree + 1, PW = degree == 4 ? 4 : 3, MAXIT = degree == 4 ? 10 : 8;      // PW: row reach of a pair in the record layout (the p = 2 kernel runs on the p = 3 tile)
    rec_items.clear(); rec_patch.assign(np, RecPatch{0, 0, 0, 0}); rec_rows = 0;
    // seg_len <= 0: whole strips (no partial sums at segment ends: 11 % fewer record bytes than 24-element items) unless the model is
    // too small to fill the device that way -- then the strips are cut until there are a few items per SIMD
    int64_t nstrips = 0;
    for (int s = 0; s < n_owned; ++s) nstrips += patches[s].nelu;
    // (one wave per SIMD = 1024 items in flight on MI355X: the launch runs in rounds, and a last round that is half empty costs as much as a full one -- among the
    //  cuts that give at least ~4 items per SIMD the one with the fullest rounds is taken, e.g. one GPU's eighth of C4, 1 536 strips: 2 x 1 536 = 3 full rounds
    //  rather than 3 x 1 536 = 4.5)
    int cut = 1;
    if (seg_len <= 0) {
        const int64_t slots = 1024, ns = std::max<int64_t>(nstrips, 1);
        const int c0 = int(std::min<int64_t>(64, (2 * slots + ns - 1) / ns)), c1 = int(std::min<int64_t>(64, (4 * slots + ns - 1) / ns));
        double best = -1.0;
        for (int c = c0; c <= std::max(c0, c1); ++c) {
            const int64_t items = ns * c, rounds = (items + slots - 1) / slots;
            const double eff = double(items) / double(rounds * slots) - 0.01 * (c - c0);          // fuller rounds first, fewer segments among equals
            if (eff > best) { best = eff; cut = c; }
        }
    }
    for (int s = 0; s < n_owned; ++s) {
        const PatchDev& P = patches[s];
        int nseg = seg_len > 0 ? std::max(1, (P.nelv + seg_len / 2) / std::max(seg_len, P1)) : cut;
        nseg = std::max(nseg, (P.nelv + 254) / 255);                   // the kernel keeps an item's span indices in LDS: <= 255 elements per item
        while (nseg > 1 && P.nelv / nseg < P1) --nseg;                 // every segment holds at least p + 1 elements: a pair lies in at most two
        RecPatch& R = rec_patch[s];
        R.item_off = (int)rec_items.size(); R.nseg = nseg;
        R.seg_of = (int)ints.size(); ints.resize(ints.size() + P.nelv);
        R.ev0_of = (int)ints.size(); ints.resize(ints.size() + nseg + 1);
        for (int g = 0; g <= nseg; ++g) ints[R.ev0_of + g] = int(int64_t(g) * P.nelv / nseg);
        for (int g = 0; g < nseg; ++g) for (int ev = ints[R.ev0_of + g]; ev < ints[R.ev0_of + g + 1]; ++ev) ints[R.seg_of + ev] = g;
        int rat = 0;                                                   // WalkItem::pad: non-constant weights (the walking kernel's polynomial instance needs W,alpha = 0)
        for (int64_t a = P.cp_off + 1; a < P.cp_off + int64_t(P.nu) * P.nv; ++a) if (weights[a] != weights[P.cp_off]) { rat = 1; break; }
        for (int eu = 0; eu < P.nelu; ++eu) for (int g = 0; g < nseg; ++g) {
            const int e0 = ints[R.ev0_of + g], e1 = ints[R.ev0_of + g + 1];
            rec_items.push_back({s, eu, e0, e1 - e0, g, 0, ints[P.spu + eu] - P.p, rat});
            rec_rows = std::max(rec_rows, (ints[P.spv + e1 - 1] + 1) - (ints[P.spv + e0] - P.q));     // rows first .. last of the item's windows
        }
    }
    rec_order.clear();
    for (int k = 0; k < (int)rec_items.size(); ++k) if (!rec_items[k].pad) rec_order.push_back(k);
    rec_npoly = (int)rec_order.size();
    for (int k = 0; k < (int)rec_items.size(); ++k) if (rec_items[k].pad) rec_order.push_back(k);
    if (degree == 4) rec_cp4.assign(total_cp, RecCp4{}); else rec_cp.assign(total_cp, RecCp{});
    for (int s = 0; s < n_owned; ++s) {
        const PatchDev& P = patches[s];
        const RecPatch& R = rec_patch[s];
        const int* c2v = &ints[P.c2v];
        for (int64_t a = P.cp_off; a < P.cp_off + int64_t(P.nu) * P.nv; ++a) {
            const CpDesc& c = cp_desc[a];
            const int zf = (zero[3 * a] ? 1 : 0) | (zero[3 * a + 1] ? 2 : 0) | (zero[3 * a + 2] ? 4 : 0);
            int nit = 0, rows_[10]; unsigned infos_[10];
            const int lov_a = c2v[2 * c.ja], hiv_a = c2v[2 * c.ja + 1];
            if (lov_a <= hiv_a) {
                const int g_lo = ints[R.seg_of + lov_a], g_hi = ints[R.seg_of + hiv_a];
                if (c.neu * (g_hi - g_lo + 1) > MAXIT) throw std::runtime_error("build_rec: a control point lies in more than " + std::to_string(MAXIT) + " work items");
                for (int k = 0; k < c.neu; ++k) for (int g = g_lo; g <= g_hi; ++g) {
                    const int E0 = ints[R.ev0_of + g], E1 = ints[R.ev0_of + g + 1], ivf = ints[P.spv + E0] - P.q;
                    unsigned pm = 0;
                    for (int d = 0; d < 2 * PW + 1; ++d) {
                        const int jb = c.ja - PW + d;
                        if (jb < 0 || jb >= P.nv) continue;
                        const int lo = std::max(lov_a, c2v[2 * jb]), hi = std::min(hiv_a, c2v[2 * jb + 1]);
                        if (lo <= hi && lo < E1 && hi >= E0) pm |= 1u << d;
                    }
                    const int item = R.item_off + (c.eu0 + k) * R.nseg + g;
                    rows_[nit] = item * rec_rows + (c.ja - ivf);
                    infos_[nit] = unsigned(c.bu[k] - c.i0) | unsigned(c.ia - c.bu[k]) << 8 | pm << 16;
                    ++nit;
                }
            }
            if (degree == 4) { RecCp4& rc = rec_cp4[a]; rc.flags = zf; rc.nit = nit; for (int q = 0; q < nit; ++q) { rc.it[q].row = rows_[q]; rc.it[q].info = infos_[q]; } }
            else { RecCp& rc = rec_cp[a]; rc.flags = zf; rc.nit = nit; for (int q = 0; q < nit; ++q) { rc.it[q].row = rows_[q]; rc.it[q].info = infos_[q]; } }
        }
    }
}

}  // namespace gf
