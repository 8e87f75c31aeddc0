// gf_nd_symbolic.hpp -- symbolic phase of the nested-dissection multifrontal factorisation on the host (native counterpart of goldfish_amd/_nd.py, which stays as the
// readable statement the tests compare this against, entry by entry).
//
// Recursive coordinate bisection of the control-point graph with vertex separators: a region is sorted along the longest axis of its bounding box (ties: control-point
// index), cut at the rank within `cut_window` of the median that gives the smallest separator {v in A coupled to B} (a median cut of a patch grid falls on patch interfaces,
// where the penalty coupling reaches one control-point row further: _nd.py: _best_cuts), the separator becomes the tree node, the two sides recurse (the two halves of a
// large region on two threads).  Then: fronts in post-order, elimination order inside a front along the first coordinate, boundaries bottom-up (later-eliminated neighbours
// of the front's control points + the children's boundaries), positions of the boundary control points in the parent front (the extend-add map).
// Replaces nothing of the reference directly: GOLDFISH hands K to MUMPS, whose analysis phase (orderings by METIS / SCOTCH / AMD) is the counterpart
// (GOLDFISH/utils/opt_utils.py:156-209).  Host code only: no HIP call in this file.
#pragma once
#include <algorithm>
#include <atomic>
#include <memory>
#include <cstdint>
#include <functional>
#include <future>
#include <map>
#include <stdexcept>
#include <vector>

namespace gfnd {

struct Symbolic {
    std::vector<int64_t> elim, elim_off, bnd, bnd_off, parent, order, front_of, pmap;
    int64_t nfronts = 0;
};

struct Dissector {
    int64_t ncp; const int64_t* nb_ptr; const int32_t* nb; const double* X; int dim; int64_t leaf; double window;
    std::vector<int64_t> node_of;          // tree node (heap numbering: children of r are 2 r + 1, 2 r + 2) that owns the control point
    // node of the region the control point currently lies in (-1: it has its final node).  A region's thread reads the tags of ITS vertices' neighbours, which may
    // belong to a region another thread is splitting: relaxed atomics (a foreign tag never equals the reader's node, whatever moment it is read at)
    std::unique_ptr<std::atomic<int64_t>[]> tag;
    std::vector<int32_t> pos;              // rank of the control point in its region's order along the region's axis
    int max_threads = 1;

    void split(int32_t* v, int64_t n, int64_t node, int depth) {
        if (n <= 0) return;
        if (n <= leaf) { for (int64_t k = 0; k < n; ++k) { node_of[v[k]] = node; tag[v[k]].store(-1, std::memory_order_relaxed); } return; }
        // longest axis of the bounding box (the first of equal ones), order along it (ties: control-point index)
        int axis = 0; double best = -1.0;
        for (int d = 0; d < dim; ++d) {
            double lo = X[(size_t)v[0] * dim + d], hi = lo;
            for (int64_t k = 1; k < n; ++k) { const double x = X[(size_t)v[k] * dim + d]; lo = std::min(lo, x); hi = std::max(hi, x); }
            if (hi - lo > best) { best = hi - lo; axis = d; }
        }
        std::sort(v, v + n, [&](int32_t a, int32_t b) { const double xa = X[(size_t)a * dim + axis], xb = X[(size_t)b * dim + axis]; return xa < xb || (xa == xb && a < b); });
        for (int64_t k = 0; k < n; ++k) { pos[v[k]] = (int32_t)k; tag[v[k]].store(node, std::memory_order_relaxed); }
        const int64_t cut0 = (n + 1) / 2;
        int64_t w = (int64_t)(window * (double)n); if (w < 0) w = 0;
        int64_t tmin = std::max<int64_t>(cut0 - w, 1), tmax = std::min<int64_t>(cut0 + w, n - 1);
        tmax = std::max(tmax, tmin);
        // M(v) = the largest rank among v's neighbours in the region, for the ranks below the last candidate cut
        std::vector<int32_t> M((size_t)tmax);
        for (int64_t k = 0; k < tmax; ++k) {
            const int32_t a = v[k]; int32_t m = (int32_t)k;
            for (int64_t e = nb_ptr[a]; e < nb_ptr[a + 1]; ++e) { const int32_t b = nb[e]; if (tag[b].load(std::memory_order_relaxed) == node && pos[b] > m) m = pos[b]; }
            M[(size_t)k] = m;
        }
        int64_t cut = cut0;
        if (tmax > tmin) {                 // separator size of the cut t (A = ranks < t): #{v: rank(v) < t <= M(v)}, for every t of the window from one difference array
            std::vector<int64_t> diff((size_t)(tmax - tmin + 2), 0);
            for (int64_t k = 0; k < tmax; ++k) {
                const int64_t m = M[(size_t)k];
                const int64_t t0 = std::max(k + 1, tmin), t1 = std::min(m, tmax);
                if (t1 >= t0) { ++diff[(size_t)(t0 - tmin)]; --diff[(size_t)(t1 - tmin + 1)]; }
            }
            int64_t run = 0, bsz = -1, bdist = 0;
            for (int64_t t = tmin; t <= tmax; ++t) {
                run += diff[(size_t)(t - tmin)];
                const int64_t dist = t > cut0 ? t - cut0 : cut0 - t;
                if (bsz < 0 || run < bsz || (run == bsz && dist < bdist)) { bsz = run; bdist = dist; cut = t; }      // ties: the cut nearest to the median (the lower one of two)
            }
        }
        // separator -> this node; the rest of A and B keep going
        std::vector<int32_t> A, B;
        A.reserve((size_t)cut); B.reserve((size_t)(n - cut));
        for (int64_t k = 0; k < n; ++k) {
            const int32_t a = v[k];
            if (k < cut && M[(size_t)k] >= cut) { node_of[a] = node; tag[a].store(-1, std::memory_order_relaxed); }
            else (k < cut ? A : B).push_back(a);
        }
        std::vector<int32_t>().swap(M);
        const bool fork = depth < 6 && (int64_t)A.size() > 20000 && (int64_t)B.size() > 20000 && (1 << depth) < max_threads;
        if (fork) {
            auto fut = std::async(std::launch::async, [&] { split(A.data(), (int64_t)A.size(), 2 * node + 1, depth + 1); });
            split(B.data(), (int64_t)B.size(), 2 * node + 2, depth + 1);
            fut.get();
        } else {
            split(A.data(), (int64_t)A.size(), 2 * node + 1, depth + 1);
            split(B.data(), (int64_t)B.size(), 2 * node + 2, depth + 1);
        }
    }
};

inline Symbolic nested_dissection(int64_t ncp, const int64_t* nb_ptr, const int32_t* nb, const double* X, int dim, int64_t leaf, double window, int threads) {
    if (ncp <= 0 || dim <= 0 || leaf <= 0) throw std::runtime_error("gfs_symbolic_create: empty graph, no coordinates or leaf <= 0");
    if (ncp >= (int64_t)1 << 31) throw std::runtime_error("gfs_symbolic_create: more than 2^31 control points");
    Dissector D; D.ncp = ncp; D.nb_ptr = nb_ptr; D.nb = nb; D.X = X; D.dim = dim; D.leaf = leaf; D.window = window;
    D.node_of.assign((size_t)ncp, -1); D.tag.reset(new std::atomic<int64_t>[(size_t)ncp]); D.pos.assign((size_t)ncp, 0);
    for (int64_t a = 0; a < ncp; ++a) D.tag[(size_t)a].store(0, std::memory_order_relaxed);
    D.max_threads = std::max(1, threads);
    {
        std::vector<int32_t> all((size_t)ncp);
        for (int64_t a = 0; a < ncp; ++a) all[(size_t)a] = (int32_t)a;
        D.split(all.data(), ncp, 0, 0);
    }
    // tree nodes that own control points, parents through the heap numbering, post-order (children ascending)
    std::vector<int64_t> nodes(D.node_of);
    std::sort(nodes.begin(), nodes.end()); nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
    auto present = [&](int64_t r) { return std::binary_search(nodes.begin(), nodes.end(), r); };
    auto parent_of = [&](int64_t r) -> int64_t {
        while (r > 0) { r = (r - 1) / 2; if (present(r)) return r; }
        return -1;
    };
    std::map<int64_t, std::vector<int64_t>> children;
    std::vector<int64_t> roots;
    for (int64_t r : nodes) { const int64_t p = r > 0 ? parent_of(r) : -1; if (p >= 0) children[p].push_back(r); else roots.push_back(r); }
    std::vector<int64_t> post; post.reserve(nodes.size());
    {
        std::vector<std::pair<int64_t, bool>> stack;
        for (auto it = roots.rbegin(); it != roots.rend(); ++it) stack.push_back({*it, false});
        while (!stack.empty()) {
            auto [r, seen] = stack.back(); stack.pop_back();
            if (seen) { post.push_back(r); continue; }
            stack.push_back({r, true});
            auto c = children.find(r);
            if (c != children.end()) for (auto it = c->second.rbegin(); it != c->second.rend(); ++it) stack.push_back({*it, false});
        }
    }
    const int64_t nf = (int64_t)post.size();
    std::map<int64_t, int64_t> index;
    for (int64_t i = 0; i < nf; ++i) index[post[(size_t)i]] = i;
    Symbolic S; S.nfronts = nf;
    S.front_of.resize((size_t)ncp); S.parent.assign((size_t)nf, -1);
    {
        std::vector<int64_t> idx_of(nodes.size());
        for (size_t k = 0; k < nodes.size(); ++k) idx_of[k] = index[nodes[k]];
        for (int64_t a = 0; a < ncp; ++a) S.front_of[(size_t)a] = idx_of[(size_t)(std::lower_bound(nodes.begin(), nodes.end(), D.node_of[(size_t)a]) - nodes.begin())];
    }
    for (int64_t i = 0; i < nf; ++i) { const int64_t r = post[(size_t)i]; const int64_t p = r > 0 ? parent_of(r) : -1; S.parent[(size_t)i] = p >= 0 ? index[p] : -1; }
    // elimination order: fronts in post-order, inside a front along the first coordinate (ties: control-point index)
    S.elim.resize((size_t)ncp);
    for (int64_t a = 0; a < ncp; ++a) S.elim[(size_t)a] = a;
    std::sort(S.elim.begin(), S.elim.end(), [&](int64_t a, int64_t b) {
        const int64_t fa = S.front_of[(size_t)a], fb = S.front_of[(size_t)b];
        if (fa != fb) return fa < fb;
        const double xa = X[(size_t)a * dim], xb = X[(size_t)b * dim];
        return xa < xb || (xa == xb && a < b);
    });
    S.order.resize((size_t)ncp);
    for (int64_t q = 0; q < ncp; ++q) S.order[(size_t)S.elim[(size_t)q]] = q;
    S.elim_off.assign((size_t)nf + 1, 0);
    for (int64_t a = 0; a < ncp; ++a) ++S.elim_off[(size_t)S.front_of[(size_t)a] + 1];
    for (int64_t t = 0; t < nf; ++t) S.elim_off[(size_t)t + 1] += S.elim_off[(size_t)t];
    // boundaries, bottom-up: (neighbours of the front's control points + the children's boundaries) eliminated behind the front's subtree; as elimination positions, ascending
    std::vector<std::vector<int64_t>> bnds((size_t)nf), kids((size_t)nf);
    for (int64_t t = 0; t < nf; ++t) if (S.parent[(size_t)t] >= 0) kids[(size_t)S.parent[(size_t)t]].push_back(t);
    for (int64_t t = 0; t < nf; ++t) {
        const int64_t hi = S.elim_off[(size_t)t + 1];              // post-order: the subtree of t ends with t itself
        std::vector<int64_t>& b = bnds[(size_t)t];
        for (int64_t q = S.elim_off[(size_t)t]; q < hi; ++q) {
            const int64_t a = S.elim[(size_t)q];
            for (int64_t e = nb_ptr[a]; e < nb_ptr[a + 1]; ++e) { const int64_t o = S.order[(size_t)nb[e]]; if (o >= hi) b.push_back(o); }
        }
        for (int64_t c : kids[(size_t)t]) for (int64_t o : bnds[(size_t)c]) if (o >= hi) b.push_back(o);
        std::sort(b.begin(), b.end()); b.erase(std::unique(b.begin(), b.end()), b.end());
    }
    S.bnd_off.assign((size_t)nf + 1, 0);
    for (int64_t t = 0; t < nf; ++t) S.bnd_off[(size_t)t + 1] = S.bnd_off[(size_t)t] + (int64_t)bnds[(size_t)t].size();
    S.bnd.resize((size_t)S.bnd_off[(size_t)nf]); S.pmap.assign(S.bnd.size(), 0);
    for (int64_t t = 0; t < nf; ++t) {
        const std::vector<int64_t>& b = bnds[(size_t)t];
        const int64_t p = S.parent[(size_t)t];
        for (size_t k = 0; k < b.size(); ++k) {
            S.bnd[(size_t)S.bnd_off[(size_t)t] + k] = S.elim[(size_t)b[k]];
            if (p < 0) continue;
            // position in the parent front: among its eliminated control points, or behind them in its boundary list
            const int64_t e0 = S.elim_off[(size_t)p], e1 = S.elim_off[(size_t)p + 1];
            if (b[k] >= e0 && b[k] < e1) S.pmap[(size_t)S.bnd_off[(size_t)t] + k] = b[k] - e0;
            else {
                const std::vector<int64_t>& pb = bnds[(size_t)p];
                S.pmap[(size_t)S.bnd_off[(size_t)t] + k] = (e1 - e0) + (int64_t)(std::lower_bound(pb.begin(), pb.end(), b[k]) - pb.begin());
            }
        }
    }
    return S;
}

}  // namespace gfnd
