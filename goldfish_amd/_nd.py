"""Nested-dissection ordering and the symbolic phase of the multifrontal factorisation of K (host side, NumPy).

The device solver's skyline (goldfish_amd/csrc/gf_solver.hip) costs n x bandwidth of memory and n x bandwidth^2 of work; for a shell --
a 2-D manifold of control points -- nested dissection brings that to O(n log n) and O(n^1.5): the control points are split recursively
by coordinate bisection (the physical control points are at hand), the separator of a split is the set of control points of one half
that are coupled to the other half (three control-point rows thick inside a patch, the penalty's 2 (p + 1) rows across an interface),
and every tree node becomes a dense FRONT: the control points it eliminates followed by its boundary (the not yet eliminated control
points its subtree couples to, all of them on ancestor separators).  The numeric phase (gf_solver.hip, ``gfs_create_nd``) factors the
fronts in post-order on the 64 x 64 FP64-MFMA tile kernels of the skyline solver and passes Schur complements up (extend-add).

Replaces, for large models, the MUMPS ordering / analysis phase behind GOLDFISH/utils/opt_utils.py:156-209 (solve_Ax_b / solve_ATx_b).
"""
import numpy as np


class Symbolic:
    """Fronts in post-order.  Per front t: ``elim[elim_off[t]:elim_off[t+1]]`` the control points it eliminates (in elimination order),
    ``bnd[bnd_off[t]:bnd_off[t+1]]`` its boundary control points (ascending elimination order), ``parent[t]`` (-1: root).
    ``order[cp]`` = position of a control point in the global elimination order, ``front_of[cp]`` = the front that eliminates it."""

    def __init__(self, elim, elim_off, bnd, bnd_off, parent, order, front_of):
        self.elim, self.elim_off, self.bnd, self.bnd_off = elim, elim_off, bnd, bnd_off
        self.parent, self.order, self.front_of = parent, order, front_of
        self.nfronts = parent.size

    def front_dofs(self, nb=64):
        """Per front: eliminated dofs padded to tiles, boundary dofs padded to tiles, block counts."""
        ne = 3 * np.diff(self.elim_off)
        nbd = 3 * np.diff(self.bnd_off)
        nblk_e = (ne + nb - 1) // nb
        nblk_b = (nbd + nb - 1) // nb
        return ne, nbd, nblk_e.astype(np.int64), nblk_b.astype(np.int64)

    def stats(self, nb=64):
        ne, nbd, be, bb = self.front_dofs(nb)
        bt = be + bb
        tiles = bt * (bt + 1) // 2
        # partial factorisation of a dense front: sum over eliminated block columns k of (rows below) + (rows below)(rows below + 1) / 2 tile products
        flops = 0.0
        for e, t in zip(be, bt):
            k = np.arange(e)
            r = t - 1 - k
            flops += 2.0 * nb ** 3 * float(np.sum(r + r * (r + 1) / 2.0) + e / 3.0)
        return dict(fronts=int(self.nfronts), tiles=int(tiles.sum()), bytes=int(tiles.sum()) * nb * nb * 8, flops=flops,
                    largest_front_dofs=int((nb * bt).max()), eliminated_block_columns=int(be.sum()))


def _best_cuts(ncp, act, seg, rank, start, count, cut0, rows, cols, window):
    """Per region the cut rank within ``window`` (fraction of the region's size) of the median that gives the SMALLEST separator (the control points of A coupled to
    B).  A median cut of a patch grid falls on patch interfaces, where the penalty coupling reaches p + 1 control-point rows instead of the p rows inside a patch:
    a separator there is a third wider, and the work of a front grows with the cube of that.  With M(v) = the largest rank among v's neighbours in its region, the
    separator of a cut t is {v: rank(v) < t <= M(v)} -- its size for every t of the window comes from one difference array per level.  Only the edges that cross the
    window matter (two byte-flag gathers over the edge list select them)."""
    w = np.maximum((window * count).astype(np.int64), 0)
    tmin, tmax = np.maximum(cut0 - w, 1), np.minimum(cut0 + w, count - 1)
    tmax = np.maximum(tmax, tmin)
    flag = np.zeros(ncp, np.int8)
    flag[act] = (rank < tmax[seg]).astype(np.int8) | ((rank >= tmin[seg]).astype(np.int8) << 1)
    fr, fc = flag[rows], flag[cols]
    sel = ((fr & 1) & (fc >> 1) | (fc & 1) & (fr >> 1)).astype(bool)
    gpos = np.zeros(ncp, np.int64)
    gpos[act] = np.arange(act.size)                                     # position in the (region, coordinate) order: rank + start of the region
    p1, p2 = gpos[rows[sel]], gpos[cols[sel]]
    lo, hi = np.minimum(p1, p2), np.maximum(p1, p2)
    M = np.arange(act.size, dtype=np.int64)
    o = np.argsort(lo, kind="stable")
    lo, hi = lo[o], hi[o]
    if lo.size:
        first = np.flatnonzero(np.concatenate([[True], lo[1:] != lo[:-1]]))
        M[lo[first]] = np.maximum(np.maximum.reduceat(hi, first), lo[first])
    # separator size of the cut at global position g (A = positions < g): #{v: v < g <= M(v)}
    diff = np.zeros(act.size + 2, np.int64)
    has = M > np.arange(act.size)
    np.add.at(diff, np.flatnonzero(has) + 1, 1)
    np.add.at(diff, M[has] + 1, -1)
    size = np.cumsum(diff)[:act.size + 1]
    cut = cut0.copy()
    for r in range(count.size):                                        # regions of one level: at most a few thousand
        a, b = start[r] + tmin[r], start[r] + tmax[r]
        if b <= a:
            continue
        sz = size[a:b + 1]
        best = np.flatnonzero(sz == sz.min())
        cut[r] = tmin[r] + best[np.argmin(np.abs(best + tmin[r] - cut0[r]))]       # ties: the cut nearest to the median
    return cut


def nested_dissection_native(nb_ptr, nb, coords, leaf=192, cut_window=0.04, threads=None):
    """The same symbolic phase in the solver library (csrc/gf_nd_symbolic.hpp: host C++, the halves of the large regions on threads): identical result, a fraction of
    the time (C4: seconds of NumPy passes over 66 M edges).  Returns (Symbolic, pmap): the extend-add map comes with it."""
    import ctypes as C
    import os
    from ._solver import lib
    L = lib()
    i64p, vp = C.POINTER(C.c_int64), C.c_void_p
    if not getattr(L, "_gf_symbolic_bound", False):
        L.gfs_symbolic_create.argtypes = [C.c_int64, i64p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int, C.c_int64, C.c_double, C.c_int, C.POINTER(vp)]
        L.gfs_symbolic_sizes.argtypes = [vp, i64p, i64p]
        L.gfs_symbolic_sizes.restype = None
        L.gfs_symbolic_copy.argtypes = [vp] + [i64p] * 8
        L.gfs_symbolic_copy.restype = None
        L.gfs_symbolic_destroy.argtypes = [vp]
        L.gfs_symbolic_destroy.restype = None
        L._gf_symbolic_bound = True
    nb_ptr = np.ascontiguousarray(nb_ptr, np.int64)
    nb = np.ascontiguousarray(nb, np.int32)
    ncp = nb_ptr.size - 1
    X = np.ascontiguousarray(np.asarray(coords, float).reshape(ncp, -1))
    if threads is None:
        threads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    h = vp()
    if L.gfs_symbolic_create(ncp, nb_ptr.ctypes.data_as(i64p), nb.ctypes.data_as(C.POINTER(C.c_int32)), X.ctypes.data_as(C.POINTER(C.c_double)), X.shape[1], int(leaf),
                             float(cut_window), int(threads), C.byref(h)):
        raise RuntimeError(L.gfs_last_error().decode())
    try:
        nf, nbnd = C.c_int64(0), C.c_int64(0)
        L.gfs_symbolic_sizes(h, C.byref(nf), C.byref(nbnd))
        nf, nbnd = nf.value, nbnd.value
        elim, elim_off, bnd, bnd_off = np.empty(ncp, np.int64), np.empty(nf + 1, np.int64), np.empty(nbnd, np.int64), np.empty(nf + 1, np.int64)
        parent, order, front_of, pmap = np.empty(nf, np.int64), np.empty(ncp, np.int64), np.empty(ncp, np.int64), np.empty(nbnd, np.int64)
        L.gfs_symbolic_copy(h, *[a.ctypes.data_as(i64p) for a in (elim, elim_off, bnd, bnd_off, parent, order, front_of, pmap)])
    finally:
        L.gfs_symbolic_destroy(h)
    return Symbolic(elim, elim_off, bnd, bnd_off, parent, order, front_of), pmap


def nested_dissection(nb_ptr, nb, coords, leaf=192, cut_window=0.04):
    """Recursive coordinate bisection of the control-point graph (nb_ptr, nb: neighbour lists incl. the control point itself) with
    vertex separators; ``leaf``: regions of at most that many control points are not split further.  Level-synchronous: every pass
    splits all regions of the current level with array operations over the edge list.  ``cut_window``: the cut of a region is the one with the
    smallest separator among the ranks within that fraction of the region's size around the median (0: the median itself)."""
    ncp = nb_ptr.size - 1
    X = np.asarray(coords, float).reshape(ncp, -1)
    rows = np.repeat(np.arange(ncp, dtype=np.int32), np.diff(nb_ptr))      # 32-bit indices: the passes below are gathers over the edge list (66 M edges at C4)
    cols = np.asarray(nb, np.int32)
    keep = rows < cols                                                    # the pattern is symmetric: one direction per edge, both tested below
    rows, cols = rows[keep], cols[keep]
    region = np.zeros(ncp, np.int64)              # tree node (heap numbering: children of r are 2 r + 1, 2 r + 2) a control point currently lies in
    done = np.zeros(ncp, bool)                    # True: the control point has its final tree node (a separator or a leaf)
    node_of = np.full(ncp, -1, np.int64)
    while True:
        act = np.flatnonzero(~done)
        if act.size == 0:
            break
        reg = region[act]
        order = np.argsort(reg, kind="stable")
        act, reg = act[order], reg[order]
        uniq, start, count = np.unique(reg, return_index=True, return_counts=True)
        small = count <= leaf
        # regions small enough become leaves
        seg_small = np.repeat(small, count)
        node_of[act[seg_small]] = reg[seg_small]
        done[act[seg_small]] = True
        if small.all():
            break
        act, reg = act[~seg_small], reg[~seg_small]
        uniq, start, count = uniq[~small], None, count[~small]
        start = np.concatenate([[0], np.cumsum(count)[:-1]])
        # longest axis of every region, median split along it
        seg = np.repeat(np.arange(uniq.size), count)
        ext = np.stack([np.maximum.reduceat(X[act, d], start) - np.minimum.reduceat(X[act, d], start) for d in range(X.shape[1])], 1)
        axis = ext.argmax(1)
        val = X[act, axis[seg]]
        o2 = np.lexsort((val, seg))
        act, seg = act[o2], seg[o2]
        rank = np.arange(act.size) - start[seg]
        cut = (count + 1) // 2                                         # per region: ranks below the cut form A, the others B
        if cut_window > 0.0:
            cut = _best_cuts(ncp, act, seg, rank, start, count, cut, rows, cols, cut_window)
        side = (rank >= cut[seg])                                      # False: first part (A), True: second part (B)
        side_of = np.zeros(ncp, np.int8)
        side_of[act] = side.astype(np.int8) + 1                        # 1: A, 2: B, 0: not in a region that is being split
        # separator: control points of A coupled to a control point of B of the same region.  The edge list only holds edges whose ends are both
        # still active and in the same region (pruned at the end of every pass), so the test is on the sides alone
        sr, sc = side_of[rows], side_of[cols]
        sep = np.zeros(ncp, bool)
        sep[rows[(sr == 1) & (sc == 2)]] = True
        sep[cols[(sr == 2) & (sc == 1)]] = True
        is_sep = sep[act]
        node_of[act[is_sep]] = uniq[seg[is_sep]]
        done[act[is_sep]] = True
        rest = ~is_sep
        region[act[rest]] = 2 * uniq[seg[rest]] + 1 + side[rest]
        live = ~(done[rows] | done[cols])
        rows, cols = rows[live], cols[live]
        live = region[rows] == region[cols]
        rows, cols = rows[live], cols[live]
    # post-order of the tree nodes that own control points; heap numbering gives parents by (r - 1) // 2
    nodes = np.unique(node_of)
    present = set(int(r) for r in nodes)

    def parent_of(r):
        r = (r - 1) // 2
        while r >= 0 and r not in present:
            r = (r - 1) // 2 if r > 0 else -1
        return r
    children = {int(r): [] for r in nodes}
    roots = []
    for r in nodes:
        p = parent_of(int(r)) if r > 0 else -1
        (children[p] if p >= 0 else roots).append(int(r))
    post, stack = [], [(r, False) for r in reversed(roots)]
    while stack:
        r, seen = stack.pop()
        if seen:
            post.append(r)
        else:
            stack.append((r, True))
            for c in reversed(children[r]):
                stack.append((c, False))
    index = {r: i for i, r in enumerate(post)}
    front_of = np.array([index[int(r)] for r in nodes])[np.searchsorted(nodes, node_of)].astype(np.int64)
    nf = len(post)
    parent = np.array([index[p] if (p := (parent_of(r) if r > 0 else -1)) >= 0 else -1 for r in post], np.int64)
    # elimination order: fronts in post-order, inside a front along the first principal coordinate (locality only)
    key = np.lexsort((X[:, 0], front_of))
    order = np.empty(ncp, np.int64)
    order[key] = np.arange(ncp)
    elim = key.astype(np.int64)
    elim_off = np.concatenate([[0], np.cumsum(np.bincount(front_of, minlength=nf))]).astype(np.int64)
    # boundaries, bottom-up: bnd(t) = (neighbours of elim(t) + boundaries of the children) not eliminated in the subtree of t
    hi = elim_off[1:].copy()                      # post-order: the subtree of t ends with t itself -> everything with order >= elim_off[t + 1] is outside
    bnds = [None] * nf
    kids = [[] for _ in range(nf)]
    for t in range(nf):
        if parent[t] >= 0:
            kids[parent[t]].append(t)
    nb_ptr = np.asarray(nb_ptr, np.int64)
    nbl = np.asarray(nb, np.int64)
    # the later-eliminated neighbours of every front's own control points, for all fronts at once: edges (a, b) with order[b] beyond the subtree of front_of[a]
    thr = hi[front_of]                                                     # per control point: first elimination position outside its front's subtree
    later = order[nbl] >= np.repeat(thr, np.diff(nb_ptr))
    ea = np.repeat(np.arange(ncp, dtype=np.int64), np.diff(nb_ptr))[later]
    key = np.unique(front_of[ea] * np.int64(ncp) + order[nbl[later]])      # (front, elimination position of the neighbour): sorted, duplicates dropped
    ef, eb = key // ncp, elim[key % ncp]
    own_off = np.concatenate([[0], np.cumsum(np.bincount(ef, minlength=nf))]).astype(np.int64)
    for t in range(nf):
        cand = eb[own_off[t]:own_off[t + 1]]                    # sorted by elimination order, unique
        if kids[t]:
            cand = np.unique(np.concatenate([cand] + [bnds[c] for c in kids[t]]))
            cand = cand[order[cand] >= hi[t]]
            cand = cand[np.argsort(order[cand], kind="stable")]
        bnds[t] = cand
    bnd_off = np.concatenate([[0], np.cumsum([b.size for b in bnds])]).astype(np.int64)
    bnd = np.concatenate(bnds).astype(np.int64) if nf else np.zeros(0, np.int64)
    return Symbolic(elim, elim_off, bnd, bnd_off, parent, order, front_of)


def multifrontal_reference_solve(sym, K, b):
    """Dense NumPy statement of the numeric phase (tests only: tiny models): factor the fronts in post-order, extend-add the Schur
    complements, forward / backward substitution -- what gf_solver.hip does on tiles.  K: scipy CSR over dofs (3 per control point)."""
    K = K.tocsr()
    nf = sym.nfronts
    fr = []
    upd = [None] * nf
    for t in range(nf):
        e, bd = sym.elim[sym.elim_off[t]:sym.elim_off[t + 1]], sym.bnd[sym.bnd_off[t]:sym.bnd_off[t + 1]]
        dof = np.concatenate([3 * e[:, None] + np.arange(3), 3 * bd[:, None] + np.arange(3)]).ravel() if e.size + bd.size else np.zeros(0, np.int64)
        ne = 3 * e.size
        F = np.zeros((dof.size, dof.size))
        Ke = K[dof[:ne]][:, dof].toarray()
        F[:ne, :] = Ke
        F[:, :ne] = Ke.T
        pos = {int(d): i for i, d in enumerate(dof)}
        for c in range(nf):
            if sym.parent[c] == t and upd[c] is not None:
                cd, S = upd[c]
                ii = np.array([pos[int(d)] for d in cd], np.int64)
                F[np.ix_(ii, ii)] += S
        A11, A21 = F[:ne, :ne], F[ne:, :ne]
        X = np.linalg.solve(A11, A21.T)
        upd[t] = (dof[ne:], F[ne:, ne:] - A21 @ X)
        fr.append((dof, ne, A11, A21))
    y = np.asarray(b, float).copy()
    z = np.zeros_like(y)
    for t in range(nf):                                         # forward
        dof, ne, A11, A21 = fr[t]
        z[dof[:ne]] = np.linalg.solve(A11, y[dof[:ne]])
        y[dof[ne:]] -= A21 @ z[dof[:ne]]
    x = np.zeros_like(y)
    for t in reversed(range(nf)):                               # backward
        dof, ne, A11, A21 = fr[t]
        x[dof[:ne]] = z[dof[:ne]] - np.linalg.solve(A11, A21.T @ x[dof[ne:]])
    return x
