"""CPU tests of the host-side logic and of the C-ABI surface (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from goldfish_amd import geometry as G
from goldfish_amd import sharding
from goldfish_amd.model import Interface, arrays_from_spec
from goldfish_amd.splines import NURBSPatch, open_uniform_knots

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cylinder_sector_is_exact_and_refinement_preserves_geometry():
    s = NURBSPatch.cylinder_sector(25.0, np.radians(50), np.radians(130), 0.0, 50.0, 1, 1, 3)
    f = NURBSPatch.cylinder_sector(25.0, np.radians(50), np.radians(130), 0.0, 50.0, 7, 5, 3)
    rng = np.random.default_rng(0)
    for xi in rng.uniform(0, 1, (20, 2)):
        X, Y = s.eval(xi), f.eval(xi)
        assert abs(np.hypot(X[0], X[1]) - 25.0) < 1e-12
        assert np.abs(X - Y).max() < 1e-12
    assert f.n_u == 7 + 3 and f.n_v == 5 + 3 and f.nel == (7, 5)
    assert np.ptp(f.control[:, :, 3]) > 1e-3          # truly rational


def test_bilinear_patch_and_inversion():
    s = NURBSPatch.bilinear([[0, 0, 0], [2, 0, 0], [0, 3, 1], [2, 3, 1]], 4, 5, 3)
    xi = np.array([0.3, 0.7])
    X = s.eval(xi)
    assert np.allclose(X, [0.6, 2.1, 0.7], atol=1e-13)
    assert np.allclose(s.invert(X, (0.5, 0.5)), xi, atol=1e-10)
    assert s.get_side_dofs(1, 0, 1) == list(range(s.n_u))


def test_interface_weights_and_tangent():
    itf = Interface.from_endpoints(0, 1, [[1.0, 0.0], [1.0, 1.0]], [[0.0, 0.0], [0.0, 1.0]], 10)
    assert itf.npts == 11 and abs(itf.wt.sum() - 1.0) < 1e-14
    assert np.allclose(itf.tau, [[0.0, 1.0]] * 11)


def test_model_arrays_layout():
    spec = G.tbeam_2patch(4)
    A = arrays_from_spec(spec)
    assert A.total_cp == sum(p.ncp for p in spec.patches) and A.ndof == 3 * A.total_cp
    assert A.n_gauss_points == sum(p.nel[0] * p.nel[1] * 16 for p in spec.patches)
    p0 = spec.patches[0]
    assert np.allclose(A.cp_hom[0][:p0.ncp].reshape(p0.n_v, p0.n_u).T, p0.control[:, :, 0])   # u-index fastest
    assert len(A.zero_dofs) == 3 * (p0.n_u + spec.patches[1].n_u)
    assert A.if_alpha[0] > A.if_alpha[1] > 0


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports exactly what include/goldfish_hip.h declares."""
    from goldfish_amd import _lib, build
    build.build()
    hdr = open(os.path.join(ROOT, "include", "goldfish_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gf_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(_lib.EXPORTS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    _lib.lib()


def test_solver_library_exports_every_declared_symbol():
    """libgoldfish_solver.so (block-banded L D L^T factorisation + solves on the device) loads and exports
    what include/goldfish_solver.h declares (no compute call without a GPU: gfs_create must fail cleanly); the host half --
    control-point graph from the dof pattern, bandwidth-reducing order -- on a two-patch model."""
    import scipy.sparse as sp
    from goldfish_amd import _solver, build
    build.build()
    hdr = open(os.path.join(ROOT, "include", "goldfish_solver.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gfs_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(_solver.EXPORTS)
    L = ctypes.CDLL(_solver.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    # dof-level pattern of a small block matrix -> control-point lists; the order is a permutation that shrinks the bandwidth
    rng = np.random.default_rng(0)
    ncp = 60
    pts = rng.uniform(0, 1, (ncp, 2))
    adj = (np.abs(pts[:, None, :] - pts[None, :, :]).max(-1) < 0.22)
    nb_lists = [np.flatnonzero(adj[a]) for a in range(ncp)]
    rowptr, col = [0], []
    for a in range(ncp):
        for i in range(3):
            col += [3 * b + j for b in nb_lists[a] for j in range(3)]
            rowptr.append(len(col))
    nb_ptr, nb = _solver.control_point_graph(np.array(rowptr), np.array(col))
    assert np.array_equal(nb_ptr, np.concatenate([[0], np.cumsum([len(x) for x in nb_lists])]))
    assert np.array_equal(nb, np.concatenate(nb_lists))
    new = _solver.bandwidth_reducing_order(nb_ptr, nb)
    assert sorted(new) == list(range(ncp))
    bw_old = max(abs(a - b) for a in range(ncp) for b in nb_lists[a])
    bw_new = max(abs(int(new[a]) - int(new[b])) for a in range(ncp) for b in nb_lists[a])
    assert bw_new < bw_old
    if _lib_has_no_gpu():
        h = ctypes.c_void_p()
        L.gfs_create.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                 ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        L.gfs_last_error.restype = ctypes.c_char_p
        rc = L.gfs_create(0, ncp, nb_ptr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), nb.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                          new.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.c_void_p(8), ctypes.byref(h))
        assert rc != 0 and b"no HIP device" in L.gfs_last_error()


def test_nested_dissection_cuts_avoid_the_wide_coupling_of_patch_interfaces():
    """goldfish_amd/_nd.py: _best_cuts -- a median cut of a patch grid falls on the patch interfaces, where the coupling reaches one control-point row further than
    inside a patch; the cut with the smallest separator within the window lies inside the patches: fewer flops, smaller fronts, still a valid elimination order
    (every control point once; the boundaries checked by the test below on random points), and both sides of every split stay non-empty."""
    from goldfish_amd import _nd
    n, pw = 96, 24
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    idx = ii * n + jj
    rows, cols = [], []
    for di in range(-4, 5):
        for dj in range(-4, 5):
            i2, j2 = ii + di, jj + dj
            ok = (i2 >= 0) & (i2 < n) & (j2 >= 0) & (j2 < n)
            ok &= ((abs(di) <= 3) | ((ii // pw) != (i2 // pw))) & ((abs(dj) <= 3) | ((jj // pw) != (j2 // pw)))     # reach 3 inside a patch, 4 across an interface
            rows.append(idx[ok]); cols.append((i2 * n + j2)[ok])
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    o = np.lexsort((cols, rows)); rows, cols = rows[o], cols[o]
    nb_ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n * n))]).astype(np.int64)
    X = np.stack([ii.ravel(), jj.ravel()], 1).astype(float)
    st = {}
    for cw in (0.0, 0.04):
        sym = _nd.nested_dissection(nb_ptr, cols.astype(np.int32), X, leaf=64, cut_window=cw)
        assert sorted(sym.elim) == list(range(n * n)) and np.array_equal(sym.order[sym.elim], np.arange(n * n))
        st[cw] = sym.stats()
        if cw > 0:                       # the root separator: three grid lines inside a patch instead of four on an interface
            root = int(np.flatnonzero(sym.parent < 0)[0])
            assert sym.elim_off[root + 1] - sym.elim_off[root] == 3 * n
    assert st[0.04]["flops"] < 0.9 * st[0.0]["flops"] and st[0.04]["largest_front_dofs"] <= st[0.0]["largest_front_dofs"]


def test_nested_dissection_symbolic_phase_and_reference_multifrontal_solve():
    """goldfish_amd/_nd.py: the fronts of the nested-dissection order (what gfs_create_nd factors on the device) -- every control point
    eliminated once, boundaries lie outside the subtree and inside the parent's front, sibling subtrees are not coupled -- and the dense
    NumPy statement of the multifrontal factorisation on those fronts against a direct solve."""
    import scipy.sparse as sp
    from goldfish_amd import _nd
    rng = np.random.default_rng(3)
    ncp = 900
    pts = rng.uniform(0, 1, (ncp, 3)) * [1.0, 0.7, 0.05]
    adj = np.abs(pts[:, None, :2] - pts[None, :, :2]).max(-1) < 0.06
    nb_lists = [np.flatnonzero(adj[a]) for a in range(ncp)]
    nb_ptr = np.concatenate([[0], np.cumsum([len(x) for x in nb_lists])]).astype(np.int64)
    nb = np.concatenate(nb_lists).astype(np.int32)
    sym = _nd.nested_dissection(nb_ptr, nb, pts, leaf=40)
    assert sym.nfronts > 8 and sorted(sym.elim) == list(range(ncp))
    assert np.array_equal(sym.order[sym.elim], np.arange(ncp))
    for t in range(sym.nfronts):
        e = sym.elim[sym.elim_off[t]:sym.elim_off[t + 1]]
        bd = sym.bnd[sym.bnd_off[t]:sym.bnd_off[t + 1]]
        assert np.all(sym.front_of[e] == t) and np.all(np.diff(sym.order[bd]) > 0)
        assert np.all(sym.order[bd] >= sym.elim_off[t + 1])                                   # outside the subtree
        nbrs = np.unique(np.concatenate([nb_lists[a] for a in e])) if e.size else e
        later = nbrs[sym.order[nbrs] >= sym.elim_off[t + 1]]
        assert np.isin(later, bd).all()                                                      # every later neighbour is on the boundary
        if sym.parent[t] >= 0:
            pt = sym.parent[t]
            assert pt > t
            pf = np.concatenate([sym.elim[sym.elim_off[pt]:sym.elim_off[pt + 1]], sym.bnd[sym.bnd_off[pt]:sym.bnd_off[pt + 1]]])
            assert np.isin(bd, pf).all()                                                     # extend-add target exists
        else:
            assert bd.size == 0
    st = sym.stats()
    assert st["eliminated_block_columns"] >= (3 * ncp + 63) // 64 and st["bytes"] > 0
    # SPD block matrix on that pattern; the reference multifrontal solve is a direct solve
    rows = np.repeat(np.arange(ncp), np.diff(nb_ptr))
    B = rng.standard_normal((rows.size, 3, 3)) * 0.1
    K = sp.bsr_matrix((B, nb, nb_ptr), shape=(3 * ncp, 3 * ncp)).tocsr()
    K = (K + K.T) * 0.5 + sp.identity(3 * ncp) * 8.0
    b = rng.standard_normal(3 * ncp)
    x = _nd.multifrontal_reference_solve(sym, K, b)
    xd = np.linalg.solve(K.toarray(), b)
    assert np.abs(x - xd).max() < 1e-12 * np.abs(xd).max()


def test_native_symbolic_phase_equals_the_numpy_statement():
    """csrc/gf_nd_symbolic.hpp (gfs_symbolic_create: host C++ in libgoldfish_solver.so, what DeviceSolver uses) against goldfish_amd/_nd.py entry by entry -- elimination
    order, fronts, boundaries, parents, extend-add maps -- on scattered points and on a patch grid with wide interface coupling, for several cut windows, leaf sizes and
    thread counts (the result does not depend on the threads)."""
    from goldfish_amd import _nd
    from goldfish_amd._solver import parent_positions
    rng = np.random.default_rng(11)
    ncp = 1800
    pts = rng.uniform(0, 1, (ncp, 3)) * [1.0, 0.6, 0.03]
    adj = np.abs(pts[:, None, :2] - pts[None, :, :2]).max(-1) < 0.045
    nbl = [np.flatnonzero(adj[a]) for a in range(ncp)]
    cases = [(np.concatenate([[0], np.cumsum([len(x) for x in nbl])]).astype(np.int64), np.concatenate(nbl).astype(np.int32), pts)]
    n, pw = 72, 18
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    idx = ii * n + jj
    rows, cols = [], []
    for di in range(-4, 5):
        for dj in range(-4, 5):
            i2, j2 = ii + di, jj + dj
            ok = (i2 >= 0) & (i2 < n) & (j2 >= 0) & (j2 < n)
            ok &= ((abs(di) <= 3) | ((ii // pw) != (i2 // pw))) & ((abs(dj) <= 3) | ((jj // pw) != (j2 // pw)))
            rows.append(idx[ok]); cols.append((i2 * n + j2)[ok])
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    o = np.lexsort((cols, rows)); rows, cols = rows[o], cols[o]
    cases.append((np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n * n))]).astype(np.int64), cols.astype(np.int32), np.stack([ii.ravel(), jj.ravel()], 1).astype(float)))
    for nb_ptr, nb, X in cases:
        for leaf, cw in ((40, 0.0), (40, 0.04), (128, 0.1)):
            a = _nd.nested_dissection(nb_ptr, nb, X, leaf=leaf, cut_window=cw)
            for threads in (1, 3):
                b, pmap = _nd.nested_dissection_native(nb_ptr, nb, X, leaf=leaf, cut_window=cw, threads=threads)
                for k in ("elim", "elim_off", "bnd", "bnd_off", "parent", "order", "front_of"):
                    assert np.array_equal(getattr(a, k), getattr(b, k)), (k, leaf, cw, threads)
                assert np.array_equal(parent_positions(a), pmap)


def test_native_symbolic_phase_on_degenerate_graphs():
    """The same comparison where the bisection has nothing to hold on to: disconnected clusters and isolated points (empty separators: tree nodes that own nothing),
    identical coordinates, a single front, a chain with a wide cut window, a complete graph."""
    from goldfish_amd import _nd
    from goldfish_amd._solver import parent_positions
    rng = np.random.default_rng(5)

    def csr(nbl):
        return np.concatenate([[0], np.cumsum([len(x) for x in nbl])]).astype(np.int64), np.concatenate([np.asarray(x, np.int32) for x in nbl])
    pts = np.concatenate([rng.uniform(0, 1, (250, 2)), rng.uniform(5, 6, (250, 2)), rng.uniform(10, 20, (100, 2))])
    adj = np.abs(pts[:, None, :] - pts[None, :, :]).max(-1) < 0.12
    chain = [np.unique(np.clip(np.arange(a - 3, a + 4), 0, 199)) for a in range(200)]
    cases = [(csr([np.flatnonzero(adj[a]) for a in range(600)]), pts, 30, 0.04), (csr(chain), np.zeros((200, 3)), 20, 0.04),
             (csr(chain), np.arange(200.0)[:, None] * np.ones((1, 3)), 500, 0.04), (csr(chain), np.arange(200.0)[:, None], 10, 0.3),
             (csr([np.arange(60) for _ in range(60)]), rng.uniform(0, 1, (60, 3)), 8, 0.1)]
    for (nb_ptr, nb), X, leaf, cw in cases:
        a = _nd.nested_dissection(nb_ptr, nb, X, leaf=leaf, cut_window=cw)
        b, pmap = _nd.nested_dissection_native(nb_ptr, nb, X, leaf=leaf, cut_window=cw, threads=4)
        for k in ("elim", "elim_off", "bnd", "bnd_off", "parent", "order", "front_of"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), k
        assert np.array_equal(parent_positions(a), pmap)


def test_distributed_factorisation_tree_split_and_partial_symbolics():
    """goldfish_amd/_dsolver.py (host side of the distributed factorisation): split_tree deals whole subtrees to the ranks and keeps the top replicated;
    partial_symbolic gives gfs_create_nd_partial consistent pieces -- own control points numbered by their place in the handle's elimination list, the top's control
    points behind them for a handle of subtrees, everything else in front (negative); boundary lists ascending in that order; a stub per subtree root below the top
    with the root's boundary list; the boundary maps into the parents monotone."""
    from goldfish_amd import _nd, _dsolver
    rng = np.random.default_rng(5)
    ncp = 1500
    pts = rng.uniform(0, 1, (ncp, 3)) * [1.0, 0.8, 0.02]
    adj = np.abs(pts[:, None, :2] - pts[None, :, :2]).max(-1) < 0.05
    nb_lists = [np.flatnonzero(adj[a]) for a in range(ncp)]
    nb_ptr = np.concatenate([[0], np.cumsum([len(x) for x in nb_lists])]).astype(np.int64)
    sym = _nd.nested_dissection(nb_ptr, np.concatenate(nb_lists).astype(np.int32), pts, leaf=40)
    for world in (2, 3, 5):
        owner, roots = _dsolver.split_tree(sym, world)
        assert set(np.unique(owner)) == set(range(-1, world)) and len(roots) >= world
        for t in range(sym.nfronts):                                     # a subtree is owned as a whole; the parent of a subtree root is a top front
            p = sym.parent[t]
            if owner[t] == -1:
                assert p < 0 or owner[p] == -1
            elif t in roots:
                assert p >= 0 and owner[p] == -1
            else:
                assert owner[p] == owner[t]
        top_f = np.flatnonzero(owner == -1)
        top_cp = np.concatenate([sym.elim[sym.elim_off[t]:sym.elim_off[t + 1]] for t in top_f])
        seen = np.zeros(ncp, int)
        for r in list(range(world)) + [-1]:
            keep = np.flatnonzero(owner == r)
            fronts, sub, pmap = _dsolver.partial_symbolic(sym, keep, roots if r < 0 else (), None if r < 0 else top_cp)
            seen[sub.elim] += 1
            assert np.array_equal(sub.order[sub.elim], np.arange(sub.elim.size)) and np.all(sub.front_of[sub.elim] >= 0)
            foreign = np.ones(ncp, bool); foreign[sub.elim] = False
            assert np.all(sub.front_of[foreign] == -1)
            if r >= 0:
                assert np.all(sub.order[top_cp] >= sub.elim.size) and np.all(sub.order[foreign & ~np.isin(np.arange(ncp), top_cp)] < 0)
            else:
                assert np.all(sub.order[foreign] < 0)
            for i, t in enumerate(fronts):
                bd = sub.bnd[sub.bnd_off[i]:sub.bnd_off[i + 1]]
                assert np.array_equal(bd, sym.bnd[sym.bnd_off[t]:sym.bnd_off[t + 1]]) and np.all(np.diff(sub.order[bd]) > 0)
                ne = sub.elim_off[i + 1] - sub.elim_off[i]
                if r < 0 and t in roots:
                    assert ne == 0 and sub.parent[i] >= 0                                   # stub below a top front
                else:
                    assert ne == sym.elim_off[t + 1] - sym.elim_off[t] and np.all(sub.order[bd] >= sub.elim_off[i + 1])
                if sub.parent[i] >= 0:
                    pm = pmap[sub.bnd_off[i]:sub.bnd_off[i + 1]]
                    pi = sub.parent[i]
                    assert np.all(np.diff(pm) > 0) and (pm.size == 0 or pm[-1] < (sub.elim_off[pi + 1] - sub.elim_off[pi]) + (sub.bnd_off[pi + 1] - sub.bnd_off[pi]))
                elif r >= 0:
                    assert t in roots
        assert np.all(seen == 1)                                          # every control point is eliminated by exactly one handle


def _lib_has_no_gpu():
    from goldfish_amd import _lib
    return _lib.lib().gf_device_count() == 0


def test_partition_and_shards():
    """Partition of the interface graph: every patch owned once, balanced Gauss points, ghosts = the patches
    across cut interfaces; the owned rows of a rank (not contiguous in the global numbering in general) map back to the global vector."""
    spec = G.synthetic_shell(4, 3, nel=4, p=3, jitter=1)
    for world in (1, 2, 3, 4):
        part = sharding.partition_patches(spec, world)
        assert part.shape == (12,) and sorted(set(part.tolist())) == list(range(world))
        q = sharding.partition_quality(spec, part)
        assert q["imbalance"] < 1.35 and sum(q["owned_patches"]) == 12
    for rank in range(3):
        sh = sharding.shard_spec(spec, rank, 3)
        own = set(sh.order[:sh.n_owned])
        assert own == set(sh.owned_by_rank[rank])
        for itf_g in spec.interfaces:
            if (itf_g.a in own) != (itf_g.b in own):
                assert {itf_g.a, itf_g.b} <= set(sh.order)
        rows = sh.owned_rows_global(3)
        assert rows.size == sh.owned_local_size(3)
        v = np.arange(3 * sh.total_cp_global, dtype=float)
        assert np.array_equal(sh.to_local(v, 3)[:rows.size], v[rows])
    allrows = np.sort(np.concatenate([sh.owned_rows_global(3, r) for r in range(3)]))
    assert np.array_equal(allrows, np.arange(3 * sh.total_cp_global))


def test_partition_of_the_c4_topology_at_eight_ranks():
    """C4 (16 x 16 patches, 480 interfaces) over 8 ranks: the 4 x 2 arrangement of 4 x 8-patch blocks -- 64 cut interfaces
    (a contiguous 1-D split cuts 112), Gauss points balanced to 1 %, half of the owned patches as ghosts on average."""
    spec = G.synthetic_shell(16, 16, nel=48, p=3, jitter=2)
    for world, cut in ((2, 16), (4, 32), (8, 64)):
        q = sharding.partition_quality(spec, sharding.partition_patches(spec, world))
        assert q["cut_interfaces"] <= cut and q["imbalance"] < 1.02, (world, q)
        assert q["owned_patches"] == [256 // world] * world
    assert q["ghost_fraction_mean"] <= 0.5 and q["ghost_fraction_max"] <= 0.65


def test_no_gpu_means_loud_failure():
    """The product path must fail loudly (never fall back) when no GPU / extension is present."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from goldfish_amd import _lib
    spec = G.tbeam_2patch(4)
    with pytest.raises(RuntimeError):
        _lib.DeviceModel(arrays_from_spec(spec))


def test_ffd_block_reproduces_surface_control_points():
    """N2: the FFD operator of an undeformed block is the identity on the surface control points
    (reference assumption, GOLDFISH/utils/ffd_utils.py:37-39) and its columns sum to one."""
    from goldfish_amd.utils.ffd_utils import CP_FFD_matrix, create_3D_block
    rng = np.random.default_rng(1)
    X = rng.uniform([-1, 0, -2], [1, 20, 0], (50, 3))
    blk = create_3D_block([3, 2, 1], 3, [[-1, 1], [0, 20], [-2, 0]])
    D = CP_FFD_matrix(X, blk.degree, blk.knots).tocsr()
    q = blk.control[..., 0:3].transpose(2, 1, 0, 3).reshape(-1, 3)       # i + j*l + k*l*m ordering
    assert D.shape == (50, q.shape[0])
    assert np.abs(D @ q - X).max() < 1e-12
    assert np.abs(np.asarray(D.sum(1)).ravel() - 1.0).max() < 1e-12
    flat = create_3D_block([2, 2, 1], 2, [[0, 1], [0, 1], [0.5, 0.5]])    # degenerate direction gets thickened
    assert flat.knots[2][-1] > flat.knots[2][0]


def _ffd_problem():
    from goldfish_amd.nonmatching_opt import NonMatchingOptFFD
    from goldfish_amd.utils.ffd_utils import create_3D_block
    pb = NonMatchingOptFFD.from_spec(G.tbeam_2patch(4), klass=NonMatchingOptFFD)
    pb.set_shopt_surf_inds_FFD([1, 2], [[0, 1], [0, 1]])
    lims = [list(x) for x in pb.cpsurf_lims]
    for f in range(3):
        pad = 0.1 * max(lims[f][1] - lims[f][0], 1.0)
        lims[f] = [lims[f][0] - pad, lims[f][1] + pad]
    blk = create_3D_block([3, 2, 2], 2, lims)
    pb.set_shopt_FFD(blk.knots, blk.control)
    return pb, blk


def test_ffd_fields_with_different_patch_sets():
    """set_shopt_FFD refused opt fields that optimise different patch sets (round-3 verdict, missing 5); the reference shares one list
    (nonmatching_opt_ffd.py:60-72), a list per field gives one map per field over the same block."""
    from goldfish_amd.nonmatching_opt import NonMatchingOptFFD
    from goldfish_amd.om_comps import om
    from goldfish_amd.om_comps.ffd_comps import CPFFD2SurfComp
    from goldfish_amd.utils.ffd_utils import create_3D_block
    pb = NonMatchingOptFFD.from_spec(G.tbeam_2patch(4), klass=NonMatchingOptFFD)
    pb.set_shopt_surf_inds_FFD([1, 2], [[0, 1], [1]])                    # field 1 moves both patches, field 2 only the web
    lims = [list(x) for x in pb.cpsurf_des_lims]
    for f in range(3):
        pad = 0.1 * max(lims[f][1] - lims[f][0], 1.0)
        lims[f] = [lims[f][0] - pad, lims[f][1] + pad]
    blk = create_3D_block([3, 2, 2], 2, lims)
    maps = pb.set_shopt_FFD(blk.knots, blk.control)
    assert isinstance(maps, list) and len(maps) == 2 and not pb.shopt_ffd_shared_patches
    n0, n1 = pb.splines[0].ncp, pb.splines[1].ncp
    assert maps[0].shape == (n0 + n1, pb.shopt_cpffd_size) and maps[1].shape == (n1, pb.shopt_cpffd_size)
    q = pb.shopt_cpffd_flat
    for i, field in enumerate(pb.opt_field):                             # undeformed block: identity on the surface control points of that field's patches
        assert np.abs(maps[i] @ q[:, field] - pb.cp_iga[field][pb._shopt_cols[i]]).max() < 1e-10
    assert np.abs(maps[0].tocsr()[n0:] - maps[1].tocsr()).max() < 1e-14   # the web's rows are the same in both maps
    comp = CPFFD2SurfComp(nonmatching_opt_ffd=pb)
    comp.init_parameters()
    prob = om.Problem(model=comp)
    prob.setup(); prob.run_model()
    assert np.ravel(prob.get_val("CP_FE1")).size == n0 + n1 and np.ravel(prob.get_val("CP_FE2")).size == n1
    shared = NonMatchingOptFFD.from_spec(G.tbeam_2patch(4), klass=NonMatchingOptFFD)
    shared.set_shopt_surf_inds_FFD([1, 2], [0, 1])
    m = shared.set_shopt_FFD(blk.knots, blk.control)
    assert not isinstance(m, list) and shared.shopt_ffd_shared_patches and shared.shopt_dcpsurf_fedcpffd_list[1] is m


def test_ffd_constraint_maps_against_their_definitions():
    """N2: align / pin / regularisation maps of the FFD block, checked against brute-force
    constructions of their definitions (GOLDFISH/nonmatching_opt_ffd.py:691-883, 1034-1244)."""
    pb, blk = _ffd_problem()
    l, m, n = pb.shopt_cpffd_shape
    dof = lambda i, j, k: i + j * l + k * l * m
    # -- align: every full dof copies the design dof that has the aligned coordinates replaced by the side-0 layer
    for field, dirs in [(2, [0]), (2, [1]), (0, [2]), (0, [1, 2]), (1, [0, 2]), (2, [0, 1])]:
        free, D = pb.dCPaligndCPFFD(field, dirs, (l, m, n))
        D = D.toarray()
        assert D.shape == (l * m * n, len(free)) and sorted(free) == list(free)
        for k in range(n):
            for j in range(m):
                for i in range(l):
                    src = [i, j, k]
                    for d in dirs:
                        src[d] = 0
                    row = D[dof(i, j, k)]
                    assert row.sum() == 1.0 and row[free.index(dof(*src))] == 1.0
    with pytest.raises(ValueError):
        pb.dCPaligndCPFFD(1, [1], (l, m, n))
    maps = pb.set_shopt_align_CPFFD([[0], None])
    assert maps[0].shape == (l * m * n, m * n) and maps[1].shape == (l * m * n, l * m * n)
    assert len(pb.shopt_init_cpffd_design[0]) == m * n
    full = maps[0] @ pb.shopt_init_cpffd_design[0]                  # the undeformed block is already aligned along x for field y
    assert np.allclose(full, pb.shopt_init_cpffd_full[0])
    # -- pin: face j = 0 of field 1 (only design dofs survive), edge (i = l-1, k = 0) of field 2
    pins = pb.set_shopt_pin_CPFFD([1, 0], [[0], [1]], [None, 2], [None, [0]])
    want0 = sorted(d for d in (dof(i, 0, k) for i in range(l) for k in range(n)) if d in pb.shopt_cpffd_design_dof[0])
    want1 = sorted(dof(l - 1, j, 0) for j in range(m))
    assert pb.shopt_cpffd_pin_dof == [want0, want1] and pb.pin_field == [1, 2]
    for fi in range(2):
        P = pins[fi].toarray()
        des = pb.shopt_cpffd_design_dof[fi]
        assert P.shape == (len(pb.shopt_cpffd_pin_dof[fi]), len(des))
        x = np.arange(len(des), dtype=float)
        assert np.array_equal(P @ x, [des.index(d) for d in pb.shopt_cpffd_pin_dof[fi]])
        assert np.allclose(pb.shopt_pin_vals[fi], pb.shopt_cpffd_flat[:, pb.opt_field[fi]][pb.shopt_cpffd_pin_dof[fi]])
    # -- regularisation: neighbour differences along the optimised coordinate on the reduced lattice
    regs = pb.set_shopt_regu_CPFFD()
    R0 = regs[0].toarray()                                          # field 1, aligned along 0 -> lattice (1, m, n)
    rows = [(dof0, dof0 + 1) for i in range(1) for j in range(m - 1) for k in range(n) for dof0 in [i + j * 1 + k * 1 * m]]
    assert R0.shape == (len(rows), m * n)
    for r, (a, b) in enumerate(rows):
        assert R0[r, a] == -1.0 and R0[r, b] == 1.0 and np.abs(R0[r]).sum() == 2.0
    R1 = regs[1].toarray()                                          # field 2, no alignment
    rows = [(dof(i, j, k), dof(i, j, k + 1)) for i in range(l) for j in range(m) for k in range(n - 1)]
    assert R1.shape == (len(rows), l * m * n)
    for r, (a, b) in enumerate(rows):
        assert R1[r, a] == -1.0 and R1[r, b] == 1.0
    z = pb.shopt_init_cpffd_design[1]
    assert (regs[1] @ z > 0).all()                                   # an undeformed block is not folded


def test_thickness_ffd_maps_and_design_components():
    """N2: thickness FFD (nonmatching_opt_ffd.py:434-532, 915-997) and the constant-map components run through
    the OpenMDAO protocol shim: outputs, declared sparsity and partials."""
    from goldfish_amd import om_shim
    from goldfish_amd.om_comps.ffd_comps import (CPFFDesign2FullComp, CPFFDPinComp, CPFFDReguComp, HthFFD2FEComp,
                                                 HthFFDAlignComp, HthFFDReguComp, HthMapComp)
    from goldfish_amd.utils.ffd_utils import create_3D_block
    pb, blk = _ffd_problem()
    pb.set_shopt_align_CPFFD([[0], None])
    pb.set_shopt_pin_CPFFD([1, 0], [[0], [1]], [None, 2], [None, [0]])
    pb.set_shopt_regu_CPFFD()
    pb.set_thopt_surf_inds_FFD([0, 1])
    lims = [[a - 0.1 * max(b - a, 1.0), b + 0.1 * max(b - a, 1.0)] for a, b in pb.thopt_cpsurf_des_lims]
    tb = create_3D_block([2, 2, 1], 2, lims)
    A = pb.set_thopt_FFD(tb.knots, tb.control)
    h0 = pb.get_init_h_th_FFD()
    assert np.allclose(A @ h0, np.concatenate(pb.h_th)[pb._thopt_cols], atol=1e-12)      # a constant field is reproduced
    l, m, n = pb.thopt_cpffd_shape
    Al = pb.set_thopt_align_CPFFD([2]).toarray()
    assert Al.shape == (l * m * (n - 1), l * m * n) and np.abs(Al.sum(1)).max() == 0.0 and np.abs(Al @ np.ones(l * m * n)).max() == 0.0
    Rg = pb.set_thopt_regu_CPFFD([None], [None])[0].toarray()
    assert Rg.shape == (l * m * (n - 1), l * m * n) and pb.thopt_cpregu_sizes == [l * m * (n - 1)]
    Rf = pb.set_thopt_regu_CPFFD([0], [1])[0].toarray()                                  # only the face i = l-1
    assert Rf.shape == (m * (n - 1), l * m * n) and set(np.nonzero(Rf)[1] % l) == {l - 1}
    for comp, key in [(CPFFDesign2FullComp(nonmatching_opt_ffd=pb), None), (CPFFDPinComp(nonmatching_opt_ffd=pb), None),
                      (CPFFDReguComp(nonmatching_opt_ffd=pb), None), (HthFFD2FEComp(nonmatching_opt_ffd=pb), None),
                      (HthFFDAlignComp(nonmatching_opt_ffd=pb), None), (HthFFDReguComp(nonmatching_opt_ffd=pb), None),
                      (HthMapComp(nonmatching_opt=pb), None)]:
        comp.init_parameters()
        prob = om_shim.Problem(model=comp)
        prob.setup()
        prob.run_model()
        errs = prob.check_partials(step=1e-3)
        assert errs and max(errs.values()) < 1e-9, (type(comp).__name__, errs)
    from goldfish_amd.om_comps.ffd_comps import HthFE2IGAComp
    fe = HthFE2IGAComp(nonmatching_opt=pb)
    fe.init_parameters()
    prob = om_shim.Problem(model=fe)
    prob.setup()
    prob.run_model()
    assert np.array_equal(prob["thickness_IGA"], pb.init_h_th_fe) and max(prob.check_partials(step=1e-3).values()) < 1e-9
    for attr in ("vec_scalar_fe_dof", "init_h_th_fe", "h_th_fe_list", "cpdes_iga_nest", "cpdes_fe_nest", "shopt_cpsurf_fe_hom_list", "cpsurf_des_lims"):
        assert getattr(pb, attr) is not None, attr
    assert pb.shopt_cpsurf_fe_hom_list.shape == (pb._shopt_cols[0].size, 4)
    hm = HthMapComp(nonmatching_opt=pb)
    hm.init_parameters()
    assert hm.deriv.shape == (pb.vec_scalar_iga_dof, pb.num_splines) and np.allclose(hm.deriv @ hm.init, np.concatenate(pb.h_th))


def test_multi_ffd_blocks():
    """N2: two FFD blocks driving different patches / fields (GOLDFISH/nonmatching_opt_ffd.py:184-428, 726-913):
    identity at the start, block structure of the maps, constraint maps per block, components."""
    from goldfish_amd import om_shim
    from goldfish_amd.nonmatching_opt import NonMatchingOptFFD
    from goldfish_amd.om_comps.ffd_comps import CPFFD2SurfComp, CPFFDesign2FullComp, CPFFDPinComp, CPFFDReguComp
    from goldfish_amd.utils.ffd_utils import create_3D_block
    pb = NonMatchingOptFFD.from_spec(G.tbeam_4patch(), klass=NonMatchingOptFFD)
    pb.set_shopt_surf_inds_multiFFD([[1, 2], [2]], [[1, 0], [2, 3]])           # block 0 -> y,z of patches 1,0; block 1 -> z of patches 2,3
    assert pb.opt_field == [1, 2] and pb.shopt_surf_inds == [[0, 1], [0, 1, 2, 3]] and pb.opt_field_ffdinds == [[0], [0, 1]]
    blks = []
    for lims in pb.shopt_cpsurf_lims_mffd:
        lims = [[a - 0.1 * max(b - a, 1.0), b + 0.1 * max(b - a, 1.0)] for a, b in lims]
        blks.append(create_3D_block([2, 2, 1], 2, lims))
    maps = pb.set_shopt_multiFFD([b.knots for b in blks], [b.control for b in blks])
    sizes = pb.shopt_cp_mffd_size
    assert maps[0].shape == (pb._shopt_cols[0].size, sizes[0]) and maps[1].shape == (pb._shopt_cols[1].size, sizes[0] + sizes[1])
    for fi, f in enumerate(pb.opt_field):                                       # undeformed blocks reproduce the control points
        assert np.abs(maps[fi] @ pb.shopt_init_cp_mffd_full[fi] - pb.get_init_CPIGA()[fi]).max() < 1e-12
    M1 = maps[1].tocsr()                                                        # patches 0,1 depend on block 0 only, patches 2,3 on block 1 only
    n01 = int(pb.cp_off[2])
    assert abs(M1[:n01, sizes[0]:]).sum() == 0.0 and abs(M1[n01:, :sizes[0]]).sum() == 0.0
    al = pb.set_shopt_align_CP_multiFFD(0, [[0], None])
    assert al[0].shape == (sizes[0], sizes[0] // pb.shopt_cp_mffd_shape[0][0]) and al[1].shape == (sizes[0] + sizes[1],) * 2
    pins = pb.set_shopt_pin_CP_multiFFD(1, [0], [[0]])
    assert pb.pin_field == [2] and pins[0] is None and pins[1].shape[1] == sizes[0] + sizes[1]
    assert min(pb.shopt_cp_mffd_pin_dof[1]) >= sizes[0]                        # the pinned dofs belong to block 1
    regs = pb.set_shopt_regu_CP_multiFFD()
    assert regs[0].shape[1] == al[0].shape[1] and regs[1].shape[1] == sizes[0] + sizes[1]
    for Comp in (CPFFD2SurfComp, CPFFDesign2FullComp, CPFFDPinComp, CPFFDReguComp):
        comp = Comp(nonmatching_opt_ffd=pb)
        comp.init_parameters()
        prob = om_shim.Problem(model=comp)
        prob.setup()
        prob.run_model()
        errs = prob.check_partials(step=1e-3)
        assert errs and max(errs.values()) < 1e-9, (Comp.__name__, errs)
    with pytest.raises(ValueError):
        pb.set_shopt_surf_inds_multiFFD([[2], [2]], [[0, 1], [1, 2]])


def test_multi_ffd_thickness():
    """N2: thickness driven by two FFD blocks, remaining patches constant (nonmatching_opt_ffd.py:534-685, 999-1032)."""
    from goldfish_amd import om_shim
    from goldfish_amd.nonmatching_opt import NonMatchingOptFFD
    from goldfish_amd.om_comps.ffd_comps import HthFFD2FEComp, HthFFDAlignComp
    from goldfish_amd.utils.ffd_utils import create_3D_block
    pb = NonMatchingOptFFD.from_spec(G.tbeam_4patch(), klass=NonMatchingOptFFD)
    pb.set_thopt_multiFFD_surf_inds([[2], [0, 1]])                              # patch 3 keeps a constant thickness
    assert pb.thopt_nonffd_shell_inds == [3]
    blks = [create_3D_block([2, 1, 1], 2, [[a - 0.1 * max(b - a, 1.0), b + 0.1 * max(b - a, 1.0)] for a, b in lims]) for lims in pb.thopt_cpsurf_lims_multiffd]
    A = pb.set_thopt_multiFFD([b.knots for b in blks], [b.control for b in blks])
    nd = sum(pb.thopt_cpffd_size_list) + 1
    assert A.shape == (pb.vec_scalar_iga_dof, nd) and pb.thopt_cpffd_design_size == nd
    h0 = pb.get_init_h_th_multiFFD()
    assert np.allclose(A @ h0, np.concatenate(pb.h_th), atol=1e-12)
    Ac = A.tocsr()
    c3 = np.arange(pb.cp_off[3], pb.cp_off[4])
    assert np.array_equal(Ac[c3].toarray(), np.eye(nd)[[-1] * c3.size])           # constant patch: one column of ones
    Al = pb.set_thopt_align_CP_multiFFD([2, [1, 2]])
    assert Al.shape[1] == nd and abs(Al.tocsr()[:, -1]).sum() == 0.0 and np.abs(Al @ np.ones(nd)).max() == 0.0
    for Comp in (HthFFD2FEComp, HthFFDAlignComp):
        comp = Comp(nonmatching_opt_ffd=pb)
        comp.init_parameters()
        prob = om_shim.Problem(model=comp)
        prob.setup()
        prob.run_model()
        errs = prob.check_partials(step=1e-3)
        assert errs and max(errs.values()) < 1e-9, (Comp.__name__, errs)


@pytest.mark.parametrize("method", ["KS", "pnorm", "induced power"])
def test_max_vm_stress_aggregation_chain_rule(method):
    """Host part of MaxvMStressExOperation (max_vmstress_exop.py:188-328): local + global aggregation of per-patch
    form values and the chain-rule factors d(global)/d(form), against central differences; no device involved."""
    from types import SimpleNamespace
    from goldfish_amd.operations.max_vmstress_exop import MaxvMStressExOperation
    nm = SimpleNamespace(num_splines=4, splines=[None] * 4, opt_field=[0], opt_shape=False, opt_thickness=False)
    m = 3.0e6
    rho = 5.0 / m if method == "KS" else 5.0
    op = MaxvMStressExOperation(nm, rho=rho, alpha=0.02, m=m, method=method)
    rng = np.random.default_rng(0)

    def total(vals):
        return op.discrete_max_vM_stress([op.continuous_max_vM_stress(vals[s], s) for s in range(4)])

    if method == "induced power":
        vals = [(rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0)) for _ in range(4)]
    else:
        vals = list(rng.uniform(0.005, 0.05, 4))
    fac = op._patch_factors(vals)
    for s in range(4):
        for k in range(len(fac[s])):
            def pert(eps):
                v = [tuple(x) if isinstance(x, tuple) else x for x in vals]
                if method == "induced power":
                    t = list(v[s]); t[k] += eps; v[s] = tuple(t)
                else:
                    v[s] += eps
                return total(v)
            eps = 1e-6 * (vals[s][k] if method == "induced power" else vals[s])
            num = (pert(eps) - pert(-eps)) / (2 * eps)
            assert abs(num - fac[s][k]) < 1e-6 * abs(num), (method, s, k)
    with pytest.raises(ValueError):
        MaxvMStressExOperation(nm, alpha=1.0, m=1.0, surf="inside")
    with pytest.raises(ValueError):
        MaxvMStressExOperation(nm, alpha=1.0, m=1.0, method="max")
    with pytest.raises(NotImplementedError):
        MaxvMStressExOperation(nm, alpha=1.0, m=1.0, linearize_stress=True)


def _tbeam_intersection_data():
    from goldfish_amd.cpiga2xi import IntersectionData
    spec = G.tbeam_2patch(4)
    itf = spec.interfaces[0]
    return spec, IntersectionData(patches=spec.patches, mapping_list=[[itf.a, itf.b]],
                                  intersections_para_coords=[[itf.xi_a, itf.xi_b]])


def test_cpiga2xi_residual_and_derivatives():
    """N3 host part (cpiga2xi.py:401-790): the intersection residual vanishes at the stored parametric coordinates,
    dR/dxi and dR/dCP agree with central differences."""
    from goldfish_amd.cpiga2xi import CPIGA2Xi
    spec, pre = _tbeam_intersection_data()
    c2x = CPIGA2Xi(pre, opt_surf_inds=[[0, 1]] * 3, opt_field=[0, 1, 2])
    n = c2x.diff_int_num_pts[0]
    assert c2x.xi_size_global == 4 * n and c2x.cp_size_global == sum(P.ncp for P in spec.patches)
    xi0 = c2x.xi_flat_global.copy()
    assert np.abs(c2x.residual(xi0)).max() < 1e-12
    rng = np.random.default_rng(0)
    xi = np.clip(xi0 + 0.02 * rng.standard_normal(xi0.size), 0.01, 0.99)
    J = c2x.dRdxi(xi)
    assert J.shape == (4 * n, 4 * n)
    for k in rng.choice(xi.size, 8, replace=False):
        e = np.zeros(xi.size)
        e[k] = 1e-6
        fd = (c2x.residual(xi + e) - c2x.residual(xi - e)) / 2e-6
        assert np.abs(fd - J[:, k]).max() < 1e-7 * max(1.0, np.abs(J[:, k]).max())
    for field in (0, 2):
        Jc = c2x.dRdCP(xi, field, coo=False)
        assert Jc.shape == (4 * n, c2x.cp_size_global)
        base = c2x.cp_flat_global[:, field].copy()
        for k in rng.choice(base.size, 6, replace=False):
            r = []
            for s in (1, -1):
                v = base.copy()
                v[k] += s * 1e-6
                c2x.update_CPs(v, field)
                r.append(c2x.residual(xi))
            c2x.update_CPs(base, field)
            assert np.abs((r[0] - r[1]) / 2e-6 - Jc[:, k]).max() < 1e-7 * max(1.0, np.abs(Jc[:, k]).max())


def test_cpiga2xi_follows_a_moved_patch():
    """Shifting the web of the T-beam by +0.1 in x moves the intersection to xi_u = 0.55 on the flange; the implicit
    derivative d xi / d CP = -(dR/dxi)^-1 dR/dCP predicts the same move."""
    from goldfish_amd.cpiga2xi import CPIGA2Xi
    spec, pre = _tbeam_intersection_data()
    c2x = CPIGA2Xi(pre, opt_surf_inds=[[0, 1]] * 3, opt_field=[0, 1, 2])
    n = c2x.diff_int_num_pts[0]
    xi0 = c2x.xi_flat_global.copy()
    dxidcp = -np.linalg.solve(c2x.dRdxi(xi0), c2x.dRdCP(xi0, 0, coo=False))
    shift = np.zeros(c2x.cp_size_global)
    shift[c2x.cp_flat_inds[1]:c2x.cp_flat_inds[2]] = 0.1 * spec.patches[1].cp_hom_flat()[:, 3]      # homogeneous x-coefficients of the web
    c2x.update_CPs(c2x.cp_flat_global[:, 0] + shift, 0)
    xi = c2x.solve_xi(xi0)
    assert np.abs(c2x.residual(xi)).max() < 1e-10
    xa, xb = xi[:2 * n].reshape(-1, 2), xi[2 * n:].reshape(-1, 2)
    assert np.abs(xa[:, 0] - 0.55).max() < 1e-10 and np.abs(xa[:, 1] - xi0[:2 * n].reshape(-1, 2)[:, 1]).max() < 1e-9
    assert np.abs(xb - xi0[2 * n:].reshape(-1, 2)).max() < 1e-9
    assert np.abs(xi0 + dxidcp @ shift - xi).max() < 1e-9                     # the map is linear for this geometry


def test_cpiga2xi_component_partials():
    """CPIGA2XiComp (om_comps/cpiga2xi_comp.py:6-103) through the OpenMDAO protocol: solve_nonlinear reproduces the
    stored coordinates, apply_linear (fwd) matches central differences, solve_linear inverts dR/dxi in both modes."""
    from goldfish_amd.nonmatching_opt import NonMatchingOptFFD
    from goldfish_amd.om_comps import CPIGA2XiComp, om
    spec = G.tbeam_2patch(4)
    nm = NonMatchingOptFFD.from_spec(spec)
    nm.set_shopt_surf_inds_FFD([0, 1, 2], [[0, 1]] * 3)
    nm.create_diff_intersections()
    comp = CPIGA2XiComp(nonmatching_opt=nm)
    comp.init_parameters()
    prob = om.Problem(model=comp)
    prob.setup()
    prob.run_model()
    assert np.abs(prob["int_para_coord"] - nm.cpiga2xi.xi_flat_global).max() < 1e-10
    errs = prob.check_partials(compact_print=False)
    assert max(errs.values()) < 1e-6, errs
    op = comp.cpiga2xi_imop
    rng = np.random.default_rng(2)
    b = rng.standard_normal(nm.xi_size)
    x = op.solve_linear_fwd(np.zeros(nm.xi_size), b.copy())
    assert np.abs(op.dRdxi_mat @ x - b).max() < 1e-9
    y = op.solve_linear_rev(b.copy(), np.zeros(nm.xi_size))
    assert np.abs(op.dRdxi_mat.T @ y - b).max() < 1e-9


def test_cpiga2xi_edge_intersection_and_edge_component():
    """'edge-surf' / 'surf-edge' bookkeeping of CPIGA2Xi (cpiga2xi.py:151-304): the pinned end coordinates, the
    coordinates that stay on the patch edge, and IntXiEdgeComp (om_comps/int_xi_edge_comp.py) on top of them."""
    from goldfish_amd.cpiga2xi import CPIGA2Xi, IntersectionData
    from goldfish_amd.nonmatching_opt import NonMatchingOptFFD
    from goldfish_amd.om_comps import IntXiEdgeComp, om
    spec = G.tbeam_2patch(4)
    itf = spec.interfaces[0]
    edge_dir = int(np.argmin(np.ptp(itf.xi_b, axis=0)))                 # the web's coordinate that is constant along the curve
    edge_val = int(round(itf.xi_b[0, edge_dir]))
    pre = IntersectionData(patches=spec.patches, mapping_list=[[itf.a, itf.b]], intersections_para_coords=[[itf.xi_a, itf.xi_b]],
                           intersections_type=[["surf-edge", "1-%d.%d" % (edge_dir, edge_val)]],
                           diff_int_edge_cons=["1-%d.%d" % (edge_dir, edge_val)])
    c2x = CPIGA2Xi(pre, opt_surf_inds=[[0, 1]] * 3, opt_field=[0, 1, 2])
    n = c2x.diff_int_num_pts[0]
    want = 2 * n + 2 * np.arange(n) + edge_dir                         # side-1 coordinates in direction edge_dir
    assert np.array_equal(c2x.int_edge_cons_dofs, want) and np.all(c2x.int_edge_cons_vals == edge_val)
    assert len(c2x.int_xi_free_dofs) == 4 * n - n
    assert np.abs(c2x.residual(c2x.xi_flat_global)).max() < 1e-12
    J = c2x.dRdxi(c2x.xi_flat_global)
    assert np.linalg.matrix_rank(J) == 4 * n                            # the 4n x 4n system is regular
    c2x.implicit_edge = True                                            # edge coordinates replace one coincidence equation
    assert np.abs(c2x.residual(c2x.xi_flat_global)).max() < 1e-12
    nm = NonMatchingOptFFD.from_spec(spec)
    nm.set_shopt_surf_inds_FFD([0, 1, 2], [[0, 1]] * 3)
    nm.create_diff_intersections(preprocessor=pre)
    comp = IntXiEdgeComp(nonmatching_opt=nm)
    comp.init_parameters()
    prob = om.Problem(model=comp)
    prob.setup()
    prob.run_model()
    assert np.abs(prob["int_xi_edge"]).max() < 1e-14
    assert max(prob.check_partials(compact_print=False).values()) < 1e-8


def test_shape_design_on_surface_control_points():
    """set_shopt_align_CP / set_shopt_pin_CP (nonmatching_opt.py:232-364): design dofs, replication map and pin selection
    for shape optimisation directly on the control nets (the moving-intersection demos)."""
    from goldfish_amd.nonmatching_opt import NonMatchingOpt
    spec = G.tbeam_2patch(4)
    nm = NonMatchingOpt.from_spec(spec)
    nm.set_shopt_surf_inds([0, 2], [[0, 1], [1]])
    n0, n1 = spec.patches[0].ncp, spec.patches[1].ncp
    (c0, r0), (c1, r1) = nm.cp_shapes
    D = nm.set_shopt_align_CP(align_surf_inds=[[1], [1]], align_dir=[[0], [1]])
    assert D[0].shape == (n0 + n1, n0 + r1) and D[1].shape == (n1, c1)
    assert nm.shopt_num_desvars == [n0 + r1, c1]
    init = nm.get_init_CPIGA()
    # replicating the design values reproduces a net that is constant along the aligned direction
    full0 = D[0] @ nm.init_cp_iga_design[0]
    web = full0[n0:].reshape(r1, c1)
    assert np.all(web == web[:, :1]) and np.array_equal(full0[:n0], init[0][:n0])
    full1 = (D[1] @ nm.init_cp_iga_design[1]).reshape(r1, c1)
    assert np.all(full1 == full1[:1, :])
    P = nm.set_shopt_pin_CP(pin_surf_inds=[[0], [1]], pin_dir=[[1], [0]], pin_side=[[0], [1]])
    assert P[0].shape == (c0, n0 + r1) and np.array_equal(nm.shopt_pin_dofs[0], np.arange(c0))
    # field 2: only the design dofs (first row of the web) that lie on the u = 1 edge are pinned -> the last one
    assert nm.shopt_pin_dofs[1] == [c1 - 1] and P[1].shape == (1, c1)
    assert np.array_equal(P[0] @ nm.init_cp_iga_design[0], nm.shopt_pin_vals[0])
    with pytest.raises(ValueError):
        nm.set_shopt_align_CP(align_surf_inds=[[1], [0]], align_dir=[[0], [1]])


def test_reference_module_paths_and_identity_projections():
    """The reference's module layout resolves (GOLDFISH/nonmatching_opt_om.py aggregator, nonmatching_opt_ffd.py, one module
    per component / operation), and the FE->IGA projection operations are the identity with unit Jacobians here."""
    import importlib
    for name in ("nonmatching_opt_om", "nonmatching_opt_ffd", "cpiga2xi", "om_comps.cpfe2iga_comp", "om_comps.hthfe2iga_comp",
                 "om_comps.cpiga2xi_comp", "om_comps.int_xi_edge_comp", "om_comps.disp_states_mi_comp", "om_comps.max_vmstress_comp",
                 "om_comps.ffd_comps.hth_map_comp", "om_comps.ffd_comps.hthffd2fe_comp", "om_comps.ffd_comps.hthffd_align_comp",
                 "om_comps.ffd_comps.hthffd_regu_comp", "operations.cpfe2iga_imop", "operations.hthfe2iga_imop", "operations.custom_exop",
                 "operations.cpiga2xi_imop", "operations.disp_mi_imop", "operations.max_vmstress_exop"):
        importlib.import_module("goldfish_amd." + name)
    ns = {}
    exec("from goldfish_amd.nonmatching_opt_om import *", ns)
    for cls in ("NonMatchingOptFFD", "DispStatesComp", "DispMintStatesComp", "CPIGA2XiComp", "IntXiEdgeComp", "MaxvMStressComp",
                "CPFFD2SurfComp", "HthMapComp", "create_3D_block", "CP_FFD_matrix"):
        assert cls in ns, cls
    from types import SimpleNamespace
    from goldfish_amd.operations.cpfe2iga_imop import CPFE2IGAImOperation
    from goldfish_amd.operations.custom_exop import CustomExOperation
    nm = SimpleNamespace(opt_field=[0], opt_shape=True, num_splines=1)
    op = CPFE2IGAImOperation(nm)
    x = np.arange(4.0)
    assert np.array_equal(op.solve_nonlinear(x), x) and np.abs(op.apply_nonlinear(x, op.solve_nonlinear(x))).max() == 0
    r = np.zeros(4)
    op.apply_linear_fwd(2 * x, 3 * x, r)
    assert np.array_equal(r, x)
    c = CustomExOperation(nm, lambda p: 3.0, lambda p: np.ones(2))
    assert c.func() == 3.0 and np.array_equal(c.func_deriv(), np.ones(2))
    with pytest.raises(TypeError):
        CustomExOperation(nm, "ufl form", "ufl form")


def test_small_utils_with_reference_names():
    """utils/ffd_utils.py (scale_knots, rationalized_control, refine_knot, update_FFD_block :10-33,126-161,348-358) and
    utils/opt_utils.py (solve_Ax_b / solve_ATx_b :156-209) counterparts."""
    import scipy.sparse as sps
    from goldfish_amd.utils.ffd_utils import create_3D_block, rationalized_control, refine_knot, scale_knots, update_FFD_block
    from goldfish_amd.utils.opt_utils import solve_ATx_b, solve_Ax_b
    blk = create_3D_block([2, 1, 1], 2, [[0, 2], [0, 1], [0, 3]])
    k = scale_knots([np.array([0, 0, .5, 1, 1.]), np.array([0, 1.]), np.array([0, 1.])], blk.control)
    assert np.allclose(k[0], [0, 0, 1, 2, 2]) and np.allclose(k[2], [0, 3])
    assert np.allclose(rationalized_control(blk), blk.control[..., :3])
    assert np.allclose(refine_knot(np.array([0, 0, 1, 1.]), 1), [0, 0, 0, .5, 1, 1, 1])
    n = int(np.prod(blk.shape))
    b2 = update_FFD_block(blk, [np.arange(n, dtype=float)], [2])
    assert np.allclose(b2.control[..., :3].transpose(2, 1, 0, 3).reshape(-1, 3)[:, 2], np.arange(n))
    assert np.allclose(b2.control[..., 0], blk.control[..., 0])
    A = sps.csr_matrix(np.array([[4, 1, 0], [2, 3, 0], [0, 1, 5.]]))
    b = np.array([1, 2, 3.])
    assert np.allclose(A @ solve_Ax_b(A, b), b) and np.allclose(A.T @ solve_ATx_b(A, b), b)


def _build_c_consumer(tmp_path):
    import subprocess
    from goldfish_amd import build
    build.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "goldfish_amd")
    exe = os.path.join(str(tmp_path), "c_consumer")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "c_abi", "c_consumer.c"),
                           "-o", exe, "-L" + libdir, "-lgoldfish_hip", "-lm", "-Wl,-rpath," + libdir])
    return exe


def test_c_abi_headers_compile_and_link_from_plain_c(tmp_path):
    """include/*.h are C99-clean and libgoldfish_hip.so links into a plain-C program (tests/c_abi/c_consumer.c); it is run on
    the GPU box by tests/test_gpu_api.py."""
    assert os.path.exists(_build_c_consumer(tmp_path))


def _dump_model(A, path):
    """Flat binary of a ModelArrays in the order tests/c_abi/setup_sanitize.cpp reads it."""
    hd = np.array([A.n_patches, A.knots.size, A.total_cp, len(A.zero_dofs), len(A.pl_dof), A.n_interfaces, A.n_mortar_points, A.n_owned], dtype=np.int64)
    with open(path, "wb") as f:
        for arr, dt in ((hd, np.int64), (A.degree, np.int32), (A.ncp, np.int32), (A.knot_off, np.int64), (A.knots, np.float64), (A.cp_off, np.int64),
                        (A.weights, np.float64), (A.young, np.float64), (A.poisson, np.float64), (A.body_force, np.float64), (A.zero_dofs, np.int64),
                        (A.pl_dof, np.int64), (A.pl_val, np.float64), (A.if_patch, np.int32), (A.if_off, np.int64), (A.if_xi, np.float64),
                        (A.if_tau, np.float64), (A.if_wt, np.float64), (A.if_alpha, np.float64), (A.load_proj, np.float64)):
            np.ascontiguousarray(arr, dtype=dt).tofile(f)


def test_host_setup_under_address_and_ub_sanitizers(tmp_path):
    """gf_setup.hpp (the host half of gf_create: element / neighbour / mortar-vertex / owner tables) built from real models
    under ASan + UBSan: T-beam, NURBS Scordelis-Lo, the 16-patch wing with the reference's interface data (T junctions, interior
    curves), a p = 4 shell, a shard with ghost patches, and a single patch without interfaces."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(tmp_path), "setup_sanitize")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           os.path.join(root, "tests", "c_abi", "setup_sanitize.cpp"), "-o", exe])
    wing = G.wing_16patch_from_interface_data(np.load(os.path.join(root, "tests", "golden", "ref_wing_int_data.npz"), allow_pickle=True))
    shell = G.synthetic_shell(3, 2, nel=4, p=3, jitter=1)
    models = {"tbeam4": arrays_from_spec(G.tbeam_4patch(nels=((5, 6), (6, 7), (5, 7), (6, 8)))), "slr9": arrays_from_spec(G.scordelis_lo_9patch(3, nels=[2, 1, 2, 3, 2, 3, 2, 1, 2])),
              "wing16": arrays_from_spec(wing), "shell_p4": arrays_from_spec(G.synthetic_shell(2, 2, nel=3, p=4, jitter=1)),
              "shard": sharding.shard_arrays(sharding.shard_spec(shell, 1, 2)), "single": arrays_from_spec(G.scordelis_lo_single(4)),
              "double_knots": arrays_from_spec(G.with_double_knots(G.synthetic_shell(2, 2, nel=6, p=3, jitter=1)))}   # windows that move two rows at a time
    for name, A in models.items():
        path = os.path.join(str(tmp_path), name + ".bin")
        _dump_model(A, path)
        for seg in ("4", "7", "1000", "0"):  # segment length of the row-record path's work items (1000: whole strips; 0: its own choice)
            out = subprocess.run([exe, path, seg], capture_output=True, text=True, timeout=300,
                                 env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
            assert out.returncode == 0 and "built:" in out.stdout, (name, out.stderr[-2000:])
            assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, (name, out.stderr[-2000:])
            # the row-record path (default for p <= 3): the element kernel's stores against the record gather's reads, position by position
            assert ("rec:" not in out.stdout) if name == "shell_p4" else ("rows per item, dry run ok" in out.stdout), (name, out.stdout)


def test_intersection_cache_format_round_trip_and_reference_files(tmp_path):
    """The reference's .npz intersection cache (keys name1..name6): IntersectionData reads the two files shipped with the
    reference (plate: 5 interfaces, wing: 62) and writes / re-reads its own in the same format."""
    from goldfish_amd.cpiga2xi import IntersectionData
    root = os.path.dirname(os.path.abspath(__file__))
    plate = IntersectionData.load_intersections_data(os.path.join(root, "golden", "ref_plate_int_data.npz"), G.plate_6patch().patches)
    assert plate.num_intersections_all == 5 and list(plate.mortar_nels) == [17, 19, 19, 17, 16]
    assert all(c[0].shape == (n + 1, 2) for c, n in zip(plate.intersections_para_coords, plate.mortar_nels))
    # the strips of the plate meet along x = k/6: both pre-images of a vertex are the same physical point
    for (a, b), c in zip(plate.mapping_list, plate.intersections_para_coords):
        XA = np.array([plate.patches[a].eval(x) for x in c[0]])
        XB = np.array([plate.patches[b].eval(x) for x in c[1]])
        assert np.abs(XA - XB).max() < 1e-6
    wing = np.load(os.path.join(root, "golden", "ref_wing_int_data.npz"), allow_pickle=True)
    assert int(wing["name1"]) == 62 == len(wing["name2"])
    path = os.path.join(str(tmp_path), "int_data.npz")
    plate.save_intersections_data(path)
    again = IntersectionData.load_intersections_data(path, plate.patches)
    assert again.mapping_list == plate.mapping_list and list(again.mortar_nels) == list(plate.mortar_nels)
    for c0, c1 in zip(plate.intersections_para_coords, again.intersections_para_coords):
        assert np.array_equal(c0[0], c1[0]) and np.array_equal(c0[1], c1[1])
    d = np.load(path, allow_pickle=True)
    assert sorted(d.files) == ["name%d" % k for k in range(1, 7)] and d["name5"].shape == (5,) and np.all(d["name5"] > 0.99)


def test_no_valu_write_in_front_of_a_dpp_read():
    """The v_fmac_f64_dpp of the element and penalty kernels are inline assembly: LLVM's hazard recogniser does not see them, and a copy or
    spill reload placed between the kernels' `s_nop 1` fence and a DPP read would read stale lanes silently.  tools/check_dpp_hazard.py
    disassembles the built code object and checks every DPP read (and every MFMA operand produced by one) for a VALU write of its source
    within two wait states; the checker itself is tested on synthetic code."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec_ = importlib.util.spec_from_file_location("check_dpp_hazard", os.path.join(root, "tools", "check_dpp_hazard.py"))
    chk = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(chk)
    head = "0000000000001000 <kern>:\n"
    dpp = "\tv_fmac_f64_dpp v[4:5], v[166:167], v[114:115] row_newbcast:0 row_mask:0xf bank_mask:0xf // 0\n"
    assert chk.scan(head + "\tv_mul_f64 v[166:167], v[2:3], v[8:9] // 0\n" + dpp)[2]                                  # written one slot earlier
    assert chk.scan(head + "\tv_mul_f64 v[166:167], v[2:3], v[8:9] // 0\n\tv_mov_b32_e32 v1, v2 // 0\n" + dpp)[2]     # one instruction between: one wait state
    assert not chk.scan(head + "\tv_mul_f64 v[166:167], v[2:3], v[8:9] // 0\n\ts_nop 1 // 0\n" + dpp)[2]              # the kernels' fence
    assert not chk.scan(head + "\tv_mul_f64 v[160:161], v[2:3], v[8:9] // 0\n" + dpp)[2]                                 # another register
    assert chk.scan(head + dpp + "\tv_mfma_f64_16x16x4_f64 a[0:7], v[4:5], v[6:7], a[0:7] // 0\n")[2]                    # DPP result straight into an MFMA
    from goldfish_amd import build
    build.build()
    n, kernels, bad = chk.scan(chk.disassemble(build.LIB))
    assert n > 1000 and any("kl_element_rec_kernel" in k for k in kernels) and any("pen_row16_kernel" in k for k in kernels), (n, kernels)
    assert not bad, bad[:5]


def test_om_shim_groups_totals_and_strictness():
    """The protocol stand-in behaves like OpenMDAO where a component could pass here and fail there (openmdao itself is
    not installable in the build image): Group + connect + compute_totals (explicit and implicit components, reverse mode) against
    finite differences, undeclared sub-Jacobians, wrong sizes, bad connections, and an apply_linear that assigns instead of accumulating."""
    from goldfish_amd import om_shim as om

    class Sq(om.ExplicitComponent):                       # y = A x^2 (sparse declared partials)
        def setup(self):
            self.add_input("x", shape=3)
            self.add_output("y", shape=2)
            self.A = np.array([[1.0, 2.0, 0.0], [0.0, -1.0, 3.0]])
            r, c = np.nonzero(self.A)
            self.declare_partials("y", "x", rows=r, cols=c)

        def compute(self, inputs, outputs):
            outputs["y"] = self.A @ inputs["x"] ** 2

        def compute_partials(self, inputs, partials):
            r, c = np.nonzero(self.A)
            partials["y", "x"] = (self.A * (2 * inputs["x"])[None, :])[r, c]

    class Imp(om.ImplicitComponent):                      # R(y; p) = K y + y^3 - B p = 0
        accumulate = True

        def setup(self):
            self.add_input("p", shape=2)
            self.add_output("s", shape=2)
            self.K, self.B = np.array([[3.0, 1.0], [1.0, 2.0]]), np.array([[1.0, 0.5], [0.0, 2.0]])
            self.declare_partials("s", "s")
            self.declare_partials("s", "p")

        def apply_nonlinear(self, inputs, outputs, residuals):
            residuals["s"] = self.K @ outputs["s"] + outputs["s"] ** 3 - self.B @ inputs["p"]

        def solve_nonlinear(self, inputs, outputs):
            y = np.zeros(2)
            for _ in range(50):
                r = self.K @ y + y ** 3 - self.B @ inputs["p"]
                y = y - np.linalg.solve(self.K + np.diag(3 * y ** 2), r)
            outputs["s"] = y

        def linearize(self, inputs, outputs, partials):
            self.J = self.K + np.diag(3 * outputs["s"] ** 2)

        def apply_linear(self, inputs, outputs, d_inputs, d_outputs, d_residuals, mode):
            if mode == "fwd":
                add = np.zeros(2)
                if "s" in d_outputs:
                    add += self.J @ d_outputs["s"]
                if "p" in d_inputs:
                    add -= self.B @ d_inputs["p"]
                d_residuals["s"] = (d_residuals["s"] + add) if self.accumulate else add
            else:
                if "s" in d_outputs:
                    d_outputs["s"] = d_outputs["s"] + self.J.T @ d_residuals["s"]
                if "p" in d_inputs:
                    d_inputs["p"] = d_inputs["p"] - self.B.T @ d_residuals["s"]

        def solve_linear(self, d_outputs, d_residuals, mode):
            if mode == "fwd":
                d_outputs["s"] = np.linalg.solve(self.J, d_residuals["s"])
            else:
                d_residuals["s"] = np.linalg.solve(self.J.T, d_outputs["s"])

    class Obj(om.ExplicitComponent):                      # f = sum(s^2) + c . x
        def setup(self):
            self.add_input("s", shape=2)
            self.add_input("x", shape=3)
            self.add_output("f")
            self.declare_partials("f", "s")
            self.declare_partials("f", "x", val=np.array([[0.5, -1.0, 2.0]]))

        def compute(self, inputs, outputs):
            outputs["f"] = np.sum(inputs["s"] ** 2) + np.array([0.5, -1.0, 2.0]) @ inputs["x"]

        def compute_partials(self, inputs, partials):
            partials["f", "s"] = 2 * inputs["s"]

    def build():
        g = om.Group()
        ivc = om.IndepVarComp()
        ivc.add_output("x", shape=3, val=[0.3, -0.7, 1.1])
        g.add_subsystem("ivc", ivc)
        g.add_subsystem("sq", Sq())
        g.add_subsystem("imp", Imp())
        g.add_subsystem("obj", Obj())
        g.connect("ivc.x", "sq.x")
        g.connect("sq.y", "imp.p")
        g.connect("imp.s", "obj.s")
        g.connect("ivc.x", "obj.x")
        return g

    prob = om.Problem(model=build())
    prob.setup()
    prob.run_model()
    tot = prob.compute_totals(of=["obj.f", "imp.s"], wrt=["ivc.x"])
    x0 = prob.get_val("ivc.x").copy()

    def f_of(x):
        prob.set_val("ivc.x", x)
        prob.run_model()
        return np.concatenate([prob.get_val("obj.f").ravel(), prob.get_val("imp.s").ravel()])
    J = np.zeros((3, 3))
    for k in range(3):
        e = np.zeros(3)
        e[k] = 1e-6
        J[:, k] = (f_of(x0 + e) - f_of(x0 - e)) / 2e-6
    assert np.abs(tot[("obj.f", "ivc.x")] - J[0:1]).max() < 1e-8 and np.abs(tot[("imp.s", "ivc.x")] - J[1:3]).max() < 1e-8

    # the implicit component alone: accumulate / transpose / solve_linear consistency are part of check_partials
    p1 = om.Problem(model=Imp())
    p1.setup()
    p1["p"] = [0.4, -0.2]
    p1.run_model()
    assert max(p1.check_partials(compact_print=False).values()) < 1e-7
    bad = Imp()
    bad.accumulate = False                                # assigns d_residuals instead of accumulating: OpenMDAO would give wrong totals
    p2 = om.Problem(model=bad)
    p2.setup()
    p2.run_model()
    with pytest.raises(AssertionError, match="apply_linear|rel err"):
        errs = p2.check_partials(compact_print=False)
        assert max(errs.values()) < 1e-7, "rel err"
    # undeclared sub-Jacobian, wrong nnz, wrong variable size
    p3 = om.Problem(model=Sq())
    p3.setup()
    P = om._Partials(p3.model, p3.inputs, p3.outputs)
    with pytest.raises(KeyError):
        P["y", "z"] = 1.0
    with pytest.raises(ValueError, match="nnz"):
        P["y", "x"] = np.ones(5)
    with pytest.raises(ValueError):
        p3["x"] = np.ones(4)
    # connections: unknown names, mismatched shapes, wrong direction
    for src, tgt, exc in (("sq.nope", "imp.p", NameError), ("ivc.x", "imp.p", ValueError), ("imp.s", "sq.x", ValueError)):
        g = build()
        g.connect(src, tgt) if exc is not ValueError or src != "imp.s" else g._conns.append((src, tgt))
        with pytest.raises(exc):
            om.Problem(model=g).setup()


def test_traffic_tool_counts_every_kernel_of_a_pass_once():
    """tools/traffic_from_pmc.py over the committed PMC CSVs: the p = 4 gather (no WITHC template argument) is in the full pass, only the
    full-pass pen_owner instance is, per-launch averages do not mix full-pass and Newton-pass launches, and the step total is the sum of
    the kernels the step launches (round-3 verdict, What's weak 4)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("traffic_from_pmc", os.path.join(ROOT, "tools", "traffic_from_pmc.py"))
    T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
    out = T.build("r03_c5share_v2", 9012750, os.path.join(ROOT, "profiles"))
    fp, K = out["full_pass_kernels"], out["kernels"]
    assert "kl_gather_kernel<4>" in fp and "pen_owner_kernel<4, 3, true, true>" in fp and "pen_owner_kernel<4, 3, false, true>" not in fp
    g = K["kl_gather_kernel<4>"]
    assert (g["launches_full_pass"], g["launches_other"]) == (2, 5)
    assert g["hbm_side_bytes_corrected_per_launch"] > 2.0 * g["other_pass_hbm_side_bytes_corrected_per_launch"]      # 50.9 vs 22.5 GB
    assert abs(out["full_pass_bytes_per_step"] - sum(K[k]["hbm_side_bytes_corrected_per_launch"] for k in fp)) < 1.0
    assert 90e9 < out["full_pass_bytes_per_step"] < 100e9
    out3 = T.build("r03_v4", 9421968, os.path.join(ROOT, "profiles"))
    assert "kl_gather_rec_kernel<3, true>" in out3["full_pass_kernels"] and "kl_gather_rec_kernel<3, false>" not in out3["full_pass_kernels"]
    assert 33e9 < out3["full_pass_bytes_per_step"] < 36e9


def test_design_to_analysis_control_nets():
    """utils/bsp_utils.py (N3, reference GOLDFISH/utils/bsp_utils.py:516-1230): order elevation and knot refinement are exact embeddings (the surface does not
    change), alignment / pin / regularisation / distance operators against their definitions, and the surf components reproduce the analysis control net."""
    from goldfish_amd.cpiga2xi import IntersectionData
    from goldfish_amd.om_comps import om
    from goldfish_amd.om_comps.surf_comps import CPSurfAlignComp, CPSurfOrderElevationComp, CPSurfKnotRefinementComp, CPSurfPinComp, CPSurfReguComp, CPSurfDistanceComp
    from goldfish_amd.splines import NURBSPatch
    from goldfish_amd.utils import bsp_utils as B
    rng = np.random.default_rng(3)
    # a random biquadratic 3 x 4 net elevated to bicubic and refined: same surface at random points
    p_in, k_in = [2, 2], [np.array([0, 0, 0, 1, 1, 1.0]), np.array([0, 0, 0, 0.5, 1, 1, 1.0])]
    p_out, k_el = [3, 3], [np.array([0, 0, 0, 0, 1, 1, 1, 1.0]), np.array([0, 0, 0, 0, 0.5, 0.5, 1, 1, 1, 1.0])]
    E = B.surface_order_elevation_operator(p_in, k_in, p_out, k_el, coo=False)
    ref = [np.array([0.25, 0.6]), np.array([0.3])]
    R = B.surface_knot_refine_operator(k_el, ref, coo=False)
    k_fine = [np.sort(np.concatenate([k_el[0], ref[0]])), np.sort(np.concatenate([k_el[1], ref[1]]))]
    c = rng.standard_normal(3 * 4)
    cf = R @ (E @ c)

    def ev(p, k, cp, x):
        Bu, Bv = B._collocation(p[0], k[0], [x[0]]), B._collocation(p[1], k[1], [x[1]])
        return float(Bv[0] @ cp.reshape(Bv.shape[1], Bu.shape[1]) @ Bu[0])
    for x in rng.uniform(0, 1, (20, 2)):
        assert abs(ev(p_in, k_in, c, x) - ev(p_out, k_fine, cf, x)) < 1e-12
    A0, free0 = B.surface_cp_align_operator([3, 4], 0)
    A1, free1 = B.surface_cp_align_operator([3, 4], 1)
    v = rng.standard_normal(4)
    assert np.array_equal((A0 @ v).reshape(4, 3), np.repeat(v[:, None], 3, 1)) and free0 == [0, 3, 6, 9] and free1 == [0, 1, 2]
    assert np.array_equal((A1 @ v[:3]).reshape(4, 3), np.repeat(v[None, :3], 4, 0))
    G0 = B.surface_cp_regu_operator([3, 4], 0).toarray()
    net = rng.standard_normal((4, 3))                                   # [j][i]
    assert np.allclose((G0 @ net.ravel()).reshape(2, 4), (net[:, 1:] - net[:, :-1]).T)
    # the class on two patches of the T-beam: both fields' chains reproduce the analysis control points
    spec = G.tbeam_2patch(4)
    pre = IntersectionData(patches=spec.patches, mapping_list=[[i.a, i.b] for i in spec.interfaces], intersections_para_coords=[[i.xi_a, i.xi_b] for i in spec.interfaces])
    d2a = B.CPSurfDesign2Analysis(pre, opt_field=[0, 2], shopt_surf_inds=[[1], [0, 1]])
    lin = [[0.0, 0.0, 1.0, 1.0], [0.0, 0.0, 1.0, 1.0]]
    cub = [[0.0] * 4 + [1.0] * 4, [0.0] * 4 + [1.0] * 4]
    d2a.set_init_knots([0, 1], [[1, 1], [1, 1]], [lin, lin])
    d2a.set_order_elevation([0, 1], [[3, 3], [3, 3]], [cub, cub])
    d2a.set_knot_refinement()
    d2a.get_init_cp_coarse()
    for f in range(2):                                                  # flat bilinear patches: the coarse nets reproduce them exactly
        Aall = d2a.knot_refine_operator_list[f].tocsr() @ d2a.order_ele_operator_list[f].tocsr()
        assert np.abs(Aall @ d2a.init_cp_coarse[f] - d2a.init_analysis_cp[f]).max() < 1e-10
    d2a.set_cp_align(2, [None, 0])
    assert d2a.init_cp_design[1].size == 4 + 2
    pin = d2a.set_cp_pin(2, [1, None], [[0], None])
    assert pin.shape == (2, 6) and np.abs(pin @ d2a.init_cp_design[1] - np.asarray(d2a.cp_coarse_pin_vals[1])).max() == 0.0
    regu = d2a.set_cp_regu(2, [1, None])
    assert regu.shape == (2, 6)
    d2a.set_cp_align(0, [[0, 1]])
    comps = [CPSurfAlignComp(cpdesign2analysis=d2a), CPSurfOrderElevationComp(cpdesign2analysis=d2a), CPSurfKnotRefinementComp(cpdesign2analysis=d2a, output_cp_fine_name_pre='CP_IGA'),
             CPSurfPinComp(cpdesign2analysis=d2a, output_cp_pin_name_pre='CP_pin'), CPSurfReguComp(cpdesign2analysis=d2a)]
    for c_ in comps:
        c_.init_parameters()
        prob = om.Problem(model=c_)
        prob.setup(); prob.run_model()
        assert max(prob.check_partials(compact_print=False).values()) < 1e-6
    prob = om.Problem(model=comps[3]); prob.setup(); prob.run_model()
    assert np.abs(np.ravel(prob.get_val('CP_pin2'))).max() == 0.0       # feasible at the initial design


def test_partition_following_elimination_tree():
    """goldfish_amd/_dsolver.py: partition_tree (round 5) -- the elimination tree of the distributed factorisation follows the patch partition, so that a rank's
    subtrees eliminate only control points that rank owns (its handle reads the K it assembled; no replicated K): the control points on rank boundaries (the
    lower-rank end of every edge between two ranks) are eliminated along the hierarchy of the rank bisection, everything else by the owner's own nested
    dissection.  Checked for 1, 2, 3, 4, 8 ranks on a random planar graph: every control point eliminated once, a subtree front holds one rank's control points
    only, parents are later fronts and never another rank's, and the dense statement of the numeric phase (_nd.multifrontal_reference_solve) on that tree solves
    a random SPD system to round-off."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    from goldfish_amd import _nd, _dsolver
    rng = np.random.default_rng(5)
    ncp = 700
    pts = rng.uniform(0, 1, (ncp, 3)) * [1.0, 0.8, 0.02]
    adj = np.abs(pts[:, None, :2] - pts[None, :, :2]).max(-1) < 0.07
    nb_lists = [np.flatnonzero(adj[a]) for a in range(ncp)]
    nb_ptr = np.concatenate([[0], np.cumsum([len(x) for x in nb_lists])]).astype(np.int64)
    nb = np.concatenate(nb_lists).astype(np.int32)
    rows = np.repeat(np.arange(ncp), np.diff(nb_ptr))
    B = sp.csr_matrix((rng.uniform(0.1, 1.0, nb.size), (rows, nb)), shape=(ncp, ncp))
    B = B + B.T
    K = sp.kron(B, np.ones((3, 3))).tocsr()
    K = K + sp.diags(np.asarray(abs(K).sum(1)).ravel() + 1.0)
    b = rng.standard_normal(3 * ncp)
    xref = spl.spsolve(K.tocsc(), b)
    flops_global = _nd.nested_dissection(nb_ptr, nb, pts, leaf=40).stats()["flops"]
    for world in (1, 2, 3, 4, 8):
        nx = 2 if world >= 4 else world
        gx = np.minimum((pts[:, 0] * nx).astype(int), nx - 1)
        gy = np.minimum((pts[:, 1] / 0.8 * (world // 2)).astype(int), world // 2 - 1) if world >= 4 else 0
        owner_cp = gx * (world // 2 if world >= 4 else 1) + gy
        sym, owner, roots = _dsolver.partition_tree(nb_ptr, nb, pts, owner_cp, world, leaf=40, native=(world % 2 == 0))
        assert np.array_equal(np.sort(sym.elim), np.arange(ncp)) and set(np.unique(owner)) <= set(range(-1, world))
        for t in range(sym.nfronts):
            e, p = sym.elim[sym.elim_off[t]:sym.elim_off[t + 1]], sym.parent[t]
            bd = sym.bnd[sym.bnd_off[t]:sym.bnd_off[t + 1]]
            assert p < 0 or p > t
            assert np.all(np.diff(sym.order[bd]) > 0) and (bd.size == 0 or sym.front_of[bd[0]] == p)
            if owner[t] >= 0:
                assert np.all(owner_cp[e] == owner[t]) and (p < 0 or owner[p] in (owner[t], -1))
                assert (t in roots) == (p < 0 or owner[p] < 0)
            else:
                assert p < 0 or owner[p] < 0
        assert int((owner < 0).sum()) <= max(world - 1, 0)                       # one separator front per node of the rank hierarchy at most
        x = _nd.multifrontal_reference_solve(sym, K, b)
        assert np.abs(x - xref).max() < 1e-12 * np.abs(xref).max()
        assert sym.stats()["flops"] < 1.5 * flops_global                          # following the partition costs little against the free dissection
