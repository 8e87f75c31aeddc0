"""ctypes binding of libgoldfish_solver.so (include/goldfish_solver.h): direct solves with K that stay on the device.

Replaces the per-call MUMPS factorisation of GOLDFISH/utils/opt_utils.py:156-209 (solve_Ax_b / solve_ATx_b): the control
points are renumbered once on the host (reverse Cuthill-McKee on the neighbour graph, whose pattern never changes during
an optimisation), K's values are read in place from libgoldfish_hip's buffer, and factorisation (block-banded L D L^T,
64 x 64 tiles on the FP64 matrix pipe), substitutions and iterative refinement run on the GPU -- hand-written HIP, no
rocSOLVER / rocBLAS.  No CPU fallback: raises when the library or a GPU is missing."""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GF_SOLVER_LIB", os.path.join(_HERE, "libgoldfish_solver.so"))   # GF_SOLVER_LIB: A/B builds while tuning
EXPORTS = ["gfs_last_error", "gfs_create", "gfs_create_nd", "gfs_destroy", "gfs_refactor", "gfs_prepare_refactor", "gfs_solve", "gfs_solve_dev", "gfs_set_general", "gfs_solve_transposed",
           "gfs_solve_transposed_dev", "gfs_info", "gfs_solve_multi", "gfs_solve_multi_dev", "gfs_create_nd_partial", "gfs_schur_doubles", "gfs_export_schur",
           "gfs_set_schur_source", "gfs_get_fbnd", "gfs_set_fbnd", "gfs_get_fbnd_packed", "gfs_set_fbnd_packed", "gfs_set_row_mask", "gfs_x_ptr", "gfs_forward_dev", "gfs_backward_dev", "gfs_symbolic_create", "gfs_symbolic_sizes",
           "gfs_symbolic_copy", "gfs_symbolic_destroy"]
_L = None


def lib():
    global _L
    if _L is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgoldfish_solver.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
        from ._lib import one_hip_runtime
        one_hip_runtime()                                      # one HIP runtime per process: torch's, when torch is installed (see _lib.one_hip_runtime)
        L = C.CDLL(LIB_PATH)
        i32p, i64p, dp, vp = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_void_p
        L.gfs_last_error.restype = C.c_char_p
        L.gfs_create.argtypes = [C.c_int, C.c_int64, i64p, i32p, i32p, vp, C.POINTER(vp)]
        L.gfs_create_nd.argtypes = [C.c_int, C.c_int64, i64p, i32p, vp, C.c_int64] + [i64p] * 8 + [C.POINTER(vp)]
        L.gfs_destroy.argtypes = [vp]
        L.gfs_destroy.restype = None
        L.gfs_refactor.argtypes = [vp]
        L.gfs_prepare_refactor.argtypes = [vp]
        L.gfs_solve.argtypes = [vp, dp, dp, C.c_int, dp]
        L.gfs_solve_dev.argtypes = [vp, vp, vp, C.c_int, dp]
        L.gfs_set_general.argtypes = [vp, C.c_int]
        L.gfs_solve_transposed.argtypes = [vp, dp, dp, C.c_int, dp]
        L.gfs_solve_transposed_dev.argtypes = [vp, vp, vp, C.c_int, dp]
        L.gfs_info.argtypes = [vp, dp]
        L.gfs_solve_multi.argtypes = [vp, C.c_int, dp, dp, C.c_int, dp, C.c_int]
        L.gfs_solve_multi_dev.argtypes = [vp, C.c_int, vp, vp, C.c_int, dp, C.c_int]
        _L = L
    return _L


def control_point_graph(rowptr, col):
    """Control-point-level neighbour lists (nb_ptr, nb) from the dof-level pattern of K (gf_pattern(GF_MAT_K): row 3 a + i
    lists the columns 3 b + j of a's neighbours b in ascending order)."""
    rowptr, col = np.asarray(rowptr, np.int64), np.asarray(col, np.int64)
    ncp = (rowptr.size - 1) // 3
    starts, ends = rowptr[0:3 * ncp:3], rowptr[1:3 * ncp + 1:3]
    deg = (ends - starts) // 3
    nb_ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    idx = np.repeat(starts, deg) + 3 * (np.arange(nb_ptr[-1]) - np.repeat(nb_ptr[:-1], deg))
    return nb_ptr, (col[idx] // 3).astype(np.int32)


def half_bandwidth(nb_ptr, nb, new_index):
    """Largest distance (in control points) between two coupled control points in the order ``new_index``."""
    rows = np.repeat(np.arange(nb_ptr.size - 1), np.diff(nb_ptr))
    return int(np.abs(new_index[rows].astype(np.int64) - new_index[nb].astype(np.int64)).max()) if nb.size else 0


def bandwidth_reducing_order(nb_ptr, nb, coords=None):
    """new_index[a] = position of control point a in the factorisation order: the candidate with the smallest bandwidth among
    reverse Cuthill-McKee on the neighbour graph and -- when the control points' coordinates are given -- plane sweeps along the
    principal axes of the point cloud.  RCM starts at a corner of a plate-like model and sweeps it diagonally (C4: 5.6 k control
    points of half bandwidth); the sweep along the long axis has straight fronts (2.5 k), i.e. less than half the band memory and
    a fifth of the factorisation work."""
    ncp = nb_ptr.size - 1
    G = sp.csr_matrix((np.ones(nb.size, np.int8), nb, nb_ptr), shape=(ncp, ncp))
    perm = reverse_cuthill_mckee(G, symmetric_mode=True)           # perm[new] = old
    best = np.empty(ncp, np.int32)
    best[perm] = np.arange(ncp, dtype=np.int32)
    if coords is not None and ncp > 1:
        X = np.asarray(coords, float).reshape(ncp, -1)
        X = X - X.mean(0)
        _, _, Vt = np.linalg.svd(X[:: max(1, ncp // 200000)], full_matrices=False)     # principal axes (a sample is enough)
        bw = half_bandwidth(nb_ptr, nb, best)
        for ax in range(Vt.shape[0]):
            proj = X @ Vt[ax]
            other = X @ Vt[(ax + 1) % Vt.shape[0]]
            order = np.lexsort((other, proj))                      # primary key: position along the axis
            cand = np.empty(ncp, np.int32)
            cand[order] = np.arange(ncp, dtype=np.int32)
            b = half_bandwidth(nb_ptr, nb, cand)
            if b < bw:
                best, bw = cand, b
    return best


def parent_positions(sym):
    """For every boundary control point of every front: its position in the parent front's numbering (the parent's eliminated control
    points first, then the parent's boundary) -- the index map of the extend-add (gfs_create_nd: ``pmap``)."""
    pmap = np.zeros(sym.bnd.size, np.int64)
    ne = np.diff(sym.elim_off)
    for t in range(sym.nfronts):
        p = sym.parent[t]
        b = sym.bnd[sym.bnd_off[t]:sym.bnd_off[t + 1]]
        if p < 0 or b.size == 0:
            continue
        ob = sym.order[b]
        inside = sym.front_of[b] == p
        pb = sym.bnd[sym.bnd_off[p]:sym.bnd_off[p + 1]]
        pos = np.where(inside, ob - sym.elim_off[p], ne[p] + np.searchsorted(sym.order[pb], ob))
        pmap[sym.bnd_off[t]:sym.bnd_off[t + 1]] = pos
    return pmap


ND_MIN_CP = int(os.environ.get("GF_SOLVER_ND_MIN_CP", "5000"))     # models with more control points factor by nested dissection (when coordinates are given)


class DeviceSolver:
    """K x = b and K^T x = b with the K of a goldfish_amd._lib.DeviceModel; factors resident in HBM.  ``general`` (a K that is not symmetric: the
    load stiffness of a follower pressure): the symmetric part is factored and preconditions the refinement against K / K^T itself
    (gfs_set_general); ``max_refine`` then defaults to 40 steps instead of 3 (the refinement is the solver for the skew part and stops by itself when the
    residual no longer drops).
    ``method``: "skyline" (block skyline after RCM: small and medium models), "nd" (nested-dissection multifrontal: goldfish_amd/_nd.py +
    gfs_create_nd; needs the control points' coordinates), "auto": nd above ND_MIN_CP control points."""

    def __init__(self, dev_model, max_refine=None, coords=None, method="auto", leaf=128, general=False):
        from . import _lib
        self.general = bool(general)
        self.D, self.max_refine = dev_model, (40 if general else 3) if max_refine is None else max_refine
        if hasattr(dev_model, "cp_graph"):                      # the library's own control-point-level lists (a ninth of the dof-level pattern)
            self.nb_ptr, self.nb = dev_model.cp_graph()
        else:                                                   # sharded model: the global dof-level pattern gathered from the ranks
            rowptr, col = dev_model.pattern(_lib.MAT_K)
            self.nb_ptr, self.nb = control_point_graph(rowptr, col)
            del rowptr, col
        ncp = self.nb_ptr.size - 1
        self.n = 3 * ncp
        if method == "auto":
            method = "nd" if (coords is not None and ncp >= ND_MIN_CP) else "skyline"
        self.method = method
        # K's values on this device in the layout of dev_model.pattern(MAT_K): the library's own buffer, or (sharded model) the replicated global K
        dK = dev_model.k_values_ptr()
        h = C.c_void_p()
        i64 = lambda a: np.ascontiguousarray(a, np.int64).ctypes.data_as(C.POINTER(C.c_int64))
        if method == "nd":
            from . import _nd
            if coords is None:
                raise ValueError("DeviceSolver(method='nd') needs the control points' coordinates")
            leaf = int(os.environ.get("GF_SOLVER_LEAF", leaf))            # measurement switch (tools/solver_bench.py)
            if os.environ.get("GF_ND", "native") == "python":       # the NumPy statement of the symbolic phase (what the tests compare the native one against)
                self.sym = sym = _nd.nested_dissection(self.nb_ptr, self.nb, coords, leaf=leaf)
                pmap = parent_positions(sym)
            else:
                self.sym, pmap = _nd.nested_dissection_native(self.nb_ptr, self.nb, coords, leaf=leaf)
                sym = self.sym
            keep = [np.ascontiguousarray(a, np.int64) for a in (sym.elim, sym.elim_off, sym.bnd, sym.bnd_off, sym.parent, sym.order, sym.front_of, pmap)]
            rc = lib().gfs_create_nd(int(dev_model.device), ncp, self.nb_ptr.ctypes.data_as(C.POINTER(C.c_int64)), self.nb.ctypes.data_as(C.POINTER(C.c_int32)),
                                     C.c_void_p(dK), sym.nfronts, *[i64(a) for a in keep], C.byref(h))
        else:
            self.new_index = bandwidth_reducing_order(self.nb_ptr, self.nb, coords)
            rc = lib().gfs_create(int(dev_model.device), ncp, self.nb_ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                  self.nb.ctypes.data_as(C.POINTER(C.c_int32)), self.new_index.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(dK), C.byref(h))
        if rc:
            raise RuntimeError(lib().gfs_last_error().decode())
        self.h = h
        self.rel_residual = self.backward_error = None
        self.small_pivot = False
        if self.general and lib().gfs_set_general(self.h, 1):
            raise RuntimeError(lib().gfs_last_error().decode())
        self.refactor()                                       # numeric factors of the current K

    def close(self):
        if getattr(self, "h", None):
            lib().gfs_destroy(self.h)
            self.h = None

    __del__ = close

    def prepare(self):
        """The factors at hand will not be used again: start clearing the factor storage now (gfs_prepare_refactor: 9 - 11 ms of HBM writes at C4 that the next refactor()
        then skips).  Pays when the device is idle until then; beside an assembly pass it does not (include/goldfish_solver.h has the measurement)."""
        if lib().gfs_prepare_refactor(self.h):
            raise RuntimeError(lib().gfs_last_error().decode())

    def refactor(self):
        """Numeric factorisation of the values of K currently on the device (after a new assembly)."""
        self.D.sync()                                         # the assembly runs on the model's stream, the solver on its own
        if hasattr(self.D, "refresh_k_values"):                 # sharded model: gather the owned value rows of all ranks into the replicated K (collective)
            self.D.refresh_k_values()
        if lib().gfs_refactor(self.h):
            raise RuntimeError(lib().gfs_last_error().decode())

    def solve(self, b, transpose=False, max_refine=None):
        """``max_refine``: refinement sweeps for this call (default: the solver's; 0 = substitutions only -- a Newton correction does not need the last decade of the
        linear residual, and every sweep reads the factors twice: C4 0.048 s instead of 0.13 s)."""
        b = np.ascontiguousarray(b, float)
        if b.size != self.n:
            raise ValueError("DeviceSolver.solve: expected %d values, got %d" % (self.n, b.size))
        x, rr = np.empty(self.n), C.c_double(0.0)
        dp = C.POINTER(C.c_double)
        fn = lib().gfs_solve_transposed if transpose else lib().gfs_solve
        if fn(self.h, b.ctypes.data_as(dp), x.ctypes.data_as(dp), int(self.max_refine if max_refine is None else max_refine), C.byref(rr)):
            raise RuntimeError(lib().gfs_last_error().decode())
        self.rel_residual = rr.value
        inf = self.info()
        self.backward_error, self.small_pivot = inf["backward_error"], inf["small_pivot"]
        return x

    MAX_RHS = 8

    def solve_multi(self, B, transpose=False, max_refine=None):
        """X[k] = K^{-1} B[k] (or K^{-T} B[k]) for the rows of B in ONE call (gfs_solve_multi): in the nested-dissection mode groups of three right-hand sides
        share one pass over the factors (the sweeps are bound by the factor bytes) and the groups run next to each other on the device; the result of a row is
        bitwise the one ``solve`` gives; ``rel_residuals`` holds one value per row, ``backward_error`` the largest."""
        B = np.ascontiguousarray(np.atleast_2d(B), float)
        if B.shape[1] != self.n:
            raise ValueError("DeviceSolver.solve_multi: expected rows of %d values, got %d" % (self.n, B.shape[1]))
        X = np.empty_like(B)
        rr_all = []
        dp = C.POINTER(C.c_double)
        for k0 in range(0, B.shape[0], self.MAX_RHS):
            blk = np.ascontiguousarray(B[k0:k0 + self.MAX_RHS])
            out, rr = np.empty_like(blk), np.zeros(blk.shape[0])
            if lib().gfs_solve_multi(self.h, blk.shape[0], blk.ctypes.data_as(dp), out.ctypes.data_as(dp), int(self.max_refine if max_refine is None else max_refine), rr.ctypes.data_as(dp), int(bool(transpose))):
                raise RuntimeError(lib().gfs_last_error().decode())
            X[k0:k0 + blk.shape[0]] = out
            rr_all.append(rr)
            inf = self.info()
            self.backward_error = inf["backward_error"] if k0 == 0 else max(self.backward_error, inf["backward_error"])
            self.small_pivot = inf["small_pivot"]
        self.rel_residuals = np.concatenate(rr_all)
        self.rel_residual = float(self.rel_residuals.max())
        return X

    def info(self):
        v = (C.c_double * 8)()
        lib().gfs_info(self.h, v)
        return {"half_bandwidth": int(v[0]), "block_columns": int(v[1]), "tiles_per_block_row": int(v[2]), "device_bytes": int(v[3]),
                "factor_flops": float(v[4]), "small_pivot": bool(v[5]), "backward_error": float(v[6]), "norm_K": float(v[7])}

    @property
    def device_bytes(self):
        return self.info()["device_bytes"]
