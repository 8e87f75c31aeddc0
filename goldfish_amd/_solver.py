"""ctypes binding of libgoldfish_solver.so (include/goldfish_solver.h): direct solves with K that stay on the device.

Replaces the per-call MUMPS factorisation of GOLDFISH/utils/opt_utils.py:156-209 (solve_Ax_b / solve_ATx_b) after the
first solve: ordering + symbolic factorisation once on the host (SuperLU, symmetric mode), then numeric re-factorisation and
triangular solves on the GPU (rocSOLVER csrrf) for every later Newton step and adjoint solve.  No CPU fallback: raises
when the library or a GPU is missing."""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgoldfish_solver.so")
EXPORTS = ["gfs_last_error", "gfs_create", "gfs_destroy", "gfs_refactor", "gfs_solve", "gfs_nnz_factors", "gfs_device_bytes"]
_L = None


def lib():
    global _L
    if _L is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgoldfish_solver.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
        L = C.CDLL(LIB_PATH)
        i32p, dp, vp = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p
        L.gfs_last_error.restype = C.c_char_p
        L.gfs_create.argtypes = [C.c_int, C.c_int64, C.c_int64, i32p, i32p, vp, C.c_int64, i32p, i32p, dp, i32p, i32p, C.POINTER(vp)]
        L.gfs_destroy.argtypes = [vp]
        L.gfs_refactor.argtypes = [vp]
        L.gfs_solve.argtypes = [vp, dp, dp]
        L.gfs_nnz_factors.restype = C.c_int64
        L.gfs_nnz_factors.argtypes = [vp]
        L.gfs_device_bytes.restype = C.c_int64
        L.gfs_device_bytes.argtypes = [vp]
        _L = L
    return _L


def host_symbolic(K, seed=0):
    """Ordering + symbolic factorisation on the host.  Returns (T, pivP, pivQ): the structural pattern of (L - I) + U as CSR
    and the row / column orders of P K Q = L U in rocSOLVER's convention.

    scipy returns the SuperLU factors without the fill entries that happen to be numerically zero (and K has many exact zeros,
    e.g. membrane-bending coupling of flat patches at u = 0), so the pattern is taken from the factors of a GENERIC symmetric
    positive definite matrix with K's structure (random symmetric off-diagonals, dominant diagonal); the numeric factors of
    the real K are then computed on the device (DeviceSolver calls refactor())."""
    n = K.shape[0]
    P = sp.csr_matrix(K)
    rng = np.random.default_rng(seed)
    G = sp.csr_matrix((rng.uniform(0.5, 1.5, P.nnz), P.indices, P.indptr), shape=(n, n))
    G = G + G.T                                               # symmetric, no cancellations (all positive)
    G.setdiag(0.0)
    G = (G + sp.diags(np.asarray(abs(G).sum(1)).ravel() + 1.0)).tocsc()
    lu = spla.splu(G, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    Lc, Uc = lu.L.tocoo(), lu.U.tocoo()
    off = Lc.row != Lc.col                                   # drop L's unit diagonal structurally
    T = sp.coo_matrix((np.concatenate([Lc.data[off], Uc.data]), (np.concatenate([Lc.row[off], Uc.row]), np.concatenate([Lc.col[off], Uc.col]))),
                      shape=(n, n)).tocsr()
    T.sort_indices()
    # SuperLU: Pr K Pc = L U with Pr[perm_r[i], i] = 1 and Pc[i, perm_c[i]] = 1, i.e. row i of K becomes row perm_r[i] and
    # column j becomes column perm_c[j]; rocSOLVER wants the source index of every permuted row / column: the inverses
    pivP = np.argsort(lu.perm_r).astype(np.int32)
    pivQ = np.argsort(lu.perm_c).astype(np.int32)
    return T, pivP, pivQ


class DeviceSolver:
    """K x = b (= K^T x = b) with the K of a goldfish_amd._lib.DeviceModel, factors resident in HBM."""

    def __init__(self, dev_model, _pivots=None):
        from . import _lib
        self.D = dev_model
        rowptr, col = dev_model.pattern(_lib.MAT_K)
        vals = dev_model.values(_lib.MAT_K)
        n = rowptr.size - 1
        K = sp.csr_matrix((vals, col, rowptr), shape=(n, n))
        T, pivP, pivQ = host_symbolic(K)
        if _pivots is not None:
            pivP, pivQ = _pivots(pivP, pivQ)
        self.n, self.nnzT = n, T.nnz
        ptrA, indA = rowptr.astype(np.int32), np.ascontiguousarray(col, np.int32)
        ptrT, indT, valT = T.indptr.astype(np.int32), T.indices.astype(np.int32), np.ascontiguousarray(T.data, float)
        i32p, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        dK = _lib.lib().gf_device_ptr(dev_model.h, _lib.BUF_VAL_K)
        h = C.c_void_p()
        rc = lib().gfs_create(int(dev_model.device), n, ptrA[-1], ptrA.ctypes.data_as(i32p), indA.ctypes.data_as(i32p), C.c_void_p(dK),
                              T.nnz, ptrT.ctypes.data_as(i32p), indT.ctypes.data_as(i32p), valT.ctypes.data_as(dp),
                              pivP.ctypes.data_as(i32p), pivQ.ctypes.data_as(i32p), C.byref(h))
        if rc:
            raise RuntimeError(lib().gfs_last_error().decode())
        self.h = h
        self.refactor()                                       # numeric factors of the current K, computed on the device

    def close(self):
        if getattr(self, "h", None):
            lib().gfs_destroy(self.h)
            self.h = None

    __del__ = close

    def refactor(self):
        """Numeric re-factorisation with the values of K currently on the device (after a new assembly)."""
        if lib().gfs_refactor(self.h):
            raise RuntimeError(lib().gfs_last_error().decode())

    def solve(self, b):
        b = np.ascontiguousarray(b, float)
        if b.size != self.n:
            raise ValueError("DeviceSolver.solve: expected %d values, got %d" % (self.n, b.size))
        x = np.empty(self.n)
        dp = C.POINTER(C.c_double)
        if lib().gfs_solve(self.h, b.ctypes.data_as(dp), x.ctypes.data_as(dp)):
            raise RuntimeError(lib().gfs_last_error().decode())
        return x

    @property
    def device_bytes(self):
        return lib().gfs_device_bytes(self.h)
