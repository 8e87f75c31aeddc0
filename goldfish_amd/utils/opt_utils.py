"""NumPy/SciPy counterparts of the PETSc helpers in GOLDFISH/utils/opt_utils.py that callers of the problem classes use:
vectors are plain ndarrays and matrices scipy.sparse here, so most of them are trivial.  The state solves of the hot path do
NOT go through these functions: they use NonMatchingOpt.solve_K (host SuperLU or the device re-factorisation)."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def array2petsc_vec(ndarray, comm=None):
    """opt_utils.py:9-26."""
    return np.array(ndarray, float)


def get_petsc_vec_array(petsc_vec, comm=None):
    """opt_utils.py:28-54."""
    return np.asarray(petsc_vec, float)


def update_nest_vec(vec_array, nest_vec, comm=None):
    """opt_utils.py:70-104: in-place copy."""
    nest_vec[:] = np.asarray(vec_array, float)
    return nest_vec


def solve_Ax_b(A, b, array=False, comm=None):
    """opt_utils.py:156-181 (MUMPS LU there): sparse direct solve of A x = b on the host."""
    return spla.splu(sp.csc_matrix(A)).solve(np.asarray(b, float))


def solve_ATx_b(A, b, array=False, comm=None):
    """opt_utils.py:183-209: A^T x = b."""
    return spla.splu(sp.csc_matrix(A).T.tocsc()).solve(np.asarray(b, float))


def PETSc_ksp_solve(A, x, b, ksp_type="cg", pc_type="jacobi", max_it=10000, rtol=1e-15):
    """opt_utils.py:104-131 (KSP cg + pc, defined there and never called by the reference).  ``A``: a goldfish_amd._lib.DeviceModel whose K is assembled -- the
    iteration then runs on the device (goldfish_amd/_krylov.py: DevicePCG, products through gf_apply_dev) -- or a scipy.sparse matrix (host CG, the same
    preconditioners).  ``x`` is overwritten and returned.  ``pc_type``: "jacobi" | "bjacobi" | "none".  Raises if ``ksp_type`` is not "cg" (K is symmetric
    positive definite on this path); warns when the iteration ends without reaching ``rtol`` (the tangents of penalty-coupled thin shells have cond(K) ~ 1e12+:
    see _krylov.py for what was measured)."""
    import warnings
    if ksp_type != "cg":
        raise NotImplementedError("PETSc_ksp_solve: only ksp_type='cg' is provided")
    b = np.asarray(b, float)
    if hasattr(A, "k_values_ptr"):
        from .._krylov import DevicePCG
        S = DevicePCG(A, pc_type=pc_type)
        sol = S.solve(b, rtol=rtol, max_it=max_it)
        it, conv, rr = S.iterations, S.converged, S.rel_residual
    else:
        A = sp.csr_matrix(A)
        if pc_type == "jacobi":
            d = 1.0 / A.diagonal()
            M = spla.LinearOperator(A.shape, lambda r: d * r)
        elif pc_type == "bjacobi":
            nb = A.shape[0] // 3
            blocks = np.stack([A[3 * k:3 * k + 3, 3 * k:3 * k + 3].toarray() for k in range(nb)])
            inv = np.linalg.inv(blocks)
            M = spla.LinearOperator(A.shape, lambda r: np.einsum("kij,kj->ki", inv, np.asarray(r).reshape(-1, 3)).ravel())
        elif pc_type == "none":
            M = None
        else:
            raise ValueError("PETSc_ksp_solve: pc_type must be 'jacobi', 'bjacobi' or 'none'")
        count = [0]
        sol, info = spla.cg(A, b, rtol=rtol, atol=0.0, maxiter=max_it, M=M, callback=lambda xk: count.__setitem__(0, count[0] + 1))
        it, conv = count[0], info == 0
        rr = float(np.linalg.norm(b - A @ sol) / max(np.linalg.norm(b), 1e-300))
    if not conv:
        warnings.warn("PETSc_ksp_solve: cg + %s ended after %d iterations at |b - A x| / |b| = %.3e > rtol %.1e" % (pc_type, it, rr, rtol), RuntimeWarning)
    x[:] = sol
    return x
