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
