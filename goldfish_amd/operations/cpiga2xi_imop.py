"""CPIGA2XiImOperation -- implicit operation xi(CP) of the moving intersections
(reference: GOLDFISH/operations/cpiga2xi_imop.py:19-91; method names and in-place semantics of the reference).
Host-side: the systems have four unknowns per mortar vertex."""
from scipy.sparse import csc_matrix
from scipy.sparse.linalg import splu


class CPIGA2XiImOperation(object):

    def __init__(self, nonmatching_opt):
        self.nonmatching_opt = nonmatching_opt
        self.preprocessor = nonmatching_opt.preprocessor
        self.cpiga2xi = nonmatching_opt.cpiga2xi
        self.opt_field = nonmatching_opt.opt_field

    def apply_nonlinear(self, xi_flat):
        """Residual of the intersection equations (:26-27)."""
        return self.cpiga2xi.residual(xi_flat)

    def solve_nonlinear(self, xi_flat_init):
        """Root from an initial guess (:29-30)."""
        return self.cpiga2xi.solve_xi(xi_flat_init)

    def linearize(self, xi_flat, coo=True):
        """Jacobians wrt xi and wrt every optimised coordinate field, LU of both dR/dxi and its transpose (:32-44)."""
        c2x = self.cpiga2xi
        self.dRdxi_mat = c2x.dRdxi(xi_flat, coo=True).tocsr()
        self.dRdCP_mat_list = [c2x.dRdCP(xi_flat, f, coo=True).tocsr() for f in self.opt_field]
        self.lu_fwd, self.lu_rev = splu(csc_matrix(self.dRdxi_mat)), splu(csc_matrix(self.dRdxi_mat.T))
        return self.dRdxi_mat, self.dRdCP_mat_list

    def _pairs(self, d_inputs_array_list, d_outputs_array):
        """(Jacobian block, direction) pairs that are present: the state block first, then the fields in opt_field order."""
        pairs = [] if d_outputs_array is None else [(self.dRdxi_mat, d_outputs_array)]
        if d_inputs_array_list is not None:
            pairs += list(zip(self.dRdCP_mat_list, d_inputs_array_list))
        return pairs

    def apply_linear_fwd(self, d_inputs_array_list=None, d_outputs_array=None, d_residuals_array=None):
        """d_residuals += dR/dxi d_xi + sum_f dR/dCP_f d_cp_f (:46-62)."""
        if d_residuals_array is not None:
            for J, dx in self._pairs(d_inputs_array_list, d_outputs_array):
                d_residuals_array[:] += J @ dx
        return d_residuals_array

    def apply_linear_rev(self, d_inputs_array_list=None, d_outputs_array=None, d_residuals_array=None):
        """d_xi += (dR/dxi)^T d_res, d_cp_f += (dR/dCP_f)^T d_res (:64-80)."""
        if d_residuals_array is not None:
            for J, dx in self._pairs(d_inputs_array_list, d_outputs_array):
                dx[:] += J.T @ d_residuals_array
        return d_inputs_array_list, d_outputs_array

    def solve_linear_fwd(self, d_outputs_array, d_residuals_array):
        d_outputs_array[:] = self.lu_fwd.solve(d_residuals_array)
        return d_outputs_array

    def solve_linear_rev(self, d_outputs_array, d_residuals_array):
        d_residuals_array[:] = self.lu_rev.solve(d_outputs_array)
        return d_residuals_array
