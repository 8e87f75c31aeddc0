"""CPIGA2XiImOperation -- implicit operation xi(CP) of the moving intersections
(reference: GOLDFISH/operations/cpiga2xi_imop.py:19-91; same method names and in-place semantics).
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
        """cpiga2xi_imop.py:26-27."""
        return self.cpiga2xi.residual(xi_flat)

    def solve_nonlinear(self, xi_flat_init):
        """cpiga2xi_imop.py:29-30."""
        return self.cpiga2xi.solve_xi(xi_flat_init)

    def linearize(self, xi_flat, coo=True):
        """cpiga2xi_imop.py:32-44: dR/dxi, dR/dCP_f and the two LU factorisations."""
        self.dRdxi_mat = self.cpiga2xi.dRdxi(xi_flat, coo=True).tocsr()
        self.dRdCP_mat_list = [self.cpiga2xi.dRdCP(xi_flat, field, coo=True).tocsr() for field in self.opt_field]
        self.lu_fwd = splu(csc_matrix(self.dRdxi_mat))
        self.lu_rev = splu(csc_matrix(self.dRdxi_mat.T))
        return self.dRdxi_mat, self.dRdCP_mat_list

    def apply_linear_fwd(self, d_inputs_array_list=None, d_outputs_array=None, d_residuals_array=None):
        """cpiga2xi_imop.py:46-62."""
        if d_residuals_array is not None:
            if d_outputs_array is not None:
                d_residuals_array[:] += self.dRdxi_mat @ d_outputs_array
            if d_inputs_array_list is not None:
                for i, field in enumerate(self.opt_field):
                    d_residuals_array[:] += self.dRdCP_mat_list[i] @ d_inputs_array_list[i]
        return d_residuals_array

    def apply_linear_rev(self, d_inputs_array_list=None, d_outputs_array=None, d_residuals_array=None):
        """cpiga2xi_imop.py:64-80."""
        if d_residuals_array is not None:
            if d_outputs_array is not None:
                d_outputs_array[:] += self.dRdxi_mat.T @ d_residuals_array
            if d_inputs_array_list is not None:
                for i, field in enumerate(self.opt_field):
                    d_inputs_array_list[i][:] += self.dRdCP_mat_list[i].T @ d_residuals_array
        return d_inputs_array_list, d_outputs_array

    def solve_linear_fwd(self, d_outputs_array, d_residuals_array):
        """cpiga2xi_imop.py:82-84."""
        d_outputs_array[:] = self.lu_fwd.solve(d_residuals_array)
        return d_outputs_array

    def solve_linear_rev(self, d_outputs_array, d_residuals_array):
        """cpiga2xi_imop.py:86-88."""
        d_residuals_array[:] = self.lu_rev.solve(d_outputs_array)
        return d_residuals_array
