"""CPFE2IGAImOperation -- in the reference the implicit L2 projection ``Mc^T (Mc x_iga - x_fe) = 0`` of the FE control-point
functions onto IGA dofs (GOLDFISH/operations/cpfe2iga_imop.py:63-94).  The IGA control points are the only representation
here, so Mc is the identity: residual x_iga - x_fe, unit Jacobians.  Same method names."""
import numpy as np


class CPFE2IGAImOperation(object):

    def __init__(self, nonmatching_opt):
        self.nonmatching_opt = nonmatching_opt
        self.opt_field = nonmatching_opt.opt_field
        self.opt_shape = nonmatching_opt.opt_shape

    def apply_nonlinear(self, cp_fe_array, cp_iga_array, field=None):
        """cpfe2iga_imop.py:63-77."""
        return np.asarray(cp_iga_array, float) - np.asarray(cp_fe_array, float)

    def solve_nonlinear(self, cp_fe_array, field=None):
        """cpfe2iga_imop.py:79-94."""
        return np.array(cp_fe_array, float)

    def apply_linear_fwd(self, d_inputs_array=None, d_outputs_array=None, d_residuals_array=None, field=None):
        if d_residuals_array is not None:
            if d_outputs_array is not None:
                d_residuals_array[:] += d_outputs_array
            if d_inputs_array is not None:
                d_residuals_array[:] -= d_inputs_array
        return d_residuals_array

    def apply_linear_rev(self, d_inputs_array=None, d_outputs_array=None, d_residuals_array=None, field=None):
        if d_residuals_array is not None:
            if d_outputs_array is not None:
                d_outputs_array[:] += d_residuals_array
            if d_inputs_array is not None:
                d_inputs_array[:] -= d_residuals_array
        return d_inputs_array, d_outputs_array

    def solve_linear_fwd(self, d_outputs_array, d_residuals_array, field=None):
        d_outputs_array[:] = d_residuals_array
        return d_outputs_array

    def solve_linear_rev(self, d_outputs_array, d_residuals_array, field=None):
        d_residuals_array[:] = d_outputs_array
        return d_residuals_array
