"""DispImOpeartion -- implicit operation for the displacement states
(reference: GOLDFISH/operations/disp_imop.py:3-142; the class name keeps the
reference's spelling).  Same call signatures and in-place semantics; Jacobian
products run on the GPU (csr_apply kernels, one gf_apply_many call per OM call), direct solves on the device
(NonMatchingOpt.solve_K)."""
import numpy as np

from .. import _lib


class DispImOpeartion(object):

    def __init__(self, nonmatching_opt):
        self.nonmatching_opt = nonmatching_opt
        self.comm = nonmatching_opt.comm
        self.opt_shape = nonmatching_opt.opt_shape
        self.opt_field = nonmatching_opt.opt_field
        self.opt_thickness = nonmatching_opt.opt_thickness
        self.var_thickness = nonmatching_opt.var_thickness
        self.use_aero_pressure = nonmatching_opt.use_aero_pressure
        self._lu = None

    # residual / state solve -------------------------------------------------------------
    def apply_nonlinear(self):
        """disp_imop.py:33-36."""
        return self.nonmatching_opt.RIGA()

    def solve_nonlinear(self, max_it=30, rtol=1e-3):
        """disp_imop.py:38-44."""
        _, u = self.nonmatching_opt.solve_nonlinear_nonmatching_problem(
            max_it=max_it, zero_mortar_funcs=True, rtol=rtol, iga_dofs=True)
        return u.copy()

    def linearize(self):
        """disp_imop.py:46-56: dR/du, dR/dCP_f for every opt field, dR/dh -- one fused device pass."""
        flags = _lib.ASM_K
        if self.opt_shape:
            flags |= _lib.ASM_DRDCP
        if self.opt_thickness:
            flags |= _lib.ASM_DRDH
        self.nonmatching_opt._assemble(flags)
        self._lu = None

    # Jacobian-vector products -------------------------------------------------------------
    def _cp_full(self, i, x):
        nm = self.nonmatching_opt
        full = np.zeros(nm.vec_scalar_iga_dof)
        full[nm._shopt_cols[i]] = x
        return full

    def _h_full(self, x):
        nm = self.nonmatching_opt
        return np.asarray(x, float) if self.var_thickness else np.repeat(np.asarray(x, float), nm.vec_scalar_iga_dof_list)

    def apply_linear_fwd(self, d_inputs_array_list=None, d_outputs_array=None, d_residuals_array=None):
        """disp_imop.py:58-97: d_residuals += K du + sum_f dR/dCP_f dcp_f + dR/dh dh -- one device call (gf_apply_many)."""
        dev = self.nonmatching_opt.dev
        if d_residuals_array is not None:
            which, xs = [], []
            if d_outputs_array is not None:
                which.append(_lib.MAT_K); xs.append(d_outputs_array)
            if d_inputs_array_list is not None:
                if self.opt_shape:
                    for i, field in enumerate(self.opt_field):
                        which.append(_lib.MAT_DRDCP0 + field); xs.append(self._cp_full(i, d_inputs_array_list[i]))
                if self.opt_thickness:
                    which.append(_lib.MAT_DRDH); xs.append(self._h_full(d_inputs_array_list[len(self.opt_field)]))
            if which:
                acc = np.zeros(self.nonmatching_opt.vec_iga_dof)
                dev.apply_many(which, xs, [acc])
                d_residuals_array[:] += acc
        return d_residuals_array

    def apply_linear_rev(self, d_inputs_array_list=None, d_outputs_array=None, d_residuals_array=None):
        """disp_imop.py:99-128: d_outputs += K^T d_res, d_inputs[f] += (dR/dCP_f)^T d_res, ... -- one device call."""
        nm, dev = self.nonmatching_opt, self.nonmatching_opt.dev
        if d_residuals_array is not None:
            dres = np.ascontiguousarray(d_residuals_array, float)
            which, ys, sinks = [], [], []
            if d_outputs_array is not None:
                which.append(_lib.MAT_K); ys.append(np.zeros(nm.vec_iga_dof)); sinks.append(("u", None))
            if d_inputs_array_list is not None:
                if self.opt_shape:
                    for i, field in enumerate(self.opt_field):
                        which.append(_lib.MAT_DRDCP0 + field); ys.append(np.zeros(nm.vec_scalar_iga_dof)); sinks.append(("cp", i))
                if self.opt_thickness:
                    which.append(_lib.MAT_DRDH); ys.append(np.zeros(nm.vec_scalar_iga_dof)); sinks.append(("h", None))
            if which:
                dev.apply_many(which, [dres], ys, transpose=True)
            for (kind, i), acc in zip(sinks, ys):
                if kind == "u":
                    d_outputs_array[:] += acc
                elif kind == "cp":
                    d_inputs_array_list[i][:] += acc[nm._shopt_cols[i]]
                else:
                    if not self.var_thickness:
                        acc = np.add.reduceat(acc, nm.cp_off[:-1])
                    d_inputs_array_list[len(self.opt_field)][:] += acc
        return d_inputs_array_list, d_outputs_array

    # direct solves: NonMatchingOpt.solve_K -- host SuperLU or device re-factorisation --------------
    def solve_linear_fwd(self, d_outputs_array, d_residuals_array):
        """disp_imop.py:130-135: d_outputs = K^{-1} d_residuals."""
        d_outputs_array[:] = self.nonmatching_opt.solve_K(d_residuals_array)
        return d_outputs_array

    def solve_linear_rev(self, d_outputs_array, d_residuals_array):
        """disp_imop.py:137-142: d_residuals = K^{-T} d_outputs (K is symmetric including its Dirichlet treatment unless a follower
        pressure contributes its load stiffness)."""
        d_residuals_array[:] = self.nonmatching_opt.solve_K(d_outputs_array, transpose=True)
        return d_residuals_array
