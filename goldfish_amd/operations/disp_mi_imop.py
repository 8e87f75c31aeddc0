"""DispMintImOpeartion -- implicit operation for the displacement states with moving intersections
(reference: GOLDFISH/operations/disp_mi_imop.py:3-126; the class name keeps the reference's spelling).
The last entry of ``d_inputs_array_list`` is the vector of intersection parametric coordinates."""
import numpy as np

from .. import _lib
from .disp_imop import DispImOpeartion


class DispMintImOpeartion(DispImOpeartion):

    def __init__(self, nonmatching_opt, save_files=False):
        super().__init__(nonmatching_opt)
        self.save_files = save_files

    def linearize(self):
        """disp_mi_imop.py:34-42: dR/du and dR/dCP_f in one device pass, dR/dxi from the mortar-vertex kernel."""
        self.nonmatching_opt._assemble(_lib.ASM_K | _lib.ASM_DRDCP)
        self._dRigadxi = None                         # the matrix itself is built when somebody asks for it (forward products, tests): reverse products do not need it
        self._lin_dev = self.nonmatching_opt.dev
        self._lin_state = getattr(self.nonmatching_opt, "_state_version", 0)

    @property
    def dRigadxi(self):
        """d R_IGA / d xi as a sparse matrix (disp_mi_imop.py:40; one host copy of the per-vertex blocks)."""
        sv = (getattr(self.nonmatching_opt, "_state_version", 0), id(self.nonmatching_opt.dev))
        if getattr(self, "_dRigadxi", None) is None or getattr(self, "_dRigadxi_state", None) != sv:      # always the matrix of the CURRENT state, like the device-side reverse product
            self._dRigadxi, self._dRigadxi_state = self.nonmatching_opt.dRIGAdxi(), sv
        return self._dRigadxi

    def stale(self):
        """True when the device model was re-created or its state changed (update_transfer_matrices patches the vertex tables of a moved interface in place) after
        the last linearize: the matrices of that linearisation are gone."""
        nm = self.nonmatching_opt
        return getattr(self, "_lin_dev", None) is not nm.dev or getattr(self, "_lin_state", None) != getattr(nm, "_state_version", 0)

    def apply_linear_fwd(self, d_inputs_array_list=None, d_outputs_array=None, d_residuals_array=None):
        """disp_mi_imop.py:44-73."""
        nm, dev = self.nonmatching_opt, self.nonmatching_opt.dev
        if d_residuals_array is not None:
            acc = np.zeros(nm.vec_iga_dof)
            if d_outputs_array is not None:
                dev.apply(_lib.MAT_K, d_outputs_array, acc)
            if d_inputs_array_list is not None:
                for i, field in enumerate(self.opt_field):
                    dev.apply(_lib.MAT_DRDCP0 + field, self._cp_full(i, d_inputs_array_list[i]), acc)
                acc += self.dRigadxi @ np.asarray(d_inputs_array_list[-1], float)
            d_residuals_array[:] += acc
        return d_residuals_array

    def apply_linear_rev(self, d_inputs_array_list=None, d_outputs_array=None, d_residuals_array=None):
        """disp_mi_imop.py:75-104."""
        nm, dev = self.nonmatching_opt, self.nonmatching_opt.dev
        if d_residuals_array is not None:
            dres = np.ascontiguousarray(d_residuals_array, float)
            if d_outputs_array is not None:
                acc = np.zeros(nm.vec_iga_dof)
                dev.apply(_lib.MAT_K, dres, acc, transpose=True)
                d_outputs_array[:] += acc
            if d_inputs_array_list is not None:
                for i, field in enumerate(self.opt_field):
                    acc = np.zeros(nm.vec_scalar_iga_dof)
                    dev.apply(_lib.MAT_DRDCP0 + field, dres, acc, transpose=True)
                    d_inputs_array_list[i][:] += acc[nm._shopt_cols[i]]
                d_inputs_array_list[-1][:] += nm.dRIGAdxi_rev(dres)               # formed on the device: 6 doubles per mortar vertex come back
        return d_inputs_array_list, d_outputs_array
