"""DispMintStatesComp -- displacement states with moving intersections: R(u; CP_IGA, xi) = 0
(reference: GOLDFISH/om_comps/disp_states_mi_comp.py:6-117; option names, defaults, variable names and shapes of the reference)."""
import numpy as np

from ._design_io import _REQUIRED
from .disp_states_comp import StatesComp
from ..operations.disp_mi_imop import DispMintImOpeartion


class DispMintStatesComp(StatesComp):
    OPTIONS = (('nonmatching_opt', _REQUIRED), ('input_cp_iga_name_pre', 'CP_IGA'), ('input_xi_name', 'int_para'),
               ('output_u_name', 'displacements'))

    def init_parameters(self, save_files=False, nonlinear_solver_rtol=1e-3, nonlinear_solver_max_it=10):
        self._read_options()
        self._counters(save_files, nonlinear_solver_rtol, nonlinear_solver_max_it)
        self.disp_mint_state_imop = self._imop = DispMintImOpeartion(self.nonmatching_opt)
        self._init_design(thickness=False)
        self.input_xi_shape = self.nonmatching_opt.xi_size
        self.init_xi = self.nonmatching_opt.cpiga2xi.xi_flat_global
        self._last_xi = None

    def setup(self):
        self._add_design_inputs(self.output_u_name)
        self.add_input(self.input_xi_name, shape=self.input_xi_shape, val=self.init_xi)
        self._state_setup()
        self.declare_partials(self.output_u_name, self.input_xi_name)

    def _input_names(self):
        return self._design_names() + [self.input_xi_name]

    def update_inputs_outpus(self, inputs, outputs):
        """disp_states_mi_comp.py:54-60 (the reference's spelling); the device model is re-created only when the parametric
        coordinates changed."""
        xi = np.asarray(inputs[self.input_xi_name], float).ravel()
        if self._last_xi is None or not np.array_equal(xi, self._last_xi):
            self.nonmatching_opt.update_xi(xi)
            self.nonmatching_opt.update_transfer_matrices()
            self._last_xi = xi.copy()
        self._push_design(inputs)
        self.nonmatching_opt.update_uIGA(outputs[self.output_u_name])

    def apply_linear(self, inputs, outputs, d_inputs, d_outputs, d_residuals, mode):
        self.update_inputs_outpus(inputs, outputs)
        if self._imop.stale():                                # the device model was re-created since linearize
            self._imop.linearize()
        self._products(d_inputs, d_outputs, d_residuals, mode)
