"""DispMintStatesComp -- displacement states with moving intersections: R(u; CP_IGA, xi) = 0
(reference: GOLDFISH/om_comps/disp_states_mi_comp.py:6-117; same option and variable names)."""
import numpy as np

from . import om
from ..operations.disp_mi_imop import DispMintImOpeartion


class DispMintStatesComp(om.ImplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt')
        self.options.declare('input_cp_iga_name_pre', default='CP_IGA')
        self.options.declare('input_xi_name', default='int_para')
        self.options.declare('output_u_name', default='displacements')

    def init_parameters(self, save_files=False, nonlinear_solver_rtol=1e-3, nonlinear_solver_max_it=10):
        self.nonmatching_opt = self.options['nonmatching_opt']
        self.input_cp_iga_name_pre = self.options['input_cp_iga_name_pre']
        self.input_xi_name = self.options['input_xi_name']
        self.output_u_name = self.options['output_u_name']
        self.save_files = save_files
        self.nonlinear_solver_max_it = nonlinear_solver_max_it
        self.nonlinear_solver_rtol = nonlinear_solver_rtol
        self.disp_mint_state_imop = DispMintImOpeartion(self.nonmatching_opt)
        self.opt_field = self.nonmatching_opt.opt_field
        self.input_cp_shapes = [len(d) for d in self.nonmatching_opt.cpdes_iga_dofs_full]
        self.input_xi_shape = self.nonmatching_opt.xi_size
        self.output_shape = self.nonmatching_opt.vec_iga_dof
        self.init_cp_iga = self.nonmatching_opt.get_init_CPIGA()
        self.init_xi = self.nonmatching_opt.cpiga2xi.xi_flat_global
        self.input_cp_iga_name_list = [self.input_cp_iga_name_pre + str(f) for f in self.opt_field]
        self._last_xi = None

    def setup(self):
        for i, field in enumerate(self.opt_field):
            self.add_input(self.input_cp_iga_name_list[i], shape=self.input_cp_shapes[i], val=self.init_cp_iga[i])
        self.add_input(self.input_xi_name, shape=self.input_xi_shape, val=self.init_xi)
        self.add_output(self.output_u_name, shape=self.output_shape)
        for i, field in enumerate(self.opt_field):
            self.declare_partials(self.output_u_name, self.input_cp_iga_name_list[i])
        self.declare_partials(self.output_u_name, self.input_xi_name)
        self.declare_partials(self.output_u_name, self.output_u_name)

    def update_inputs_outpus(self, inputs, outputs):
        """disp_states_mi_comp.py:54-60 (the reference's spelling); the device model is re-created only when the
        parametric coordinates changed."""
        xi = np.asarray(inputs[self.input_xi_name], float).ravel()
        if self._last_xi is None or not np.array_equal(xi, self._last_xi):
            self.nonmatching_opt.update_xi(xi)
            self.nonmatching_opt.update_transfer_matrices()
            self._last_xi = xi.copy()
        for i, field in enumerate(self.opt_field):
            self.nonmatching_opt.update_CPIGA(inputs[self.input_cp_iga_name_list[i]], field)
        self.nonmatching_opt.update_uIGA(outputs[self.output_u_name])

    def apply_nonlinear(self, inputs, outputs, residuals):
        self.update_inputs_outpus(inputs, outputs)
        residuals[self.output_u_name] = self.disp_mint_state_imop.apply_nonlinear()

    def solve_nonlinear(self, inputs, outputs):
        self.update_inputs_outpus(inputs, outputs)
        outputs[self.output_u_name] = self.disp_mint_state_imop.solve_nonlinear(self.nonlinear_solver_max_it,
                                                                               self.nonlinear_solver_rtol)

    def linearize(self, inputs, outputs, partials):
        self.update_inputs_outpus(inputs, outputs)
        self.disp_mint_state_imop.linearize()

    def apply_linear(self, inputs, outputs, d_inputs, d_outputs, d_residuals, mode):
        self.update_inputs_outpus(inputs, outputs)
        if self.disp_mint_state_imop.stale():                 # the device model was re-created since linearize
            self.disp_mint_state_imop.linearize()
        d_inputs_array_list = [d_inputs[n] for n in self.input_cp_iga_name_list if n in d_inputs]
        if self.input_xi_name in d_inputs:
            d_inputs_array_list.append(d_inputs[self.input_xi_name])
        d_inputs_array_list = d_inputs_array_list or None
        d_outputs_array = d_outputs[self.output_u_name] if self.output_u_name in d_outputs else None
        d_residuals_array = d_residuals[self.output_u_name] if self.output_u_name in d_residuals else None
        if mode == 'fwd':
            self.disp_mint_state_imop.apply_linear_fwd(d_inputs_array_list, d_outputs_array, d_residuals_array)
        elif mode == 'rev':
            self.disp_mint_state_imop.apply_linear_rev(d_inputs_array_list, d_outputs_array, d_residuals_array)

    def solve_linear(self, d_outputs, d_residuals, mode):
        if mode == 'fwd':
            self.disp_mint_state_imop.solve_linear_fwd(d_outputs[self.output_u_name], d_residuals[self.output_u_name])
        if mode == 'rev':
            self.disp_mint_state_imop.solve_linear_rev(d_outputs[self.output_u_name], d_residuals[self.output_u_name])
