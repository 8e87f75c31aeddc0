"""DispStatesComp -- implicit component solving for the displacement states
(reference: GOLDFISH/om_comps/disp_states_comp.py:6-144; identical option names, defaults,
variable names and shapes)."""
import numpy as np

from . import om
from ..operations.disp_imop import DispImOpeartion


class DispStatesComp(om.ImplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt')
        self.options.declare('input_cp_iga_name_pre', default='CP_IGA')
        self.options.declare('input_h_th_name', default='thickness')
        self.options.declare('output_u_name', default='displacements')

    def init_parameters(self, save_files=False, nonlinear_solver_rtol=1e-3, nonlinear_solver_max_it=30):
        """disp_states_comp.py:14-49 (must be called before setup, like the reference)."""
        self.nonmatching_opt = self.options['nonmatching_opt']
        self.input_cp_iga_name_pre = self.options['input_cp_iga_name_pre']
        self.input_h_th_name = self.options['input_h_th_name']
        self.output_u_name = self.options['output_u_name']
        self.save_files = save_files
        self.nonlinear_solver_max_it = nonlinear_solver_max_it
        self.nonlinear_solver_rtol = nonlinear_solver_rtol
        self.major_iter_ind = 0
        self.func_eval_ind = 0
        self.func_eval_major_ind = []
        self.disp_state_imop = DispImOpeartion(self.nonmatching_opt)
        self.opt_field = self.nonmatching_opt.opt_field
        self.opt_shape = self.nonmatching_opt.opt_shape
        self.opt_thickness = self.nonmatching_opt.opt_thickness
        self.var_thickness = self.nonmatching_opt.var_thickness
        self.output_shape = self.nonmatching_opt.vec_iga_dof
        if self.opt_shape:
            # like the reference, CP_IGA<field> is sized over the optimised patches of that field
            # (all patches in every demo, disp_states_comp.py:37 / SURVEY.md 8(b))
            self.init_cp_iga = self.nonmatching_opt.get_init_CPIGA()
            self.input_cp_shapes = [v.size for v in self.init_cp_iga]
            self.input_cp_iga_name_list = [self.input_cp_iga_name_pre + str(f) for f in self.opt_field]
        if self.opt_thickness:
            if self.var_thickness:
                self.input_h_th_shape = self.nonmatching_opt.vec_scalar_iga_dof
                self.init_h_th = self.nonmatching_opt.init_h_th_iga
            else:
                self.input_h_th_shape = self.nonmatching_opt.h_th_dof
                self.init_h_th = self.nonmatching_opt.init_h_th

    def setup(self):
        self.add_output(self.output_u_name, shape=self.output_shape)
        self.declare_partials(self.output_u_name, self.output_u_name)
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                self.add_input(self.input_cp_iga_name_list[i], shape=self.input_cp_shapes[i], val=self.init_cp_iga[i])
                self.declare_partials(self.output_u_name, self.input_cp_iga_name_list[i])
        if self.opt_thickness:
            self.add_input(self.input_h_th_name, shape=self.input_h_th_shape, val=self.init_h_th)
            self.declare_partials(self.output_u_name, self.input_h_th_name)

    def update_inputs_outpus(self, inputs, outputs):
        """disp_states_comp.py:68-79 (name kept)."""
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                self.nonmatching_opt.update_CPIGA(inputs[self.input_cp_iga_name_list[i]], field)
        if self.opt_thickness:
            if self.var_thickness:
                self.nonmatching_opt.update_h_th_IGA(inputs[self.input_h_th_name])
            else:
                self.nonmatching_opt.update_h_th(inputs[self.input_h_th_name])
        self.nonmatching_opt.update_uIGA(outputs[self.output_u_name])

    def apply_nonlinear(self, inputs, outputs, residuals):
        self.update_inputs_outpus(inputs, outputs)
        residuals[self.output_u_name] = self.disp_state_imop.apply_nonlinear()

    def solve_nonlinear(self, inputs, outputs):
        self.update_inputs_outpus(inputs, outputs)
        outputs[self.output_u_name] = self.disp_state_imop.solve_nonlinear(
            self.nonlinear_solver_max_it, self.nonlinear_solver_rtol)
        self.func_eval_ind += 1

    def linearize(self, inputs, outputs, partials):
        self.update_inputs_outpus(inputs, outputs)
        self.disp_state_imop.linearize()
        self.func_eval_major_ind += [self.func_eval_ind - 1]
        self.major_iter_ind += 1

    def apply_linear(self, inputs, outputs, d_inputs, d_outputs, d_residuals, mode):
        """disp_states_comp.py:107-134.  The reference re-pushes inputs here; the state is
        already on the device after linearize, so that redundant update is skipped."""
        d_inputs_array_list = []
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                if self.input_cp_iga_name_list[i] in d_inputs:
                    d_inputs_array_list += [d_inputs[self.input_cp_iga_name_list[i]]]
        if self.opt_thickness:
            if self.input_h_th_name in d_inputs:
                d_inputs_array_list += [d_inputs[self.input_h_th_name]]
        if len(d_inputs_array_list) == 0:
            d_inputs_array_list = None
        d_outputs_array = d_outputs[self.output_u_name] if self.output_u_name in d_outputs else None
        d_residuals_array = d_residuals[self.output_u_name] if self.output_u_name in d_residuals else None
        if mode == 'fwd':
            self.disp_state_imop.apply_linear_fwd(d_inputs_array_list, d_outputs_array, d_residuals_array)
        elif mode == 'rev':
            self.disp_state_imop.apply_linear_rev(d_inputs_array_list, d_outputs_array, d_residuals_array)

    def solve_linear(self, d_outputs, d_residuals, mode):
        d_outputs_array = d_outputs[self.output_u_name]
        d_residuals_array = d_residuals[self.output_u_name]
        if mode == 'fwd':
            self.disp_state_imop.solve_linear_fwd(d_outputs_array, d_residuals_array)
        if mode == 'rev':
            self.disp_state_imop.solve_linear_rev(d_outputs_array, d_residuals_array)
