"""CPIGA2XiComp -- implicit component: intersection parametric coordinates as states of the control points
(reference: GOLDFISH/om_comps/cpiga2xi_comp.py:6-103; same option and variable names)."""
from . import om
from ..operations.cpiga2xi_imop import CPIGA2XiImOperation


class CPIGA2XiComp(om.ImplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt')
        self.options.declare('input_cp_iga_name_pre', default='CP_IGA')
        self.options.declare('output_xi_name', default='int_para_coord')

    def init_parameters(self):
        self.nonmatching_opt = self.options['nonmatching_opt']
        self.opt_field = self.nonmatching_opt.opt_field
        self.input_cp_iga_name_pre = self.options['input_cp_iga_name_pre']
        self.output_xi_name = self.options['output_xi_name']
        self.cpiga2xi_imop = CPIGA2XiImOperation(self.nonmatching_opt)
        self.input_cp_shapes = [len(d) for d in self.nonmatching_opt.cpdes_iga_dofs_full]
        self.init_cp_iga = self.nonmatching_opt.get_init_CPIGA()
        self.output_shape = self.cpiga2xi_imop.cpiga2xi.xi_size_global
        self.input_cp_iga_name_list = [self.input_cp_iga_name_pre + str(f) for f in self.opt_field]

    def setup(self):
        for i, field in enumerate(self.opt_field):
            self.add_input(self.input_cp_iga_name_list[i], shape=self.input_cp_shapes[i], val=self.init_cp_iga[i])
        self.add_output(self.output_xi_name, shape=self.output_shape, val=self.cpiga2xi_imop.cpiga2xi.xi_flat_global)
        for i, field in enumerate(self.opt_field):
            self.declare_partials(self.output_xi_name, self.input_cp_iga_name_list[i])
        self.declare_partials(self.output_xi_name, self.output_xi_name)

    def update_inputs(self, inputs):
        for i, field in enumerate(self.opt_field):
            self.cpiga2xi_imop.cpiga2xi.update_CPs(inputs[self.input_cp_iga_name_list[i]], field)

    def apply_nonlinear(self, inputs, outputs, residuals):
        self.update_inputs(inputs)
        residuals[self.output_xi_name] = self.cpiga2xi_imop.apply_nonlinear(outputs[self.output_xi_name])

    def solve_nonlinear(self, inputs, outputs):
        self.update_inputs(inputs)
        outputs[self.output_xi_name] = self.cpiga2xi_imop.solve_nonlinear(self.cpiga2xi_imop.cpiga2xi.xi_flat_global)

    def linearize(self, inputs, outputs, partials):
        self.update_inputs(inputs)
        self.cpiga2xi_imop.linearize(outputs[self.output_xi_name])

    def apply_linear(self, inputs, outputs, d_inputs, d_outputs, d_residuals, mode):
        self.update_inputs(inputs)
        d_inputs_array_list = [d_inputs[n] for n in self.input_cp_iga_name_list if n in d_inputs] or None
        d_outputs_array = d_outputs[self.output_xi_name] if self.output_xi_name in d_outputs else None
        d_residuals_array = d_residuals[self.output_xi_name] if self.output_xi_name in d_residuals else None
        if mode == 'fwd':
            self.cpiga2xi_imop.apply_linear_fwd(d_inputs_array_list, d_outputs_array, d_residuals_array)
        elif mode == 'rev':
            self.cpiga2xi_imop.apply_linear_rev(d_inputs_array_list, d_outputs_array, d_residuals_array)

    def solve_linear(self, d_outputs, d_residuals, mode):
        if mode == 'fwd':
            self.cpiga2xi_imop.solve_linear_fwd(d_outputs[self.output_xi_name], d_residuals[self.output_xi_name])
        if mode == 'rev':
            self.cpiga2xi_imop.solve_linear_rev(d_outputs[self.output_xi_name], d_residuals[self.output_xi_name])
