"""CPIGA2XiComp -- implicit component: intersection parametric coordinates as states of the control points
(reference: GOLDFISH/om_comps/cpiga2xi_comp.py:6-103; option and variable names of the reference)."""
from ._design_io import _REQUIRED
from .disp_states_comp import StatesComp
from ..operations.cpiga2xi_imop import CPIGA2XiImOperation


class CPIGA2XiComp(StatesComp):
    OPTIONS = (('nonmatching_opt', _REQUIRED), ('input_cp_iga_name_pre', 'CP_IGA'), ('output_xi_name', 'int_para_coord'))
    STATE_OPTION = 'output_xi_name'

    def init_parameters(self):
        self._read_options()
        self.cpiga2xi_imop = self._imop = CPIGA2XiImOperation(self.nonmatching_opt)
        self.opt_field = self.nonmatching_opt.opt_field
        self.input_cp_shapes = [len(d) for d in self.nonmatching_opt.cpdes_iga_dofs_full]
        self.init_cp_iga = self.nonmatching_opt.get_init_CPIGA()
        self.output_shape = self._imop.cpiga2xi.xi_size_global
        self.input_cp_iga_name_list = [self.input_cp_iga_name_pre + str(f) for f in self.opt_field]

    def setup(self):
        xi = self.output_xi_name
        for name, shape, val in zip(self.input_cp_iga_name_list, self.input_cp_shapes, self.init_cp_iga):
            self.add_input(name, shape=shape, val=val)
            self.declare_partials(xi, name)
        self.add_output(xi, shape=self.output_shape, val=self._imop.cpiga2xi.xi_flat_global)
        self.declare_partials(xi, xi)

    def _input_names(self):
        return self.input_cp_iga_name_list

    def update_inputs(self, inputs):
        for name, field in zip(self.input_cp_iga_name_list, self.opt_field):
            self._imop.cpiga2xi.update_CPs(inputs[name], field)

    def apply_nonlinear(self, inputs, outputs, residuals):
        self.update_inputs(inputs)
        residuals[self.output_xi_name] = self._imop.apply_nonlinear(outputs[self.output_xi_name])

    def solve_nonlinear(self, inputs, outputs):
        self.update_inputs(inputs)
        outputs[self.output_xi_name] = self._imop.solve_nonlinear(self._imop.cpiga2xi.xi_flat_global)

    def linearize(self, inputs, outputs, partials):
        self.update_inputs(inputs)
        self._imop.linearize(outputs[self.output_xi_name])

    def apply_linear(self, inputs, outputs, d_inputs, d_outputs, d_residuals, mode):
        self.update_inputs(inputs)
        self._products(d_inputs, d_outputs, d_residuals, mode)
