"""ComplianceComp -- explicit component for the compliance
(reference: GOLDFISH/om_comps/compliance_comp.py:8-157; same option and variable names)."""
import numpy as np

from . import om
from ..operations.compliance_exop import ComplianceExOperation


class ComplianceComp(om.ExplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt')
        self.options.declare('forces')
        self.options.declare('input_cp_iga_name_pre', default='CP_IGA')
        self.options.declare('input_h_th_name', default='thickness')
        self.options.declare('input_u_name', default='displacements')
        self.options.declare('output_c_name', default='compliance')

    def init_parameters(self, c_regu=None):
        self.nonmatching_opt = self.options['nonmatching_opt']
        self.forces = self.options['forces']
        self.input_cp_iga_name_pre = self.options['input_cp_iga_name_pre']
        self.input_h_th_name = self.options['input_h_th_name']
        self.input_u_name = self.options['input_u_name']
        self.output_c_name = self.options['output_c_name']
        self.c_exop = ComplianceExOperation(self.nonmatching_opt, self.forces, c_regu)
        self.opt_shape = self.nonmatching_opt.opt_shape
        self.opt_thickness = self.nonmatching_opt.opt_thickness
        self.input_u_shape = self.nonmatching_opt.vec_iga_dof
        self.init_disp_array = np.ones(self.nonmatching_opt.vec_iga_dof)          # compliance_comp.py:36
        if self.opt_shape:
            self.opt_field = self.nonmatching_opt.opt_field
            self.input_cp_shapes = [len(d) for d in self.nonmatching_opt.cpdes_iga_dofs_full]
            self.init_cp_iga = self.nonmatching_opt.get_init_CPIGA()
            self.input_cp_iga_name_list = [self.input_cp_iga_name_pre + str(f) for f in self.opt_field]

    def setup(self):
        self.add_output(self.output_c_name)
        self.add_input(self.input_u_name, shape=self.input_u_shape, val=self.init_disp_array)
        self.declare_partials(self.output_c_name, self.input_u_name)
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                self.add_input(self.input_cp_iga_name_list[i], shape=self.input_cp_shapes[i], val=self.init_cp_iga[i])
                self.declare_partials(self.output_c_name, self.input_cp_iga_name_list[i])

    def update_inputs(self, inputs):
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                self.nonmatching_opt.update_CPIGA(inputs[self.input_cp_iga_name_list[i]], field)
        self.nonmatching_opt.update_uIGA(inputs[self.input_u_name])

    def compute(self, inputs, outputs):
        self.update_inputs(inputs)
        outputs[self.output_c_name] = self.c_exop.cpl()

    def compute_partials(self, inputs, partials):
        self.update_inputs(inputs)
        partials[self.output_c_name, self.input_u_name] = self.c_exop.dcplduIGA(apply_bcs=False)
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                partials[self.output_c_name, self.input_cp_iga_name_list[i]] = self.c_exop.dcpldCPIGA(field)
