"""IntEnergyComp -- explicit component for the internal energy
(reference: GOLDFISH/om_comps/int_energy_comp.py:7-104)."""
import numpy as np

from . import om
from ..operations.int_energy_exop import IntEnergyExOperation


class IntEnergyComp(om.ExplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt')
        self.options.declare('input_cp_iga_name_pre', default='CP_IGA')
        self.options.declare('input_h_th_name', default='thickness')
        self.options.declare('input_u_name', default='displacements')
        self.options.declare('output_wint_name', default='w_int')

    def init_parameters(self):
        self.nonmatching_opt = self.options['nonmatching_opt']
        self.input_cp_iga_name_pre = self.options['input_cp_iga_name_pre']
        self.input_h_th_name = self.options['input_h_th_name']
        self.input_u_name = self.options['input_u_name']
        self.output_wint_name = self.options['output_wint_name']
        self.wint_exop = IntEnergyExOperation(self.nonmatching_opt)
        self.opt_field = self.nonmatching_opt.opt_field
        self.opt_shape = self.nonmatching_opt.opt_shape
        self.opt_thickness = self.nonmatching_opt.opt_thickness
        self.var_thickness = self.nonmatching_opt.var_thickness
        self.input_u_shape = self.nonmatching_opt.vec_iga_dof
        self.init_disp_array = self.nonmatching_opt.u_iga.copy()
        if self.opt_shape:
            self.init_cp_iga = self.nonmatching_opt.get_init_CPIGA()
            self.input_cp_shapes = [len(d) for d in self.nonmatching_opt.cpdes_iga_dofs_full]
            self.input_cp_iga_name_list = [self.input_cp_iga_name_pre + str(f) for f in self.opt_field]
        if self.opt_thickness:
            if self.var_thickness:
                self.input_h_th_shape = self.nonmatching_opt.vec_scalar_iga_dof
                self.init_h_th = self.nonmatching_opt.init_h_th_iga
            else:
                self.input_h_th_shape = self.nonmatching_opt.h_th_dof
                self.init_h_th = self.nonmatching_opt.init_h_th

    def setup(self):
        self.add_output(self.output_wint_name)
        self.add_input(self.input_u_name, shape=self.input_u_shape, val=self.init_disp_array)
        self.declare_partials(self.output_wint_name, self.input_u_name)
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                self.add_input(self.input_cp_iga_name_list[i], shape=self.input_cp_shapes[i], val=self.init_cp_iga[i])
                self.declare_partials(self.output_wint_name, self.input_cp_iga_name_list[i])
        if self.opt_thickness:
            self.add_input(self.input_h_th_name, shape=self.input_h_th_shape, val=self.init_h_th)
            self.declare_partials(self.output_wint_name, self.input_h_th_name)

    def update_inputs(self, inputs):
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                self.nonmatching_opt.update_CPIGA(inputs[self.input_cp_iga_name_list[i]], field)
        if self.opt_thickness:
            if self.var_thickness:
                self.nonmatching_opt.update_h_th_IGA(inputs[self.input_h_th_name])
            else:
                self.nonmatching_opt.update_h_th(inputs[self.input_h_th_name])
        self.nonmatching_opt.update_uIGA(inputs[self.input_u_name])

    def compute(self, inputs, outputs):
        self.update_inputs(inputs)
        outputs[self.output_wint_name] = self.wint_exop.Wint()

    def compute_partials(self, inputs, partials):
        self.update_inputs(inputs)
        partials[self.output_wint_name, self.input_u_name] = self.wint_exop.dWintduIGA(apply_bcs=False)
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                partials[self.output_wint_name, self.input_cp_iga_name_list[i]] = self.wint_exop.dWintdCPIGA(field)
        if self.opt_thickness:
            partials[self.output_wint_name, self.input_h_th_name] = self.wint_exop.dWintdh_th()
