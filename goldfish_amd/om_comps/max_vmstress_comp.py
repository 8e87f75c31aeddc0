"""MaxvMStressComp -- explicit component for the aggregated maximum von Mises stress
(reference: GOLDFISH/om_comps/max_vmstress_comp.py:7-117; same option and variable names)."""
import numpy as np

from . import om
from ..operations.max_vmstress_exop import MaxvMStressExOperation


class MaxvMStressComp(om.ExplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt')
        self.options.declare('rho', default=1.)
        self.options.declare('alpha', default=None)
        self.options.declare('m', default=None)
        self.options.declare('surf', default='top')
        self.options.declare('method', default='pnorm')
        self.options.declare('linearize_stress', default=False)
        self.options.declare('input_u_name', default='displacements')
        self.options.declare('input_cp_iga_name_pre', default='CP_IGA')
        self.options.declare('input_h_th_name', default='thickness')
        self.options.declare('output_max_vM_name', default='max_vM_stress')

    def init_parameters(self):
        self.nonmatching_opt = self.options['nonmatching_opt']
        self.rho = self.options['rho']
        self.alpha = self.options['alpha']
        self.m = self.options['m']
        self.surf = self.options['surf']
        self.method = self.options['method']
        self.linearize_stress = self.options['linearize_stress']
        self.input_cp_iga_name_pre = self.options['input_cp_iga_name_pre']
        self.input_h_th_name = self.options['input_h_th_name']
        self.input_u_name = self.options['input_u_name']
        self.output_max_vM_name = self.options['output_max_vM_name']
        self.max_vm_exop = MaxvMStressExOperation(self.nonmatching_opt, self.rho, self.alpha, self.m, self.surf,
                                                  self.method, self.linearize_stress)
        self.input_u_shape = self.nonmatching_opt.vec_iga_dof
        self.init_disp_array = self.nonmatching_opt.u_iga.copy()
        self.opt_field = self.nonmatching_opt.opt_field
        self.opt_shape = self.nonmatching_opt.opt_shape
        self.opt_thickness = self.nonmatching_opt.opt_thickness
        self.var_thickness = self.nonmatching_opt.var_thickness
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
        self.add_output(self.output_max_vM_name)
        self.add_input(self.input_u_name, shape=self.input_u_shape, val=self.init_disp_array)
        self.declare_partials(self.output_max_vM_name, self.input_u_name)
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                self.add_input(self.input_cp_iga_name_list[i], shape=self.input_cp_shapes[i], val=self.init_cp_iga[i])
                self.declare_partials(self.output_max_vM_name, self.input_cp_iga_name_list[i])
        if self.opt_thickness:
            self.add_input(self.input_h_th_name, shape=self.input_h_th_shape, val=self.init_h_th)
            self.declare_partials(self.output_max_vM_name, self.input_h_th_name)

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
        outputs[self.output_max_vM_name] = self.max_vm_exop.max_vM_stress_global()

    def compute_partials(self, inputs, partials):
        self.update_inputs(inputs)
        partials[self.output_max_vM_name, self.input_u_name] = self.max_vm_exop.dmax_vMduIGA_global(array=True, apply_bcs=True)
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                partials[self.output_max_vM_name, self.input_cp_iga_name_list[i]] = self.max_vm_exop.dmax_vMdCPIGA_global(field, array=True)
        if self.opt_thickness:
            partials[self.output_max_vM_name, self.input_h_th_name] = self.max_vm_exop.dmax_vMdh_th_global(array=True)
