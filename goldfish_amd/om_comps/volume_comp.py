"""VolumeComp -- explicit component for the material volume
(reference: GOLDFISH/om_comps/volume_comp.py:7-90)."""
from . import om
from ..operations.volume_exop import VolumeExOperation


class VolumeComp(om.ExplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt')
        self.options.declare('vol_surf_inds', default=None)
        self.options.declare('input_cp_iga_name_pre', default='CP_IGA')
        self.options.declare('input_h_th_name', default='thickness')
        self.options.declare('output_vol_name', default='volume')

    def init_parameters(self):
        self.nonmatching_opt = self.options['nonmatching_opt']
        self.vol_surf_inds = self.options['vol_surf_inds']
        self.input_cp_iga_name_pre = self.options['input_cp_iga_name_pre']
        self.input_h_th_name = self.options['input_h_th_name']
        self.output_vol_name = self.options['output_vol_name']
        self.vol_exop = VolumeExOperation(self.nonmatching_opt, self.vol_surf_inds)
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
        self.add_output(self.output_vol_name)
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                self.add_input(self.input_cp_iga_name_list[i], shape=self.input_cp_shapes[i], val=self.init_cp_iga[i])
                self.declare_partials(self.output_vol_name, self.input_cp_iga_name_list[i])
        if self.opt_thickness:
            self.add_input(self.input_h_th_name, shape=self.input_h_th_shape, val=self.init_h_th)
            self.declare_partials(self.output_vol_name, self.input_h_th_name)

    def update_inputs(self, inputs):
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                self.nonmatching_opt.update_CPIGA(inputs[self.input_cp_iga_name_list[i]], field)
        if self.opt_thickness:
            if self.var_thickness:
                self.nonmatching_opt.update_h_th_IGA(inputs[self.input_h_th_name])
            else:
                self.nonmatching_opt.update_h_th(inputs[self.input_h_th_name])

    def compute(self, inputs, outputs):
        self.update_inputs(inputs)
        outputs[self.output_vol_name] = self.vol_exop.volume()

    def compute_partials(self, inputs, partials):
        self.update_inputs(inputs)
        if self.opt_shape:
            for i, field in enumerate(self.opt_field):
                partials[self.output_vol_name, self.input_cp_iga_name_list[i]] = self.vol_exop.dvoldCPIGA(field)
        if self.opt_thickness:
            partials[self.output_vol_name, self.input_h_th_name] = self.vol_exop.dvoldh_th()
