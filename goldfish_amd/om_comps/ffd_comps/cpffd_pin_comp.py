"""CPFFDPinComp -- values of the pinned design FFD control points (linear equality constraint)
(reference: GOLDFISH/om_comps/ffd_comps/cpffd_pin_comp.py:5-66; same option and variable names)."""
from .. import om


class CPFFDPinComp(om.ExplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt_ffd')
        self.options.declare('input_cpffd_design_name_pre', default='CP_FFD')
        self.options.declare('output_cppin_name_pre', default='CP_FFD_pin')

    def init_parameters(self):
        nm = self.nonmatching_opt_ffd = self.options['nonmatching_opt_ffd']
        self.input_cpffd_design_name_pre = self.options['input_cpffd_design_name_pre']
        self.output_cppin_name_pre = self.options['output_cppin_name_pre']
        self.opt_field = nm.opt_field
        self.pin_field = nm.pin_field
        multi = getattr(nm, 'shopt_multiffd', False)
        self.init_cpffd = nm.shopt_init_cp_mffd_design if multi else nm.shopt_init_cpffd_design
        self.derivs = [None if d is None else d.tocoo() for d in (nm.shopt_dcppindcp_mffd if multi else nm.shopt_dcppindcpffd)]
        self.field_inds = [self.opt_field.index(f) for f in self.pin_field]
        self.input_shapes = [self.derivs[k].shape[1] for k in self.field_inds]
        self.output_shapes = [self.derivs[k].shape[0] for k in self.field_inds]
        self.input_cpffd_name_list = [self.input_cpffd_design_name_pre + str(f) for f in self.pin_field]
        self.output_cppin_name_list = [self.output_cppin_name_pre + str(f) for f in self.pin_field]

    def setup(self):
        for i, field in enumerate(self.pin_field):
            k = self.field_inds[i]
            self.add_input(self.input_cpffd_name_list[i], shape=self.input_shapes[i], val=self.init_cpffd[k])
            self.add_output(self.output_cppin_name_list[i], shape=self.output_shapes[i])
            self.declare_partials(self.output_cppin_name_list[i], self.input_cpffd_name_list[i],
                                  val=self.derivs[k].data, rows=self.derivs[k].row, cols=self.derivs[k].col)

    def compute(self, inputs, outputs):
        for i, field in enumerate(self.pin_field):
            outputs[self.output_cppin_name_list[i]] = self.derivs[self.field_inds[i]] * inputs[self.input_cpffd_name_list[i]]

    # the partials are constant and declared in setup (COO values in rows / cols order): no compute_partials
