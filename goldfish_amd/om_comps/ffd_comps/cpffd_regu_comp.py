"""CPFFDReguComp -- differences of neighbouring design FFD control points (linear inequality constraint > 0)
(reference: GOLDFISH/om_comps/ffd_comps/cpffd_regu_comp.py:5-56; same option and variable names)."""
from .. import om


class CPFFDReguComp(om.ExplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt_ffd')
        self.options.declare('input_cpffd_design_name_pre', default='CP_FFD')
        self.options.declare('output_cpregu_name_pre', default='CP_FFD_regu')

    def init_parameters(self):
        nm = self.nonmatching_opt_ffd = self.options['nonmatching_opt_ffd']
        self.input_cpffd_design_name_pre = self.options['input_cpffd_design_name_pre']
        self.output_cpregu_name_pre = self.options['output_cpregu_name_pre']
        self.opt_field = nm.opt_field
        multi = getattr(nm, 'shopt_multiffd', False)
        self.derivs = [d.tocoo() for d in (nm.shopt_dcpregudcp_mffd if multi else nm.shopt_dcpregudcpffd)]
        self.init_cpffd = nm.shopt_init_cp_mffd_design if multi else nm.shopt_init_cpffd_design
        self.input_shapes = [m.shape[1] for m in self.derivs]
        self.output_shapes = [m.shape[0] for m in self.derivs]
        self.input_cpffd_name_list = [self.input_cpffd_design_name_pre + str(f) for f in self.opt_field]
        self.output_cpregu_name_list = [self.output_cpregu_name_pre + str(f) for f in self.opt_field]

    def setup(self):
        for i, field in enumerate(self.opt_field):
            self.add_input(self.input_cpffd_name_list[i], shape=self.input_shapes[i], val=self.init_cpffd[i])
            self.add_output(self.output_cpregu_name_list[i], shape=self.output_shapes[i])
            self.declare_partials(self.output_cpregu_name_list[i], self.input_cpffd_name_list[i],
                                  val=self.derivs[i].data, rows=self.derivs[i].row, cols=self.derivs[i].col)

    def compute(self, inputs, outputs):
        for i, field in enumerate(self.opt_field):
            outputs[self.output_cpregu_name_list[i]] = self.derivs[i] * inputs[self.input_cpffd_name_list[i]]

    # the partials are constant and declared in setup (COO values in rows / cols order): no compute_partials
