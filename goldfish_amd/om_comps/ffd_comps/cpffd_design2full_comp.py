"""CPFFDesign2FullComp -- design FFD control points (after alignment) -> full FFD control points
(reference: GOLDFISH/om_comps/ffd_comps/cpffd_design2full_comp.py:5-57; same option and variable names)."""
from .. import om


class CPFFDesign2FullComp(om.ExplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt_ffd')
        self.options.declare('input_cpffd_design_name_pre', default='CP_FFD_design')
        self.options.declare('output_cpffd_full_name_pre', default='CP_FFD_full')

    def init_parameters(self):
        nm = self.nonmatching_opt_ffd = self.options['nonmatching_opt_ffd']
        self.input_cpffd_design_name_pre = self.options['input_cpffd_design_name_pre']
        self.output_cpffd_full_name_pre = self.options['output_cpffd_full_name_pre']
        self.opt_field = nm.opt_field
        multi = getattr(nm, 'shopt_multiffd', False)
        self.deriv = [d.tocoo() for d in (nm.shopt_dcpaligndcp_mffd if multi else nm.shopt_dcpaligndcpffd)]
        self.init_cpffd = nm.shopt_init_cp_mffd_design if multi else nm.shopt_init_cpffd_design
        self.input_shapes = [m.shape[1] for m in self.deriv]
        self.output_shapes = [m.shape[0] for m in self.deriv]
        self.input_cpffd_name_list = [self.input_cpffd_design_name_pre + str(f) for f in self.opt_field]
        self.output_cpalign_name_list = [self.output_cpffd_full_name_pre + str(f) for f in self.opt_field]

    def setup(self):
        for i, field in enumerate(self.opt_field):
            self.add_input(self.input_cpffd_name_list[i], shape=self.input_shapes[i], val=self.init_cpffd[i])
            self.add_output(self.output_cpalign_name_list[i], shape=self.output_shapes[i])
            self.declare_partials(self.output_cpalign_name_list[i], self.input_cpffd_name_list[i],
                                  val=self.deriv[i].data, rows=self.deriv[i].row, cols=self.deriv[i].col)

    def compute(self, inputs, outputs):
        for i, field in enumerate(self.opt_field):
            outputs[self.output_cpalign_name_list[i]] = self.deriv[i] * inputs[self.input_cpffd_name_list[i]]

    # the partials are constant and declared in setup (COO values in rows / cols order): no compute_partials
