"""CPFFD2SurfComp -- constant linear map FFD control points -> surface control points
(reference: GOLDFISH/om_comps/ffd_comps/cpffd2surf_comp.py:6-62; same option and variable names)."""
from .. import om


class CPFFD2SurfComp(om.ExplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt_ffd')
        self.options.declare('input_cpffd_name_pre', default='CP_FFD')
        self.options.declare('output_cpsurf_name_pre', default='CP_FE')

    def init_parameters(self):
        self.nonmatching_opt_ffd = self.options['nonmatching_opt_ffd']
        self.input_cpffd_name_pre = self.options['input_cpffd_name_pre']
        self.output_cpsurf_name_pre = self.options['output_cpsurf_name_pre']
        nm = self.nonmatching_opt_ffd
        self.opt_field = nm.opt_field
        if getattr(nm, 'shopt_multiffd', False):
            self.derivs = [d.tocoo() for d in nm.shopt_dcpsurf_fedcp_mffd]
            self.input_shapes = [len(d) for d in nm.shopt_cp_mffd_design_dof_full]
            self.init_cpffd = nm.shopt_init_cp_mffd_full
        else:
            deriv = nm.shopt_dcpsurf_fedcpffd
            self.derivs = [deriv] * len(self.opt_field)
            self.input_shapes = [len(d) for d in nm.shopt_cpffd_design_dof_full]
            self.init_cpffd = [nm.shopt_cpffd_flat[:, f] for f in self.opt_field]
        self.output_shapes = [c.size for c in nm._shopt_cols]
        self.input_cpffd_name_list = [self.input_cpffd_name_pre + str(f) for f in self.opt_field]
        self.output_cpsurf_name_list = [self.output_cpsurf_name_pre + str(f) for f in self.opt_field]

    def setup(self):
        for i, field in enumerate(self.opt_field):
            self.add_input(self.input_cpffd_name_list[i], shape=self.input_shapes[i], val=self.init_cpffd[i])
            self.add_output(self.output_cpsurf_name_list[i], shape=self.output_shapes[i])
            self.declare_partials(self.output_cpsurf_name_list[i], self.input_cpffd_name_list[i],
                                  val=self.derivs[i].data, rows=self.derivs[i].row, cols=self.derivs[i].col)

    def compute(self, inputs, outputs):
        for i, field in enumerate(self.opt_field):
            outputs[self.output_cpsurf_name_list[i]] = self.derivs[i] * inputs[self.input_cpffd_name_list[i]]

    def compute_partials(self, inputs, partials):
        for i, field in enumerate(self.opt_field):
            partials[self.output_cpsurf_name_list[i], self.input_cpffd_name_list[i]] = self.derivs[i].toarray()
