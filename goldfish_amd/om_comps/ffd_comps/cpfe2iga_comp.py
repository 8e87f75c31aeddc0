"""CPFE2IGAComp -- in the reference an implicit L2 projection of the FE control-point functions onto
the IGA control points (GOLDFISH/om_comps/cpfe2iga_comp.py, operations/cpfe2iga_imop.py:63-94:
Mc^T Mc x = Mc^T x_fe).  Assembly happens directly in IGA dofs here, so "FE control points" are the
IGA control points and the projection is the identity; the component is kept (as an explicit
pass-through with unit Jacobian) so that the demos' group wiring CP_FE<f> -> CP_IGA<f> still works."""
import numpy as np

from .. import om


class CPFE2IGAComp(om.ExplicitComponent):

    def initialize(self):
        self.options.declare('nonmatching_opt')
        self.options.declare('input_cp_fe_name_pre', default='CP_FE')
        self.options.declare('output_cp_iga_name_pre', default='CP_IGA')

    def init_parameters(self):
        self.nonmatching_opt = self.options['nonmatching_opt']
        self.opt_field = self.nonmatching_opt.opt_field
        self.init_cp_iga = self.nonmatching_opt.get_init_CPIGA()
        self.input_cp_fe_name_list = [self.options['input_cp_fe_name_pre'] + str(f) for f in self.opt_field]
        self.output_cp_iga_name_list = [self.options['output_cp_iga_name_pre'] + str(f) for f in self.opt_field]

    def setup(self):
        for i, field in enumerate(self.opt_field):
            n = self.init_cp_iga[i].size
            self.add_input(self.input_cp_fe_name_list[i], shape=n, val=self.init_cp_iga[i])
            self.add_output(self.output_cp_iga_name_list[i], shape=n, val=self.init_cp_iga[i])
            self.declare_partials(self.output_cp_iga_name_list[i], self.input_cp_fe_name_list[i],
                                  val=np.ones(n), rows=np.arange(n), cols=np.arange(n))

    def compute(self, inputs, outputs):
        for i, field in enumerate(self.opt_field):
            outputs[self.output_cp_iga_name_list[i]] = inputs[self.input_cp_fe_name_list[i]]

    # the partials are constant and declared in setup (COO values in rows / cols order): no compute_partials
