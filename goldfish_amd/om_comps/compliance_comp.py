"""ComplianceComp -- explicit component for the compliance
(reference: GOLDFISH/om_comps/compliance_comp.py:8-157; option names, defaults, variable names and shapes of the reference;
the compliance does not depend on the thickness)."""
import numpy as np

from ._design_io import _REQUIRED, FunctionalComp
from ..operations.compliance_exop import ComplianceExOperation


class ComplianceComp(FunctionalComp):
    OPTIONS = (('nonmatching_opt', _REQUIRED), ('forces', _REQUIRED), ('input_cp_iga_name_pre', 'CP_IGA'),
               ('input_h_th_name', 'thickness'), ('input_u_name', 'displacements'), ('output_c_name', 'compliance'))
    OUTPUT_OPTION = 'output_c_name'
    USES_THICKNESS = False

    def _operation(self, c_regu=None):
        self.c_exop = ComplianceExOperation(self.nonmatching_opt, self.forces, c_regu)

    def _initial_u(self):
        return np.ones(self.nonmatching_opt.vec_iga_dof)                  # compliance_comp.py:36

    def _value(self):
        return self.c_exop.cpl()

    def _du(self):
        return self.c_exop.dcplduIGA(apply_bcs=True)          # compliance_comp.py:130 of the reference

    def _dcp(self, field):
        return self.c_exop.dcpldCPIGA(field)
