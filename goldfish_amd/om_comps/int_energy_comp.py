"""IntEnergyComp -- explicit component for the internal energy
(reference: GOLDFISH/om_comps/int_energy_comp.py:7-104; option names, defaults, variable names and shapes of the reference)."""
from ._design_io import _REQUIRED, FunctionalComp
from ..operations.int_energy_exop import IntEnergyExOperation


class IntEnergyComp(FunctionalComp):
    OPTIONS = (('nonmatching_opt', _REQUIRED), ('input_cp_iga_name_pre', 'CP_IGA'), ('input_h_th_name', 'thickness'),
               ('input_u_name', 'displacements'), ('output_wint_name', 'w_int'))
    OUTPUT_OPTION = 'output_wint_name'

    def _operation(self, wint_regu=None):
        """int_energy_comp.py:16-23: init_parameters(wint_regu=None) hands the per-patch regularisation terms to the operation."""
        self.wint_exop = IntEnergyExOperation(self.nonmatching_opt, wint_regu)

    def _value(self):
        return self.wint_exop.Wint()

    def _du(self):
        # Dirichlet rows zeroed, as the reference does (int_energy_comp.py:94, apply_bcs=True): the constrained displacements are not
        # free variables, and dR/dh keeps its untreated Dirichlet rows (nonmatching_opt.py:1012-1014) -- a reaction-sized entry here
        # would reach the total derivative through them (tests/test_gpu_api.py: ThicknessOptGroup totals)
        return self.wint_exop.dWintduIGA(apply_bcs=True)

    def _dcp(self, field):
        return self.wint_exop.dWintdCPIGA(field)

    def _dh(self):
        return self.wint_exop.dWintdh_th()
