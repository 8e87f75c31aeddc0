"""IntEnergyReguComp -- explicit component for the regularised internal energy
(reference: demos_om/shape_opt/eVTOL/int_energy_regu_comp.py; IntEnergyComp with the operation swapped and the extra option
``regu_para``)."""
from .int_energy_comp import IntEnergyComp
from ..operations.int_energy_regu_exop import IntEnergyReguExOperation


class IntEnergyReguComp(IntEnergyComp):
    OPTIONS = IntEnergyComp.OPTIONS + (('regu_para', 1.0e-1),)

    def _operation(self):
        self.wint_exop = IntEnergyReguExOperation(self.nonmatching_opt, self.regu_para)
