"""MaxvMStressComp -- explicit component for the aggregated maximum von Mises stress
(reference: GOLDFISH/om_comps/max_vmstress_comp.py:7-117; option names, defaults, variable names and shapes of the reference;
the displacement partial has its Dirichlet rows zeroed, as there)."""
from ._design_io import _REQUIRED, FunctionalComp
from ..operations.max_vmstress_exop import MaxvMStressExOperation


class MaxvMStressComp(FunctionalComp):
    OPTIONS = (('nonmatching_opt', _REQUIRED), ('rho', 1.), ('alpha', None), ('m', None), ('surf', 'top'), ('method', 'pnorm'),
               ('linearize_stress', False), ('input_u_name', 'displacements'), ('input_cp_iga_name_pre', 'CP_IGA'),
               ('input_h_th_name', 'thickness'), ('output_max_vM_name', 'max_vM_stress'))
    OUTPUT_OPTION = 'output_max_vM_name'

    def _operation(self):
        self.max_vm_exop = MaxvMStressExOperation(self.nonmatching_opt, self.rho, self.alpha, self.m, self.surf, self.method,
                                                  self.linearize_stress)

    def _value(self):
        return self.max_vm_exop.max_vM_stress_global()

    def _du(self):
        return self.max_vm_exop.dmax_vMduIGA_global(array=True, apply_bcs=True)

    def _dcp(self, field):
        return self.max_vm_exop.dmax_vMdCPIGA_global(field, array=True)

    def _dh(self):
        return self.max_vm_exop.dmax_vMdh_th_global(array=True)
