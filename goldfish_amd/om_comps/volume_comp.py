"""VolumeComp -- explicit component for the material volume
(reference: GOLDFISH/om_comps/volume_comp.py:7-90; option names, defaults, variable names and shapes of the reference)."""
from ._design_io import _REQUIRED, FunctionalComp
from ..operations.volume_exop import VolumeExOperation


class VolumeComp(FunctionalComp):
    OPTIONS = (('nonmatching_opt', _REQUIRED), ('vol_surf_inds', None), ('input_cp_iga_name_pre', 'CP_IGA'),
               ('input_h_th_name', 'thickness'), ('output_vol_name', 'volume'))
    OUTPUT_OPTION = 'output_vol_name'
    USES_U = False

    def _operation(self):
        self.vol_exop = VolumeExOperation(self.nonmatching_opt, self.vol_surf_inds)

    def _value(self):
        return self.vol_exop.volume()

    def _dcp(self, field):
        return self.vol_exop.dvoldCPIGA(field)

    def _dh(self):
        return self.vol_exop.dvoldh_th()
