"""CPFFD2SurfComp -- constant linear map FFD control points -> surface control points, one per optimised coordinate field
(reference: GOLDFISH/om_comps/ffd_comps/cpffd2surf_comp.py:6-62; option and variable names of the reference)."""
from .._design_io import _REQUIRED, LinearMapsComp


class CPFFD2SurfComp(LinearMapsComp):
    OPTIONS = (('nonmatching_opt_ffd', _REQUIRED), ('input_cpffd_name_pre', 'CP_FFD'), ('output_cpsurf_name_pre', 'CP_FE'))

    def _build(self):
        nm = self.nonmatching_opt_ffd
        self.opt_field = nm.opt_field
        if getattr(nm, 'shopt_multiffd', False):
            self.derivs, self.init_cpffd = [d.tocoo() for d in nm.shopt_dcpsurf_fedcp_mffd], nm.shopt_init_cp_mffd_full
        else:
            self.derivs = list(getattr(nm, 'shopt_dcpsurf_fedcpffd_list', [nm.shopt_dcpsurf_fedcpffd] * len(self.opt_field)))
            self.init_cpffd = [nm.shopt_cpffd_flat[:, f] for f in self.opt_field]
        self.input_cpffd_name_list = [self.input_cpffd_name_pre + str(f) for f in self.opt_field]
        self.output_cpsurf_name_list = [self.output_cpsurf_name_pre + str(f) for f in self.opt_field]
        return [(i, o, A, x0, None) for i, o, A, x0 in zip(self.input_cpffd_name_list, self.output_cpsurf_name_list, self.derivs, self.init_cpffd)]
