"""Design -> analysis control-net components of the moving-intersection shape optimisation (reference: GOLDFISH/om_comps/surf_comps/*.py; option names,
defaults and variable names of the reference).  Every one is a constant sparse map of a goldfish_amd.utils.bsp_utils.CPSurfDesign2Analysis."""
import numpy as np

from .._design_io import _REQUIRED, LinearMapsComp


class _D2AMaps(LinearMapsComp):
    IN_OPT, OUT_OPT = None, None

    def _fields_maps(self, d2a):
        """[(field, A, initial input, offset b or None)]"""
        raise NotImplementedError

    def _build(self):
        d2a = self.cpdesign2analysis
        self.opt_field = d2a.opt_field
        pre_in, pre_out = getattr(self, self.IN_OPT), getattr(self, self.OUT_OPT)
        return [(pre_in + str(f), pre_out + str(f), A, x0, b) for f, A, x0, b in self._fields_maps(d2a)]


class CPSurfAlignComp(_D2AMaps):
    """cpsurf_align_comp.py:5-57: CP_coarse = A_align CP_design + diff_vec."""
    OPTIONS = (('cpdesign2analysis', _REQUIRED), ('diff_vec', None), ('input_cp_design_name_pre', 'CP_design'), ('output_cp_coarse_name_pre', 'CP_coarse'))
    IN_OPT, OUT_OPT = 'input_cp_design_name_pre', 'output_cp_coarse_name_pre'

    def _fields_maps(self, d2a):
        dv = self.diff_vec if self.diff_vec is not None else [None] * len(d2a.opt_field)
        return [(f, d2a.cp_coarse_align_deriv_list[i], d2a.init_cp_design[i], None if dv[i] is None else -np.asarray(dv[i], float)) for i, f in enumerate(d2a.opt_field)]


class CPSurfOrderElevationComp(_D2AMaps):
    """cpsurf_order_elevation_comp.py:5-48."""
    OPTIONS = (('cpdesign2analysis', _REQUIRED), ('input_cp_coarse_name_pre', 'CP_coarse'), ('output_cp_order_ele_name_pre', 'CP_order_ele'))
    IN_OPT, OUT_OPT = 'input_cp_coarse_name_pre', 'output_cp_order_ele_name_pre'

    def _fields_maps(self, d2a):
        return [(f, d2a.order_ele_operator_list[i], d2a.init_cp_coarse[i], None) for i, f in enumerate(d2a.opt_field)]


class CPSurfKnotRefinementComp(_D2AMaps):
    """cpsurf_knot_refienment_comp.py:5-46."""
    OPTIONS = (('cpdesign2analysis', _REQUIRED), ('input_cp_order_ele_name_pre', 'CP_order_ele'), ('output_cp_fine_name_pre', 'CP_fine'))
    IN_OPT, OUT_OPT = 'input_cp_order_ele_name_pre', 'output_cp_fine_name_pre'

    def _fields_maps(self, d2a):
        return [(f, d2a.knot_refine_operator_list[i], d2a.order_ele_operator_list[i].tocsr() @ d2a.init_cp_coarse[i], None) for i, f in enumerate(d2a.opt_field)]


class CPSurfPinComp(_D2AMaps):
    """cpsurf_pin_comp.py:6-60: A_pin CP_design - pinned values (= 0 at a feasible design)."""
    OPTIONS = (('cpdesign2analysis', _REQUIRED), ('input_cp_design_name_pre', 'CP_design'), ('output_cp_pin_name_pre', 'CP_regu'))
    IN_OPT, OUT_OPT = 'input_cp_design_name_pre', 'output_cp_pin_name_pre'

    def _fields_maps(self, d2a):
        return [(f, d2a.cp_coarse_pin_deriv_list[d2a.opt_field.index(f)], d2a.init_cp_design[d2a.opt_field.index(f)],
                 np.asarray(d2a.cp_coarse_pin_vals[d2a.opt_field.index(f)], float)) for f in d2a.cp_coarse_pin_field]


class CPSurfReguComp(_D2AMaps):
    """cpsurf_regu_comp.py:6-61: differences of neighbouring design control points."""
    OPTIONS = (('cpdesign2analysis', _REQUIRED), ('input_cp_design_name_pre', 'CP_design'), ('output_cp_regu_name_pre', 'CP_regu'))
    IN_OPT, OUT_OPT = 'input_cp_design_name_pre', 'output_cp_regu_name_pre'

    def _fields_maps(self, d2a):
        return [(f, d2a.cp_coarse_regu_deriv_list[d2a.opt_field.index(f)], d2a.init_cp_design[d2a.opt_field.index(f)], None) for f in d2a.cp_coarse_regu_field]


class CPSurfDistanceComp(_D2AMaps):
    """cpsurf_distance_comp.py:6-61: differences between the design control points of consecutive patches."""
    OPTIONS = (('cpdesign2analysis', _REQUIRED), ('input_cp_design_name_pre', 'CP_design'), ('output_cp_dist_name_pre', 'CP_regu'))
    IN_OPT, OUT_OPT = 'input_cp_design_name_pre', 'output_cp_dist_name_pre'

    def _fields_maps(self, d2a):
        return [(f, d2a.cp_coarse_dist_deriv_list[d2a.opt_field.index(f)], d2a.init_cp_design[d2a.opt_field.index(f)], None) for f in d2a.cp_coarse_dist_field]
