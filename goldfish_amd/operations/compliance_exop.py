"""ComplianceExOperation -- compliance of the non-matching structure and its partials
(reference: GOLDFISH/operations/compliance_exop.py:3-99): C = sum_s int forces[s] . u dA with the
homogeneous displacement function (the reference uses ``spline_funcs`` un-rationalised, :24-26)."""
import numpy as np


class ComplianceExOperation(object):

    def __init__(self, nonmatching_opt, forces, c_regu=None):
        if c_regu is not None:
            raise NotImplementedError("regularisation terms are SURVEY.md 8(f) N4")
        self.nonmatching_opt = nonmatching_opt
        self.num_splines = nonmatching_opt.num_splines
        self.splines = nonmatching_opt.splines
        self.opt_field = nonmatching_opt.opt_field
        self.opt_shape = nonmatching_opt.opt_shape
        self.forces = np.asarray(forces, float).reshape(self.num_splines, 3)
        if self.opt_shape:
            self.shopt_surf_inds = nonmatching_opt.shopt_surf_inds

    def _c(self, apply_bcs=True):
        return self.nonmatching_opt.compliance(self.forces, apply_bcs=apply_bcs)

    def cpl(self):
        """compliance_exop.py:50-54."""
        return float(self._c()["C"])

    def dcplduIGA(self, array=True, apply_bcs=True):
        """compliance_exop.py:56-68."""
        return self._c(apply_bcs)["dCdu"]

    def dcpldCPIGA(self, field, array=True):
        """compliance_exop.py:70-99."""
        nm = self.nonmatching_opt
        return self._c()["dCdcp"][field][nm._shopt_cols[self.opt_field.index(field)]]
