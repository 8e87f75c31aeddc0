"""ComplianceExOperation -- compliance of the non-matching structure and its partials
(reference: GOLDFISH/operations/compliance_exop.py:3-99): C = sum_s int forces[s] . u dA with the
homogeneous displacement function (the reference uses ``spline_funcs`` un-rationalised, :24-26).  ``c_regu`` (compliance_exop.py:9-28: a per-patch list of
forms added to the compliance form, ``None`` entries for patches without one): the device path evaluates ``ShapeRegu`` terms (the shape regularisation the
reference's demos build, operations/int_energy_exop.py) -- they depend on the control points only, so they enter cpl() and dcpldCPIGA()."""
import numpy as np

from .int_energy_exop import regu_fields, regu_values


class ComplianceExOperation(object):

    def __init__(self, nonmatching_opt, forces, c_regu=None):
        self.nonmatching_opt = nonmatching_opt
        self.num_splines = nonmatching_opt.num_splines
        self.splines = nonmatching_opt.splines
        self.opt_field = nonmatching_opt.opt_field
        self.opt_shape = nonmatching_opt.opt_shape
        self.forces = np.asarray(forces, float).reshape(self.num_splines, 3)
        self.c_regu, self._regu_fields = regu_fields(nonmatching_opt, c_regu, "c_regu")
        if self.opt_shape:
            self.shopt_surf_inds = nonmatching_opt.shopt_surf_inds

    def _c(self, apply_bcs=True):
        return self.nonmatching_opt.compliance(self.forces, apply_bcs=apply_bcs)

    def cpl(self):
        """compliance_exop.py:50-54."""
        return float(self._c()["C"]) + float(sum(r["value"] for r in regu_values(self.nonmatching_opt, self._regu_fields, "c_regu")))

    def dcplduIGA(self, array=True, apply_bcs=True):
        """compliance_exop.py:56-68."""
        return self._c(apply_bcs)["dCdu"]

    def dcpldCPIGA(self, field, array=True):
        """compliance_exop.py:70-99."""
        nm = self.nonmatching_opt
        cols = nm._shopt_cols[self.opt_field.index(field)]
        g = self._c()["dCdcp"][field][cols]
        for r in regu_values(nm, self._regu_fields, "c_regu"):
            g = g + r["dcp"][field][cols]
        return g
