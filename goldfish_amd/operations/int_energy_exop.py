"""IntEnergyExOperation -- internal (strain) energy of the non-matching structure and its
partials (reference: GOLDFISH/operations/int_energy_exop.py:3-107).  One device pass
(gf_functionals) produces the value and every gradient."""
import numpy as np


class IntEnergyExOperation(object):

    def __init__(self, nonmatching_opt, wint_regu=None):
        if wint_regu is not None:
            raise NotImplementedError("regularisation terms are SURVEY.md 8(f) N4")
        self.nonmatching_opt = nonmatching_opt
        self.num_splines = nonmatching_opt.num_splines
        self.splines = nonmatching_opt.splines
        self.opt_shape = nonmatching_opt.opt_shape
        self.opt_thickness = nonmatching_opt.opt_thickness
        if self.opt_shape:
            self.opt_field = nonmatching_opt.opt_field
            self.shopt_surf_inds = nonmatching_opt.shopt_surf_inds

    def _f(self, apply_bcs=True):
        return self.nonmatching_opt.functionals(apply_bcs=apply_bcs)

    def Wint(self):
        """int_energy_exop.py:55-59."""
        return float(self._f()["Wint"])

    def dWintduIGA(self, array=True, apply_bcs=True):
        """int_energy_exop.py:61-73 (Dirichlet rows zeroed by FE2IGA(..., apply_bcs))."""
        return self._f(apply_bcs)["dWdu"]

    def dWintdCPIGA(self, field, array=True):
        """int_energy_exop.py:75-90."""
        nm = self.nonmatching_opt
        return self._f()["dWdcp"][field][nm._shopt_cols[self.opt_field.index(field)]]

    def dWintdh_th(self, extract=False, array=True):
        """int_energy_exop.py:92-107."""
        nm = self.nonmatching_opt
        g = self._f()["dWdh"]
        return g if nm.var_thickness else np.add.reduceat(g, nm.cp_off[:-1])
