"""VolumeExOperation -- material volume int h dA and its partials
(reference: GOLDFISH/operations/volume_exop.py:3-84)."""
import numpy as np


class VolumeExOperation(object):

    def __init__(self, nonmatching_opt, vol_surf_inds=None):
        self.nonmatching_opt = nonmatching_opt
        self.num_splines = nonmatching_opt.num_splines
        self.splines = nonmatching_opt.splines
        self.opt_shape = nonmatching_opt.opt_shape
        self.opt_thickness = nonmatching_opt.opt_thickness
        self.vol_surf_inds = list(range(self.num_splines)) if vol_surf_inds is None else list(vol_surf_inds)
        if len(self.vol_surf_inds) != self.num_splines:
            raise NotImplementedError("volume of a subset of patches is not on the device path yet")
        if self.opt_shape:
            self.opt_field = nonmatching_opt.opt_field
            self.shopt_surf_inds = nonmatching_opt.shopt_surf_inds

    def _f(self):
        return self.nonmatching_opt.functionals()

    def volume(self):
        """volume_exop.py:46-50."""
        return float(self._f()["volume"])

    def dvoldh_th(self, array=True):
        """volume_exop.py:52-66."""
        nm = self.nonmatching_opt
        g = self._f()["dVdh"]
        return g if nm.var_thickness else np.add.reduceat(g, nm.cp_off[:-1])

    def dvoldCPIGA(self, field, array=True):
        """volume_exop.py:68-84."""
        nm = self.nonmatching_opt
        return self._f()["dVdcp"][field][nm._shopt_cols[self.opt_field.index(field)]]
